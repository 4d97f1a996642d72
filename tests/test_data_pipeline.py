"""Input side of the path: file-name label parsing vs the reference's Dataset constructors
(fixture datasets.npz) on CPU; uint8 -> normalised network input on the GPU."""
import numpy as np
import pytest
import torch

from densebox_amd import data as DT
from oracle import densebox_oracle as O


def test_label_parsers_match_reference_datasets(golden):
    g = golden('datasets')
    for i, name in enumerate(g['db_names']):
        b, v, l = DT.parse_densebox_label(str(name))
        assert np.array_equal(b, g['db_bbox'][i]) and np.array_equal(v, g['db_vert'][i]) and np.array_equal(l, g['db_lab'][i])
        ob, neg = O.parse_label_name(str(name), 12)
        assert np.array_equal(np.concatenate([b, v]), ob if not neg else np.zeros(12, np.float32)) and neg == (l[0] == 0)
    for i, name in enumerate(g['lm_names']):
        b, v = DT.parse_lm_label(str(name))
        assert np.array_equal(b, g['lm_bbox'][i]) and np.array_equal(v, g['lm_vert'][i])
    for i, name in enumerate(g['lp_names']):
        assert np.array_equal(DT.parse_bbox_label(str(name)), g['lp_bbox'][i])
    bb, vv, ll = DT.collate_labels([str(n) for n in g['db_names']])
    assert tuple(bb.shape) == (6, 4) and tuple(vv.shape) == (6, 8) and tuple(ll.shape) == (6, 1)
    assert bb.dtype == torch.float32 and np.array_equal(bb.numpy(), g['db_bbox'])
    with pytest.raises(ValueError):
        DT.parse_densebox_label('no_label_here.jpg')


@pytest.mark.gpu
def test_uint8_input_equals_normalised_fp32_input():
    """net(uint8 NHWC) must equal net(ToTensor+Normalize(uint8)) -- bit-identical network input, hence identical maps."""
    import densebox_amd as D
    from densebox_amd import synth
    rs = np.random.RandomState(3)
    u8 = torch.from_numpy(rs.randint(0, 256, size=(2, 240, 240, 3)).astype(np.uint8))
    ref_in = O.normalize_u8(u8)
    for dtype in ('f32', 'bf16'):
        net = D.DenseBox(synth.vgg19_standin(seed=0))
        synth.fill_params_(net, 11)
        net = net.cuda().eval()
        net.compute_dtype = dtype
        with torch.no_grad():
            a = net(u8.cuda())
            x0_u8 = net.engine().read_activation('x0', 0, 3).clone()
            b = net(ref_in.cuda())
            x0_f = net.engine().read_activation('x0', 0, 3).clone()
        assert torch.equal(x0_u8, x0_f), dtype           # ((u8/255) - mean) / std in fp32, then the same rounding
        for p, q in zip(a, b):
            assert torch.equal(p, q)
    assert torch.equal(x0_f.cpu(), ref_in.to(torch.bfloat16).float())
