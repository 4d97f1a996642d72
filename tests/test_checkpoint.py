"""Checkpoint interop: the reference saves `net.state_dict()` (DenseBox.py:2206) and loads it with
load_state_dict (:1989-1994).  tests/golden/checkpoints.npz was captured from the imported reference (oracle/gen_golden.py
checkpoints): key order, per-entry shape / dtype / sums, alias groups, the small tensors, and the forward of a net carrying
those weights.  CPU: a state_dict of the build is entry-for-entry what the reference writes.  GPU: a net restored from such a
file keeps the aliases across .cuda() and reproduces the reference forward on the HIP path."""
import io

import numpy as np
import pytest
import torch

import densebox_amd as D
from densebox_amd import synth

KINDS = ('DenseBox', 'DenseBoxLM', 'DenseBoxLMLOC')


def _reference_format_file(g, kind):
    """A .pth byte stream with exactly the entries the reference wrote (checked against the fixture)."""
    a = getattr(D, kind)(synth.vgg19_standin(seed=0))
    synth.fill_params_(a, int(g['ckpt_seed']))
    sd = a.state_dict()
    keys = [str(k) for k in g[kind + '_keys']]
    assert list(sd.keys()) == keys
    for i, k in enumerate(keys):
        t = sd[k]
        assert ','.join(str(v) for v in t.shape) == str(g[kind + '_shapes'][i]), k
        assert str(t.dtype) == str(g[kind + '_dtypes'][i]), k
        assert abs(float(t.double().sum()) - float(g[kind + '_sums'][i])) <= 1e-9 * max(1.0, float(g[kind + '_l1'][i])), k
        assert abs(float(t.double().abs().sum()) - float(g[kind + '_l1'][i])) <= 1e-9 * max(1.0, float(g[kind + '_l1'][i])), k
        vk = '%s_val_%s' % (kind, k)
        if vk in g.files:
            assert np.array_equal(t.numpy(), g[vk]), k
    # the same keys share storage as in the reference's dict
    groups = sorted(sorted(int(v) for v in str(s).split(';')) for s in g[kind + '_alias'])
    ptr = {}
    for i, k in enumerate(keys):
        ptr.setdefault(sd[k].data_ptr(), []).append(i)
    assert sorted(v for v in ptr.values() if len(v) > 1) == groups
    buf = io.BytesIO()
    torch.save(sd, buf)
    buf.seek(0)
    return buf, keys, groups


@pytest.mark.parametrize('kind', KINDS)
def test_state_dict_is_what_the_reference_writes(golden, kind):
    g = golden('checkpoints')
    buf, keys, groups = _reference_format_file(g, kind)
    b = getattr(D, kind)(synth.vgg19_standin(seed=1))
    missing = b.load_state_dict(torch.load(buf), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    sd = b.state_dict()
    for grp in groups:                                   # aliases survive the load
        assert len({sd[keys[i]].data_ptr() for i in grp}) == 1


@pytest.mark.gpu
@pytest.mark.parametrize('kind', KINDS)
def test_restored_checkpoint_runs_on_the_hip_path(golden, kind):
    g = golden('checkpoints')
    buf, keys, groups = _reference_format_file(g, kind)
    net = getattr(D, kind)(synth.vgg19_standin(seed=1))
    net.load_state_dict(torch.load(buf), strict=True)
    net = net.cuda().eval()
    net.compute_dtype = 'f32'
    sd = net.state_dict()
    for grp in groups:                                   # .cuda() keeps one tensor per alias group
        assert len({sd[keys[i]].data_ptr() for i in grp}) == 1 and sd[keys[grp[0]]].is_cuda
    with torch.no_grad():
        outs = net(synth.synth_images(1, 240, 240, seed=6).cuda())
    n_out = sum(1 for k in g.files if k.startswith(kind + '_out_'))
    assert len(outs) == n_out
    for i, o in enumerate(outs):
        ref = torch.from_numpy(g['%s_out_%d' % (kind, i)])
        err = (o.cpu() - ref).abs().max().item()
        assert err <= 1e-4 * max(1.0, ref.abs().max().item()), (kind, i, err)
    # and the reverse direction: what the build saves from the GPU loads into a CPU module of the reference's layout
    buf2 = io.BytesIO()
    torch.save(net.state_dict(), buf2)
    buf2.seek(0)
    c = getattr(D, kind)(synth.vgg19_standin(seed=2))
    c.load_state_dict(torch.load(buf2, map_location='cpu'), strict=True)
    for (n1, p1), (n2, p2) in zip(net.named_parameters(), c.named_parameters()):
        assert n1 == n2 and torch.equal(p1.cpu(), p2)
