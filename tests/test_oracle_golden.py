"""Pin the CPU oracle against vectors captured from the reference itself
(tests/golden/*.npz, made by oracle/gen_golden.py).  Integer/index bookkeeping
must be bit-exact; fp32 maps within 1e-5 relative (same ATen kernels, different
call structure)."""
import numpy as np
import pytest
import torch

from conftest import unpack
from densebox_amd import synth
from oracle import densebox_oracle as O


# --------------------------------------------------------------------------- labels / masks
def test_label_maps_bit_exact(golden):
    g = golden('labels')
    bbox, vert, lab, ok = g['bbox'], g['vert'], g['lab'], g['ok_rows']
    B = bbox.shape[0]
    assert np.array_equal(O.init_score_map(bbox), unpack(g['score_map'], (B, 1, 60, 60)))
    assert np.array_equal(O.init_score_map(bbox, lab), unpack(g['score_pn'], (B, 1, 60, 60)))
    assert np.array_equal(O.init_loc_map(bbox), g['loc_map'])
    assert np.array_equal(O.init_loc_map(bbox, lab), g['loc_pn'])
    assert np.array_equal(O.init_lm_heatmap(vert[ok]), unpack(g['lm_heat'], (len(ok), 4, 60, 60)))
    assert np.array_equal(O.init_lm_heatmap(vert, lab), unpack(g['lm_heat_pn'], (B, 4, 60, 60)))
    assert np.array_equal(O.init_lm_locmap(vert[ok]), g['lm_loc'])
    assert np.array_equal(O.init_lm_locmap(vert, lab), g['lm_loc_pn'])


def test_masks_bit_exact(golden):
    g = golden('labels')
    bbox, vert, lab = g['bbox'], g['vert'], g['lab']
    B = bbox.shape[0]
    for tag, labels in (('', None), ('_pn', lab)):
        gt = O.init_score_map(bbox, labels)
        pos = O.nonzero4(gt)
        assert np.array_equal(pos, g['pos_idx' + tag])
        m = gt.copy()
        O.mask_by_sel(m, pos, g['neg_idx'])
        assert np.array_equal(m, unpack(g['mask_sel' + tag], (B, 1, 60, 60)))
        O.mask_gray_zone_cls(m, bbox, labels)
        assert np.array_equal(m, unpack(g['mask_gray' + tag], (B, 1, 60, 60)))
    heat = O.init_lm_heatmap(vert, lab)
    lm_mask = heat.copy()
    for i in range(4):
        view = lm_mask[:, i:i + 1]
        pos = O.nonzero4(heat[:, i:i + 1])
        O.mask_by_sel(view, pos, g['lm_neg_idx'][i])
        O.mask_gray_zone_lm(view, pos)
    assert np.array_equal(lm_mask, unpack(g['lm_mask'], (B, 4, 60, 60)))
    gtp = O.init_score_map(bbox, lab)
    assert np.array_equal(O.gen_neg_loss(g['neg_loss_in'], gtp), g['neg_loss_out'])


def test_neg_counts():
    # DenseBox.py:2074/2081; SURVEY 8(a10): N=2, 42 positives -> neg_num 21, half 11
    assert O.neg_counts(42, 2) == (21, 11)
    assert O.neg_counts(0, 4) == (0, 0)
    assert O.neg_counts(5, 2) == (3, 2)      # int(2.5+0.5)=3; int(1.5+0.5)=2


# --------------------------------------------------------------------------- networks
KINDS = ['DenseBox', 'DenseBoxLM', 'DenseBoxLMLOC']


def _oracle_params(kind, seed):
    import densebox_amd as D
    net = getattr(D, kind)(synth.vgg19_standin(seed=0))
    synth.fill_params_(net, seed)
    return net, O.params_of(net)


@pytest.mark.parametrize('kind', KINDS)
def test_forward_matches_reference(golden, kind):
    g = golden('net_' + kind)
    net, P = _oracle_params(kind, int(g['param_seed']))
    # the product module must expose the reference's exact (aliased) state_dict key list
    assert list(net.state_dict().keys()) == [str(k) for k in g['keys']]
    assert [n for n, _ in net.named_parameters()] == [str(k) for k in g['param_names']]
    sums = np.array([float(p.double().sum()) for _, p in net.named_parameters()])
    assert np.allclose(sums, g['param_sums'], rtol=1e-12, atol=1e-12)      # (the float64 sum's order depends on the host's thread count)
    with torch.no_grad():
        outs = O.forward(kind, P, synth.synth_images(2, 240, 240, seed=3))
        for i, o in enumerate(outs):
            ref = g['out240_%d' % i]
            assert o.shape == ref.shape
            assert np.allclose(o.numpy(), ref, rtol=1e-5, atol=1e-5), (kind, i, np.abs(o.numpy() - ref).max())
        outs = O.forward(kind, P, synth.synth_images(1, 100, 132, seed=4))
        for i, o in enumerate(outs):
            ref = g['outodd_%d' % i]
            assert o.shape == ref.shape == (1, ref.shape[1], 25, 33)
            assert np.allclose(o.numpy(), ref, rtol=1e-5, atol=1e-5)


# --------------------------------------------------------------------------- training-step captures
def _run_capture(golden, name, step=0):
    g = golden(name)
    kind = str(g['kind'])
    net, P = _oracle_params(kind, int(g['param_seed']))
    n, batch = int(g['n_patch']), int(g['batch'])
    x, _, _, _ = synth.synth_batch(n, seed=int(g['seed']))
    sl = slice(step * batch, (step + 1) * batch)
    return g, kind, P, x[sl], g['bbox'][sl], g['vert'][sl], g['lab'][sl]


def _dropmasks(g, kind, batch):
    if str(g['dropout']) != 'mask':
        return None
    names = [h for h, _ in O.HEADS[kind]]
    return {h: torch.from_numpy(unpack(g['dropmask_%d' % i], (batch, 512, 60, 60))) for i, h in enumerate(names)}


@pytest.mark.parametrize('name', ['train_DenseBox', 'train_DenseBox_dropout', 'train_DenseBoxLM',
                                  'train_DenseBoxLMLOC'])
def test_training_step_matches_reference(golden, name):
    g, kind, P, x, bbox, vert, lab = _run_capture(golden, name)
    N = x.shape[0]
    for p in P.values():
        p.requires_grad_(True)
    outs = O.forward(kind, P, x, _dropmasks(g, kind, N))
    for i, o in enumerate(outs):
        o.retain_grad()
        assert np.allclose(o.detach().numpy(), g['s0_out_%d' % i], rtol=2e-5, atol=2e-5)
    neg0 = g['s0_neg_idx_0']
    half = neg0.shape[1] // 2
    kw = {k[3:]: float(g[k]) for k in g.files if k.startswith('kw_')}
    lm_rand = None
    if kind != 'DenseBox':
        lm_rand = np.stack([g['s0_neg_idx_%d' % (1 + i)][:, 1:] for i in range(4)])
    res = O.loss_step(kind, outs, bbox, vert, lab, rand_neg=neg0[:, half:], lm_rand_neg=lm_rand, **kw)
    # index bookkeeping: bit-exact
    assert res['half'] == half
    assert np.array_equal(res['neg_idx'], neg0)
    assert np.array_equal(res['mask_sel'], unpack(g['s0_mask_sel_0'], (N, 1, 60, 60)))
    gray = 's0_mask_gray_zone_cls_pn_0' if kind == 'DenseBoxLMLOC' else 's0_mask_gray_zone_cls_0'
    assert np.array_equal(res['mask'], unpack(g[gray], (N, 1, 60, 60)))
    if kind != 'DenseBox':
        for i in range(4):
            assert np.array_equal(res['lm_neg_idx'][i], g['s0_neg_idx_%d' % (1 + i)])
        # the capture of channel 3's gray-zone call sees the view of the finished [N,4] mask's channel 3
        assert np.array_equal(res['lm_mask'][:, 3:4], unpack(g['s0_mask_gray_zone_lm_3'], (N, 1, 60, 60)))
    # loss + gradients
    assert np.isclose(float(res['loss']), float(g['s0_loss']), rtol=2e-6)
    res['loss'].backward()
    for i, o in enumerate(outs):
        assert np.allclose(o.grad.numpy(), g['s0_dout_%d' % i], rtol=1e-4, atol=1e-4)
    for n_, p in P.items():
        if 's0_gnone_' + n_ in g.files:
            assert p.grad is None            # conv3_3 is never executed (DenseBox.py:193-195)
            continue
        gr = p.grad.numpy()
        stat = g['s0_gstat_' + n_]
        flat = gr.reshape(-1).astype(np.float64)
        assert np.isclose(np.abs(flat).sum(), stat[1], rtol=1e-4), n_
        if 's0_g_' + n_ in g.files:
            ref = g['s0_g_' + n_]
            assert np.allclose(gr, ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max()), n_
        else:
            ref = g['s0_gsub_' + n_]
            assert np.allclose(gr.reshape(-1)[::997], ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max()), n_


def test_sgd_two_steps(golden):
    g = golden('train_DenseBox')
    lr = float(g['lr'])
    for n_ in ('conv1_1_1.weight', 'conv5_2_det.weight', 'conv5_2_loc.bias'):
        p = torch.from_numpy(g['s0_pb_' + n_])
        buf = None
        for s in range(int(g['n_steps'])):
            assert np.array_equal(p.numpy(), g['s%d_pb_%s' % (s, n_)]) or s > 0
            p, buf = O.sgd_step(p, torch.from_numpy(g['s%d_g_%s' % (s, n_)]), buf, lr)
            ref = g['s%d_pa_%s' % (s, n_)]
            assert np.allclose(p.numpy(), ref, rtol=1e-6, atol=1e-9), (n_, s)
    assert [O.adjust_lr(e) for e in (0, 4, 5, 9, 10, 14, 15, 99)] == [1e-9, 1e-9, 2e-9, 2e-9, 4e-9, 4e-9, 1e-9, 1e-9]


# --------------------------------------------------------------------------- decode / NMS
def test_decode_bit_exact(golden):
    g = golden('decode')
    s, l, hm, ll = g['a_s'], g['a_l'], g['a_hm'], g['a_ll']
    assert np.array_equal(O.parse_det(s, l, 240, 240, 10), g['a_parse_output'])
    assert np.array_equal(O.parse_det(s, l, 240, 240, 10), g['a_parse_out_MN'])
    assert np.array_equal(O.parse_det(s, l, 240, 240, 50), g['a_parse_out_MN_K50'])
    assert np.array_equal(O.parse_det(s, l, 240, 240, 10, lm_heat=hm), g['a_parse_DetLM'])
    assert np.array_equal(O.parse_det(s, l, 240, 240, 10, lm_heat=hm, lm_loc=ll), g['a_parse_DetLMLOC'])
    s, l, hm, ll = g['b_s'], g['b_l'], g['b_hm'], g['b_ll']
    assert np.array_equal(O.parse_det(s, l, 101, 134, 10), g['b_parse_out_MN'])
    assert np.array_equal(O.parse_det(s, l, 101, 134, 7, lm_heat=hm), g['b_parse_DetLM'])
    assert np.array_equal(O.parse_det(s, l, 101, 134, 7, lm_heat=hm, lm_loc=ll), g['b_parse_DetLMLOC'])


def test_nms_bit_exact(golden):
    g = golden('decode')
    for th, key in ((0.4, 'nms_keep_04'), (0.0, 'nms_keep_00'), (0.7, 'nms_keep_07')):
        assert O.nms(g['nms_in'], th) == list(g[key])
    assert O.nms(g['nms_big_in'], 0.4) == list(g['nms_big_keep'])
    for k_ in ('a_parse_output', 'a_parse_DetLM', 'a_parse_DetLMLOC', 'a_parse_out_MN_K50'):
        assert O.nms(g[k_], 0.4) == list(g[k_ + '_keep'])


def test_1080p_whole_image(golden):
    """Whole-image FCN inference (DenseBox.py:3772-3799) on a 1080x1920 input -- slow on CPU (~10 s)."""
    g = golden('net_DenseBox_1080p')
    _, P = _oracle_params('DenseBox', 11)
    with torch.no_grad():
        s, l = O.forward('DenseBox', P, synth.synth_images(1, 1080, 1920, seed=5))
    assert tuple(s.shape) == tuple(g['score_shape'])
    assert np.allclose(s[0, 0, ::9, ::8].numpy(), g['score_sub'], rtol=1e-5, atol=1e-5)
    assert np.allclose(l[0, :, ::9, ::8].numpy(), g['loc_sub'], rtol=1e-5, atol=1e-5)
    dets = O.parse_det(s.numpy(), l.numpy(), 1080, 1920, 10)
    assert np.allclose(dets, g['dets'], rtol=1e-5, atol=1e-4)
    assert O.nms(g['dets'], 0.4) == list(g['keep'])
