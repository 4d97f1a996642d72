"""GPU parity of the label/mask ops and the fused loss kernel (bit-exact index bookkeeping)."""
import numpy as np
import pytest
import torch

from conftest import unpack
import densebox_amd.labels as LB
from densebox_amd.loss import densebox_loss
from densebox_amd import _lib, synth
from oracle import densebox_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def test_label_ops_bit_exact(golden):
    g = golden('labels')
    bbox, vert, lab, ok = T(g['bbox']), T(g['vert']), T(g['lab']), g['ok_rows']
    B = bbox.size(0)

    def eq(a, ref):
        assert np.array_equal(a.cpu().numpy(), ref)
    eq(LB.init_score_map(bbox, B, ratio=0.3), unpack(g['score_map'], (B, 1, 60, 60)))
    eq(LB.init_score(bbox, lab, ratio=0.3), unpack(g['score_pn'], (B, 1, 60, 60)))
    eq(LB.init_loc_map(bbox, B), g['loc_map'])
    eq(LB.init_loc(bbox, lab), g['loc_pn'])
    eq(LB.init_lm_heatmap(vert[ok], len(ok)), unpack(g['lm_heat'], (len(ok), 4, 60, 60)))
    eq(LB.init_lm_heatmap_pn(vert, lab), unpack(g['lm_heat_pn'], (B, 4, 60, 60)))
    eq(LB.init_lm_locmap(vert[ok], len(ok)), g['lm_loc'])
    eq(LB.init_lm_locmap_pn(vert, lab), g['lm_loc_pn'])
    with pytest.raises(IndexError):          # landmark x=239 -> index 60 (reference raises too, DenseBox.py:1821)
        LB.init_lm_heatmap(vert, B)


def test_mask_ops_bit_exact(golden):
    g = golden('labels')
    bbox, vert, lab = T(g['bbox']), T(g['vert']), T(g['lab'])
    B = bbox.size(0)
    for tag, labels in (('', None), ('_pn', lab)):
        gt = LB.init_score_map(bbox, labels=labels)
        pos = torch.nonzero(gt)
        assert np.array_equal(pos.cpu().numpy(), g['pos_idx' + tag])
        m = gt.clone()
        LB.mask_by_sel(m, pos, T(g['neg_idx']))
        assert np.array_equal(m.cpu().numpy(), unpack(g['mask_sel' + tag], (B, 1, 60, 60)))
        if labels is None:
            LB.mask_gray_zone_cls(m, bbox, ratio=0.3, gray_border=2.0)
        else:
            LB.mask_gray_zone_cls_pn(m, bbox, lab, ratio=0.3, gray_border=2.0)
        assert np.array_equal(m.cpu().numpy(), unpack(g['mask_gray' + tag], (B, 1, 60, 60)))
    heat = LB.init_lm_heatmap_pn(vert, lab)
    chans = []
    for i in range(4):
        gti = heat[:, i:i + 1].contiguous()
        mi = gti.clone()
        pos = torch.nonzero(gti)
        LB.mask_by_sel(mi, pos, T(g['lm_neg_idx'][i]))
        LB.mask_gray_zone_lm(mi, pos, i, gray_border=2.0)
        chans.append(mi)
    assert np.array_equal(torch.cat(chans, 1).cpu().numpy(), unpack(g['lm_mask'], (B, 4, 60, 60)))
    lo = T(g['neg_loss_in']).cuda()
    assert np.array_equal(LB.gen_neg_loss(lo, LB.init_score(bbox, lab)).cpu().numpy(), g['neg_loss_out'])


def test_positive_count_host_equals_device(golden):
    g = golden('labels')
    bbox, lab = T(g['bbox']), T(g['lab'])
    for labels in (None, lab):
        host = LB.positive_count(bbox, labels)
        dev = torch.empty(bbox.size(0), dtype=torch.int32, device='cuda')
        bb, lb = bbox.cuda(), (labels.cuda() if labels is not None else None)    # keep alive across the launch
        _lib.check(_lib.lib().dbx_count_positives(_lib.ptr(bb), _lib.ptr(lb), bbox.size(0), _lib.ptr(dev), _lib.stream_ptr()))
        maps = LB.init_score_map(bbox, labels=labels).sum(dim=(1, 2, 3)).cpu().numpy()
        assert np.array_equal(host, dev.cpu().numpy()) and np.array_equal(host, maps.astype(np.int64))


CAPS = ['train_DenseBox', 'train_DenseBox_dropout', 'train_DenseBoxLM', 'train_DenseBoxLMLOC']


@pytest.mark.parametrize('name', CAPS)
def test_fused_loss_vs_reference_capture(golden, name):
    """Feed the reference's own captured network outputs to the fused kernel: indices and masks bit-exact,
    loss and dL/dout to fp32 round-off."""
    g = golden(name)
    kind = str(g['kind'])
    n = int(g['batch'])
    outs = []
    i = 0
    while 's0_out_%d' % i in g.files:
        outs.append(T(g['s0_out_%d' % i]).cuda().requires_grad_(True))
        i += 1
    neg0 = g['s0_neg_idx_0']
    half = neg0.shape[1] // 2
    lm_rand = None if kind == 'DenseBox' else np.stack([g['s0_neg_idx_%d' % (1 + j)][:, 1:] for j in range(4)])
    kw = {k[3:]: float(g[k]) for k in g.files if k.startswith('kw_')}
    loss, dbg = densebox_loss(kind, tuple(outs), g['bbox'][:n], g['vert'][:n], g['lab'][:n], rand_neg_indices=neg0[:, half:],
                              lm_rand_neg_indices=lm_rand, return_debug=True, **kw)
    assert dbg['half'] == half
    assert np.array_equal(dbg['neg_idx'].cpu().numpy(), neg0)
    gray = 's0_mask_gray_zone_cls_pn_0' if kind == 'DenseBoxLMLOC' else 's0_mask_gray_zone_cls_0'
    assert np.array_equal(dbg['mask_cls'].cpu().numpy(), unpack(g[gray], (n, 1, 60, 60)))
    if kind != 'DenseBox':
        for j in range(4):
            assert np.array_equal(dbg['lm_neg_idx'][j].cpu().numpy(), g['s0_neg_idx_%d' % (1 + j)])
        for j in range(4):                      # every landmark channel's mask after mining + its gray zone (channel views)
            assert np.array_equal(dbg['mask_lm'][:, j:j + 1].cpu().numpy(), unpack(g['s0_mask_gray_zone_lm_%d' % j], (n, 1, 60, 60))), j
    assert np.isclose(float(loss.detach()), float(g['s0_loss']), rtol=2e-6)
    loss.backward()
    # dL/d(out) of the loss alone: the oracle's autograd on the same leaf tensors.  (The captured s0_dout of the score /
    # landmark outputs additionally contains the gradient flowing back through the refine branch, which belongs to
    # the network backward and is checked in test_hip_backward.py.)
    leaf = [o.detach().cpu().clone().requires_grad_(True) for o in outs]
    res = O.loss_step(kind, tuple(leaf), g['bbox'][:n], g['vert'][:n], g['lab'][:n], rand_neg=neg0[:, half:],
                      lm_rand_neg=lm_rand, **kw)
    res['loss'].backward()
    assert np.isclose(float(loss.detach()), float(res['loss'].detach()), rtol=2e-6)
    direct = {'DenseBox': (0, 1), 'DenseBoxLM': (1, 3), 'DenseBoxLMLOC': (1, 2, 4)}[kind]
    for i, o in enumerate(outs):
        ref = leaf[i].grad.numpy()
        assert np.allclose(o.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(ref).max())), i
        if i in direct:       # outputs that do not feed the refine branch: the captured gradient is the direct one
            cap = g['s0_dout_%d' % i]
            assert np.allclose(o.grad.cpu().numpy(), cap, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(cap).max())), i


@pytest.mark.parametrize('half', [1, 11, 48, 49, 130])
def test_mining_paths_match_the_oracle_for_small_and_large_k(golden, half):
    """The hard-negative top-K runs as a register tournament for K <= 48 and as K block-wide arg-max rounds above: both must give the
    oracle's index list (descending loss; the reference's torch.topk leaves the order of exactly equal losses unspecified, so the maps are
    continuous random numbers: no ties above the K-th loss) and masks bit for bit, and its loss -- K is forced through positive_num_global."""
    g = golden('train_DenseBoxLMLOC')
    kind = 'DenseBoxLMLOC'
    n = 3
    rs = np.random.RandomState(100 + half)
    shapes = [(n, 1, 60, 60), (n, 1, 60, 60), (n, 4, 60, 60), (n, 4, 60, 60), (n, 8, 60, 60)]
    outs_np = [rs.randn(*sh).astype(np.float32) for sh in shapes]
    rand_neg = np.stack([rs.choice(3600, half, replace=False) for _ in range(n)])
    lm_rand = np.stack([np.stack([rs.choice(3600, 1, replace=False) for _ in range(n)]) for _ in range(4)])
    P = 2 * half * n                                                                             # neg_counts(P, n) -> half
    assert LB.neg_counts(P, n)[1] == half
    outs = [T(o).cuda().requires_grad_(True) for o in outs_np]
    loss, dbg = densebox_loss(kind, tuple(outs), g['bbox'][:n], g['vert'][:n], g['lab'][:n], rand_neg_indices=rand_neg,
                              lm_rand_neg_indices=lm_rand, return_debug=True, batch_global=n, positive_num_global=P)
    leaf = [T(o).requires_grad_(True) for o in outs_np]
    res = O.loss_step(kind, tuple(leaf), g['bbox'][:n], g['vert'][:n], g['lab'][:n], rand_neg=rand_neg, lm_rand_neg=lm_rand,
                      batch_global=n, positive_num_global=P)
    assert dbg['half'] == half == res['half']
    assert np.array_equal(dbg['neg_idx'].cpu().numpy(), res['neg_idx'])
    assert np.array_equal(dbg['mask_cls'].cpu().numpy(), res['mask'])
    assert np.isclose(float(loss.detach()), float(res['loss'].detach()), rtol=2e-6)
    loss.backward(); res['loss'].backward()
    for o, l in zip(outs, leaf):
        ref = l.grad.numpy()
        assert np.allclose(o.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(ref).max()))


def test_all_negative_batch_has_no_mined_negatives():
    """A batch without a single positive patch (DenseBox.py:2074: neg_num = int(0 / N + 0.5) = 0; only train_densebox_online, whose labels
    say so -- the bbox-driven loops always find the pixel of an all-zero box): nothing is mined, the score / bbox maps contribute no loss and
    no gradient, the landmark channels keep their one random negative each -- loss, index lists (empty) and dL/dout against the oracle."""
    n, kind = 3, 'DenseBoxLMLOC'
    x, bbox, vert, lab = synth.synth_batch(n, seed=5, neg_frac=1.0)
    assert int(LB.positive_count(bbox, lab).sum()) == 0 and LB.neg_counts(0, n) == (0, 0)
    assert int(LB.positive_count(bbox, None).sum()) == n           # (the bbox-driven count: one pixel per all-zero box)
    rs = np.random.RandomState(3)
    shapes = [(n, 1, 60, 60), (n, 4, 60, 60)] if kind == 'DenseBox' else [(n, 1, 60, 60), (n, 1, 60, 60), (n, 4, 60, 60), (n, 4, 60, 60), (n, 8, 60, 60)]
    outs_np = [rs.randn(*sh).astype(np.float32) for sh in shapes]
    rand_neg = np.zeros((n, 0), dtype=np.int64)
    lm_rand = None if kind == 'DenseBox' else np.stack([np.stack([rs.choice(3600, 1, replace=False) for _ in range(n)]) for _ in range(4)])
    outs = [T(o).cuda().requires_grad_(True) for o in outs_np]
    loss, dbg = densebox_loss(kind, tuple(outs), bbox.numpy(), vert.numpy(), lab.numpy(), rand_neg_indices=rand_neg,
                              lm_rand_neg_indices=lm_rand, return_debug=True)
    leaf = [T(o).requires_grad_(True) for o in outs_np]
    res = O.loss_step(kind, tuple(leaf), bbox.numpy(), vert.numpy(), lab.numpy(), rand_neg=rand_neg, lm_rand_neg=lm_rand)
    assert dbg['half'] == 0 == res['half'] and dbg['neg_idx'].numel() == 0 and res['neg_idx'].size == 0
    assert np.array_equal(dbg['mask_cls'].cpu().numpy(), res['mask']) and not res['mask'].any()
    lo = float(res['loss'].detach())
    assert abs(float(loss.detach()) - lo) <= 2e-6 * max(1.0, abs(lo))
    loss.backward()
    if res['loss'].requires_grad:
        res['loss'].backward()
    for o, l in zip(outs, leaf):
        ref = l.grad.numpy() if l.grad is not None else np.zeros(o.shape, np.float32)
        got = o.grad.cpu().numpy() if o.grad is not None else np.zeros(o.shape, np.float32)
        assert np.allclose(got, ref, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(ref).max()))
