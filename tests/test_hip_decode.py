"""GPU parity of top-K decode + NMS (bit-exact float64 rows and keep lists, DenseBox.py:3114-3443)."""
import numpy as np
import pytest
import torch

from densebox_amd import synth
from densebox_amd import decode as DC
import densebox_amd as D

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def test_decode_bit_exact(golden):
    g = golden('decode')
    s, l, hm, ll = T(g['a_s']), T(g['a_l']), T(g['a_hm']), T(g['a_ll'])
    assert np.array_equal(DC.parse_output(s, l, K=10), g['a_parse_output'])
    assert np.array_equal(DC.parse_out_MN(s, l, 240, 240, K=10), g['a_parse_out_MN'])
    assert np.array_equal(DC.parse_out_MN(s.cuda(), l.cuda(), 240, 240, K=50), g['a_parse_out_MN_K50'])
    assert np.array_equal(DC.parse_DetLM(s, l, hm, 240, 240, K=10), g['a_parse_DetLM'])
    assert np.array_equal(DC.parse_DetLMLOC(s, l, hm, ll, 240, 240, K=10), g['a_parse_DetLMLOC'])
    s, l, hm, ll = T(g['b_s']), T(g['b_l']), T(g['b_hm']), T(g['b_ll'])
    assert np.array_equal(DC.parse_out_MN(s, l, 101, 134, K=10), g['b_parse_out_MN'])
    assert np.array_equal(DC.parse_DetLM(s, l, hm, 101, 134, K=7), g['b_parse_DetLM'])
    assert np.array_equal(DC.parse_DetLMLOC(s, l, hm, ll, 101, 134, K=7), g['b_parse_DetLMLOC'])
    with pytest.raises(AssertionError):       # shape assert of the reference (DenseBox.py:3311)
        DC.parse_out_MN(s, l, 240, 240, K=10)


def test_nms_bit_exact(golden):
    g = golden('decode')
    for th, key in ((0.4, 'nms_keep_04'), (0.0, 'nms_keep_00'), (0.7, 'nms_keep_07')):
        assert DC.NMS(g['nms_in'], th) == list(g[key])
    assert DC.NMS(g['nms_big_in'], 0.4) == list(g['nms_big_keep'])
    for k_ in ('a_parse_output', 'a_parse_DetLM', 'a_parse_DetLMLOC', 'a_parse_out_MN_K50'):
        assert DC.NMS(g[k_], 0.4) == list(g[k_ + '_keep'])


def test_detect_1080p(golden):
    """Config 5: whole-image 1920x1080 forward + on-GPU top-K/NMS.  fp32 compute so the top-K set is the reference's."""
    g = golden('net_DenseBox_1080p')
    net = D.DenseBox(synth.vgg19_standin(seed=0))
    synth.fill_params_(net, 11)
    net = net.cuda().eval()
    net.compute_dtype = 'f32'
    dets, keep = net.detect(synth.synth_images(1, 1080, 1920, seed=5).cuda(), K=10, nms_thresh=0.4)
    assert dets.shape == (10, 5) and dets.dtype == np.float64
    assert np.allclose(dets, g['dets'], rtol=1e-4, atol=2e-3)
    assert keep == list(g['keep'])


def test_detect_graph_replay_across_shapes_and_training():
    """detect() caches one hipGraph per (shape, K, dtype, weights): shapes A, B, A again, then a training step that changes the
    weights, then A -- every replayed result must equal the eager result of the same call (round-1 advisor: a replay once read
    buffers a later shape had re-planned)."""
    import os
    import densebox_amd as D
    from densebox_amd import synth, labels as LB
    from densebox_amd.optim import SGD
    net = D.DenseBoxLMLOC(synth.vgg19_standin(seed=0))
    synth.fill_params_(net, 11)
    net = net.cuda().eval()
    net.compute_dtype = 'f16'
    xa = synth.synth_images(1, 240, 240, seed=1).cuda()
    xb = synth.synth_images(1, 192, 320, seed=2).cuda()

    def both(x, K):
        os.environ['DBX_GRAPH'] = '0'
        try:
            de, ke = net.detect(x, K=K, nms_thresh=0.4)
        finally:
            os.environ.pop('DBX_GRAPH', None)
        dg, kg = net.detect(x, K=K, nms_thresh=0.4)       # first call captures, later calls replay
        dg2, kg2 = net.detect(x, K=K, nms_thresh=0.4)
        assert np.array_equal(np.asarray(de), np.asarray(dg)) and list(ke) == list(kg), 'graph != eager'
        assert np.array_equal(np.asarray(dg), np.asarray(dg2)) and list(kg) == list(kg2), 'replay != capture'
        return np.asarray(de).copy()
    a1 = both(xa, 10)
    b1 = both(xb, 10)
    a2 = both(xa, 10)                                     # back to A: its graph must still own valid buffers
    assert np.array_equal(a1, a2)
    b2 = both(xb, 25)                                     # another K on shape B
    assert b2.shape[0] == 25 and b1.shape[0] == 10
    # a training step in between: new weights, new plan -- the cached graphs must notice and re-capture
    net.train()
    net.compute_dtype = 'f16'
    opt = SGD(net.parameters(), lr=2e-9, momentum=0.9, weight_decay=5e-8)
    x, bbox, vert, lab = synth.synth_batch(2, seed=7, neg_frac=0.0)
    outs = net(x.cuda())
    loss = net.loss(outs, bbox, vert, lab)
    loss.backward()
    opt.step()
    net.eval()
    a3 = both(xa, 10)
    assert not np.array_equal(a1, a3), 'detections did not change after a weight update (stale graph)'


def test_nms_and_detect_survive_nan_scores():
    """A diverged network (NaN / Inf maps) must not turn into an out-of-range row index on the GPU (found in round 3: NaN scores
    made the rank sort of the NMS a non-permutation and the kernel read dets[garbage])."""
    d = np.random.RandomState(0).rand(40, 5) * 100
    d[:, 2:4] += d[:, 0:2]
    d[::3, 4] = np.nan
    keep = DC.NMS(d, 0.4)
    assert all(0 <= k < 40 for k in keep)
    s = torch.full((1, 1, 60, 60), float('nan')); l = torch.zeros(1, 4, 60, 60)
    out = DC.parse_out_MN(s.cuda(), l.cuda(), 240, 240, K=10)
    assert out.shape == (10, 5)


@pytest.mark.parametrize('case', ['random', 'quantised', 'constant', 'signed_zero'])
def test_topk_select_path_matches_rounds_and_reference_order(case):
    """K in (48, 1024] takes the radix-select path (smaller and larger K: arg-max rounds): the ranking must be the reference order
    bit for bit on both -- larger score first, LOWER INDEX on ties -- also when the K-th score has more copies than places (ties cut by index), on all-equal
    maps and for +0 / -0 (equal as floats)."""
    rs = np.random.RandomState({'random': 1, 'quantised': 2, 'constant': 3, 'signed_zero': 4}[case])
    rows, cols = 67, 120
    n = rows * cols
    if case == 'random':
        sc = rs.randn(n).astype(np.float32)
    elif case == 'quantised':
        sc = (rs.randint(0, 7, size=n) / 4.0).astype(np.float32)          # 7 distinct values: hundreds of ties per value
    elif case == 'constant':
        sc = np.full(n, 0.25, np.float32)
    else:
        sc = np.where(rs.rand(n) < 0.5, np.float32(0.0), np.float32(-0.0)).astype(np.float32)
        sc[rs.randint(0, n, 30)] = -1.0
    loc = rs.randn(4, n).astype(np.float32)
    s = torch.from_numpy(sc).view(1, 1, rows, cols).cuda()
    l = torch.from_numpy(loc).view(1, 4, rows, cols).cuda()
    # reference order: value descending, index ascending on ties (what K arg-max rounds with "lower index wins" produce)
    order = np.lexsort((np.arange(n), -sc.astype(np.float64)))
    for K in (1, 2, 10, 16, 17, 40, 49, 300, 1000, 1024, 1500):
        dets, topk, keep = DC._run(s, l, rows * 4, cols * 4, K)
        got = topk.cpu().numpy()
        assert np.array_equal(got, order[:K]), (case, K, np.nonzero(got != order[:K])[0][:5])
        d = dets.cpu().numpy()
        assert np.array_equal(d[:, 4], sc[order[:K]].astype(np.float64))
        xi = (order[:K] % cols).astype(np.float32)
        assert np.array_equal(d[:, 0], (xi - loc[0, order[:K]]).astype(np.float32).astype(np.float64) * 4.0)
