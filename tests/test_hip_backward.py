"""GPU parity of the full training step (forward -> fused loss -> HIP backward -> fused SGD) against the
gradients / parameter updates captured from the reference's own training loops."""
import os
import numpy as np
import pytest
import torch

from conftest import unpack
from densebox_amd import synth
from densebox_amd.optim import SGD, adjust_LR
import densebox_amd as D

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _setup(golden, name, dtype):
    g = golden(name)
    kind = str(g['kind'])
    net = getattr(D, kind)(synth.vgg19_standin(seed=0))
    synth.fill_params_(net, int(g['param_seed']))
    net = net.cuda().train()
    net.compute_dtype = dtype
    n = int(g['batch'])
    if str(g['dropout']) == 'mask':
        net.dropout_masks = {h: T(unpack(g['dropmask_%d' % i], (n, 512, 60, 60))) for i, h in enumerate(['det', 'loc'])}
    else:
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    x, _, _, _ = synth.synth_batch(int(g['n_patch']), seed=int(g['seed']))
    return g, kind, net, n, x


def _step(g, kind, net, n, x, step):
    sl = slice(step * n, (step + 1) * n)
    outs = net(x[sl].cuda())
    p = 's%d_' % step
    neg0 = g[p + 'neg_idx_0']
    half = neg0.shape[1] // 2
    lm_rand = None if kind == 'DenseBox' else np.stack([g[p + 'neg_idx_%d' % (1 + j)][:, 1:] for j in range(4)])
    kw = {k[3:]: float(g[k]) for k in g.files if k.startswith('kw_')}
    loss = net.loss(outs, g['bbox'][sl], g['vert'][sl], g['lab'][sl], rand_neg_indices=neg0[:, half:],
                    lm_rand_neg_indices=lm_rand, **kw)
    return outs, loss


def _check_grads(g, net, step):
    """Every layer above the last max-pool on its gradient path (conv4_*, heads, refine) must match element-wise
    to fp32 round-off.  Below a pool the gradient is discontinuous in the activations: a 2x2 window whose two
    largest values differ in the 7th digit can route its gradient to the other pixel on the GPU (observed: 2 of
    460k windows), so there the check is the relative L2 error of the whole tensor."""
    p = 's%d_' % step
    for name, prm in net.named_parameters():
        if p + 'gnone_' + name in g.files:
            assert prm.grad is None, name          # conv3_3 is never executed (DenseBox.py:193-195)
            continue
        gr = prm.grad.detach().float().cpu().numpy()
        stat = g[p + 'gstat_' + name]
        l1 = np.abs(gr.reshape(-1).astype(np.float64)).sum()
        assert np.isclose(l1, stat[1], rtol=2e-3), (name, l1, stat[1])
        if p + 'g_' + name in g.files:
            ref, got = g[p + 'g_' + name], gr
        else:
            ref, got = g[p + 'gsub_' + name], gr.reshape(-1)[::997]
        ref64, got64 = ref.reshape(-1).astype(np.float64), got.reshape(-1).astype(np.float64)
        rel_l2 = np.linalg.norm(got64 - ref64) / max(np.linalg.norm(ref64), 1e-300)
        rel_max = np.abs(got64 - ref64).max() / max(np.abs(ref64).max(), 1e-300)
        below_pool = name.startswith(('conv1_', 'conv2_', 'conv3_'))
        if below_pool:
            assert rel_l2 <= 5e-3 and rel_max <= 3e-2, (name, rel_l2, rel_max)
        else:
            assert rel_max <= 2e-4, (name, rel_max)


@pytest.mark.parametrize('name', ['train_DenseBox', 'train_DenseBox_dropout', 'train_DenseBoxLM', 'train_DenseBoxLMLOC'])
def test_training_step_f32_vs_reference(golden, name):
    g, kind, net, n, x = _setup(golden, name, 'f32')
    outs, loss = _step(g, kind, net, n, x, 0)
    for i, o in enumerate(outs):
        ref = g['s0_out_%d' % i]
        assert np.allclose(o.detach().cpu().numpy(), ref, rtol=0, atol=1e-4 * max(1.0, np.abs(ref).max()))
    assert np.isclose(float(loss.detach()), float(g['s0_loss']), rtol=1e-4)
    loss.backward()
    _check_grads(g, net, 0)


def test_two_sgd_steps_f32_vs_reference(golden):
    """forward/backward/step twice: momentum buffer, weight decay and the version-counter invalidation of the
    packed-weight cache are all on the path (DenseBox.py:2001-2004, :2186-2187)."""
    g, kind, net, n, x = _setup(golden, 'train_DenseBox', 'f32')
    opt = SGD(net.parameters(), lr=float(g['lr']), momentum=0.9, weight_decay=5e-8)
    for step in range(int(g['n_steps'])):
        opt.zero_grad()
        outs, loss = _step(g, kind, net, n, x, step)
        assert np.isclose(float(loss.detach()), float(g['s%d_loss' % step]), rtol=2e-3), step
        loss.backward()
        opt.step()
        for pname, prm in net.named_parameters():
            key = 's%d_pa_%s' % (step, pname)
            if key in g.files:
                ref, before = g[key], g['s%d_pb_%s' % (step, pname)]
                upd = np.abs(ref - before).max()
                assert np.abs(prm.detach().cpu().numpy() - ref).max() <= 5e-3 * upd + 1e-9, (pname, step)
    assert [adjust_LR(opt, e) for e in (0, 5, 10, 15)] == [1e-9, 2e-9, 4e-9, 1e-9]
    assert opt.param_groups[0]['lr'] == 1e-9


def test_sgd_kernel_exact_against_captured_grads(golden):
    """The fused SGD kernel alone, fed the reference's captured gradients: parameters must match to 1 ulp-ish."""
    g = golden('train_DenseBox')
    names = ['conv1_1_1.weight', 'conv5_2_det.weight', 'conv5_2_loc.bias']
    ps = [torch.nn.Parameter(T(g['s0_pb_' + n_]).cuda()) for n_ in names]
    opt = SGD(ps, lr=float(g['lr']), momentum=0.9, weight_decay=5e-8)
    for step in range(int(g['n_steps'])):
        for p, n_ in zip(ps, names):
            p.grad = T(g['s%d_g_%s' % (step, n_)]).cuda()
        v0 = ps[0]._version
        opt.step()
        assert ps[0]._version > v0
        for p, n_ in zip(ps, names):
            ref = g['s%d_pa_%s' % (step, n_)]
            assert np.allclose(p.detach().cpu().numpy(), ref, rtol=1e-6, atol=1e-9), (n_, step)


def _oracle_step(kind, P, x, g, sl, step, masks=None, hard=None, lm_hard=None):
    """The CPU oracle's training step (test infrastructure; pinned to the reference by tests/test_oracle_golden.py) on the fixture's patches
    with the fixture's random negatives: (loss, {param: gradient}, loss_step's result dict)."""
    from oracle import densebox_oracle as O
    p = 's%d_' % step
    neg0 = g[p + 'neg_idx_0']
    half = neg0.shape[1] // 2
    lm_rand = None if kind == 'DenseBox' else np.stack([g[p + 'neg_idx_%d' % (1 + j)][:, 1:] for j in range(4)])
    kw = {k[3:]: float(g[k]) for k in g.files if k.startswith('kw_')}
    for v in P.values():
        v.grad = None
    outs = O.forward(kind, P, x[sl], dropout_masks=masks)
    res = O.loss_step(kind, outs, g['bbox'][sl], g['vert'][sl], g['lab'][sl], rand_neg=neg0[:, half:], lm_rand_neg=lm_rand,
                      hard_neg=hard, lm_hard_neg=lm_hard, **kw)
    res['loss'].backward()
    return float(res['loss'].detach()), {k: v.grad.numpy().copy() for k, v in P.items() if v.grad is not None}, res


def _cpu_params(net):
    return {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.named_parameters()}


def _hash_masks(eng, kind):
    """The keep masks the forward's counter-based hash drew, read back from the hidden map (a dropped element is exactly 0)."""
    from densebox_amd.engine import _HEADS
    keep = (eng.read_activation('hid') != 0)
    frac = keep.float().mean().item()
    assert 0.49 < frac < 0.51, frac
    return {h: keep[:, 512 * i:512 * (i + 1)].float().cpu() for i, (h, _) in enumerate(_HEADS[kind])}


# Per-tensor bars of the 16-bit training step against the fp32 reference step: relative L2 error of every parameter gradient and
# | ||g|| / ||g_ref|| - 1 | (a scale error, or a wrong tenth of one gradient, fails both; a cosine of 0.995 -- round 5's check -- passes them).
# The error grows with the depth of the backward chain (every gradient map is rounded to 16 bits, ReLU gates and 2x2 arg-maxes of near-ties
# flip), so the bars are per layer group: measured on this fixture (profiles/r06_grad_errors.txt) + ~30 %.  (rel-L2 f16, bf16), prefix match.
# (heads: 0.3-3e-3 on most draws of the dropout mask; a near-tie of the refine branch's 2x2 arg-max over the score / landmark maps routes a
#  whole pixel's gradient elsewhere and gave 1e-2 on the det head in one of three runs -- hence the same 1.5e-2 as the conv4 group)
LOWP_BARS = [(('conv5_', 'conv6_'), 1.5e-2, 5e-2), (('conv4_',), 1.5e-2, 5e-2), (('conv3_', 'conv2_2'), 1.8e-2, 5.5e-2),
             (('conv2_1',), 2.1e-2, 6e-2), (('conv1_2',), 3.6e-2, 1.1e-1), (('conv1_1',), 6.5e-2, 2.5e-1)]
LOWP_NORM = {'f16': 5e-3, 'bf16': 3.5e-2}        # (measured <= 3.2e-3 / 2.5e-2)


def _lowp_bar(name, dtype):
    for pre, f16, bf16 in LOWP_BARS:
        if name.startswith(pre):
            return f16 if dtype == 'f16' else bf16
    raise KeyError(name)


@pytest.mark.parametrize('dropout', ['hash', 'off'])
@pytest.mark.parametrize('dtype', ['f16', 'bf16'])
def test_training_step_16bit_default_path_vs_oracle_per_tensor(golden, dtype, dropout):
    """The step bench.py times -- 16-bit, heads backward by linearity (lin_bwd), hidden gradient generated in its consumers (heads_gen),
    hash dropout -- against the fp32 oracle on the same patches, the same random negatives and the SAME dropout mask (exported from the
    hidden map): every gradient within LOWP_BARS.  'off': Dropout disabled, the reference-captured gradients themselves are the bar's
    reference.  The hard negatives are the 16-bit step's own selection (its top-k over its own losses; the rule is pinned bit-exactly in
    test_hip_loss.py) -- at most a few of them differ from the fp32 selection, asserted here."""
    g, kind, net, n, x = _setup(golden, 'train_DenseBoxLMLOC', dtype)
    P = _cpu_params(net)
    if dropout == 'hash':
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.5
        net.dropout_masks = None
        torch.manual_seed(4321)                       # (the hash stream starts from torch's seed: one fixed draw of the mask)
        net.engine()._seed_state = None
    sl = slice(0, n)
    outs = net(x[sl].cuda())
    neg0 = g['s0_neg_idx_0']
    half = neg0.shape[1] // 2
    lm_rand = np.stack([g['s0_neg_idx_%d' % (1 + j)][:, 1:] for j in range(4)])
    kw = {k[3:]: float(g[k]) for k in g.files if k.startswith('kw_')}
    loss, dbg = net.loss(outs, g['bbox'][sl], g['vert'][sl], g['lab'][sl], rand_neg_indices=neg0[:, half:], lm_rand_neg_indices=lm_rand,
                         return_debug=True, **kw)
    loss.backward()
    eng = net.engine()
    Pl = eng.last_plan
    assert Pl.heads_gen and 'd_hid' not in Pl.B and 'd_ups' not in Pl.B and Pl.drop_hash == (dropout == 'hash')
    masks = _hash_masks(eng, kind) if dropout == 'hash' else None
    hard = dbg['neg_idx'][:, :half].cpu().numpy()
    lm_hard = dbg['lm_neg_idx'][:, :, :1].cpu().numpy()
    lo, go, res = _oracle_step(kind, P, x, g, sl, 0, masks=masks, hard=hard, lm_hard=lm_hard)
    # the 16-bit selection against the oracle's own: same set up to near-ties
    differ = sum(len(set(a) ^ set(b)) // 2 for a, b in zip(hard.tolist(), res['hard_own'].tolist()))
    assert differ <= max(2, hard.size // 20), (differ, hard.size)
    assert abs(float(loss.detach()) - lo) <= (2e-3 if dtype == 'f16' else 1.5e-2) * abs(lo), (float(loss.detach()), lo)
    bar_norm = LOWP_NORM[dtype]
    worst, bad = (0.0, 0.0), []
    for name, prm in net.named_parameters():
        if name not in go:
            assert prm.grad is None, name
            continue
        ref = go[name].reshape(-1).astype(np.float64)
        got = prm.grad.detach().float().cpu().numpy().reshape(-1).astype(np.float64)
        assert np.isfinite(got).all(), name
        nr = np.linalg.norm(ref)
        if nr == 0:
            assert np.linalg.norm(got) == 0, name
            continue
        rel = np.linalg.norm(got - ref) / nr
        ratio = abs(np.linalg.norm(got) / nr - 1.0)
        worst = (max(worst[0], rel), max(worst[1], ratio))
        if os.environ.get('DBX_PRINT_GRAD_ERR') == '1':
            print('\n%-5s %-5s %-24s rel_l2 %.3e  norm-1 %.3e' % (dtype, dropout, name, rel, ratio), end='')
        if not (rel <= _lowp_bar(name, dtype) and ratio <= bar_norm):
            bad.append((name, rel, ratio))
    assert not bad, bad
    if dropout == 'off':
        # ... and against the reference-captured gradients themselves (the oracle is pinned to them at 1e-6; this closes the loop on the GPU box)
        for name, prm in net.named_parameters():
            key = 's0_g_' + name
            if key in g.files and np.abs(g[key]).max() > 0:
                ref = g[key].reshape(-1).astype(np.float64)
                got = prm.grad.detach().float().cpu().numpy().reshape(-1).astype(np.float64)
                assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 1.2 * _lowp_bar(name, dtype), name
    print('worst rel-L2 / norm ratio', dtype, dropout, worst)


@pytest.mark.parametrize('name', ['train_DenseBox', 'train_DenseBox_dropout', 'train_DenseBoxLM', 'train_DenseBoxLMLOC'])
def test_training_step_f32_with_the_16bit_backward_structure_vs_reference(golden, name, monkeypatch):
    """DBX_F32_LIN=1: the exact-fp32 kernels run the backward STRUCTURE of the benchmarked 16-bit step -- conv4_4's part of the heads' first
    convs on conv4_4's own grid after one transposed up-sampling of the hidden gradient (dbx_head2_backward_up, wgrad column slices, the two
    data-gradient parts) and, without an injected mask, the hidden gradient GENERATED inside its two 60x60 consumers (the fp32 reference
    instantiations of dbx_heads1_wgrad_gen / dbx_heads1_dgrad_gen; the stored d_hid is poisoned with NaN behind its producer) -- against
    the gradients captured from the reference's own training loop at the plain fp32 path's tolerance.  Pins the algebra and the plumbing
    (slices, scale, seeds, W2 pointers, per-head k) of what bench.py times to the reference."""
    monkeypatch.setenv('DBX_F32_LIN', '1')
    g, kind, net, n, x = _setup(golden, name, 'f32')
    outs, loss = _step(g, kind, net, n, x, 0)
    assert np.isclose(float(loss.detach()), float(g['s0_loss']), rtol=1e-4)
    loss.backward()
    Pl = net.engine().last_plan
    assert Pl.flags[0] and Pl.flags[4] and 'd_ups' not in Pl.B and Pl.heads_gen
    if str(g['dropout']) != 'mask':
        bh = Pl.B['d_hid']
        assert bool(torch.isnan(Pl.ws[bh.off:bh.off + 4096].view(torch.float32)).all())      # generated, not read: the buffer is poison
    _check_grads(g, net, 0)


def test_training_step_f32_generated_structure_with_hash_dropout_vs_oracle(golden, monkeypatch):
    """The same structure with the forward's hash dropout on (the seed / scale / keep-bit plumbing of the generators): fp32 kernels against
    the oracle fed the exported mask, element-wise at fp32 tolerance."""
    monkeypatch.setenv('DBX_F32_LIN', '1')
    g, kind, net, n, x = _setup(golden, 'train_DenseBoxLMLOC', 'f32')
    P = _cpu_params(net)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.5
    net.dropout_masks = None
    torch.manual_seed(4321)
    net.engine()._seed_state = None
    outs, loss = _step(g, kind, net, n, x, 0)
    loss.backward()
    eng = net.engine()
    assert eng.last_plan.drop_hash and eng.last_plan.heads_gen
    lo, go, _ = _oracle_step(kind, P, x, g, slice(0, n), 0, masks=_hash_masks(eng, kind))
    assert abs(float(loss.detach()) - lo) <= 1e-4 * abs(lo)
    for name, prm in net.named_parameters():
        if name not in go:
            assert prm.grad is None, name
            continue
        ref, got = go[name].reshape(-1).astype(np.float64), prm.grad.detach().float().cpu().numpy().reshape(-1).astype(np.float64)
        rel_l2 = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300)
        rel_max = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300)
        if name.startswith(('conv1_', 'conv2_', 'conv3_')):          # below a max-pool: see _check_grads
            assert rel_l2 <= 5e-3 and rel_max <= 3e-2, (name, rel_l2, rel_max)
        else:
            assert rel_max <= 6e-4, (name, rel_max)       # (2e-4 against the captured fixtures; the dropout scale of 2 doubles the summands: 3.6e-4 measured)


def test_dataparallel_world1_equals_plain_autograd(golden):
    """The data-parallel step (flat gradient buffer, readiness callbacks, global mining constants) with one rank must
    reproduce the plain autograd path bit-for-bit (same kernels, same order)."""
    from densebox_amd.dist import DataParallel
    g, kind, net, n, x = _setup(golden, 'train_DenseBoxLMLOC', 'f32')
    outs, loss = _step(g, kind, net, n, x, 0)
    loss.backward()
    ref = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    opt = SGD(net.parameters(), lr=float(g['lr']))
    dp = DataParallel(net, opt)
    neg0 = g['s0_neg_idx_0']
    half = neg0.shape[1] // 2
    lm_rand = np.stack([g['s0_neg_idx_%d' % (1 + j)][:, 1:] for j in range(4)])
    l2 = dp.step(x[:n].cuda(), g['bbox'][:n], g['vert'][:n], g['lab'][:n], rand_neg_indices=neg0[:, half:],
                 lm_rand_neg_indices=lm_rand)
    assert float(l2.detach()) == float(loss.detach())
    for k, p in net.named_parameters():
        if k in ref:
            assert torch.equal(p.grad, ref[k]), k
            assert not torch.equal(p.detach(), before[k]), k          # SGD moved it
        else:
            assert p.grad is None and torch.equal(p.detach(), before[k])   # conv3_3 untouched (DenseBox.py:193-195)
    net.engine().grad_sink = None


def test_hash_dropout_equals_injected_mask(golden):
    """Training-mode Dropout without a mask buffer: the keep bits are a counter-based hash evaluated in the forward
    epilogue and again in backward.  Recover the mask from the hidden activations, inject it through the buffer path
    (which the reference-captured dropout fixture pins) and require identical outputs and gradients."""
    g, kind, net, n, x = _setup(golden, 'train_DenseBoxLM', 'f32')
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.5
    net.dropout_masks = None
    outs, loss = _step(g, kind, net, n, x, 0)
    loss.backward()
    eng = net.engine()
    assert eng.last_plan.drop_hash
    hid = eng.read_activation('hid').clone()
    g_hash = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    o_hash = [o.detach().clone() for o in outs]
    keep = (hid != 0)
    frac = keep.float().mean().item()
    assert 0.49 < frac < 0.51, frac                          # p = 0.5
    # second forward draws a different mask
    outs2, _ = _step(g, kind, net, n, x, 0)
    assert not torch.equal(outs2[0], o_hash[0])
    # inject the recovered mask: buffer path
    for p in net.parameters():
        p.grad = None
    net.dropout_masks = {h: keep[:, 512 * i:512 * (i + 1)].to(torch.uint8) for i, h in enumerate(['det', 'loc', 'landmark'])}
    outs3, loss3 = _step(g, kind, net, n, x, 0)
    assert not eng.last_plan.drop_hash
    for a, b in zip(outs3, o_hash):
        assert torch.equal(a.detach(), b)
    loss3.backward()
    for k, p in net.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, g_hash[k]), k


@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
def test_fused_first_layer_gradient_equals_two_kernel_path(golden, dtype, monkeypatch):
    """The 16-bit training step folds conv1_1's weight gradient into conv1_2's data-gradient kernel (dbx_conv_dgrad_wgrad1) and
    pool1 into conv1_2's forward.  With DBX_FUSE_WG1=0 the step runs the separate data-gradient and weight-gradient kernels:
    every other gradient must be bitwise identical, conv1_1's equal up to the fp32 summation order of its 3.7 M-pixel sum."""
    g, kind, net, n, x = _setup(golden, 'train_DenseBoxLMLOC', dtype)

    def grads(fuse):
        monkeypatch.setenv('DBX_FUSE_WG1', fuse)
        for p in net.parameters():
            p.grad = None
        _, loss = _step(g, kind, net, n, x, 0)
        loss.backward()
        return float(loss.detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    la, ga = grads('1')
    lb, gb = grads('0')
    assert la == lb
    for k in ga:
        if k.startswith('conv1_1_1'):
            rel = float((ga[k].double() - gb[k].double()).norm() / (gb[k].double().norm() + 1e-30))
            assert rel <= 2e-5, (k, rel)
        else:
            assert torch.equal(ga[k], gb[k]), k


def test_eval_and_train_forward_agree_through_fused_pool(golden, monkeypatch):
    """Inference and training both skip the full-resolution conv1_2 map (dbx_conv_forward_pool_idx writes the pooled output, training
    also the arg-max nibbles its pooling backward reads); DBX_POOL_IDX=0 brings the map and the activation-reading backward back.
    The pooled activation that everything downstream reads is the same bit for bit in all three."""
    if os.environ.get('DBX_POOL_IDX') == '0':
        pytest.skip('this process runs without the arg-max nibbles: the default plan this test starts from does not exist')
    g, kind, net, n, x = _setup(golden, 'train_DenseBoxLMLOC', 'f16')
    xs = x[:n].cuda()
    eng = net.engine()
    with torch.no_grad():
        net(xs)
    p_train = eng.read_activation('p1').clone()
    # no full-resolution conv1_2 map exists in this plan (its name aliases conv1_1's output, which the fused call never writes)
    assert eng.last_plan.pool_idx is not None and eng.last_plan.a12_alias and eng.last_plan.B['a12'] is eng.last_plan.B['a11']
    monkeypatch.setenv('DBX_POOL_IDX', '0')
    eng.plans = {}
    with torch.no_grad():
        net(xs)
    assert eng.last_plan.pool_idx is None and not eng.last_plan.a12_alias and float(eng.read_activation('a12').abs().sum()) > 0
    assert torch.equal(p_train, eng.read_activation('p1'))
    monkeypatch.delenv('DBX_POOL_IDX')
    eng.plans = {}
    net.eval()
    with torch.no_grad():
        net(xs)
    p_eval = eng.read_activation('p1').clone()
    assert eng.last_plan.a12_alias
    assert torch.equal(p_train, p_eval) and float(p_eval.abs().sum()) > 0


@pytest.mark.parametrize('dtype', ['f32', 'f16'])
def test_pool_backward_from_argmax_nibbles_equals_activation_reading_backward(golden, dtype, monkeypatch):
    """A training step with the arg-max nibbles (default) against DBX_POOL_IDX=0 (pooling backward re-reads the activations, conv1_2
    writes its full-resolution map): same loss, every gradient bitwise equal -- the nibbles of the fused conv1_2 + pool1 kernel are
    taken from the rounded values, exactly what the stored map would give."""
    g, kind, net, n, x = _setup(golden, 'train_DenseBoxLMLOC', dtype)

    def grads(flag):
        monkeypatch.setenv('DBX_POOL_IDX', flag)
        net.engine().plans = {}
        for p in net.parameters():
            p.grad = None
        _, loss = _step(g, kind, net, n, x, 0)
        loss.backward()
        return float(loss.detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    la, ga = grads('1')
    lb, gb = grads('0')
    assert la == lb and set(ga) == set(gb)
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


@pytest.mark.parametrize('dtype', ['f32', 'f16'])
def test_refine_branch_by_linearity_equals_the_three_convs(golden, dtype, monkeypatch):
    """Training step with the refine branch computed by its linear structure (default: one folded 7x7 conv forward, dbx_refine_backward)
    against DBX_REFINE_LINEAR=0 (conv6_1 / conv6_2 / up-sampling / conv6_3 and their autograd-order backward on the MFMA kernels): same
    loss and gradients up to fp32 summation order in f32; in f16 the new path has no 16-bit rounding inside the branch, so the
    conv6_x gradients agree to the rounding of the old path."""
    g, kind, net, n, x = _setup(golden, 'train_DenseBoxLMLOC', dtype)

    def grads(flag):
        monkeypatch.setenv('DBX_REFINE_LINEAR', flag)
        net.engine().plans = {}
        for p in net.parameters():
            p.grad = None
        _, loss = _step(g, kind, net, n, x, 0)
        loss.backward()
        return float(loss.detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    la, ga = grads('1')
    lb, gb = grads('0')
    tol = 2e-5 if dtype == 'f32' else 2e-2
    assert abs(la - lb) <= (1e-6 if dtype == 'f32' else 2e-3) * abs(lb), (la, lb)
    assert set(ga) == set(gb)
    for k in ga:
        rel = float((ga[k].double() - gb[k].double()).norm() / (gb[k].double().norm() + 1e-30))
        assert rel <= tol, (k, rel)


@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
def test_heads_backward_with_generated_hidden_gradient_equals_the_in_memory_form(golden, dtype, monkeypatch):
    """Default 16-bit training plans hold no hidden-gradient buffer: dbx_head2_backward_up leaves d_hid out and its two 60x60 consumers
    generate it (dbx_heads1_wgrad_gen / dbx_heads1_dgrad_gen).  Against DBX_HEADS_GEN=0 (d_hid written and read back), same hash-dropout
    seed: identical loss (the forward is the same), gradients equal up to the rounding of W2 to the compute dtype inside the
    generating MFMAs (the in-memory form multiplies by the fp32 W2)."""
    g, kind, net, n, x = _setup(golden, 'train_DenseBoxLMLOC', dtype)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.5
    net.dropout_masks = None

    def grads(flag):
        monkeypatch.setenv('DBX_HEADS_GEN', flag)
        eng = net.engine()
        eng.plans = {}
        torch.manual_seed(1234)
        eng._seed_state = None
        for p in net.parameters():
            p.grad = None
        _, loss = _step(g, kind, net, n, x, 0)
        loss.backward()
        P = eng.last_plan
        assert P.drop_hash and P.heads_gen == (flag == '1') and ('d_hid' in P.B) == (flag == '0')
        return float(loss.detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    la, ga = grads('1')
    lb, gb = grads('0')
    assert la == lb and set(ga) == set(gb)
    tol = 6e-3 if dtype == 'f16' else 4e-2
    worst = 0.0
    for k in ga:
        rel = float((ga[k].double() - gb[k].double()).norm() / (gb[k].double().norm() + 1e-30))
        worst = max(worst, rel)
        assert rel <= tol, (k, rel)
    assert worst > 0            # (the two forms are different kernels)


@pytest.mark.parametrize('dtype', ['f16', 'bf16', 'f32'])
@pytest.mark.parametrize('name', ['train_DenseBoxLMLOC', 'train_DenseBox'])
def test_sgd_with_repacking_folded_in_equals_update_then_pack(golden, name, dtype, monkeypatch):
    """optim.SGD.step() on an engine's parameters is ONE launch (dbx_sgd_pack_step: the update and every packed image of every parameter);
    DBX_SGD_PACK=0 keeps dbx_sgd_step + dbx_pack_multi.  Three steps each way from the same start (momentum on, a weight decay large
    enough to show): losses, parameters, momentum buffers and every packed weight / bias image must be bitwise equal, the fused run must
    not launch the re-packing (the engine's signature is current after step()), and a parameter changed behind the engine's back falls
    back to the regular path."""
    runs = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('DBX_SGD_PACK', mode)
        g, kind, net, n, x = _setup(golden, name, dtype)
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        net.dropout_masks = None
        opt = SGD(net.parameters(), lr=2e-9, momentum=0.9, weight_decay=1e-3)
        eng = net.engine()
        losses, fused = [], []
        for it in range(3):
            opt.zero_grad()
            _, loss = _step(g, kind, net, n, x, 0)
            loss.backward()
            sig = eng._wsig
            opt.step()
            fused.append(eng._wsig is not sig)             # the fused step installs the post-update signature
            losses.append(float(loss.detach()))
        # the next forward's weights, as the engine will use them
        with torch.no_grad():
            outs = net(x[:n].cuda())
        torch.cuda.synchronize()
        packed = {str(k): v[1].clone() for k, v in eng.wcache.items() if torch.is_tensor(v[1])}
        packed.update({'b' + str(k): v[1].clone() for k, v in eng.bias_cache.items()})
        runs[mode] = (losses, {k: p.detach().clone() for k, p in net.named_parameters()},
                      [opt.bufs[id(p)].clone() for p in opt.params if id(p) in opt.bufs], packed, [o.clone() for o in outs], fused)
    a, b = runs['1'], runs['0']
    assert all(a[5]) and not any(b[5])
    assert a[0] == b[0] and a[0][0] != a[0][2]
    for k in a[1]:
        assert bool(torch.isfinite(a[1][k]).all()) and torch.equal(a[1][k], b[1][k]), k
    assert len(a[2]) == len(b[2]) and all(torch.equal(u, v) for u, v in zip(a[2], b[2]))
    assert set(a[3]) == set(b[3]) and len(a[3]) > 20
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k
    assert all(torch.equal(u, v) for u, v in zip(a[4], b[4]))
    # a parameter written behind the engine's back: the fused path declines, the regular path re-packs
    monkeypatch.setenv('DBX_SGD_PACK', '1')
    g, kind, net, n, x = _setup(golden, name, dtype)
    opt = SGD(net.parameters(), lr=2e-9, momentum=0.9, weight_decay=1e-3)
    _, loss = _step(g, kind, net, n, x, 0)
    loss.backward()
    with torch.no_grad():
        next(net.parameters()).mul_(1.0)
    sig = net.engine()._wsig
    opt.step()
    assert net.engine()._wsig is sig
    # the engine finds its parameters through a registry on the side: nothing rides on the Parameter objects (they pickle / deep-copy)
    import copy
    import pickle
    p0 = next(net.parameters())
    assert not [k for k in vars(p0) if 'dbx' in k]
    assert torch.equal(pickle.loads(pickle.dumps(p0)).cpu(), p0.detach().cpu()) and torch.equal(copy.deepcopy(p0), p0)


def test_f16_overflow_guard_skips_the_step_and_counts_it(golden, monkeypatch):
    """f16 keeps dL/d(pre-activation) in 16-bit frames and the loss is an un-normalised sum (DenseBox.py:2917).  A step whose residuals
    overflow them (here: lambda_det = 1e9 pushes dL/d(score) past 65504) produces non-finite weight gradients; under the guard
    (dbx_grad_guard + dbx_sgd_pack_step_guarded, on by default for f16 steps of dist.DataParallel) it changes NOTHING -- parameters, momentum
    buffers, packed weights -- and is counted on the device; the next, sane step updates as usual and equals the same step taken from the same
    state without the bad one in between.  bf16 / fp32 steps run the plain update (fp32's range)."""
    from densebox_amd.dist import DataParallel

    def fresh():
        g, kind, net, n, x = _setup(golden, 'train_DenseBoxLMLOC', 'f16')
        opt = SGD(net.parameters(), lr=1e-7, momentum=0.9, weight_decay=5e-8)
        return g, kind, net, n, x, opt, DataParallel(net, opt)

    def dp_step(dp, g, n, x, **kw):
        neg0 = g['s0_neg_idx_0']
        half = neg0.shape[1] // 2
        lm_rand = np.stack([g['s0_neg_idx_%d' % (1 + j)][:, 1:] for j in range(4)])
        return dp.step(x[:n].cuda(), g['bbox'][:n], g['vert'][:n], g['lab'][:n], rand_neg_indices=neg0[:, half:], lm_rand_neg_indices=lm_rand, **kw)

    g, kind, net, n, x, opt, dp = fresh()
    assert dp.guard_f16
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    dp_step(dp, g, n, x, lambda_det=1e9)
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(dp.reducer.flat).all())            # the overflow reached the weight gradients
    assert dp.skipped_steps() == 1
    for k, p in net.named_parameters():
        assert torch.equal(p.detach(), before[k]), k
    assert all(float(b.abs().max()) == 0.0 for b in opt.bufs.values())
    l1 = dp_step(dp, g, n, x)                                         # a sane step from the untouched state ...
    torch.cuda.synchronize()
    assert dp.skipped_steps() == 1 and bool(torch.isfinite(dp.reducer.flat).all())
    g2, kind2, net2, n2, x2, opt2, dp2 = fresh()                      # ... equals the first step of a fresh run, bit for bit
    l2 = dp_step(dp2, g2, n2, x2)
    torch.cuda.synchronize()
    assert float(l1.detach()) == float(l2.detach()) and dp2.skipped_steps() == 0
    moved = 0
    for (k, a), (_, b) in zip(net.named_parameters(), net2.named_parameters()):
        assert torch.equal(a.detach(), b.detach()), k
        moved += int(not torch.equal(a.detach(), before[k]))
    assert moved > 30
    # bf16: the guard is not armed (and DBX_F16_GUARD=0 disarms it for f16)
    net2.compute_dtype = 'bf16'
    dp_step(dp2, g2, n2, x2)
    assert not opt2._guard_on
    net.engine().grad_sink = None
    net2.engine().grad_sink = None
