"""GPU data-parallel equivalence: WORLD ranks (2 and 8 = configs[3]'s world size; gloo transport, all on cuda:0 -- RCCL needs one
device per rank, the reducer/step logic is backend-agnostic) each train on their slice of a batch; parameters after the step must
equal the single-process step on the concatenated batch, and every rank must hold the same parameters (sum-loss => all-reduce SUM without division; global
positive count => identical mining constants)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
KIND, N_PER, LR = 'DenseBoxLMLOC', 2, 1e-8


def _data(WORLD):
    from densebox_amd import synth, labels as LB
    N = N_PER * WORLD
    x, bbox, vert, lab = synth.synth_batch(N, seed=31, neg_frac=0.3)
    lab[0, 0] = 1.0 if float(bbox[0].abs().sum()) > 0 else 0.0
    P = int(LB.positive_count(bbox, lab).sum())
    _, half = LB.neg_counts(P, N)
    rn = synth.synth_rand_neg_indices(N, half, seed=5)
    lrn = synth.synth_rand_neg_indices(4 * N, 1, seed=6).reshape(4, N, 1)
    return x, bbox, vert, lab, rn, lrn


def _make_net():
    import densebox_amd as D
    from densebox_amd import synth
    net = getattr(D, KIND)(synth.vgg19_standin(seed=0))
    synth.fill_params_(net, 11)
    net = net.cuda().train()
    net.compute_dtype = 'f32'
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return net


def _rank_main(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if world > 2:
        import time
        time.sleep(0.3 * rank)           # one device context after the other (see tests/test_zz_world8.py)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from densebox_amd.dist import DataParallel
        from densebox_amd.optim import SGD
        x, bbox, vert, lab, rn, lrn = _data(world)
        net = _make_net()
        dp = DataParallel(net, SGD(net.parameters(), lr=LR), bucket_bytes=4 << 20)
        from densebox_amd import labels as LB
        mine = slice(rank * N_PER, (rank + 1) * N_PER)
        assert dp.global_positive_num(bbox[mine], lab[mine]) == int(LB.positive_count(bbox, lab).sum())    # the int64 all-reduce at this world size
        sl = slice(rank * N_PER, (rank + 1) * N_PER)
        loss = dp.step(x[sl].cuda(), bbox[sl], vert[sl], lab[sl], rand_neg_indices=rn[sl], lm_rand_neg_indices=lrn[:, sl])
        tot = loss.detach().double().cpu().clone()
        dist.all_reduce(tot)
        torch.cuda.synchronize()
        # every rank holds the same parameters after the step (bitwise: same all-reduced gradient, same update)
        sums = torch.stack([p.detach().double().sum() for p in net.parameters()]).cpu()
        gathered = [torch.zeros_like(sums) for _ in range(world)]
        dist.all_gather(gathered, sums)
        assert all(torch.equal(g, gathered[0]) for g in gathered), 'ranks diverged after one step'
        if rank == 0:
            torch.save({'loss': float(tot), 'params': {k: p.detach().cpu() for k, p in net.named_parameters()},
                        'grads': {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None}}, out_path)
    finally:
        dist.destroy_process_group()


def run_ranks_equal_one_process(tmp_path, WORLD, attempts=1):
    """Retries ONLY a rank killed by a signal (eight device contexts time-slicing one virtual device died in driver code about once in
    three runs: ProcessExitedException with a signal, no Python exception).  An exception RAISED inside a rank -- the in-rank assertions
    ('ranks diverged after one step', the global positive count) or any error the library reports -- is a real failure and propagates
    at once.  More than one attempt is reported as a warning with the signal of every retried run."""
    import time
    import warnings
    from torch.multiprocessing import ProcessExitedException
    retried = []
    out = None
    for attempt in range(attempts):
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        out = str(tmp_path / ('dp%d.pt' % attempt))
        try:
            mp.spawn(_rank_main, args=(WORLD, port, out), nprocs=WORLD, join=True)
            break
        except ProcessExitedException as e:
            sig = getattr(e, 'signal_name', None)
            if sig is None or attempts == 1:                # a non-zero exit code without a signal is the rank's own doing: no retry
                raise
            retried.append('attempt %d: rank %s killed by %s' % (attempt, getattr(e, 'error_index', '?'), sig))
            out = None
            time.sleep(5.0)                                 # (the dead ranks' device contexts are torn down asynchronously: failures came in streaks)
    if out is None:
        # Measured in round 5 (gpurun_w8.sh: nine runs under three kernel selections, incl. DBX_WS=0 DBX_P8=0 = round 1's LDS kernels
        # only): ~40 % of all eight-rank attempts on ONE virtual GPU die in driver code (HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION or a memory
        # fault in torch's fill kernel) whatever kernels the step uses, while worlds of one and two never do -- the box cannot time-slice
        # eight contexts reliably.  Reported, never counted as a pass.
        pytest.xfail('world-%d run: every one of %d attempts was killed by a signal (%s)' % (WORLD, attempts, '; '.join(retried)))
    if retried:
        warnings.warn('world-%d run needed %d attempts (%s)' % (WORLD, len(retried) + 1, '; '.join(retried)))
    got = torch.load(out)
    # single process, concatenated batch
    from densebox_amd.optim import SGD
    x, bbox, vert, lab, rn, lrn = _data(WORLD)
    net = _make_net()
    opt = SGD(net.parameters(), lr=LR)
    outs = net(x.cuda())
    loss = net.loss(outs, bbox, vert, lab, rand_neg_indices=rn, lm_rand_neg_indices=lrn)
    loss.backward()
    ref_g = {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None}
    opt.step()
    assert np.isclose(got['loss'], float(loss.detach()), rtol=1e-6)
    assert set(got['grads']) == set(ref_g)
    for k, g in ref_g.items():
        scale = float(g.abs().max()) + 1e-30
        assert float((got['grads'][k] - g).abs().max()) <= 2e-5 * scale, k      # fp32 summation order only
    for k, p in net.named_parameters():
        assert torch.allclose(got['params'][k], p.detach().cpu(), rtol=0, atol=1e-6 * float(p.abs().max()) + 1e-12), k


def test_two_ranks_equal_one_process(tmp_path):
    run_ranks_equal_one_process(tmp_path, 2)
