"""world_size-2 CPU (gloo) tests of the data-parallel layer: flat-buffer bucketed all-reduce, the global positive
count, and the algebra that makes N ranks equal one rank on the concatenated batch (DenseBox.py:2074, :2917)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import densebox_amd as D
from densebox_amd import synth, labels as LB
from densebox_amd.dist import GradReducer, DataParallel
from oracle import densebox_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(fn, world=2):
    port = _free_port()
    mp.spawn(_entry, args=(world, port, fn), nprocs=world, join=True)


def _entry(rank, world, port, fn):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        fn(rank, world)
    finally:
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------- reducer
def _reducer_case(rank, world):
    torch.manual_seed(0)
    named = [('w%d' % i, torch.nn.Parameter(torch.zeros(s))) for i, s in enumerate([(7, 3), (5,), (64, 9), (1,), (33, 2)])]
    order = ['w4', 'w3', 'w2', 'w1', 'w0', 'not_a_param']
    red = GradReducer(named, order, bucket_bytes=256)          # 64-element buckets -> several collectives
    red.begin()
    for name in order[:-1]:
        red.grad_view(name).copy_(torch.full_like(red.grad_view(name), float(rank + 1)) * (1 + int(name[1])))
        red.ready([name])
    red.finish()
    for name in order[:-1]:
        want = sum(r + 1 for r in range(world)) * (1 + int(name[1]))
        assert torch.all(red.grad_view(name) == want), (name, red.grad_view(name).flatten()[:3], want)
    assert red.region(['w4', 'w3']).numel() == 67
    # a second step reuses the buffer
    red.begin()
    red.flat.fill_(float(rank))
    red.ready(order[:-1])
    red.finish()
    assert torch.all(red.flat == sum(range(world)))


def test_grad_reducer_bucketed_allreduce():
    _run(_reducer_case)


# ----------------------------------------------------------------------------------------------- positive count + broadcast
def _count_case(rank, world):
    n = 6
    bbox, vert, lab = synth.synth_labels(n * world, seed=4, neg_frac=0.3)
    net = D.DenseBoxLMLOC(synth.vgg19_standin(seed=rank))        # ranks start from DIFFERENT weights
    synth.fill_params_(net, 100 + rank)
    from densebox_amd.optim import SGD
    dp = DataParallel(net, SGD(net.parameters(), lr=1e-9), bucket_bytes=1 << 20)
    ref = D.DenseBoxLMLOC(synth.vgg19_standin(seed=0))
    synth.fill_params_(ref, 100)
    for (_, a), (_, b) in zip(net.named_parameters(), ref.named_parameters()):
        assert torch.equal(a, b)                                   # broadcast from rank 0
    sl = slice(rank * n, (rank + 1) * n)
    got = dp.global_positive_num(bbox[sl], lab[sl])
    assert got == int(LB.positive_count(bbox, lab).sum())
    # flat gradient layout follows the engine's readiness order and covers every trainable parameter once
    total = sum(p.numel() for nme, p in net.named_parameters() if 'conv3_3' not in nme)
    assert dp.reducer.flat.numel() == total


def test_global_positive_count_and_broadcast():
    _run(_count_case)


def _prefetch_case(rank, world):
    """The next step's positive count in flight while this step runs (it depends on the labels only): the prefetched collective is
    consumed by the next global_positive_num(), two prefetches never overlap, a batch other than the prefetched one raises."""
    n = 5
    bbox, vert, lab = synth.synth_labels(2 * n * world, seed=6, neg_frac=0.3)
    net = D.DenseBoxLMLOC(synth.vgg19_standin(seed=0))
    from densebox_amd.optim import SGD
    dp = DataParallel(net, SGD(net.parameters(), lr=1e-9))
    a = slice(rank * n, (rank + 1) * n)
    b = slice((world + rank) * n, (world + rank + 1) * n)
    want_a = int(LB.positive_count(bbox[:world * n], lab[:world * n]).sum())
    want_b = int(LB.positive_count(bbox[world * n:], lab[world * n:]).sum())
    dp.prefetch_positive_num(bbox[a], lab[a])
    assert dp._pf is not None
    assert dp.global_positive_num(bbox[a], lab[a]) == want_a and dp._pf is None
    dp.prefetch_positive_num(bbox[a], lab[a])
    dp.prefetch_positive_num(bbox[b], lab[b])                      # (the first one is drained, never two in flight)
    assert dp.global_positive_num(bbox[b], lab[b]) == want_b
    assert dp.global_positive_num(bbox[a], lab[a]) == want_a       # nothing pending: the blocking path
    dp.prefetch_positive_num(bbox[a], lab[a])
    assert int(LB.positive_count(bbox[a], lab[a]).sum()) > 0
    with pytest.raises(RuntimeError):                              # twice the rows = twice the local positives: not the prefetched batch
        dp.global_positive_num(torch.cat([bbox[a], bbox[a]]), torch.cat([lab[a], lab[a]]))
    dist.barrier()


def test_prefetched_positive_count():
    _run(_prefetch_case)


# ----------------------------------------------------------------------------------------------- sharded loss == global loss
def _loss_case(rank, world):
    kind, n = 'DenseBoxLMLOC', 3
    N = n * world
    rs = np.random.RandomState(7)
    shapes = [(N, 1), (N, 1), (N, 4), (N, 4), (N, 8)]
    outs = [torch.from_numpy(rs.randn(s[0], s[1], 60, 60).astype(np.float32) * 3) for s in shapes]
    bbox, vert, lab = synth.synth_labels(N, seed=11, neg_frac=0.34)
    P = int(LB.positive_count(bbox, lab).sum())
    _, half = LB.neg_counts(P, N)
    rn = synth.synth_rand_neg_indices(N, half, seed=1).numpy()
    lrn = synth.synth_rand_neg_indices(4 * N, 1, seed=2).reshape(4, N, 1).numpy()
    full = [o.clone().requires_grad_(True) for o in outs]
    res = O.loss_step(kind, tuple(full), bbox.numpy(), vert.numpy(), lab.numpy(), rand_neg=rn, lm_rand_neg=lrn)
    res['loss'].backward()
    sl = slice(rank * n, (rank + 1) * n)
    mine = [o[sl].clone().requires_grad_(True) for o in outs]
    r = O.loss_step(kind, tuple(mine), bbox.numpy()[sl], vert.numpy()[sl], lab.numpy()[sl], rand_neg=rn[sl],
                    lm_rand_neg=lrn[:, sl], batch_global=N, positive_num_global=P)
    assert r['half'] == half
    r['loss'].backward()
    tot = r['loss'].detach().double().clone()
    dist.all_reduce(tot)                                           # SUM, no division: the loss is a sum over patches
    assert torch.isclose(tot, res['loss'].detach().double(), rtol=1e-6)
    for a, b in zip(mine, full):
        assert torch.allclose(a.grad, b.grad[sl], rtol=0, atol=0)   # per-sample terms are independent given (N, P)
    assert np.array_equal(r['neg_idx'], res['neg_idx'][sl])


def test_sharded_loss_equals_global_loss():
    _run(_loss_case)
