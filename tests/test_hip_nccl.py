"""The real RCCL backend on the one GPU a test box has: a world of one rank with collectives forced on (communicator creation,
asynchronous all-reduce on the RCCL stream ordered behind the compute stream, wait() ordering, the gloo side group for the
positive count).  SUM over one rank is the identity, so the step must equal the plain autograd step bit for bit."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _net(kind='DenseBoxLMLOC'):
    import densebox_amd as D
    from densebox_amd import synth
    net = getattr(D, kind)(synth.vgg19_standin(seed=0))
    synth.fill_params_(net, 11)
    net = net.cuda().train()
    net.compute_dtype = 'bf16'
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return net


def test_rccl_world_of_one_step_equals_plain_step():
    from densebox_amd import synth, labels as LB
    from densebox_amd.dist import DataParallel
    from densebox_amd.optim import SGD
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        n = 4
        x, bbox, vert, lab = synth.synth_batch(n, seed=9, neg_frac=0.25)
        _, half = LB.neg_counts(int(LB.positive_count(bbox, lab).sum()), n)
        rn = synth.synth_rand_neg_indices(n, half, seed=1)
        lrn = synth.synth_rand_neg_indices(4 * n, 1, seed=2).reshape(4, n, 1)
        # plain step
        a = _net()
        opt_a = SGD(a.parameters(), lr=1e-8)
        outs = a(x.cuda())
        la = a.loss(outs, bbox, vert, lab, rand_neg_indices=rn, lm_rand_neg_indices=lrn)
        la.backward()
        ga = {k: p.grad.clone() for k, p in a.named_parameters() if p.grad is not None}
        opt_a.step()
        # RCCL step: small buckets so that several asynchronous all-reduces are in flight during backward
        b = _net()
        dp = DataParallel(b, SGD(b.parameters(), lr=1e-8), bucket_bytes=1 << 20, always_reduce=True)
        assert dp.collective and dp.ctl is not None and dist.get_backend() == 'nccl'
        lb = dp.step(x.cuda(), bbox, vert, lab, rand_neg_indices=rn, lm_rand_neg_indices=lrn)
        torch.cuda.synchronize()
        assert float(la.detach()) == float(lb.detach())
        for k, p in b.named_parameters():
            if k in ga:
                assert torch.equal(p.grad, ga[k]), k
        for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            assert torch.equal(pa, pb), k
        # the module is usable the reference way again after close()
        dp.close()
        outs = b(x.cuda())
        b.loss(outs, bbox, vert, lab, rand_neg_indices=rn, lm_rand_neg_indices=lrn).backward()
        assert all(p.grad is not None for k, p in b.named_parameters() if k in ga)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_c_abi_dp_entry_points_world_of_one():
    """dbx_dp_*: the C ABI's own RCCL path (for callers without torch.distributed): id -> communicator -> in-place SUM
    all-reduce of a flat fp32 gradient buffer (identity in a world of one) -> destroy."""
    import ctypes as C
    import torch
    from densebox_amd import _lib
    from densebox_amd._lib import check, ptr, stream_ptr
    L = _lib.lib()
    torch.cuda.set_device(0)
    ident = C.create_string_buffer(128)
    check(L.dbx_dp_unique_id(ident))
    comm = C.c_void_p()
    check(L.dbx_dp_init(ident, 0, 1, C.byref(comm)))
    assert comm.value
    g = torch.randn(1 << 20, device='cuda')
    ref = g.clone()
    for lo in range(0, g.numel(), 1 << 18):                                    # bucketed, like the reducer
        check(L.dbx_dp_allreduce_sum_f32(comm, C.c_void_p(g.data_ptr() + 4 * lo), 1 << 18, stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(g, ref)
    assert L.dbx_dp_init(ident, 1, 1, C.byref(C.c_void_p())) != 0             # rank out of range: error, not a hang
    check(L.dbx_dp_destroy(comm))
