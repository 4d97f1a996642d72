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


def _dp_two_ranks_main(rank, world, id_path, out_path):
    """One of two processes on the ONE GPU of the box driving the C ABI's own RCCL path (no torch.distributed anywhere)."""
    import ctypes as C
    import json
    import threading
    import time
    import torch
    from densebox_amd import _lib
    from densebox_amd._lib import stream_ptr
    res = {'rank': rank}

    def expire():
        res['hung'] = True
        json.dump(res, open(out_path % rank, 'w'))
        os._exit(3)
    t = threading.Timer(90.0, expire)
    t.daemon = True
    t.start()
    torch.cuda.set_device(0)
    torch.zeros(1, device='cuda')
    L = _lib.lib()
    ident = C.create_string_buffer(128)
    if rank == 0:
        rc = L.dbx_dp_unique_id(ident)
        assert rc == 0, L.dbx_last_error()
        with open(id_path + '.tmp', 'wb') as f:
            f.write(ident.raw)
        os.replace(id_path + '.tmp', id_path)
    else:
        for _ in range(600):
            if os.path.exists(id_path):
                break
            time.sleep(0.05)
        ident = C.create_string_buffer(open(id_path, 'rb').read(), 128)
    comm = C.c_void_p()
    rc = L.dbx_dp_init(ident, rank, world, C.byref(comm))
    res['init_rc'] = rc
    if rc != 0:
        res['error'] = L.dbx_last_error().decode()
    else:
        g = torch.full((1 << 18,), float(rank + 1), device='cuda')
        rc2 = L.dbx_dp_allreduce_sum_f32(comm, C.c_void_p(g.data_ptr()), g.numel(), stream_ptr())
        torch.cuda.synchronize()
        res['allreduce_rc'] = rc2
        res['sum_ok'] = bool((g == float(world * (world + 1) // 2)).all())
        L.dbx_dp_destroy(comm)
    t.cancel()
    json.dump(res, open(out_path % rank, 'w'))


def test_c_abi_dp_entry_points_two_ranks_on_one_gpu(tmp_path):
    """dbx_dp_* beyond a world of one, as far as a one-GPU box goes: two processes share cuda:0, exchange the 128-byte id through a file
    and call dbx_dp_init(rank, 2).  RCCL either builds the two-rank communicator on the shared device -- then the in-place SUM all-reduce
    must give 1 + 2 on both ranks -- or rejects it: then BOTH ranks must get a non-zero status with a dbx_last_error() line that names the
    RCCL error, promptly (a hang is killed by a 90-s watchdog inside each rank and reported as xfail: the environment's, not the
    library's).  Which of the two happened is printed.  (The N > 1 run on N devices is the driver's; bench.py --gpus N rides torch's
    "nccl" group, this entry-point family is for callers without torch.distributed.)"""
    import json
    import torch.multiprocessing as mp
    from torch.multiprocessing import ProcessExitedException
    idp, outp = str(tmp_path / 'rccl_id.bin'), str(tmp_path / 'rank%d.json')
    hung = False
    try:
        mp.spawn(_dp_two_ranks_main, args=(2, idp, outp), nprocs=2, join=True)
    except ProcessExitedException as e:
        hung = getattr(e, 'exit_code', None) == 3
        if not hung:
            raise
    res = [json.load(open(outp % r)) for r in range(2) if os.path.exists(outp % r)]
    print('dbx_dp two ranks on one GPU:', res)
    if hung or any(r.get('hung') for r in res):
        pytest.xfail('RCCL did not return from a two-rank init on one device within 90 s: %r' % res)
    assert len(res) == 2
    if all(r['init_rc'] == 0 for r in res):
        assert all(r['allreduce_rc'] == 0 and r['sum_ok'] for r in res), res
    else:
        # a rejected communicator: every rank that failed says why, in the library's one-line form
        for r in res:
            if r['init_rc'] != 0:
                assert 'RCCL error' in r['error'] and 'dp init' in r['error'], r
