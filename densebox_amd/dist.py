"""Data-parallel training over the GPUs of one node: one process per GPU, gradients all-reduced
(SUM, no division -- the reference loss is a sum over the batch, DenseBox.py:2917) with RCCL over xGMI
through torch.distributed (backend "nccl" is RCCL on ROCm), overlapped with the rest of backward.

The reference has no multi-GPU path; ground truth = its single-process result on the
concatenated global batch.  The one cross-sample coupling is ``neg_num = int(positive_num / N + 0.5)``
(DenseBox.py:2074), which uses the positives of the WHOLE batch: ranks all-reduce one int64 over a
gloo side group (host memory, no GPU sync) and pass the global batch size to the loss.
"""
import contextlib
import json
import os
import sys
import threading
from datetime import timedelta

import torch
import torch.distributed as dist


def rendezvous_timeout_s():
    """Bound on rendezvous and on the first (preflight) collective: DBX_DIST_RENDEZVOUS_S, default 120 s -- the torch default of ten
    minutes turns a mis-configured first multi-GPU run into a silent hang."""
    return float(os.environ.get('DBX_DIST_RENDEZVOUS_S', '120'))


def dist_timeout_s():
    """Bound on every steady-state collective of the process group AND of the gloo control group (DBX_DIST_TIMEOUT_S, default 1800 s =
    torch's own default for gloo).  Deliberately much longer than the rendezvous bound: ranks legitimately drift apart by minutes when
    one of them does rank-0-only work between steps (a checkpoint, an evaluation pass, bench.py's rank-0 inference and CPU-baseline legs);
    a caller whose rank-0-only work can exceed it raises the variable or puts a barrier-free region around that work."""
    return float(os.environ.get('DBX_DIST_TIMEOUT_S', '1800'))


def rccl_info():
    """What a failure report needs to say about the collective set-up (also part of the bench line)."""
    env = {k: v for k, v in os.environ.items()
           if k.startswith(('NCCL_', 'RCCL_', 'HSA_ENABLE_IPC', 'MASTER_', 'DBX_DIST_')) or k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    info = {'backend': dist.get_backend() if dist.is_initialized() else None,
            'world': dist.get_world_size() if dist.is_initialized() else int(os.environ.get('WORLD_SIZE', '1')),
            'timeout_s': dist_timeout_s(), 'rendezvous_timeout_s': rendezvous_timeout_s(), 'env': env,
            'gpus_visible': torch.cuda.device_count() if torch.cuda.is_available() else 0}
    return info


def report_failure(what, exc=None, rank=None):
    """ONE parsable line instead of a hang or a bare traceback: rank 0 prints it on stdout (where the bench line would have been), the
    other ranks on stderr."""
    rank = int(os.environ.get('RANK', '0')) if rank is None else rank
    line = json.dumps({'error': what + ((': %s: %s' % (type(exc).__name__, str(exc)[:600])) if exc is not None else ''),
                       'rank': rank, 'rccl': rccl_info()})
    print(line, file=sys.stdout if rank == 0 else sys.stderr, flush=True)


@contextlib.contextmanager
def watchdog(seconds, what, rank=None):
    """A collective that never completes cannot be interrupted from Python: when `seconds` pass inside the block, report and leave the
    process (exit code 3) so that the launcher tears the job down."""
    def expire():
        report_failure('%s did not complete within %.0f s' % (what, seconds), rank=rank)
        os._exit(3)
    t = threading.Timer(seconds, expire)
    t.daemon = True
    t.start()
    try:
        yield
    finally:
        t.cancel()


def preflight(dev):
    """First collective of the job on the data path's backend, under the watchdog: every rank contributes rank + 1."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    world, rank = dist.get_world_size(), dist.get_rank()
    with watchdog(rendezvous_timeout_s(), 'first all-reduce (%s, world %d)' % (dist.get_backend(), world), rank):
        t = torch.full((1024,), float(rank + 1), device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t)
        if t.is_cuda:
            torch.cuda.synchronize()
        want = world * (world + 1) / 2.0
        if float(t[0]) != want or float(t[-1]) != want:
            raise RuntimeError('first all-reduce returned %r, expected %r' % (float(t[0]), want))


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank).
    The rendezvous is bounded by rendezvous_timeout_s(), the group's collectives afterwards by dist_timeout_s()."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # DBX_DIST_BACKEND=gloo lets several ranks share one GPU (functional testing of the N>1 path on a 1-GPU box)
            backend = os.environ.get('DBX_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        # rendezvous under the SHORT bound (a rank that never shows up fails the job in two minutes); once every rank is in, the group's
        # collectives get the long one (torch keeps ONE timeout per group for both; _set_pg_timeout is its own hook for changing it)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timedelta(seconds=rendezvous_timeout_s()))
        try:
            from torch.distributed.distributed_c10d import _set_pg_timeout
            _set_pg_timeout(timedelta(seconds=dist_timeout_s()))
        except Exception as e:                                   # (a torch without the hook: the short bound stays, and is reported)
            print('densebox_amd.dist: collectives keep the rendezvous timeout of %.0f s (%s)' % (rendezvous_timeout_s(), e), file=sys.stderr)
    return rank, world, local


class GradReducer:
    """Flat fp32 gradient buffer + bucketed asynchronous all-reduce.

    ``order`` is the list of parameter names in the order backward produces them (deepest layers first), so each
    bucket is a contiguous range of the flat buffer that can be sent while shallower layers are still computing.
    """

    def __init__(self, named_params, order, bucket_bytes=8 << 20, group=None, always_reduce=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # always_reduce: run the collectives even in a world of one (exercises the RCCL communicator, its stream and the
        # wait() ordering on a single-GPU box; SUM over one rank is the identity)
        self.collective = self.world > 1 or (always_reduce and dist.is_initialized())
        named = dict(named_params)
        self.order = [n for n in order if n in named]
        dev = next(iter(named.values())).device
        total = sum(named[n].numel() for n in self.order)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views, self.range = {}, {}
        off = 0
        for n in self.order:
            p = named[n]
            self.views[n] = self.flat[off:off + p.numel()].view_as(p)
            self.range[n] = (off, off + p.numel())
            off += p.numel()
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._ready_hi = 0
        self._sent_hi = 0
        self._works = []

    def grad_view(self, name):
        return self.views[name]

    def region(self, names):
        """Contiguous flat range covering consecutive parameters (used for the fused heads' weight gradient)."""
        lo, hi = self.range[names[0]][0], self.range[names[-1]][1]
        assert hi - lo == sum(self.range[n][1] - self.range[n][0] for n in names), 'parameters are not adjacent'
        return self.flat[lo:hi]

    in_step = False          # True while DataParallel.step() drives the backward (autograd then returns no gradients)

    def begin(self):
        self._ready_hi = self._sent_hi = 0
        self._works = []

    def ready(self, names):
        """Called by the engine right after the kernels producing these gradients were enqueued.  Outside DataParallel.step()
        (a plain ``loss.backward()`` on a wrapped module) nothing is sent: there is no begin() / finish() around that backward, and
        autograd hands out clones of the flat views while an in-place collective would still be running on them."""
        if not self.in_step:
            return
        for n in names:
            self._ready_hi = max(self._ready_hi, self.range[n][1])
        if self.collective and self._ready_hi - self._sent_hi >= self.bucket_elems:
            self._send(self._ready_hi)

    def _send(self, hi):
        if hi > self._sent_hi:
            # async_op=True: the RCCL stream waits for the compute stream's work enqueued so far, then runs concurrently
            self._works.append(dist.all_reduce(self.flat[self._sent_hi:hi], op=dist.ReduceOp.SUM, group=self.group,
                                               async_op=True))
            self._sent_hi = hi

    def finish(self):
        if self.collective:
            self._send(self.flat.numel())
            for w in self._works:
                w.wait()          # makes the current stream wait for the collective; does not block the host
        self._works = []


class DataParallel:
    """Wraps a densebox_amd network for data-parallel training.  ``step(...)`` = forward, fused loss with the global
    mining constants, backward with overlapped all-reduce, fused SGD."""

    def __init__(self, net, optimizer, bucket_bytes=8 << 20, always_reduce=False):
        self.net, self.opt = net, optimizer
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.ctl = None
        self.collective = self.world > 1 or (always_reduce and dist.is_initialized())
        if self.collective:
            for p in net.parameters():                       # replicate rank 0's weights
                dist.broadcast(p.data, src=0)
            # tiny host-side control collectives (positive counts) go over gloo: no device sync on the data path
            self.ctl = dist.new_group(backend='gloo', timeout=timedelta(seconds=dist_timeout_s())) if dist.get_backend() != 'gloo' \
                else dist.group.WORLD
        eng = net.engine()
        self.reducer = GradReducer(net.named_parameters(), eng.grad_order(), bucket_bytes, always_reduce=always_reduce)
        eng.grad_sink = self.reducer
        self._pf = None            # a prefetched global positive count: (local count, tensor, work)
        # f16 steps run under the optimizer's overflow guard (optim.SGD.enable_guard; DBX_F16_GUARD=0: off); other types keep the plain update
        self.guard_f16 = os.environ.get('DBX_F16_GUARD', '1') != '0' and hasattr(optimizer, 'enable_guard')

    def close(self):
        """Detach from the network: its engine writes gradients to ordinary tensors again (plain autograd training)."""
        eng = self.net.engine()
        if eng.grad_sink is self.reducer:
            eng.grad_sink = None

    def prefetch_positive_num(self, bbox, labels=None):
        """Start the global positive count (DenseBox.py:2070-2074) of the NEXT step's batch: it depends on the labels only, so its int64
        all-reduce on the gloo control group can fly while this step's kernels drain.  Call it right after step(); the next
        global_positive_num() / step() consumes it.  EVERY rank must prefetch (or none): the pending collective is consumed
        unconditionally, a batch other than the prefetched one raises."""
        from . import labels as LB
        if self._pf is not None:
            self._consume_prefetch()                      # (never two in flight: the collective order stays the call order on every rank)
        p = int(LB.positive_count(bbox, labels).sum())
        if self.collective:
            t = torch.tensor([p], dtype=torch.int64)
            self._pf = (p, t, dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.ctl, async_op=True))
        else:
            self._pf = (p, None, None)

    def drain_prefetch(self):
        """Wait for a pending prefetch and drop its result (end of a run: no collective may be left in flight at destroy time)."""
        if self._pf is not None:
            self._consume_prefetch()

    def _consume_prefetch(self):
        p, t, w = self._pf
        self._pf = None
        if w is not None:
            w.wait()
            return p, int(t.item())
        return p, p

    def global_positive_num(self, bbox, labels=None):
        from . import labels as LB
        p = int(LB.positive_count(bbox, labels).sum())
        if self._pf is not None:
            local, total = self._consume_prefetch()
            if local != p:
                raise RuntimeError('prefetch_positive_num() was given another batch than this step (local positives %d != %d)' % (local, p))
            return total
        if self.collective:
            t = torch.tensor([p], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.ctl)
            p = int(t.item())
        return p

    def skipped_steps(self):
        """f16 steps the overflow guard skipped so far (optim.SGD.skipped_steps; synchronises: call it where the loss is read).  What to do
        about a non-zero count is the caller's policy; the documented fallback is net.compute_dtype = 'bf16' (same MFMA rate, fp32's range)."""
        return self.opt.skipped_steps() if hasattr(self.opt, 'skipped_steps') else 0

    def _stage(self, dev, items):
        """Host-side loss inputs (boxes, vertices, labels, mining draws; ~10 KB) -> device through ONE pinned buffer and one
        asynchronous copy, enqueued BEFORE the forward.  A pageable ``.to(device)`` inside the loss would block the host until
        the whole forward has drained (stream order), and the backward would then start with an empty launch queue."""
        import numpy as np
        host = []
        for t, dt in items:
            if t is None or (torch.is_tensor(t) and t.is_cuda):
                host.append(None)
            else:
                host.append(torch.as_tensor(np.ascontiguousarray(t) if isinstance(t, np.ndarray) else t).to(dt).contiguous())
        nbytes = sum((h.numel() * h.element_size() + 15) // 16 * 16 for h in host if h is not None)
        if nbytes == 0:
            return [t for t, _ in items]
        st = self.__dict__.get('_stage_buf')
        if st is None or st[0].numel() < nbytes:
            st = (torch.empty(2 * nbytes, dtype=torch.uint8).pin_memory(), torch.empty(2 * nbytes, dtype=torch.uint8, device=dev),
                  torch.cuda.Event())
            self._stage_buf = st
        pin, dbuf, ev = st
        ev.synchronize()                                   # the previous step's copy has left the pinned buffer
        out, off = [], 0
        for (t, dt), h in zip(items, host):
            if h is None:
                out.append(t); continue
            nb = h.numel() * h.element_size()
            pin[off:off + nb].view(dt).view(h.shape).copy_(h)
            out.append(dbuf[off:off + nb].view(dt).view(h.shape))
            off += (nb + 15) // 16 * 16
        dbuf[:off].copy_(pin[:off], non_blocking=True)
        ev.record()
        return out

    def step(self, x, bbox, vertices=None, labels=None, rand_neg_indices=None, lm_rand_neg_indices=None,
             positive_num_global=None, **loss_kw):
        net = self.net
        n_global = x.size(0) * self.world
        if positive_num_global is None:
            positive_num_global = self.global_positive_num(bbox, labels if net.KIND == 'DenseBoxLMLOC' else None)
        if x.is_cuda:
            import numpy as np
            from . import labels as LB
            n = x.size(0)
            rs = loss_kw.get('rng') or np.random
            _, half = LB.neg_counts(int(positive_num_global), n_global)
            if rand_neg_indices is None:                   # same draws, same order as densebox_loss would make
                rand_neg_indices = np.stack([rs.choice(3600, half, replace=False) for _ in range(n)]) if half else \
                    np.zeros((n, 0), np.int64)
            if net.KIND != 'DenseBox' and lm_rand_neg_indices is None:
                lm_rand_neg_indices = np.stack([np.stack([rs.choice(3600, 1, replace=False) for _ in range(n)])
                                                for _ in range(4)])
            bbox, vertices, labels, rand_neg_indices, lm_rand_neg_indices = self._stage(
                x.device, [(bbox, torch.float32), (vertices, torch.float32), (labels, torch.float32),
                           (rand_neg_indices, torch.int64), (lm_rand_neg_indices, torch.int64)])
        if self.guard_f16:
            self.opt.enable_guard(self.reducer.flat, on=net.resolved_dtype(True) == 'f16')
        self.opt.zero_grad(set_to_none=True)
        self.reducer.begin()
        net.engine().sample_offset = self.rank * x.size(0)     # ranks draw different dropout masks (engine._next_drop_seed)
        self.reducer.in_step = True
        try:
            outs = net(x)
            loss = net.loss(outs, bbox, vertices, labels, rand_neg_indices, lm_rand_neg_indices,
                            batch_global=n_global, positive_num_global=positive_num_global, **loss_kw)
            direct = getattr(loss, '_dbx_direct', None)
            if direct is not None and all(t.requires_grad for t in direct[0]):
                torch.autograd.backward(direct[0], direct[1])      # == loss.backward() (gradient 1.0), without the scaling kernels
            else:
                loss.backward()
        finally:
            self.reducer.in_step = False
        self.reducer.finish()
        for name, p in net.named_parameters():
            p.grad = self.reducer.views.get(name)          # None for conv3_3 (never executed)
        self.opt.step()
        return loss
