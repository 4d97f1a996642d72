"""Fused multi-tensor SGD with the semantics of the reference's optimizer
(``torch.optim.SGD(net.parameters(), lr, momentum=0.9, weight_decay=5e-8)``, DenseBox.py:2001-2004)
and its epoch schedule ``adjust_LR`` (DenseBox.py:1345-1365).  One HIP launch updates every tensor -- and, for the parameters of a
network that has run a training step on the HIP engine, also refreshes the packed 16-bit copies its next forward needs."""
import ctypes as C

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


class SGD:
    def __init__(self, params, lr, momentum=0.9, weight_decay=5e-8):
        self.params = [p for p in params]
        self.param_groups = [{'lr': lr, 'momentum': momentum, 'weight_decay': weight_decay, 'params': self.params}]
        self.bufs = {}
        self._table = None
        self._key = None
        # overflow guard of f16 training (dbx_grad_guard): set by enable_guard() -- dist.DataParallel does for f16 steps
        self.guard = None          # int32[2] on the device: [last step id with a non-finite gradient, number of skipped steps]
        self._guard_flat = None
        self._guard_on = False
        self._grads_in_flat = False
        self._step_id = 0

    def enable_guard(self, flat_grads, on=True):
        """Guard the update against non-finite gradients: `flat_grads` is the ONE contiguous fp32 buffer all gradients of a step live in
        (dist.GradReducer.flat).  A guarded step scans it (one pass, ~12 us for 46.7 MB) and, when anything in it is inf / NaN, leaves
        parameters, momentum buffers and packed weights untouched and counts the step in skipped_steps() -- all on the device, no host
        synchronisation.  Momentum buffers are created as zeros (mu * 0 + g == g: the first step's result) so that a skipped FIRST step
        leaves a defined state.  `on=False` keeps the state but runs the plain update (bf16 / fp32 steps: their range is fp32's)."""
        if self.guard is None:
            self.guard = torch.zeros(2, dtype=torch.int32, device=flat_grads.device)
        if self._guard_flat is not flat_grads:
            self._key = None                              # (the "gradients live in the flat buffer" check is part of the cached table)
        self._guard_flat = flat_grads
        self._guard_on = bool(on)

    def skipped_steps(self):
        """Number of guarded steps skipped so far (reads the device counter: synchronises -- call it where the loss is read)."""
        return int(self.guard[1].item()) if self.guard is not None else 0

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None if set_to_none else (p.grad.zero_() if p.grad is not None else None)

    def step(self):
        g = self.param_groups[0]
        live = [p for p in self.params if p.grad is not None]          # conv3_3 never gets one (DenseBox.py:193-195)
        if not live:
            return
        first = [p for p in live if id(p) not in self.bufs]
        guarded = self._guard_on and self._guard_flat is not None
        if first and (len(first) != len(live) or guarded):
            # mixed first/subsequent steps: give the newcomers a zero buffer (mu*0 + g == g)
            for p in first:
                self.bufs[id(p)] = torch.zeros_like(p, dtype=torch.float32)
            first = []
        for p in first:
            self.bufs[id(p)] = torch.empty_like(p, dtype=torch.float32)
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in live]
        key = tuple((p.data_ptr(), gr.data_ptr()) for p, gr in zip(live, grads))
        dev = live[0].device
        if key != self._key:
            # the guard scans the flat buffer: it only stands for this step when every gradient LIVES in it (a plain loss.backward() after
            # DataParallel.close() hands out separate tensors; a stale inf in the flat buffer must not skip that step)
            fl = self._guard_flat
            self._grads_in_flat = fl is not None and all(
                fl.data_ptr() <= gr.data_ptr() and gr.data_ptr() + 4 * gr.numel() <= fl.data_ptr() + 4 * fl.numel() for gr in grads)
            tab = []
            for p, gr in zip(live, grads):
                assert p.dtype == torch.float32 and p.is_contiguous() and gr.dtype == torch.float32
                tab += [p.data_ptr(), gr.data_ptr(), self.bufs[id(p)].data_ptr()]
            self._table = (torch.tensor(tab, dtype=torch.int64, device=dev),
                           torch.tensor([p.numel() for p in live], dtype=torch.int64, device=dev),
                           max(p.numel() for p in live))
            self._key = key
        ptrs, sizes, mx = self._table
        with torch.no_grad():
            gw = None
            if guarded and self._grads_in_flat:
                self._step_id += 1
                gw = self.guard
                fl = self._guard_flat
                check(_lib.lib().dbx_grad_guard(ptr(fl), fl.numel(), ptr(gw), self._step_id, stream_ptr()))
            # parameters of a network whose HIP engine has run a training step: the update and the re-packing of the weights for the
            # next forward are ONE launch (engine.sgd_pack_step -> dbx_sgd_pack_step; same bits as the two launches)
            from .engine import engine_of
            eng = engine_of(live)
            if eng is not None and eng.sgd_pack_step(live, ptrs, g['lr'], g['momentum'], g['weight_decay'], bool(first), gw, self._step_id):
                return
            check(_lib.lib().dbx_sgd_step_guarded(ptr(ptrs), ptr(sizes), len(live), mx, g['lr'], g['momentum'],
                                                  g['weight_decay'], 1 if first else 0, ptr(gw), self._step_id, stream_ptr()))
            # the kernel wrote the parameters behind autograd's back: bump their version counters so that
            # autograd and the engine's packed-weight cache see the update
            torch.autograd.graph.increment_version(live)


def adjust_LR(optimizer, epoch):
    """DenseBox.py:1345-1365: absolute LR per epoch (ignores base_lr)."""
    if epoch < 5:
        lr = 1e-9
    elif epoch < 10:
        lr = 2e-9
    elif epoch < 15:
        lr = 4e-9
    else:
        lr = 1e-9
    for grp in optimizer.param_groups:
        grp['lr'] = lr
    return lr
