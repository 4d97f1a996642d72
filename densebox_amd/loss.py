"""Fused DenseBox loss: the inline loss section of the reference training loops as one call.

  kind 'DenseBox'      train_online            DenseBox.py:2843-2918
  kind 'DenseBoxLM'    train_LM_online         DenseBox.py:2575-2723
  kind 'DenseBoxLMLOC' train_densebox_online   DenseBox.py:2023-2180

One HIP kernel (csrc/loss.hip) builds label maps, L2 terms, mines negatives, fills masks and
gray zones, reduces the weighted sums and writes dL/d(out); the returned scalar is
autograd-connected to the network outputs, so ``loss.backward()`` works as in the reference.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, labels as LB
from ._lib import LossDesc, LossIO, check, ptr, stream_ptr

_KIND_ID = {'DenseBox': 0, 'DenseBoxLM': 1, 'DenseBoxLMLOC': 2}


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loss, n_out, *outs_and_grads):
        outs, grads = outs_and_grads[:n_out], outs_and_grads[n_out:]
        ctx.grads = grads
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        return (None, None) + tuple(d * g for d in ctx.grads) + (None,) * len(ctx.grads)


def densebox_loss(kind, outputs, bbox, vertices=None, labels=None, rand_neg_indices=None, lm_rand_neg_indices=None,
                  lambda_loc=3.0, lambda_det=1.0, lambda_lm=0.5, batch_global=None, positive_num_global=None,
                  return_debug=False, rng=None):
    """outputs: the network's forward tuple (reference order for ``kind``).  bbox [N,4], vertices [N,8],
    labels [N,1] in 60-space (CPU or GPU).  rand_neg_indices [N,half] / lm_rand_neg_indices [4,N,1] are the
    ``np.random.choice`` draws of DenseBox.py:2089-2094 / :2133-2138; when None they are drawn here with
    ``rng`` (a numpy RandomState; default = numpy's global state, like the reference).
    batch_global / positive_num_global: data-parallel runs pass the global batch size and the all-reduced
    positive count so every rank mines with the reference's global ``neg_num`` (DenseBox.py:2074)."""
    kid = _KIND_ID[kind]
    if kind == 'DenseBox':
        score, loc = outputs
        lm = rf = lmloc = None
    elif kind == 'DenseBoxLM':
        score, loc, lm, rf = outputs
        lmloc = None
    else:
        score, rf, loc, lm, lmloc = outputs
    dev = score.device
    n = score.size(0)
    assert tuple(score.shape[1:]) == (1, 60, 60), 'the dense loss is defined on the 60x60 training grid'
    use_labels = kind == 'DenseBoxLMLOC'
    P = int(LB.positive_count(bbox, labels if use_labels else None).sum()) if positive_num_global is None \
        else int(positive_num_global)
    _, half = LB.neg_counts(P, n if batch_global is None else batch_global)
    rs = rng if rng is not None else np.random
    if rand_neg_indices is None:
        rand_neg_indices = np.stack([rs.choice(3600, half, replace=False) for _ in range(n)]) if half else \
            np.zeros((n, 0), np.int64)
    if kid != 0 and lm_rand_neg_indices is None:
        lm_rand_neg_indices = np.stack([np.stack([rs.choice(3600, 1, replace=False) for _ in range(n)])
                                        for _ in range(4)])
    rn = LB._dev(rand_neg_indices, dev, torch.int64).reshape(n, half)
    lrn = LB._dev(lm_rand_neg_indices, dev, torch.int64).reshape(4, n, 1) if kid != 0 else None
    bb = LB._dev(bbox, dev)
    vt = LB._dev(vertices, dev) if kid != 0 else None
    lb = LB._dev(labels, dev) if use_labels else None

    def c32(t):
        return None if t is None else t.detach().to(torch.float32).contiguous()
    o = [c32(t) for t in (score, loc, lm, rf, lmloc)]
    g = [None if t is None else torch.empty_like(t) for t in o]
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    scratch = torch.empty((_lib.lib().dbx_loss_scratch_bytes(n) + 7) // 8, dtype=torch.float64, device=dev)
    dbg = {}
    if return_debug:
        dbg['mask_cls'] = torch.empty((n, 1, 60, 60), dtype=torch.float32, device=dev)
        dbg['neg_idx'] = torch.empty((n, 2 * half), dtype=torch.int64, device=dev)
        dbg['pos_count'] = torch.empty(n, dtype=torch.int32, device=dev)
        if kid != 0:
            dbg['mask_lm'] = torch.empty((n, 4, 60, 60), dtype=torch.float32, device=dev)
            dbg['lm_neg_idx'] = torch.empty((4, n, 2), dtype=torch.int64, device=dev)
    d = LossDesc(kid, n, half, lambda_loc, lambda_det, lambda_lm, 1 if use_labels else 0)
    io = LossIO()
    for name, t in (('bbox', bb), ('vertices', vt), ('labels', lb), ('rand_neg', rn), ('lm_rand_neg', lrn),
                    ('score', o[0]), ('loc', o[1]), ('lm', o[2]), ('rf', o[3]), ('lmloc', o[4]),
                    ('d_score', g[0]), ('d_loc', g[1]), ('d_lm', g[2]), ('d_rf', g[3]), ('d_lmloc', g[4]),
                    ('loss', loss), ('mask_cls', dbg.get('mask_cls')), ('mask_lm', dbg.get('mask_lm')),
                    ('neg_idx', dbg.get('neg_idx')), ('lm_neg_idx', dbg.get('lm_neg_idx')),
                    ('pos_count', dbg.get('pos_count'))):
        setattr(io, name, t.data_ptr() if t is not None else None)
    check(_lib.lib().dbx_loss_forward_backward(C.byref(d), C.byref(io), ptr(scratch), stream_ptr()))
    live = [(t, gg) for t, gg in zip((score, loc, lm, rf, lmloc), g) if t is not None]
    out = _LossFn.apply(loss[0], len(live), *[t for t, _ in live], *[gg for _, gg in live])
    # A caller that owns the whole step (DataParallel.step) may feed dL/d(out) straight to autograd: loss.backward() is
    # torch.autograd.backward(outputs, their gradients x 1.0) -- the five d * g multiplications, the ones() and the clone go away
    out._dbx_direct = ([t for t, _ in live], [gg for _, gg in live])
    if return_debug:
        dbg['half'] = half
        dbg['grads'] = {k: gg for k, gg in zip(('score', 'loc', 'lm', 'rf', 'lmloc'), g) if gg is not None}
        return out, dbg
    return out
