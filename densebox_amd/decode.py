"""Inference tail on the GPU: top-K decode (parse_output / parse_out_MN / parse_DetLM / parse_DetLMLOC,
DenseBox.py:3114-3395) and greedy NMS (DenseBox.py:3398-3443), same names, arguments and results
(float64 ``np.ndarray`` rows in descending-score order, python list of kept row indices)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def _run(score_map, loc_map, M, N, K, lm_heat=None, lm_loc=None, nms_thresh=0.4):
    rows, cols = M // 4, N // 4
    assert score_map.size() == torch.Size([1, 1, rows, cols])
    assert loc_map.size() == torch.Size([1, 4, rows, cols])
    if lm_heat is not None:
        assert lm_heat.size() == torch.Size([1, 4, rows, cols])
    if lm_loc is not None:
        assert lm_loc.size() == torch.Size([1, 8, rows, cols])
    dev = score_map.device if score_map.is_cuda else torch.device('cuda')

    def f(t):
        return None if t is None else t.detach().to(dev, torch.float32).contiguous()
    s, l, hm, ll = f(score_map), f(loc_map), f(lm_heat), f(lm_loc)
    dc = 5 if (hm is None and ll is None) else 13
    dets = torch.empty((K, dc), dtype=torch.float64, device=dev)
    topk = torch.empty(K, dtype=torch.int64, device=dev)
    keep = torch.empty(K + 1, dtype=torch.int32, device=dev)
    L = _lib.lib()
    scratch = torch.empty(L.dbx_detect_scratch_bytes(rows, cols, K), dtype=torch.uint8, device=dev)
    check(L.dbx_detect(ptr(s), ptr(l), ptr(hm), ptr(ll), rows, cols, K, float(nms_thresh), ptr(dets), dc, ptr(topk),
                       ptr(keep), ptr(scratch), stream_ptr()))
    return dets, topk, keep


def parse_out_MN(score_map, loc_map, M, N, K=10):
    return _run(score_map, loc_map, M, N, K)[0].cpu().numpy()


def parse_output(score_map, loc_map, K=10):
    assert score_map.size() == torch.Size([1, 1, 60, 60])
    return parse_out_MN(score_map, loc_map, 240, 240, K)


def parse_DetLM(score_map, loc_map, lm_map, M, N, K=10):
    return _run(score_map, loc_map, M, N, K, lm_heat=lm_map)[0].cpu().numpy()


def parse_DetLMLOC(score_map, bbox_loc_map, lm_heat_map, lm_loc_map, M, N, K=10):
    return _run(score_map, bbox_loc_map, M, N, K, lm_heat=lm_heat_map, lm_loc=lm_loc_map)[0].cpu().numpy()


def NMS(dets, nms_thresh=0.4):
    d = torch.as_tensor(np.ascontiguousarray(dets, dtype=np.float64)).cuda()
    n, dc = d.shape
    keep = torch.empty(n + 1, dtype=torch.int32, device=d.device)
    scratch = torch.empty(5 * n + 16, dtype=torch.uint8, device=d.device)
    check(_lib.lib().dbx_nms(ptr(d), n, dc, float(nms_thresh), ptr(keep), ptr(scratch), stream_ptr()))
    k = keep.cpu().numpy()
    return [int(v) for v in k[1:1 + int(k[0])]]


_MAX_GRAPHS = 8


def _detect_eager(net, image, K, nms_thresh):
    M, N = image.size(2), image.size(3)
    with torch.no_grad():
        outs = net(image)
    kind = net.KIND
    if kind == 'DenseBox':
        dets, _, keep = _run(outs[0], outs[1], M, N, K, nms_thresh=nms_thresh)
    elif kind == 'DenseBoxLM':
        dets, _, keep = _run(outs[3], outs[1], M, N, K, lm_heat=outs[2], nms_thresh=nms_thresh)
    else:
        dets, _, keep = _run(outs[1], outs[2], M, N, K, lm_heat=outs[3], lm_loc=outs[4], nms_thresh=nms_thresh)
    return dets, keep


def detect(net, image, K=10, nms_thresh=0.4):
    """Whole-image forward -> top-K -> decode -> NMS in one go (test / test_lm / test_lmloc drivers,
    DenseBox.py:3788-3799, :3626-3643, :3709-3726): ranks by the refined score for the landmark nets.
    Returns (dets[K, 5|13] float64 ndarray, keep list).

    In eval mode the ~25 launches of one image are captured into a hipGraph per (shape, K, dtype, weight version) and
    replayed (the single-image path is launch-bound: 0.8 ms eager vs the kernels' own time); DBX_GRAPH=0 keeps it eager."""
    import os
    assert image.dim() == 4 and image.size(0) == 1
    use_graph = image.is_cuda and not net.training and os.environ.get('DBX_GRAPH', '1') != '0'
    if not use_graph:
        dets, keep = _detect_eager(net, image, K, nms_thresh)
    else:
        import collections
        cache = net.__dict__.setdefault('_detect_graphs', collections.OrderedDict())
        # weight signature: versions + storage addresses of every parameter (a replay reads the packed copies made at capture).
        # Read straight from the sub-modules' parameter dicts (sees in-place updates, .to()/.half() and replaced Parameter
        # objects; 12 us instead of the 75 us Module.parameters() spends walking the tree); the list of dicts itself is
        # rebuilt every 64 calls in case a whole sub-module was swapped.
        # A replaced sub-module (net.conv6_3_det = nn.Conv2d(...)) changes the DIRECT children's identities: their ids are part
        # of the signature (one dict walk), and the cached list is rebuilt whenever they differ.
        kids = tuple(id(m) for m in net._modules.values())
        pd = net.__dict__.get('_detect_pdicts')
        if pd is None or pd[0] <= 0 or pd[2] != kids:
            pd = [64, [m._parameters for m in net.modules() if m._parameters], kids]
            net.__dict__['_detect_pdicts'] = pd
        pd[0] -= 1
        sig = (kids,) + tuple([(p._version, p.data_ptr()) for d in pd[1] for p in d.values() if p is not None])
        key = (tuple(image.shape), image.dtype, K, float(nms_thresh), net.resolved_dtype(False))
        ent = cache.get(key)
        if ent is None or ent[0] != sig:
            static_in = image.clone()
            for _ in range(2):                       # warm: workspace plan, packed weights, scratch buffers, kernel attributes
                wd, wk = _detect_eager(net, static_in, K, nms_thresh)
            h_dets = torch.empty(wd.shape, dtype=wd.dtype).pin_memory()        # (pinned allocation is not capturable)
            h_keep = torch.empty(wk.shape, dtype=wk.dtype).pin_memory()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                dets, keep = _detect_eager(net, static_in, K, nms_thresh)
                # the two result copies are graph nodes too (pinned destinations): one replay + one stream sync per image
                # instead of two blocking .cpu() calls with their launch round trips (~60 us of idle GPU per image)
                h_dets.copy_(dets, non_blocking=True)
                h_keep.copy_(keep, non_blocking=True)
            # The captured kernels hold RAW pointers into the engine's workspace plan and packed / folded weight buffers.  The
            # engine keeps one plan and re-creates its weight caches when the dtype or mode flips, so the entry pins every
            # tensor it captured: a later forward at another shape (or a train-mode step) cannot free what a replay reads.
            ent = (sig, g, static_in, dets, keep, net._engine.captured_refs(), h_dets, h_keep)
            cache[key] = ent
            while len(cache) > _MAX_GRAPHS:            # bounded: one graph + private pool + pinned workspace per shape
                cache.popitem(last=False)
        cache.move_to_end(key)
        _, g, static_in, dets, keep, _refs, h_dets, h_keep = ent
        static_in.copy_(image)
        g.replay()
        torch.cuda.current_stream().synchronize()
        k = h_keep.numpy()
        return h_dets.numpy().copy(), [int(v) for v in k[1:1 + int(k[0])]]
    k = keep.cpu().numpy()
    return dets.cpu().numpy(), [int(v) for v in k[1:1 + int(k[0])]]
