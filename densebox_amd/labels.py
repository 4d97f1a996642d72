"""Device implementations of the reference's label-map and loss-mask functions, same names and
argument meaning (DenseBox.py:1368-1933).  Inputs may be CPU or GPU tensors; results are GPU
tensors (fp32 NCHW like the reference's).  Every map is produced by a HIP kernel of
libdensebox_hip.so; the only host arithmetic here is ``positive_count`` (integer bookkeeping
the host needs to size the mining, identical to the kernel's -- cross-checked in tests)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

HW = 60


def _dev(t, device=None, dtype=torch.float32):
    if t is None:
        return None
    t = torch.as_tensor(t)
    if not t.is_cuda:
        t = t.to(device or 'cuda')
    return t.to(dtype).contiguous()


def init_score_map(bbox, batch_size=None, ratio=0.3, labels=None):
    """DenseBox.py:1556-1584; with ``labels`` = init_score (:1587-1624)."""
    assert ratio == 0.3, 'the kernels hard-code the reference ratio 0.3'
    bbox = _dev(bbox)
    n = bbox.size(0)
    assert batch_size is None or batch_size == n
    out = torch.empty((n, 1, HW, HW), dtype=torch.float32, device=bbox.device)
    lab = _dev(labels, bbox.device)
    check(_lib.lib().dbx_init_score_map(ptr(bbox), ptr(lab), n, ptr(out), stream_ptr()))
    return out


def init_score(bboxes, labels, ratio=0.3):
    assert bboxes.size(0) == labels.size(0) and tuple(labels.shape) == (labels.size(0), 1)
    return init_score_map(bboxes, ratio=ratio, labels=labels)


def _offset(coords, c, labels):
    coords = _dev(coords)
    n = coords.size(0)
    assert tuple(coords.shape) == (n, c)
    out = torch.empty((n, c, HW, HW), dtype=torch.float32, device=coords.device)
    lab = _dev(labels, coords.device)       # keep every temporary alive until the launch is enqueued
    check(_lib.lib().dbx_init_offset_map(ptr(coords), ptr(lab), n, c, ptr(out), stream_ptr()))
    return out


def init_loc_map(bboxes, batch_size=None):
    """DenseBox.py:1627-1655: (x-x_lt, y-y_lt, x-x_rb, y-y_rb)."""
    return _offset(bboxes, 4, None)


def init_loc(bboxes, labels):
    """DenseBox.py:1658-1686."""
    return _offset(bboxes, 4, labels)


def init_lm_locmap(vertices, batch_size=None):
    """DenseBox.py:1689-1720."""
    return _offset(vertices, 8, None)


def init_lm_locmap_pn(vertices, labels):
    """DenseBox.py:1763-1799 (effective definition)."""
    return _offset(vertices, 8, labels)


def _heat(vertices, labels, clamp):
    v = _dev(vertices)
    n = v.size(0)
    assert tuple(v.shape) == (n, 8)
    out = torch.empty((n, 4, HW, HW), dtype=torch.float32, device=v.device)
    lab = _dev(labels, v.device)
    check(_lib.lib().dbx_init_lm_heatmap(ptr(v), ptr(lab), n, clamp, ptr(out), stream_ptr()))
    return out


def init_lm_heatmap(vertices, batch_size=None):
    """DenseBox.py:1802-1825.  The reference raises IndexError when a landmark rounds to 60; so does this."""
    v = torch.as_tensor(vertices).float()
    if bool(((v + 0.5).to(torch.int64) >= HW).any()):
        raise IndexError('landmark index 60 is out of bounds for dimension with size 60')
    return _heat(vertices, None, 0)


def init_lm_heatmap_pn(vertices, labels):
    """DenseBox.py:1873-1914 (effective definition: clamps to 59, skips negative patches)."""
    return _heat(vertices, labels, 1)


def mask_by_sel(loss_mask, pos_indices, neg_indices):
    """DenseBox.py:1368-1402, in place on a GPU mask [N,1,60,60]."""
    assert tuple(loss_mask.shape) == (loss_mask.size(0), 1, HW, HW) and loss_mask.is_cuda and loss_mask.is_contiguous()
    pos = _dev(pos_indices, loss_mask.device, torch.int64)
    neg = _dev(neg_indices, loss_mask.device, torch.int64)
    check(_lib.lib().dbx_mask_by_sel(ptr(loss_mask), loss_mask.size(0), ptr(pos), pos.size(0), ptr(neg),
                                     neg.size(1) if neg.dim() == 2 else 0, stream_ptr()))


def mask_gray_zone_cls(loss_mask, bboxes, ratio=0.3, gray_border=2.0):
    """DenseBox.py:1465-1504."""
    assert ratio == 0.3 and gray_border == 2.0
    assert loss_mask.is_cuda and loss_mask.is_contiguous() and tuple(loss_mask.shape[1:]) == (1, HW, HW)
    bb = _dev(bboxes, loss_mask.device)
    check(_lib.lib().dbx_mask_gray_zone_cls(ptr(loss_mask), ptr(bb), None, loss_mask.size(0), stream_ptr()))


def mask_gray_zone_cls_pn(loss_mask, bboxes, labels, ratio=0.3, gray_border=2.0):
    """DenseBox.py:1507-1553."""
    assert ratio == 0.3 and gray_border == 2.0
    assert loss_mask.is_cuda and loss_mask.is_contiguous() and tuple(loss_mask.shape[1:]) == (1, HW, HW)
    # NB: two temporaries in one call expression would be freed before the launch and could share one block
    bb, lab = _dev(bboxes, loss_mask.device), _dev(labels, loss_mask.device)
    check(_lib.lib().dbx_mask_gray_zone_cls(ptr(loss_mask), ptr(bb), ptr(lab), loss_mask.size(0), stream_ptr()))


def mask_gray_zone_lm(loss_mask, pos_indices, lm_id=0, gray_border=2.0):
    """DenseBox.py:1435-1462.  ``loss_mask`` must be a contiguous [N,1,60,60] GPU tensor (the reference passes a
    strided view of the [N,4,60,60] mask; the fused loss kernel handles that case on-chip)."""
    assert lm_id in (0, 1, 2, 3)
    assert loss_mask.is_cuda and loss_mask.is_contiguous() and tuple(loss_mask.shape[1:]) == (1, HW, HW)
    pos = _dev(pos_indices, loss_mask.device, torch.int64)
    check(_lib.lib().dbx_mask_gray_zone_lm(ptr(loss_mask), loss_mask.size(0), ptr(pos), pos.size(0), stream_ptr()))


def gen_neg_loss(loss_orig, map_gt):
    """DenseBox.py:1917-1933 (element-wise; torch op on the caller's device)."""
    assert loss_orig.size() == map_gt.size()
    return loss_orig * (1.0 - map_gt)


# ---------------------------------------------------------------------------------------------- host bookkeeping
def positive_count(bbox, labels=None):
    """Number of positive pixels per patch from the boxes alone -- the area of the centre rectangle of
    init_score_map (DenseBox.py:1572-1582) after python-slice clamping.  float32 product, float64 sums and
    int() truncation exactly as in the reference (and in loss.hip::rect_axis)."""
    b = np.asarray(torch.as_tensor(bbox).detach().cpu(), np.float32)
    n = b.shape[0]
    lab = None if labels is None else np.asarray(torch.as_tensor(labels).detach().cpu(), np.float32).reshape(n)
    out = np.zeros(n, np.int64)
    for i in range(n):
        if lab is not None and lab[i] == 0.0:
            continue
        ext = []
        for c0, c2 in ((b[i, 0], b[i, 2]), (b[i, 1], b[i, 3])):
            centre = float(np.float32(c0 + c2)) * 0.5
            rw = np.float32(np.float32(0.3) * np.float32(c2 - c0))
            org = int(centre - float(np.float32(rw * np.float32(0.5))) + 0.5)
            end = int(float(org) + float(rw) + 0.5)
            lo, hi = org, end + 1
            lo = max(lo + HW, 0) if lo < 0 else min(lo, HW)
            hi = max(hi + HW, 0) if hi < 0 else min(hi, HW)
            ext.append(max(0, hi - lo))
        out[i] = ext[0] * ext[1]
    return out


def neg_counts(positive_num, batch):
    """DenseBox.py:2074, :2081."""
    neg_num = int(float(positive_num) / float(batch) + 0.5)
    return neg_num, int(neg_num * 0.5 + 0.5)
