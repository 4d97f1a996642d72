"""Plate rectification after decode: ``perspective_transform`` (DenseBox.py:3446-3481, called from ``viz_result``
:3546-3553) on the GPU -- same name, arguments and result (uint8 image of 1.5x the input size) as the reference, which
delegates to cv2.getPerspectiveTransform / cv2.warpPerspective.  OpenCV is not part of the reference tree: the kernels
follow its published algorithm (see csrc/post_ops.hip); parity with OpenCV itself is unpinned."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, stream_ptr


def dst_rectangle(src_pts):
    """The four destination corners the reference builds (:3462-3474): the axis-aligned rectangle spanned by the
    left/right/top/bottom extremes of (left-up, right-up, right-down, left-down)."""
    assert len(src_pts) == 4
    lu, ru, rd, ld = src_pts
    min_x, max_x = min(lu[0], ld[0]), max(ru[0], rd[0])
    min_y, max_y = min(lu[1], ru[1]), max(ld[1], rd[1])
    return [[min_x, min_y], [max_x, min_y], [max_x, max_y], [min_x, max_y]]


def get_perspective_matrix(src_pts, dst_pts):
    src = np.ascontiguousarray(np.float32(src_pts).reshape(8))
    dst = np.ascontiguousarray(np.float32(dst_pts).reshape(8))
    m = np.empty(9, dtype=np.float64)
    check(_lib.lib().dbx_perspective_matrix(src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p),
                                            m.ctypes.data_as(C.c_void_p)))
    return m.reshape(3, 3)


def warp_perspective(img, M, dsize):
    """img: uint8 [H, W, C] (numpy or a CUDA tensor); dsize = (width, height) like cv2.  Returns the same kind."""
    was_np = isinstance(img, np.ndarray)
    t = torch.as_tensor(img).cuda().contiguous()
    assert t.dtype == torch.uint8 and t.dim() == 3
    h, w, c = t.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    out = torch.empty((dh, dw, c), dtype=torch.uint8, device=t.device)
    m = np.ascontiguousarray(np.asarray(M, dtype=np.float64).reshape(9))
    check(_lib.lib().dbx_warp_perspective_u8(C.c_void_p(t.data_ptr()), h, w, c, m.ctypes.data_as(C.c_void_p),
                                             C.c_void_p(out.data_ptr()), dh, dw, stream_ptr()))
    return out.cpu().numpy() if was_np else out


def perspective_transform(img, src_pts):
    """DenseBox.py:3446: warp so that the four plate corners land on their bounding rectangle; output is
    (int(W*1.5+0.5), int(H*1.5+0.5)) like the reference."""
    M = get_perspective_matrix(src_pts, dst_rectangle(src_pts))
    h, w = img.shape[0], img.shape[1]
    return warp_perspective(img, M, (int(w * 1.5 + 0.5), int(h * 1.5 + 0.5)))
