"""The data-gradient half of tools/gpu_heads_gen_bench.py (d_c34 = gate * W1c^T d_hid at batch 64, 60x60)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, 'tests')
from densebox_amd import _lib                                   # noqa: E402
from densebox_amd._lib import check, ptr, stream_ptr, ConvDesc  # noqa: E402


def bench_dgrad(L, dt, tdt, dtn, n, h, w, ks, timed):
    from test_hip_kernels import framed, _heads1_dgrad_setup
    keep, dv, gv, dhv, wp, karr, nh, use_hash, seed, wimg = _heads1_dgrad_setup(L, dt, tdt, n, h, w, ks, 3, 'hash')
    fy, ty, yv = framed(torch.zeros(n, 256, h, w), 1, tdt)
    d1 = ConvDesc(dt, 1, 1, 0, 512 * nh, 256, _lib.EPI_GATE | _lib.CONV_WFRAG, 0)
    t_mat = timed(lambda: check(L.dbx_conv_forward(C.byref(d1), C.byref(dhv), ptr(wimg), None, C.byref(yv), C.byref(gv), None, 0, stream_ptr())))
    t_gen = timed(lambda: check(L.dbx_heads1_dgrad_gen(dt, C.byref(dv), wp, karr, nh, use_hash, seed, ptr(wimg), C.byref(yv), C.byref(gv), stream_ptr())))
    print('%s d_c34 (2048 -> 256 over %d pixels, ReLU gate): d_hid from memory %.1f us, generated %.1f us' % (dtn, n * h * w, t_mat, t_gen))
