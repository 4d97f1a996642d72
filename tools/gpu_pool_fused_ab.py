"""conv1_2 + pool1 fused forward (dbx_conv_forward_pool) at batch 64: with / without the full-resolution map.
usage: python tools/gpu_pool_fused_ab.py [dtype]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densebox_amd import _lib
from densebox_amd._lib import View, ConvDesc, check, ptr, stream_ptr
dtn = sys.argv[1] if len(sys.argv) > 1 else 'f16'
N, H, c = 64, 240, 64
dt = _lib.DTYPE_ID[dtn]; L = _lib.lib()
tdt = {'bf16': torch.bfloat16, 'f16': torch.float16}[dtn]
def framed(n, h, c, pad=1):
    hp = h + 2 * pad; guard = max(8 * hp, 576 + hp) * c
    flat = torch.zeros(guard * 2 + n * hp * hp * c, dtype=tdt, device='cuda')
    t = flat[guard:guard + n * hp * hp * c].view(n, hp, hp, c)
    t[:, pad:h + pad, pad:h + pad] = torch.randn((n, h, h, c), device='cuda').to(tdt)
    return flat, t
fx, x = framed(N, H, c); fy, y = framed(N, H, c); fp, p = framed(N, H // 2, c)
xv = View(C.c_void_p(x.data_ptr()), N, H, H, 1, c, 0, c); yv = View(C.c_void_p(y.data_ptr()), N, H, H, 1, c, 0, c)
pv = View(C.c_void_p(p.data_ptr()), N, H // 2, H // 2, 1, c, 0, c)
d = ConvDesc(dt, 3, 3, 1, c, c, _lib.EPI_BIAS | _lib.EPI_RELU)
w = (torch.randn(L.dbx_conv_packed_elems(C.byref(d)), device='cuda') * 0.05).to(tdt); b = torch.zeros(c, device='cuda')
assert L.dbx_conv_pool_fusable(C.byref(d), C.byref(xv), C.byref(yv))
for full in (1, 0, 1, 0):
    for _ in range(3): check(L.dbx_conv_forward_pool(C.byref(d), C.byref(xv), ptr(w), ptr(b), C.byref(yv), C.byref(pv), full, stream_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): check(L.dbx_conv_forward_pool(C.byref(d), C.byref(xv), ptr(w), ptr(b), C.byref(yv), C.byref(pv), full, stream_ptr()))
    e1.record(); torch.cuda.synchronize()
    print('write_full=%d: %.1f us' % (full, e0.elapsed_time(e1) * 100))
