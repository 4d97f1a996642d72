"""Time dbx_conv_wgrad (kernel + split reduction) on the network's batch-64 layer shapes with HIP events.
usage: gpu_wgrad_bench.py [dtype] [iters] [layer,layer]   (DBX_WGRAD_VARIANT etc. select kernels, one process per setting)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densebox_amd import _lib
from densebox_amd._lib import View, check, ptr, stream_ptr
dtn = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dt = _lib.DTYPE_ID[dtn]; L = _lib.lib()
tdt = {'bf16': torch.bfloat16, 'f16': torch.float16}[dtn]
def framed(n, h, c, pad=1, relu=False):
    hp = h + 2 * pad; guard = max(8 * hp, 576 + 4 * hp) * c
    flat = torch.zeros(guard * 2 + n * hp * hp * c, dtype=tdt, device='cuda')
    t = flat[guard:guard + n * hp * hp * c].view(n, hp, hp, c)
    v = torch.randn((n, h, h, c), device='cuda')
    t[:, pad:h + pad, pad:h + pad] = (v.clamp_min(0) if relu else v).to(tdt)
    return flat, t
LAYERS = [('conv1_2', 240, 64, 64, 3), ('conv2_1', 120, 64, 128, 3), ('conv2_2', 120, 128, 128, 3), ('conv3_1', 60, 128, 256, 3),
          ('conv3_2', 60, 256, 256, 3), ('conv4_1', 30, 256, 512, 3), ('conv4_2', 30, 512, 512, 3), ('heads1', 60, 768, 2048, 1)]
N = 64
only = sys.argv[3].split(',') if len(sys.argv) > 3 else None
for name, H, ci, co, k in LAYERS:
    if only and name not in only: continue
    fx, x = framed(N, H, ci, relu=True); fy, y = framed(N, H, co)
    xv = View(C.c_void_p(x.data_ptr()), N, H, H, 1, ci, 0, ci); yv = View(C.c_void_p(y.data_ptr()), N, H, H, 1, co, 0, co)
    dw = torch.empty((co, ci, k, k), device='cuda'); db = torch.empty(co, device='cuda')
    sc = torch.empty(L.dbx_conv_wgrad_scratch_bytes(dt, C.byref(yv), C.byref(xv), k, k), dtype=torch.uint8, device='cuda')
    run = lambda: check(L.dbx_conv_wgrad(dt, C.byref(yv), C.byref(xv), k, k, 1 if k == 3 else 0, co, ci, ptr(dw), ptr(db), ptr(sc), 0, stream_ptr()))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * N * H * H * ci * co * k * k
    print('%-8s %4dx%-4d %4d->%-4d k%d  %8.1f us  %7.0f TFLOP/s   dw checksum %.6e' % (name, H, H, ci, co, k, us, fl / us / 1e6, float(dw.double().abs().sum())))
    del fx, fy, x, y, sc
