"""Run one conv layer a few times (for PMC collection). usage: gpu_one_conv.py H cin cout k [fwd|wgrad] [dtype] [N]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densebox_amd import _lib
from densebox_amd._lib import View, ConvDesc, check, ptr, stream_ptr
H, ci, co, k = [int(v) for v in sys.argv[1:5]]
mode = sys.argv[5] if len(sys.argv) > 5 else 'fwd'
dtn = sys.argv[6] if len(sys.argv) > 6 else 'bf16'
N = int(sys.argv[7]) if len(sys.argv) > 7 else 64
dt = _lib.DTYPE_ID[dtn]; L = _lib.lib()
tdt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[dtn]
def framed(n, h, c, pad=1):
    hp = h + 2 * pad; guard = max(8 * hp, 576 + hp) * c
    flat = torch.zeros(guard * 2 + n * hp * hp * c, dtype=tdt, device='cuda')
    t = flat[guard:guard + n * hp * hp * c].view(n, hp, hp, c)
    t[:, pad:h + pad, pad:h + pad] = torch.randn((n, h, h, c), device='cuda').to(tdt)
    return flat, t
fx, x = framed(N, H, ci); fy, y = framed(N, H, co)
pad = 1 if k == 3 else 0
xv = View(C.c_void_p(x.data_ptr()), N, H, H, 1, ci, 0, ci); yv = View(C.c_void_p(y.data_ptr()), N, H, H, 1, co, 0, co)
if mode == 'fwd':
    d = ConvDesc(dt, k, k, pad, ci, co, _lib.EPI_BIAS | _lib.EPI_RELU)
    w = (torch.randn(L.dbx_conv_packed_elems(C.byref(d)), device='cuda') * 0.05).to(tdt); b = torch.zeros(co, device='cuda')
    for _ in range(3): check(L.dbx_conv_forward(C.byref(d), C.byref(xv), ptr(w), ptr(b), C.byref(yv), None, None, 0, stream_ptr()))
else:
    dw = torch.empty((co, ci, k, k), device='cuda'); db = torch.empty(co, device='cuda')
    sc = torch.empty(L.dbx_conv_wgrad_scratch_bytes(dt, C.byref(yv), C.byref(xv), k, k), dtype=torch.uint8, device='cuda')
    for _ in range(3): check(L.dbx_conv_wgrad(dt, C.byref(yv), C.byref(xv), k, k, pad, co, ci, ptr(dw), ptr(db), ptr(sc), 0, stream_ptr()))
torch.cuda.synchronize()
