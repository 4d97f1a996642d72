"""Time 1x1 GEMM shapes through dbx_conv_forward (the kernel dbx_conv_plan picks).  usage: gpu_conv1x1_bench.py [dtype]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densebox_amd import _lib
from densebox_amd._lib import View, ConvDesc, check, ptr, stream_ptr
dtn = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dt = _lib.DTYPE_ID[dtn]; L = _lib.lib()
tdt = {'bf16': torch.bfloat16, 'f16': torch.float16}[dtn]
def framed(n, h, c, pad=1):
    hp = h + 2 * pad; guard = max(8 * hp, 576 + 4 * hp) * c
    flat = torch.zeros(guard * 2 + n * hp * hp * c, dtype=tdt, device='cuda')
    t = flat[guard:guard + n * hp * hp * c].view(n, hp, hp, c)
    t[:, pad:h + pad, pad:h + pad] = torch.randn((n, h, h, c), device='cuda').to(tdt)
    return flat, t
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
for name, H, ci, co, epi in [('heads fwd 768->2048 @60', 60, 768, 2048, _lib.EPI_BIAS | _lib.EPI_DROPHASH), ('Wc 256->2048 @60', 60, 256, 2048, _lib.EPI_BIAS | _lib.EPI_DROPHASH),
                             ('Wu 512->2048 @30', 30, 512, 2048, 0), ('dC 2048->256 @60', 60, 2048, 256, 0), ('dA 2048->512 @30', 30, 2048, 512, 0),
                             ('dgrad 2048->768 @60', 60, 2048, 768, 0),
                             ('heads fwd, bias only', 60, 768, 2048, _lib.EPI_BIAS), ('heads fwd 768->512 (one head)', 60, 768, 512, _lib.EPI_BIAS)]:
    fx, x = framed(N, H, ci); fy, y = framed(N, H, co)
    xv = View(C.c_void_p(x.data_ptr()), N, H, H, 1, ci, 0, ci); yv = View(C.c_void_p(y.data_ptr()), N, H, H, 1, co, 0, co)
    d = ConvDesc(dt, 1, 1, 0, ci, co, epi, 0x1234)
    plan = _lib.ConvPlan(); check(L.dbx_conv_plan(C.byref(d), C.byref(xv), C.byref(yv), C.byref(plan)))
    w = torch.randn(co, ci, 1, 1, device='cuda') * 0.05
    wp = torch.zeros(L.dbx_conv_packed_elems(C.byref(d)) * 2, dtype=torch.uint8, device='cuda')
    check(L.dbx_pack_weight(dt, 4 if plan.w_frag else 0, ptr(w), co, ci, 1, 1, ptr(wp), co, ci, 0, 0, stream_ptr()))
    d2 = ConvDesc(dt, 1, 1, 0, ci, co, epi | (_lib.CONV_WFRAG if plan.w_frag else 0), 0x1234)
    b = torch.zeros(co, device='cuda')
    run = lambda: check(L.dbx_conv_forward(C.byref(d2), C.byref(xv), ptr(wp), ptr(b), C.byref(yv), None, None, 0, stream_ptr()))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e2
    fl = 2.0 * N * H * H * ci * co
    print('%-26s %-34s %8.1f us %7.0f TFLOP/s' % (name, plan.name.decode(), us, fl / us / 1e6))
    del fx, fy, x, y
