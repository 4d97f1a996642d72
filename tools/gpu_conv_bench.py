"""Per-layer micro-benchmark of the conv / wgrad kernels at the bench shapes (N=64, 240x240 patches).
usage: python tools/gpu_conv_bench.py [fwd|wgrad|all] [dtype] [N]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densebox_amd import _lib
from densebox_amd._lib import View, ConvDesc, check, ptr, stream_ptr
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
dtn = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
N = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dt = _lib.DTYPE_ID[dtn]; es = _lib.ESIZE[dt]
L = _lib.lib()
tdt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[dtn]
# (name, H, cin, cout, k, pad)
LAYERS = [('conv1_2', 240, 64, 64, 3, 1), ('conv2_1', 120, 64, 128, 3, 1), ('conv2_2', 120, 128, 128, 3, 1),
          ('conv3_1', 60, 128, 256, 3, 1), ('conv3_2', 60, 256, 256, 3, 1), ('conv4_1', 30, 256, 512, 3, 1),
          ('conv4_2', 30, 512, 512, 3, 1), ('heads1', 60, 768, 2048, 1, 0), ('dfusion', 60, 2048, 512, 1, 0)]
KEEP = []
def framed(n, h, c, pad=1):
    # zero guard band before/after the frame: the weight-gradient kernel over-reads by a few frame rows
    hp = h + 2 * pad
    guard = max(8 * hp, 576 + hp) * c
    flat = torch.zeros(guard * 2 + n * hp * hp * c, dtype=tdt, device='cuda')
    KEEP.append(flat)
    t = flat[guard:guard + n * hp * hp * c].view(n, hp, hp, c)
    t[:, pad:h + pad, pad:h + pad] = torch.randn((n, h, h, c), device='cuda').to(tdt)
    return t
def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
tot = {'fwd': 0.0, 'wgrad': 0.0}
for name, H, ci, co, k, pad in LAYERS:
    KEEP.clear(); x = framed(N, H, ci); y = framed(N, H, co)
    xv = View(C.c_void_p(x.data_ptr()), N, H, H, 1, ci, 0, ci); yv = View(C.c_void_p(y.data_ptr()), N, H, H, 1, co, 0, co)
    flops = 2.0 * N * H * H * k * k * ci * co
    line = '%-8s %3dx%-3d %4d->%-4d' % (name, H, H, ci, co)
    if which in ('fwd', 'all'):
        d = ConvDesc(dt, k, k, pad, ci, co, _lib.EPI_BIAS | _lib.EPI_RELU)
        w = (torch.randn(L.dbx_conv_packed_elems(C.byref(d)), device='cuda') * 0.05).to(tdt)
        b = torch.zeros(co, device='cuda')
        us = timeit(lambda: check(L.dbx_conv_forward(C.byref(d), C.byref(xv), ptr(w), ptr(b), C.byref(yv), None, None, 0, stream_ptr())))
        line += '  fwd %8.1f us %7.1f TF' % (us, flops / us / 1e6); tot['fwd'] += us
    if which in ('wgrad', 'all') and name not in ('dfusion',):
        dw = torch.empty((co, ci, k, k), device='cuda'); db = torch.empty(co, device='cuda')
        need = L.dbx_conv_wgrad_scratch_bytes(dt, C.byref(yv), C.byref(xv), k, k)
        sc = torch.empty(need, dtype=torch.uint8, device='cuda')
        us = timeit(lambda: check(L.dbx_conv_wgrad(dt, C.byref(yv), C.byref(xv), k, k, pad, co, ci, ptr(dw), ptr(db), ptr(sc), 0, stream_ptr())))
        line += '  wgrad %8.1f us %7.1f TF' % (us, flops / us / 1e6); tot['wgrad'] += us
    print(line)
print('total us', tot)
