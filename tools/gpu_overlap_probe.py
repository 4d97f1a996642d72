"""Does an HBM-bound streaming kernel overlap with a persistent MFMA kernel when both are in flight on two streams?
all9 weight gradient of a conv4 layer (512 -> 512, 30x30, batch 64) + pool1's backward from nibbles (64 x 240 x 240 x 64).
usage: python tools/gpu_overlap_probe.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densebox_amd import _lib
from densebox_amd._lib import View, check, ptr
dt = _lib.DTYPE_ID['f16']; L = _lib.lib(); tdt = torch.float16
def framed(n, h, c, pad=1, rnd=True):
    hp = h + 2 * pad; guard = max(8 * hp, 576 + 4 * hp) * c
    flat = torch.zeros(guard * 2 + n * hp * hp * c, dtype=tdt, device='cuda')
    t = flat[guard:guard + n * hp * hp * c].view(n, hp, hp, c)
    if rnd: t[:, pad:h + pad, pad:h + pad] = torch.randn((n, h, h, c), device='cuda').to(tdt)
    return flat, t, View(C.c_void_p(t.data_ptr()), n, h, h, pad, c, 0, c)
N = 64
fz, tz, zv = framed(N, 30, 512); fx, tx, xv = framed(N, 30, 512)
dw = torch.empty((512, 512, 3, 3), device='cuda'); db = torch.empty(512, device='cuda')
sc = torch.empty(L.dbx_conv_wgrad_scratch_bytes(dt, C.byref(zv), C.byref(xv), 3, 3), dtype=torch.uint8, device='cuda')
fdy, tdy, dyv = framed(N, 120, 64, pad=0); fdx, tdx, dxv = framed(N, 240, 64, rnd=False)
idx = torch.randint(0, 255, (L.dbx_maxpool_idx_bytes(N, 240, 240, 64) + 16,), dtype=torch.uint8, device='cuda')
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def wg(s, k=4):
    for _ in range(k): check(L.dbx_conv_wgrad(dt, C.byref(zv), C.byref(xv), 3, 3, 1, 512, 512, ptr(dw), ptr(db), ptr(sc), 0, C.c_void_p(s.cuda_stream)))
def pool(s, k=8):
    for _ in range(k): check(L.dbx_maxpool2x2_bwd_idx(dt, ptr(idx), C.byref(dyv), C.byref(dxv), 0, 1, C.c_void_p(s.cuda_stream)))
def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record(cur); sa.wait_event(e0); sb.wait_event(e0)
    fn()
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    ea.record(sa); eb.record(sb); cur.wait_event(ea); cur.wait_event(eb); e1.record(cur)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
for _ in range(2): wg(sa); pool(sb)
for rep in range(3):
    ta = timed(lambda: wg(sa)); tb = timed(lambda: pool(sb))
    tab = timed(lambda: (wg(sa), pool(sb))); tba = timed(lambda: (pool(sb), wg(sa)))
    print('wgrad x4 alone %.0f us, pool_bwd x8 alone %.0f us, together (wgrad first) %.0f us, (pool first) %.0f us; sum %.0f' % (ta, tb, tab, tba, ta + tb))
