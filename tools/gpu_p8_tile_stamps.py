"""Where a tile of the heads forward (conv3x3_p8_kernel<T,1,1>) spends its time, from in-kernel s_memrealtime stamps (100 MHz) of wave 0 of every
workgroup: tile top -> K loop done -> sixteen chunks done (hash, masks, rounding, exchange, second-conv MFMAs, hidden-map stores issued) -> cross-wave
reduction done.  Library: tools/build_variant.sh tstamp conv_igemm.hip -DDBX_P8_STAMP -DP8_TILE_STAMPS; run with DBX_LIB=.../libdensebox_hip_tstamp.so
usage: python tools/gpu_p8_tile_stamps.py [cin]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from densebox_amd import _lib
from densebox_amd._lib import View, ConvDesc, check, ptr, stream_ptr
dt = _lib.DTYPE_ID['f16']; L = _lib.lib(); tdt = torch.float16
ci = int(sys.argv[1]) if len(sys.argv) > 1 else 768
N, H, co = 64, 60, 2048
def framed(n, h, c, pad=1):
    hp = h + 2 * pad; guard = max(8 * hp, 576 + 4 * hp) * c
    flat = torch.zeros(guard * 2 + n * hp * hp * c, dtype=tdt, device='cuda')
    t = flat[guard:guard + n * hp * hp * c].view(n, hp, hp, c)
    t[:, pad:h + pad, pad:h + pad] = torch.relu(torch.randn((n, h, h, c), device='cuda')).to(tdt)
    return flat, t
fx, x = framed(N, H, ci); fy, y = framed(N, H, co, pad=0)
xv = View(C.c_void_p(x.data_ptr()), N, H, H, 1, ci, 0, ci); yv = View(C.c_void_p(y.data_ptr()), N, H, H, 0, co, 0, co)
d = ConvDesc(dt, 1, 1, 0, ci, co, _lib.EPI_BIAS | _lib.EPI_DROPHASH, 0x1234)
w = torch.randn(co, ci, 1, 1, device='cuda') * 0.05
wp = torch.zeros(L.dbx_conv_packed_elems(C.byref(d)) * 2, dtype=torch.uint8, device='cuda')
check(L.dbx_pack_weight(dt, 0, ptr(w), co, ci, 1, 1, ptr(wp), co, ci, 0, 0, stream_ptr()))
b = torch.zeros(co, device='cuda')
for _ in range(3):
    check(L.dbx_conv_forward(C.byref(d), C.byref(xv), ptr(wp), ptr(b), C.byref(yv), None, None, 0, stream_ptr()))
torch.cuda.synchronize()
n = 7424
buf = (C.c_ulonglong * (4 * n))()
L.dbx_lab_p8_tile_stamps.restype = C.c_int
got = L.dbx_lab_p8_tile_stamps(buf, n)
st = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4)[:got].astype(np.int64)
ok = st[:, 0] > 0
st = st[ok]
items = np.nonzero(ok)[0]
kl, ch, rd = (st[:, 1] - st[:, 0]) * 0.01, (st[:, 2] - st[:, 1]) * 0.01, (st[:, 3] - st[:, 2]) * 0.01
print('%d stamped tiles, K = %d (%d K tiles per tile)' % (len(st), ci, ci // 64))
print('K loop   : median %.2f us  mean %.2f  p10 %.2f  p90 %.2f' % (np.median(kl), kl.mean(), np.percentile(kl, 10), np.percentile(kl, 90)))
print('chunks   : median %.2f us  mean %.2f  p10 %.2f  p90 %.2f' % (np.median(ch), ch.mean(), np.percentile(ch, 10), np.percentile(ch, 90)))
print('reduction: median %.2f us  mean %.2f  p10 %.2f  p90 %.2f' % (np.median(rd), rd.mean(), np.percentile(rd, 10), np.percentile(rd, 90)))
G = int(os.environ.get('DBX_P8_MAXWG', '256'))
first = items < G                                   # a workgroup's first tile: its K loop starts behind the prologue, no epilogue in front of it
print('K loop of a workgroup\'s FIRST tile: median %.2f us; of the later tiles: %.2f us' % (np.median(kl[first]), np.median(kl[~first])))
tot = (st[:, 3].max() - st[:, 0].min()) * 0.01
print('first stamp -> last stamp: %.1f us; per workgroup tiles x (K + chunks + reduction) = %.1f us' % (tot, (kl + ch + rd).sum() / G))
# K-tile-pair stamps (-DP8_KT_STAMPS): items 0..1023 (a workgroup's first four tiles), stamp in front of K tiles 0, 2, 4, ..
if os.environ.get('P8_KT'):
    full = np.frombuffer((C.c_ulonglong * (4 * 16384))(), dtype=np.uint64)
    big = (C.c_ulonglong * (4 * 16384))()
    L.dbx_lab_p8_tile_stamps(big, 16384)
    a = np.frombuffer(big, dtype=np.uint64).astype(np.int64)
    kt = a[4 * 16384 // 2:4 * 16384 // 2 + 1024 * 8].reshape(1024, 8)
    main = a[:4 * 1024].reshape(1024, 4)
    npair = ci // 128
    for lo, hi, name in ((0, G, 'first tiles'), (G, min(1024, 8 * G), 'later tiles')):
        k = kt[lo:hi]; m = main[lo:hi]
        segs = [np.median((k[:, i + 1] - k[:, i]) * 0.01) for i in range(npair - 1)] + [np.median((m[:, 1] - k[:, npair - 1]) * 0.01)]
        print('%-12s us per pair of K tiles: %s   (tile top -> first pair: %.2f)' % (name, ' '.join('%.2f' % v for v in segs), np.median((k[:, 0] - m[:, 0]) * 0.01)))
