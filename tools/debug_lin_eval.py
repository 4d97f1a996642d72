import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densebox_amd import _lib
from densebox_amd._lib import View, ConvDesc, check, ptr, stream_ptr
L = _lib.lib(); dt = _lib.F16; tdt = torch.float16
N, ktot = 64, 17
def framed(x_nchw, pad):
    n, c, h, w = x_nchw.shape
    hp, wp = h + 2 * pad, w + 2 * pad
    guard = max(8 * wp, 576 + 4 * wp) * c
    flat = torch.zeros(2 * guard + n * hp * wp * c, dtype=tdt, device='cuda')
    t = flat[guard:guard + n * hp * wp * c].view(n, hp, wp, c)
    t[:, pad:pad + h, pad:pad + w] = x_nchw.permute(0, 2, 3, 1).to(tdt)
    return flat, t, View(C.c_void_p(t.data_ptr()), n, h, w, pad, c, 0, c)
g = torch.Generator().manual_seed(1)
a44 = torch.randn(N, 512, 30, 30, generator=g).cuda(); c34 = torch.randn(N, 256, 60, 60, generator=g).cuda()
wa = (torch.randn(ktot, 512, 1, 1, generator=g) * 0.05).cuda(); wc = (torch.randn(ktot, 256, 1, 1, generator=g) * 0.05).cuda()
bias = torch.zeros(64, device='cuda'); bias[:ktot] = torch.randn(ktot, generator=g).cuda()
def pack(w, cin):
    d = ConvDesc(dt, 1, 1, 0, cin, 64, 0)
    out = torch.zeros(L.dbx_conv_packed_elems(C.byref(d)) * 2, dtype=torch.uint8, device='cuda')
    check(L.dbx_pack_weight(dt, 0, ptr(w), w.shape[0], w.shape[1], 1, 1, ptr(out), 64, cin, 0, 0, stream_ptr()))
    return out
wpa, wpc = pack(wa, 512), pack(wc, 256)
def run(a44_, c34_):
    fa, ta, av = framed(a44_, 1); fc, tc, cv = framed(c34_, 1)
    ga = torch.empty((N, ktot, 30, 30), device='cuda'); gv = View(C.c_void_p(ga.data_ptr()), N, 30, 30, 0, ktot, 0, ktot)
    d = ConvDesc(dt, 1, 1, 0, 512, 64, _lib.EPI_F32_NCHW)
    check(L.dbx_conv_forward(C.byref(d), C.byref(av), ptr(wpa), None, C.byref(gv), None, None, 0, stream_ptr()))
    big = torch.empty((N, ktot, 60, 60), device='cuda'); yv = View(C.c_void_p(big.data_ptr()), N, 60, 60, 0, ktot, 0, ktot)
    check(L.dbx_upsample_bilinear_nchw_f32(ptr(ga), N * ktot, 30, 30, ptr(big), 60, 60, stream_ptr()))
    up = big.clone()
    d2 = ConvDesc(dt, 1, 1, 0, 256, 64, _lib.EPI_BIAS | _lib.EPI_F32_NCHW | _lib.EPI_ACCUM)
    check(L.dbx_conv_forward(C.byref(d2), C.byref(cv), ptr(wpc), ptr(bias), C.byref(yv), None, None, 0, stream_ptr()))
    torch.cuda.synchronize()
    return ga, up, big
perm = torch.roll(torch.arange(N), 19).cuda()
r1 = run(a44, c34); r2 = run(a44, c34); r3 = run(a44[perm], c34[perm])
for name, a, b, c in zip(('ga', 'up', 'big'), r1, r2, r3):
    d = (c - a[perm]).abs().flatten(1).max(dim=1).values
    print(name, 'repeat equal', torch.equal(a, b), ' rolled max diff %.3e' % float(d.max()), 'images', (d > 0).nonzero().flatten().tolist()[:12])
# plan names
for cin, v in ((512, 30), (256, 60)):
    pass
import torch.nn.functional as F
ga1, up1, _ = r1
ga3, up3, _ = r3
d = (up3 - up1[perm]).abs()
idx = d.flatten().nonzero().flatten()
print('n diff', idx.numel(), 'first flat', idx[:5].tolist(), 'last', idx[-3:].tolist())
want1 = F.interpolate(ga1, size=(60, 60), mode='bilinear', align_corners=True)
want3 = F.interpolate(ga3, size=(60, 60), mode='bilinear', align_corners=True)
print('run1 vs torch: max %.3e ndiff %d' % (float((up1 - want1).abs().max()), int((up1 != want1).sum())))
print('run3 vs torch: max %.3e ndiff %d' % (float((up3 - want3).abs().max()), int((up3 != want3).sum())))
f = int(idx[0]); img, rem = f // (17 * 3600), f % (17 * 3600)
k, pix = rem // 3600, rem % 3600
print('first diff: rolled image', img, 'plane', k, 'y', pix // 60, 'x', pix % 60, float(up3.flatten()[f]), float(up1[perm].flatten()[f]))
