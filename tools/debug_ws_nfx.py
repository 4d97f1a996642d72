"""Debug aid: the heads' 1x1 GEMM (ws kernel, fixed epilogue) under DBX_WS_NFX=4 vs 8: saves / compares the hidden map."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from densebox_amd import _lib
from densebox_amd._lib import View, ConvDesc, check, ptr, stream_ptr
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_hip_kernels import framed, pack
L = _lib.lib(); dt = _lib.F16; tdt = torch.float16
n, h, w, nh = 3, 60, 60, 4
g = torch.Generator(device='cpu').manual_seed(77)
x = torch.relu(torch.randn(n, 768, h, w, generator=g)).cuda()
w1 = (torch.randn(512 * nh, 768, 1, 1, generator=g) * 0.05).cuda(); b1 = torch.randn(512 * nh, generator=g).cuda()
fx, tx, xv = framed(x, 1, tdt)
d = ConvDesc(dt, 1, 1, 0, 768, 512 * nh, _lib.EPI_BIAS | _lib.EPI_DROPHASH | _lib.CONV_WFRAG, 0x1234ABCD)
w1f = pack(L, dt, w1, 768, 512 * nh, mode=4)
fb, tb, hvb = framed(torch.zeros(n, 512 * nh, h, w), 0, tdt)
check(L.dbx_conv_forward(C.byref(d), C.byref(xv), ptr(w1f), ptr(b1), C.byref(hvb), None, None, 0, stream_ptr()))
torch.cuda.synchronize()
out = tb.float().cpu().numpy()
tag = os.environ.get('DBX_WS_NFX', '8')
np.save('/tmp/hid_%s.npy' % tag, out)
if os.path.exists('/tmp/hid_8.npy') and tag != '8':
    ref = np.load('/tmp/hid_8.npy')
    bad = np.abs(out - ref) > 1e-3 * (1 + np.abs(ref))
    print('mismatch fraction', bad.mean())
    if bad.any():
        idx = np.argwhere(bad)
        print('per image', [int(bad[i].sum()) for i in range(n)])
        pix = bad.any(axis=3).reshape(n, -1)    # [n, h*w]
        q = np.argwhere(pix)
        print('first bad pixels (img, flat idx):', q[:10].tolist(), '... last', q[-5:].tolist(), 'count', len(q))
        ch = bad.any(axis=(0, 1, 2)); print('bad channel blocks of 64:', sorted(set((np.argwhere(ch).ravel() // 64).tolist())))
        r = np.argwhere(pix[0]).ravel(); print('img0 bad flat pixel idx mod 32 histogram', np.bincount(r % 32, minlength=32).tolist())
        print('img0 bad rows', sorted(set((r // w).tolist()))[:40])
