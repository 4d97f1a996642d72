"""Where does the 8-phase kernel's bias + hash-dropout 1x1 GEMM differ from the ws kernel's?  (debugging aid)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import torch.nn.functional as F
from densebox_amd import _lib
from densebox_amd._lib import View, ConvDesc, check, ptr, stream_ptr
from test_hip_kernels import framed, pack, TDT
L = _lib.lib()
for dtn in ('bf16', 'f16'):
    dt, tdt = _lib.DTYPE_ID[dtn], TDT[dtn]
    n, h, w, ci, co = 16, 60, 57, 768, 1024
    g = torch.Generator(device='cpu').manual_seed(3)
    x = torch.randn(n, ci, h, w, generator=g).cuda()
    wt = (torch.randn(co, ci, 1, 1, generator=g) * (1.0 / ci) ** 0.5).cuda()
    b = torch.randn(co, generator=g).cuda()
    ref = F.conv2d(x.to(tdt).float(), wt.to(tdt).float(), b)
    fx, tx, xv = framed(x, 1, tdt)
    outs = {}
    for frag in (True, False):
        for epi in (_lib.EPI_BIAS, _lib.EPI_BIAS | _lib.EPI_DROPHASH):
            fy, ty, yv = framed(torch.zeros(n, co, h, w), 0, tdt)
            d = ConvDesc(dt, 1, 1, 0, ci, co, epi | (_lib.CONV_WFRAG if frag else 0), 0x1234)
            plan = _lib.ConvPlan()
            d0 = ConvDesc(dt, 1, 1, 0, ci, co, epi, 0x1234)
            check(L.dbx_conv_plan(C.byref(d0), C.byref(xv), C.byref(yv), C.byref(plan)))
            check(L.dbx_conv_forward(C.byref(d), C.byref(xv), ptr(pack(L, dt, wt, ci, co, mode=4 if frag else 0)), ptr(b), C.byref(yv), None, None, 0, stream_ptr()))
            torch.cuda.synchronize()
            outs[(frag, epi)] = ty.permute(0, 3, 1, 2).float()
            print(dtn, 'frag', frag, 'epi', epi, 'plan', plan.name.decode())
    for epi in (_lib.EPI_BIAS, _lib.EPI_BIAS | _lib.EPI_DROPHASH):
        a, bb = outs[(True, epi)], outs[(False, epi)]
        tol = 2e-2 if dtn == 'bf16' else 3e-3
        bad = ((a - bb).abs() > tol * (1 + a.abs()))
        print(dtn, 'epi', epi, 'mismatches', int(bad.sum()), 'of', bad.numel(), ' zero-pattern differences', int(((a == 0) != (bb == 0)).sum()))
        if epi == _lib.EPI_BIAS:
            print('   p8 vs ref max err', float((bb - ref).abs().max()), ' ws vs ref', float((a - ref).abs().max()))
        if int(bad.sum()):
            idx = bad.nonzero()
            print('   first', idx[:5].tolist())
            print('   images', sorted(set(idx[:, 0].tolist()))[:20], 'channels mod 32', sorted(set((idx[:, 1] % 32).tolist())), 'channels/32', sorted(set((idx[:, 1] // 32).tolist()))[:40])
            print('   rows', sorted(set(idx[:, 2].tolist()))[:30], 'cols', sorted(set(idx[:, 3].tolist()))[:30])
            i = idx[0]
            print('   a', float(a[i[0], i[1], i[2], i[3]]), 'b', float(bb[i[0], i[1], i[2], i[3]]), 'ref', float(ref[i[0], i[1], i[2], i[3]]))
