import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import test_hip_kernels as TK
from densebox_amd import _lib
from densebox_amd._lib import check, ptr, stream_ptr
L = _lib.lib()
for case in [TK.WG_CASES[0], TK.WG_CASES[5], TK.WG_CASES[2]]:
    n, h, w, civ, ci, co, k, pad = case
    dt, tdt = _lib.BF16, torch.bfloat16
    x = torch.randn(n, civ, h, w).cuda(); dz = torch.randn(n, max(co, 8), h, w).cuda()
    fx, tx, xv = TK.framed(x, 1, tdt); fz, tz, dzv = TK.framed(dz, 1, tdt)
    sc = torch.empty(L.dbx_conv_wgrad_scratch_bytes(dt, C.byref(dzv), C.byref(xv), k, k), dtype=torch.uint8, device='cuda')
    outs = []
    for rep in range(3):
        dw = torch.zeros(co, ci, k, k, device='cuda'); db = torch.zeros(co, device='cuda')
        check(L.dbx_conv_wgrad(dt, C.byref(dzv), C.byref(xv), k, k, pad, co, ci, ptr(dw), ptr(db), ptr(sc), 0, stream_ptr()))
        outs.append((dw.clone(), db.clone()))
    print(case, 'rep0==rep1', torch.equal(outs[0][0], outs[1][0]), 'rep1==rep2', torch.equal(outs[1][0], outs[2][0]))
    dw = outs[0][0].clone(); db = outs[0][1].clone()
    check(L.dbx_conv_wgrad(dt, C.byref(dzv), C.byref(xv), k, k, pad, co, ci, ptr(dw), ptr(db), ptr(sc), 1, stream_ptr()))
    print('   accumulate: max|dw - 2*dw0| =', (dw - 2 * outs[0][0]).abs().max().item(), ' db:', (db - 2 * outs[0][1]).abs().max().item())
