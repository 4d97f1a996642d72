"""Stage-2 heads backward, its pieces timed alone at the bench geometry (batch 64, 60x60, DenseBoxLMLOC: 4 heads, k = 1, 4, 4, 8):
dbx_head2_wgrad (streaming dW2 / db2 pass over the hidden map) and dbx_head2_backward_up (the fused pass bench.py runs).
usage: python tools/gpu_head2_bench.py [f16|bf16]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densebox_amd import _lib
from densebox_amd._lib import View, check, ptr, stream_ptr

dtn = sys.argv[1] if len(sys.argv) > 1 else 'f16'
dt = _lib.DTYPE_ID[dtn]
tdt = {'f16': torch.float16, 'bf16': torch.bfloat16}[dtn]
L = _lib.lib()
n, h, w, ks = 64, 60, 60, [1, 4, 4, 8]
nh = len(ks)
hid = torch.randn(n, h, w, 512 * nh, device='cuda').to(tdt)
dout = torch.zeros(n, h, w, 8 * nh, device='cuda', dtype=tdt)
for i, k in enumerate(ks):
    dout[..., 8 * i:8 * i + k] = torch.randn(n, h, w, k, device='cuda').to(tdt)
hv = View(C.c_void_p(hid.data_ptr()), n, h, w, 0, 512 * nh, 0, 512 * nh)
dv = View(C.c_void_p(dout.data_ptr()), n, h, w, 0, 8 * nh, 0, 8 * nh)
g44 = torch.zeros(n, 32, 32, 512 * nh, device='cuda', dtype=tdt)
gv = View(C.c_void_p(g44[:, 1:, 1:].data_ptr()), n, 30, 30, 1, 512 * nh, 0, 512 * nh)
gv = View(C.c_void_p(g44.data_ptr()), n, 30, 30, 1, 512 * nh, 0, 512 * nh)
w2 = [torch.randn(k, 512, device='cuda') * 0.05 for k in ks]
dw = [torch.empty(k, 512, device='cuda') for k in ks]
db = [torch.empty(k, device='cuda') for k in ks]
karr = (C.c_int32 * nh)(*ks)
wp = (C.c_void_p * nh)(*[t.data_ptr() for t in w2])
dwp = (C.c_void_p * nh)(*[t.data_ptr() for t in dw])
dbp = (C.c_void_p * nh)(*[t.data_ptr() for t in db])
sc = torch.empty(int(L.dbx_head2_wgrad_scratch_bytes(nh, n * h)), dtype=torch.uint8, device='cuda')
nost = View(None, n, h, w, 1, 512 * nh, 0, 512 * nh)


def timed(fn, it=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


t_wg = timed(lambda: check(L.dbx_head2_wgrad(dt, C.byref(dv), C.byref(hv), karr, nh, dwp, dbp, ptr(sc), stream_ptr())))
t_up = timed(lambda: check(L.dbx_head2_backward_up(dt, C.byref(dv), C.byref(hv), wp, karr, nh, C.byref(nost), None, 512 * nh, 1, 1234, dwp, dbp, ptr(sc),
                                                   C.byref(gv), stream_ptr())))
gb = hid.numel() * 2 / 1e9
print('%s  head2_wgrad (streaming dW2): %.1f us = %.2f TB/s over the %.2f GB hidden map;  head2_backward_up (fused, no d_hid store): %.1f us'
      % (dtn, t_wg, gb / t_wg * 1e3, gb, t_up))
