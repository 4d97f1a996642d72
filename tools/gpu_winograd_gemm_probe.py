"""what the batched-GEMM core of a Winograd F(2x2,3x3) conv4_2 would cost -- 16 GEMMs [14400 tiles x 512] x [512 x 512]
(f16, vendor library through torch.bmm) plus streaming stand-ins for the input / output transforms (their HBM traffic only) -- against
the direct conv4_2 kernel of this library on the same box.  Lower bound of a Winograd layer: no kernel of it exists in the tree."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
T = 64 * 15 * 15
V = torch.randn(16, T, 512, device='cuda').half(); U = torch.randn(16, 512, 512, device='cuda').half()
M = torch.empty(16, T, 512, device='cuda', dtype=torch.float16)
print('bmm 16 x [%d x 512] x [512 x 512] f16: %.1f us (120.8 GFLOP)' % (T, timeit(lambda: torch.bmm(V, U, out=M))))
a = torch.randn(64, 30, 30, 512, device='cuda').half()
print('input-transform traffic stand-in (read 59 MB, write 236 MB): %.1f us' % timeit(lambda: V.view(16, -1)[:, :a.numel() // 4 * 4 // 4 * 1].copy_(a.view(1, -1)[:, :V.view(16, -1).shape[1]].expand(16, -1)) if False else V.copy_(V.flip(0)[:1].expand(16, -1, -1))))
print('output-transform traffic stand-in (read 236 MB, write 59 MB): %.1f us' % timeit(lambda: a.view(-1).copy_(M.view(16, -1).sum(0)[:a.numel()]) if False else a.view(4, -1).copy_(M.view(16, -1)[:4, :a.numel() // 4])))
