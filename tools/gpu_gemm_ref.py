"""Library reference points for the 1x1 GEMM shapes of the step (torch.mm -> hipBLASLt), bf16: python tools/gpu_gemm_ref.py"""
import torch
def t(m, n, k, it=10):
    a = torch.randn(m, k, device='cuda').bfloat16(); b = torch.randn(k, n, device='cuda').bfloat16()
    for _ in range(3): torch.mm(a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): torch.mm(a, b)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / it * 1e3
    print('M=%7d N=%5d K=%5d  %8.1f us  %7.1f TFLOP/s' % (m, n, k, us, 2.0 * m * n * k / us / 1e6))
M = 64 * 60 * 60
t(M, 2048, 768); t(M, 512, 2048); t(M, 256, 2048); t(M, 768, 2048)
t(2048, 768, M); t(8192, 8192, 8192); t(64 * 30 * 30, 512, 4608); t(64 * 240 * 240, 64, 576)
