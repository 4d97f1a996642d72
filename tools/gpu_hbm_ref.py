"""HBM reference points on this box (torch library kernels): copy, write-only fill, read-only sum.  python tools/gpu_hbm_ref.py"""
import torch
def t(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device='cuda'); b = torch.empty_like(a)
af = a.view(torch.float32)
print('copy  (1 GiB read + 1 GiB write): %.0f GB/s total' % (2 * n / t(lambda: b.copy_(a)) / 1e9))
print('fill  (1 GiB write only)        : %.0f GB/s' % (n / t(lambda: a.zero_()) / 1e9))
print('sum   (1 GiB read only, fp32)   : %.0f GB/s' % (n / t(lambda: af.sum()) / 1e9))
x = torch.empty(n // 2, dtype=torch.bfloat16, device='cuda')
print('relu_ (1 GiB read + write in place, bf16): %.0f GB/s total' % (2 * n / t(lambda: x.relu_()) / 1e9))
