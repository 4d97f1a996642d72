"""Heads backward with the hidden gradient generated in its consumers vs materialised (batch 64, 60x60, four heads): per-kernel us.
usage (GPU box): python tools/gpu_heads_gen_bench.py [f16|bf16]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
sys.path.insert(0, 'tools')
from densebox_amd import _lib                                   # noqa: E402
from densebox_amd._lib import check, ptr, stream_ptr            # noqa: E402
from test_hip_kernels import framed, TDT                        # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1000 / reps


def main():
    dtn = sys.argv[1] if len(sys.argv) > 1 else 'f16'
    L = _lib.lib()
    dt, tdt = _lib.DTYPE_ID[dtn], TDT[dtn]
    n, h, w, ks = 64, 60, 60, [1, 4, 4, 8]
    nh = len(ks)
    g = torch.Generator(device='cpu').manual_seed(1)
    dout = torch.zeros(n, 8 * nh, h, w)
    for i, k in enumerate(ks):
        dout[:, 8 * i:8 * i + k] = torch.randn(n, k, h, w, generator=g)
    w2 = [(torch.randn(k, 512, generator=g) * 0.05).to(tdt).float().cuda().contiguous() for k in ks]
    fo, to, dv = framed(dout, 0, tdt)
    fx, tx, xv = framed(torch.randn(n, 256, h, w, generator=g), 1, tdt)
    fd, td, dhv = framed(torch.zeros(n, 512 * nh, h, w), 1, tdt)
    karr = (C.c_int32 * nh)(*ks)
    wp = (C.c_void_p * nh)(*[t.data_ptr() for t in w2])
    use_hash, seed = 1, 0xBEEF
    check(L.dbx_head2_dgrad(dt, C.byref(dv), wp, karr, nh, C.byref(dhv), None, 512 * nh, use_hash, seed, stream_ptr()))
    sc = torch.empty(L.dbx_conv_wgrad_scratch_bytes(dt, C.byref(dhv), C.byref(xv), 1, 1), dtype=torch.uint8, device='cuda')
    dw = torch.zeros((512 * nh, 256, 1, 1), device='cuda'); db = torch.zeros((512 * nh,), device='cuda')
    t_mat = timed(lambda: check(L.dbx_conv_wgrad_slice(dt, C.byref(dhv), C.byref(xv), 1, 1, 0, 512 * nh, 256, ptr(dw), 256, 0, ptr(db), ptr(sc), 0, stream_ptr())))
    t_gen = timed(lambda: check(L.dbx_heads1_wgrad_gen(dt, C.byref(dv), C.byref(xv), wp, karr, nh, use_hash, seed, 256, ptr(dw), 256, 0, ptr(db), ptr(sc), stream_ptr())))
    print('%s dW1 (2048 x 256 over %d pixels, incl. the split reduction): d_hid from memory %.1f us, generated %.1f us' % (dtn, n * h * w, t_mat, t_gen))
    from gpu_heads_gen_dgrad import bench_dgrad
    bench_dgrad(L, dt, tdt, dtn, n, h, w, ks, timed)


if __name__ == '__main__':
    main()
