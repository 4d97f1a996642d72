"""Calibration of the 8-phase MFMA core as a plain GEMM (tools/probe_gemm_8phase.hip): correctness against torch.mm on asymmetric random
operands (every variant, both dtypes, repeated runs compared bitwise = race screen), then TFLOP/s at 4096^3 / 8192^3 and at the conv3 / conv4
implicit-GEMM shapes, on uniform-random, post-ReLU-like (half zeros) and all-zero operands, interleaved rounds in one process, next to
hipBLASLt (torch.mm) on the same operands.   python tools/gpu_gemm_8phase.py [out.txt]
"""
import ctypes
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'labbin', 'libprobe_gemm_8phase.so'))
lib.p8_gemm.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                        ctypes.c_int, ctypes.c_void_p]
DT = {torch.float16: 0, torch.bfloat16: 1}          # DBX_F16 / DBX_BF16 (include/densebox_hip.h)
VAR = {0: '16x16x32 prio+stagger', 1: '32x32x16 prio+stagger', 2: '16x16x32 stagger, no prio', 3: '16x16x32 prio, no stagger',
       4: '16x16x32 SAFE', 5: '32x32x16 stagger, no prio', 6: '32x32x16 SAFE', 7: '16x16x32 stagger, 2 phases per K tile'}
ABL = {8: 'v2 without LDS-DMA in the loop (timing only)', 9: 'v2 without fragment reads (timing only)', 10: 'v2 with neither (timing only)'}
out_lines = []


def say(s=''):
    print(s, flush=True)
    out_lines.append(s)


def gemm(v, A, B, C, gm=4):
    M, K = A.shape
    N = B.shape[0]
    rc = lib.p8_gemm(v, DT[A.dtype], A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, gm, torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError('p8_gemm rc %d' % rc)


def make(M, N, K, dtype, kind, seed=0):
    g = torch.Generator(device='cuda'); g.manual_seed(seed)
    if kind == 'zero':
        return torch.zeros(M, K, device='cuda', dtype=dtype), torch.zeros(N, K, device='cuda', dtype=dtype)
    A = torch.rand(M, K, device='cuda', generator=g) * 2 - 1
    B = torch.rand(N, K, device='cuda', generator=g) * 2 - 1
    if kind == 'relu':                                  # activations after a ReLU: half of them zero; weights full range
        A = torch.relu(A)
    return A.to(dtype), B.to(dtype)


def check():
    ok_all = True
    for dtype in (torch.float16, torch.bfloat16):
        for (M, N, K) in ((256, 256, 128), (512, 768, 256), (1024, 512, 1024), (768, 1280, 4608)):
            A, B = make(M, N, K, dtype, 'rand', seed=M + N + K)
            ref = A.float() @ B.float().t()
            for v in VAR:
                C = torch.empty(M, N, device='cuda', dtype=dtype)
                gemm(v, A, B, C)
                torch.cuda.synchronize()
                err = (C.float() - ref).abs()
                tol = ref.abs() * (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) + 1e-2
                bad = (err > tol)
                nb = int(bad.sum())
                same = True
                for _ in range(4):                       # race screen: repeated launches must agree bitwise
                    C2 = torch.empty_like(C)
                    gemm(v, A, B, C2)
                    torch.cuda.synchronize()
                    same = same and bool((C2.view(torch.int16) == C.view(torch.int16)).all())
                ok = nb == 0 and same
                ok_all = ok_all and ok
                msg = '%s %-5s M=%5d N=%5d K=%5d  v%d %-28s max err %.3e  bad %d  repeat-bitwise %s' % (
                    'ok  ' if ok else 'FAIL', str(dtype)[6:], M, N, K, v, VAR[v], float(err.max()), nb, same)
                say(msg)
                if nb:
                    idx = bad.nonzero()
                    rows = sorted(set((idx[:, 0] % 256).tolist()))[:40]
                    cols = sorted(set((idx[:, 1] % 256).tolist()))[:40]
                    say('      bad rows mod 256 (first 40): %s' % rows)
                    say('      bad cols mod 256 (first 40): %s' % cols)
                    say('      first bad: %s  got %s  ref %s' % (idx[0].tolist(), float(C[idx[0][0], idx[0][1]]), float(ref[idx[0][0], idx[0][1]])))
    return ok_all


def time_fn(fn, it):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


def bench(M, N, K, dtype, kinds=('rand', 'relu', 'zero'), variants=(0, 1, 2, 3, 5), rounds=3, gms=(4,)):
    flop = 2.0 * M * N * K
    it = max(3, int(0.02 / (flop / 1.0e15)))             # ~20 ms per timing at 1 PFLOP/s
    say('--- M=%d N=%d K=%d %s  (%d tiles of 256x256, %.2f rounds of 256 CUs; %d launches per timing, %d interleaved rounds; TFLOP/s median [min..max])'
        % (M, N, K, str(dtype)[6:], (M // 256) * (N // 256), (M // 256) * (N // 256) / 256.0, it, rounds))
    for kind in kinds:
        A, B = make(M, N, K, dtype, kind)
        C = torch.empty(M, N, device='cuda', dtype=dtype)
        Bt = B.t()
        fns = [('hipBLASLt torch.mm', lambda: torch.mm(A, Bt, out=C))]
        for v in variants:
            for gm in gms:
                fns.append(('v%d %s gm=%d' % (v, VAR.get(v) or ABL[v], gm), (lambda v=v, gm=gm: gemm(v, A, B, C, gm))))
        for _, f in fns:
            f()
        torch.cuda.synchronize()
        res = {n: [] for n, _ in fns}
        for _ in range(rounds):
            for n, f in fns:
                res[n].append(flop / time_fn(f, it) / 1e12)
        for n, _ in fns:
            r = sorted(res[n])
            say('   %-5s %-40s %7.1f  [%7.1f .. %7.1f]' % (kind, n, r[len(r) // 2], r[0], r[-1]))


def main():
    say('device: %s' % torch.cuda.get_device_name(0))
    ok = check()
    say('correctness: %s' % ('ALL OK' if ok else 'FAILURES ABOVE'))
    if len(sys.argv) > 2 and sys.argv[2] == 'ablate':
        # what bounds the phase program: clusters of 32 MFMAs (half the barriers), and the loop without its LDS-DMA / fragment reads
        bench(4096, 4096, 4096, torch.float16, kinds=('rand',), variants=(2, 7, 8, 9, 10), rounds=3)
        bench(8192, 8192, 8192, torch.float16, kinds=('rand',), variants=(2, 7, 8, 9, 10), rounds=2)
        bench(64 * 30 * 30, 512, 4608, torch.float16, kinds=('relu',), variants=(2, 7, 8, 9, 10), rounds=3)
        with open(sys.argv[1], 'w') as f:
            f.write('\n'.join(out_lines) + '\n')
        return
    bench(4096, 4096, 4096, torch.float16)
    bench(4096, 4096, 4096, torch.bfloat16, kinds=('rand',))
    bench(8192, 8192, 8192, torch.float16, rounds=2)
    bench(8192, 8192, 8192, torch.bfloat16, kinds=('rand',), variants=(0, 1), rounds=2)
    bench(8192, 8192, 8192, torch.float16, kinds=('rand',), variants=(0,), rounds=2, gms=(1, 2, 8, 16))
    # the 3x3 stack's implicit-GEMM shapes (frame-linear pixels without halo rows x couts x 9 Cin)
    bench(64 * 30 * 32, 512, 4608, torch.float16, kinds=('rand', 'relu'), gms=(4, 8))
    bench(64 * 60 * 62, 256, 2304, torch.float16, kinds=('rand', 'relu'), gms=(4, 8))
    if len(sys.argv) > 1:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
        with open(sys.argv[1], 'w') as f:
            f.write('\n'.join(out_lines) + '\n')


if __name__ == '__main__':
    main()
