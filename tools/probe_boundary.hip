// What does a dependent kernel boundary cost on this box, by the shape of the kernel being launched?  profiles/r04_f16_timeline.txt shows a
// 10-12 us gap in front of every big MFMA kernel and ~0 in front of the small ones -- under rocprofv3.  Here every kernel stamps
// s_memrealtime (100 MHz) at the entry of its first and the exit of its last workgroup (atomicMin / atomicMax into a stamp table), so the
// gaps are measured WITHOUT a profiler; run the same binary under `rocprofv3 --kernel-trace` to see what the profiler adds.
// Chain: [small kernel] -> [kernel under test] repeated; the kernel under test varies in dynamic LDS size, block size, grid size,
// and in whether its predecessor left dirty lines.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_boundary.hip -o tools/labbin/probe_boundary
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Stamp { unsigned long long t0, t1; };

__global__ void work_kernel(Stamp* st, int idx, int spin_ticks, float* dirty, size_t dirty_elems) {
    extern __shared__ char smem[];
    const unsigned long long t = wall_clock64();
    if (threadIdx.x == 0) atomicMin(&st[idx].t0, t);
    if (dirty) for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < dirty_elems; i += (size_t)gridDim.x * blockDim.x) dirty[i] = (float)i;
    while (wall_clock64() - t < (unsigned long long)spin_ticks) { if (smem[0] == 123 && spin_ticks < 0) st[idx].t1 = 0; }
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&st[idx].t1, wall_clock64());
}

struct Cfg { const char* name; int grid, block, lds; size_t dirty_mb; };

int main() {
    const int REP = 200;
    Stamp* st; CK(hipMalloc(&st, sizeof(Stamp) * 2 * REP));
    float* dirty; CK(hipMalloc(&dirty, 256ull << 20));
    CK(hipFuncSetAttribute((const void*)work_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const Cfg cfgs[] = {
        {"256 x 256 thr, no LDS", 256, 256, 0, 0},
        {"256 x 256 thr, 32 KB LDS", 256, 256, 32 << 10, 0},
        {"256 x 256 thr, 64 KB LDS", 256, 256, 64 << 10, 0},
        {"256 x 256 thr, 80 KB LDS", 256, 256, 80 << 10, 0},
        {"256 x 512 thr, 128 KB LDS", 256, 512, 128 << 10, 0},
        {"256 x 512 thr, 160 KB LDS", 256, 512, 160 << 10, 0},
        {"256 x 512 thr, no LDS", 256, 512, 0, 0},
        {"2048 x 256 thr, no LDS", 2048, 256, 0, 0},
        {"65536 x 64 thr, no LDS", 65536, 64, 0, 0},
        {"256 x 512 thr, 128 KB LDS, predecessor dirtied 16 MB", 256, 512, 128 << 10, 16},
        {"256 x 512 thr, 128 KB LDS, predecessor dirtied 128 MB", 256, 512, 128 << 10, 128},
        {"256 x 256 thr, no LDS, predecessor dirtied 128 MB", 256, 256, 0, 128},
    };
    printf("%-58s %10s %10s %10s %12s\n", "kernel under test (behind a 256 x 256-thread small kernel)", "gap med", "gap min", "gap max", "wall / pair");
    for (const Cfg& c : cfgs) {
        std::vector<Stamp> h(2 * REP);
        for (auto& s : h) { s.t0 = ~0ull; s.t1 = 0; }
        CK(hipMemcpy(st, h.data(), sizeof(Stamp) * 2 * REP, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int warm = 0; warm < 2; ++warm) {
            if (warm) CK(hipEventRecord(e0));
            for (int i = 0; i < REP; ++i) {
                // predecessor: 5 us of spinning (+ optional dirty lines); under test: 10 us
                hipLaunchKernelGGL(work_kernel, dim3(256), dim3(256), 0, 0, st, 2 * i, 500, c.dirty_mb ? dirty : nullptr, (c.dirty_mb << 20) / 4);
                hipLaunchKernelGGL(work_kernel, dim3(c.grid), dim3(c.block), c.lds, 0, st, 2 * i + 1, 1000, (float*)nullptr, (size_t)0);
            }
            if (warm) CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            if (!warm) CK(hipMemcpy(st, h.data(), sizeof(Stamp) * 2 * REP, hipMemcpyHostToDevice));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), st, sizeof(Stamp) * 2 * REP, hipMemcpyDeviceToHost));
        std::vector<double> gap, gap2;
        for (int i = 1; i < REP; ++i) {
            gap.push_back((double)(long long)(h[2 * i + 1].t0 - h[2 * i].t1) * 0.01);          // small -> under test (us)
            gap2.push_back((double)(long long)(h[2 * i].t0 - h[2 * i - 1].t1) * 0.01);         // under test -> small
        }
        std::sort(gap.begin(), gap.end()); std::sort(gap2.begin(), gap2.end());
        printf("%-58s %8.2f us %8.2f us %8.2f us %9.2f us   (back to the small kernel: %.2f us)\n", c.name, gap[gap.size() / 2], gap.front(), gap.back(),
               ms * 1e3 / REP, gap2[gap2.size() / 2]);
    }
    // ---- the real step's pattern: the host enqueues kernels ONE CALL AT A TIME with ~30 us of host work between the calls (ctypes + Python), the
    // GPU is the bottleneck (each kernel runs 100 us), so the queue is never empty.  profiles/r05_f16_timeline.txt (rocprofv3) shows ~10 us in
    // front of the first kernel of every host call and 0 between two kernels one call enqueues back to back.  Stamps = truth.
    {
        const int REP2 = 60;
        std::vector<Stamp> h(3 * REP2);
        for (auto& s : h) { s.t0 = ~0ull; s.t1 = 0; }
        Stamp* st3; CK(hipMalloc(&st3, sizeof(Stamp) * 3 * REP2));
        CK(hipMemcpy(st3, h.data(), sizeof(Stamp) * 3 * REP2, hipMemcpyHostToDevice));
        for (int i = 0; i < REP2; ++i) {
            // one "host call": a big-LDS kernel and a small follow-up kernel back to back, then host work
            hipLaunchKernelGGL(work_kernel, dim3(256), dim3(512), 128 << 10, 0, st3, 3 * i, 10000, (float*)nullptr, (size_t)0);
            hipLaunchKernelGGL(work_kernel, dim3(256), dim3(256), 0, 0, st3, 3 * i + 1, 2000, (float*)nullptr, (size_t)0);
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 30.0) {}
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), st3, sizeof(Stamp) * 3 * REP2, hipMemcpyDeviceToHost));
        std::vector<double> g_first, g_second;
        for (int i = 5; i < REP2; ++i) {
            g_first.push_back((double)(long long)(h[3 * i].t0 - h[3 * (i - 1) + 1].t1) * 0.01);      // previous call's last kernel -> this call's first
            g_second.push_back((double)(long long)(h[3 * i + 1].t0 - h[3 * i].t1) * 0.01);           // inside one call
        }
        std::sort(g_first.begin(), g_first.end()); std::sort(g_second.begin(), g_second.end());
        printf("\nhost calls 30 us apart, GPU-bound queue (100 us + 20 us kernels per call):\n");
        printf("  in front of the first kernel of a call (256 x 512 thr, 128 KB LDS): median %.2f us  min %.2f  max %.2f\n", g_first[g_first.size() / 2], g_first.front(), g_first.back());
        printf("  between the two kernels of one call:                               median %.2f us  min %.2f  max %.2f\n", g_second[g_second.size() / 2], g_second.front(), g_second.back());
    }
    return 0;
}
