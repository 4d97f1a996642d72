// Store-pattern probe (gfx950): write-only streams of 128-byte pixels with different per-instruction footprints:
//   8 B/lane  -> 16 pixels x 32 B   (one MFMA fragment per store: the old conv epilogue)
//   16 B/lane -> 16 pixels x 64 B   (fragment pairs exchanged with v_permlane16_swap)
//   16 B/lane -> 8 pixels x 128 B   (full cache lines)
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_store_pattern.hip -o /tmp/probe_store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void probe(char* buf, long long npix) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 8 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 8;
    for (long long p0 = wave * 16; p0 < npix; p0 += nw * 16) {      // a wave owns 16 consecutive pixels (2 KB)
        if (MODE == 0) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                *(u32x2*)(buf + (p0 + (lane & 15)) * 128 + ni * 32 + (lane >> 4) * 8) = (u32x2){(unsigned)lane, (unsigned)ni};
        } else if (MODE == 1) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                *(u32x4*)(buf + (p0 + (lane & 15)) * 128 + ni * 64 + (lane >> 4) * 16) = (u32x4){(unsigned)lane, (unsigned)ni, 0u, 0u};
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                *(u32x4*)(buf + (p0 + h * 8 + (lane >> 3)) * 128 + (lane & 7) * 16) = (u32x4){(unsigned)lane, (unsigned)h, 0u, 0u};
        }
    }
}
template <int MODE> void run(const char* name, char* buf, long long npix) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe<MODE><<<2048, 512>>>(buf, npix); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 5; ++i) probe<MODE><<<2048, 512>>>(buf, npix);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-28s %.0f GB/s\n", name, 5.0 * npix * 128 / (ms * 1e-3) / 1e9);
}
int main() {
    const long long npix = 1LL << 24;                                // 2 GiB
    char* buf; (void)hipMalloc(&buf, npix * 128);
    run<0>("8 B/lane, 16 px x 32 B", buf, npix);
    run<1>("16 B/lane, 16 px x 64 B", buf, npix);
    run<2>("16 B/lane, 8 px x 128 B", buf, npix);
    return 0;
}
