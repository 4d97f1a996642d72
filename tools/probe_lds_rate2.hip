// LDS read-rate probe, round 3: the same instructions as tools/probe_lds_rate.hip but with the occupancy and issue depth the
// weight-gradient kernels really have (1, 2 or 4 waves per SIMD; 16 independent reads in flight per wave), and with the row3
// kernel's actual transpose-read address pattern (swz16<128> tile, lane (g, rsub, csub)).
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_lds_rate2.hip -o /tmp/probe_lds_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int swz128(int r, int b) { return r * 256 + (b ^ (((r & 3) << 5) | (((r >> 3) & 1) << 7))); }

template <int MODE>
__global__ __launch_bounds__(1024) void probe(unsigned* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 64 KB
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)smem)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    const char* base = smem + (wave & 3) * 16384;
    const int g = lane >> 4, rsub = (lane & 15) >> 2, csub = (lane & 3) * 8;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            u32x4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *(const u32x4*)(base + u * 1024 + lane * 16);
#pragma unroll
            for (int u = 0; u < 16; ++u) { acc0 ^= v[u].x; acc1 ^= v[u].y; acc2 ^= v[u].z; acc3 ^= v[u].w; }
        } else if (MODE == 1) {
            u32x2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *(const u32x2*)(base + u * 1024 + lane * 8);
#pragma unroll
            for (int u = 0; u < 16; ++u) { acc0 ^= v[u].x; acc1 ^= v[u].y; }
        } else if (MODE == 2) {      // transpose read, lane-linear
            u32x2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                v[u] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(base + u * 1024 + lane * 8)));
#pragma unroll
            for (int u = 0; u < 16; ++u) { acc0 ^= v[u].x; acc1 ^= v[u].y; }
        } else {                      // transpose read, the row3 kernel's pattern: rows 8 g + rsub (+4, +8, ...), channel column u
            u32x2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r0 = 8 * g + rsub + 4 * (u & 1) + 32 * ((u >> 1) & 1), cbyte = ((u >> 2) * 16) * 2 + csub;
                v[u] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(base + swz128(r0, cbyte))));
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) { acc0 ^= v[u].x; acc1 ^= v[u].y; }
        }
        asm volatile("" ::: "memory");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 ^ acc1 ^ acc2 ^ acc3;
}

template <int MODE> void run(const char* name, int bytes_per_lane, int threads) {
    unsigned* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 20000, blocks = 256;
    (void)hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe<MODE><<<blocks, threads, 65536>>>(out, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); probe<MODE><<<blocks, threads, 65536>>>(out, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double bytes_per_cu = (double)iters * 16 * threads * bytes_per_lane;
    printf("%-22s %2d waves/CU %8.3f ms  %6.1f B/clk/CU at 2.4 GHz  (%.2f clk per wave-instruction per CU)\n", name, threads / 64, ms,
           bytes_per_cu / (ms * 1e-3 * 2.4e9), (ms * 1e-3 * 2.4e9) / ((double)iters * 16 * threads / 64));
    (void)hipFree(out);
}
int main() {
    for (int th = 256; th <= 1024; th *= 2) {
        run<0>("b128 linear", 16, th); run<1>("b64 linear", 8, th); run<2>("tr16_b64 linear", 8, th); run<3>("tr16_b64 row3 pattern", 8, th);
    }
    return 0;
}
