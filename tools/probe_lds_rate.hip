// LDS read-rate probe (gfx950): bytes/clk/CU of ds_read_b128, ds_read_b64 and ds_read_b64_tr_b16 from a conflict-free layout.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_lds_rate.hip -o /tmp/probe_lds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[32768];
    for (int i = threadIdx.x; i < 8192; i += 256) ((unsigned*)smem)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    const char* base = smem + wave * 8192;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {          // b128, lane-linear (conflict-free): 1 KB per instruction
                const u32x4 v = *(const u32x4*)(base + ((u & 7) * 1024) + lane * 16);
                acc0 ^= v.x; acc1 ^= v.y; acc2 ^= v.z; acc3 ^= v.w;
            } else if (MODE == 1) {   // b64, lane-linear: 512 B per instruction
                const u32x2 v = *(const u32x2*)(base + ((u & 7) * 1024) + lane * 8);
                acc0 ^= v.x; acc1 ^= v.y;
            } else {                  // b64 transpose read, lane-linear addresses
                const short4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(base + ((u & 7) * 1024) + lane * 8));
                const u32x2 v = __builtin_bit_cast(u32x2, t);
                acc0 ^= v.x; acc1 ^= v.y;
            }
        }
        asm volatile("" ::: "memory");
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0 ^ acc1 ^ acc2 ^ acc3;
}
template <int MODE> void run(const char* name, int bytes_per_lane) {
    unsigned* out; hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 20000, blocks = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<blocks, 256>>>(out, 100); hipDeviceSynchronize();
    hipEventRecord(a); probe<MODE><<<blocks, 256>>>(out, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes_per_cu = (double)iters * 8 * 256 * bytes_per_lane;
    printf("%-10s %8.3f ms  %.1f B/clk/CU at 2.4 GHz (1 workgroup of 4 waves per CU)\n", name, ms, bytes_per_cu / (ms * 1e-3 * 2.4e9));
}
int main() { run<0>("b128", 16); run<1>("b64", 8); run<2>("tr16_b64", 8); return 0; }
