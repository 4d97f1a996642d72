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
// conv fragment-read patterns over 64-byte rows with the {0,2,3,1}[(row>>2)&3] chunk swizzle:
//   PAT 0: 16x16x32 operand, lane (fr = l&15, g = l>>4) reads chunk g of row fr + shift
//   PAT 1: 32x32x16 operand, lane (fr = l&31, h = l>>5) reads chunk h of row fr + shift
template <int PAT>
__global__ __launch_bounds__(256) void probe_frag(unsigned* out, int iters, int shift) {
    __shared__ __attribute__((aligned(16))) char smem[32768];
    for (int i = threadIdx.x; i < 8192; i += 256) ((unsigned*)smem)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = PAT == 0 ? (lane & 15) : (lane & 31), c = PAT == 0 ? (lane >> 4) : (lane >> 5);
    const int row = fr + shift;
    const int F = (0x1320 >> (((row >> 2) & 3) * 4)) & 3;
    const char* base = smem + wave * 8192 + row * 64 + ((c ^ F) << 4);
    unsigned acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32x4 v = *(const u32x4*)(base + u * (PAT == 0 ? 1024 : 2048 - 64 * 0));
            acc0 ^= v.x; acc1 ^= v.y; acc2 ^= v.z; acc3 ^= v.w;
        }
        asm volatile("" ::: "memory");
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0 ^ acc1 ^ acc2 ^ acc3;
}
template <int PAT> void run_frag(const char* name, int shift) {
    unsigned* out; (void)hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 40000, blocks = 256;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe_frag<PAT><<<blocks, 256>>>(out, 100, shift); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); probe_frag<PAT><<<blocks, 256>>>(out, iters, shift); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-22s shift %d: %.1f B/clk/CU\n", name, shift, (double)iters * 4 * 256 * 16 / (ms * 1e-3 * 2.4e9));
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
int main() {
    run<0>("b128", 16); run<1>("b64", 8); run<2>("tr16_b64", 8);
    for (int sh = 0; sh < 3; ++sh) run_frag<0>("16x16x32 frag b128", sh);
    for (int sh = 0; sh < 3; ++sh) run_frag<1>("32x32x16 frag b128", sh);
    return 0;
}
