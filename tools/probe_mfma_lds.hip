// What does the MFMA pipe sustain with ONE fat wave per SIMD in the shape of the ws kernels' K loop?  Per step: 16 v_mfma_f32_32x32x16_f16
// (2 A fragments from registers x 8 B fragments), the next step's 8 B fragments read from LDS (ds_read_b128) behind the first 8 MFMAs,
// lgkmcnt(0) at the step end, optionally a workgroup barrier every 12 steps.  Random operand bits.  No global memory in the loop.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_lds.hip -o /tmp/probe_mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// MODE 0: MFMAs only (operands in registers); 1: + LDS fragment reads; 2: + barrier every 12 steps; 3: as 2 with TWO waves per SIMD (8 waves, half the accumulators each)
template <int MODE>
__global__ __launch_bounds__(MODE == 3 ? 512 : 256) void probe(float* out, int steps) {
    constexpr int NF = MODE == 3 ? 4 : 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned x = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; ((unsigned*)smem)[i] = (x & 0x83ff83ffu) | 0x38003800u; }   // f16 pairs in [0.5, 2)
    __syncthreads();
    u32x4 wr[2][2], xf[2][NF];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; wr[s][i][j] = (x & 0x83ff83ffu) | 0x38003800u; }
    f32x16 acc[2][NF];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < NF; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
    // fragment read address: 128-byte rows, row = lane & 31 (+ 32 mi), chunk (lane >> 5) ^ ((row >> 1) & 7): the ws kernels' pattern
    const int l31 = lane & 31, h = lane >> 5;
    const int base = ((wave & 3) * 8192) + l31 * 128 + ((h ^ ((l31 >> 1) & 7)) << 4);
#pragma unroll
    for (int mi = 0; mi < NF; ++mi) xf[0][mi] = *(const u32x4*)(smem + ((base + mi * 4096) & 65535));
    for (int st = 0; st < steps; st += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int xs = half;
            const char* xp = smem + ((base + ((st + half) & 3) * 32) & 65535);
#pragma unroll
            for (int k = 0; k < 2 * NF; ++k) {
                acc[k / NF][k % NF] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wr[xs][k / NF]), __builtin_bit_cast(f16x8, xf[xs][k % NF]),
                                                                           acc[k / NF][k % NF], 0, 0, 0);
                if (MODE >= 1 && k < NF) xf[xs ^ 1][k] = *(const u32x4*)(xp + ((k * 4096) & 32767));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 0) {
#pragma unroll
                for (int mi = 0; mi < NF; ++mi) xf[xs ^ 1][mi] = xf[xs][mi];
            }
            if (MODE >= 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (MODE >= 2 && ((st + half) % 12) == 11) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < NF; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[ni][mi][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}
template <int MODE> void run(float* out, int steps, const char* what) {
    const int threads = MODE == 3 ? 512 : 256;
    (void)hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe<MODE><<<256, threads, 65536>>>(out, 240); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); probe<MODE><<<256, threads, 65536>>>(out, steps); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double mf = 256.0 * (threads / 64) * steps * (MODE == 3 ? 8 : 16);
    printf("%-64s %7.2f ms  %7.1f TFLOP/s   %.1f clk per MFMA per SIMD at 2.0 GHz\n", what, ms, mf * 32768.0 / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.0e9 / (mf / 1024.0));
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(out, 48000, "MFMA only, 1 wave/SIMD, 16 per step");
        run<1>(out, 48000, "+ 8 ds_read_b128 per step, lgkmcnt(0) at the step end");
        run<2>(out, 48000, "+ workgroup barrier every 12 steps");
        run<3>(out, 48000, "same, 2 waves/SIMD (8 MFMAs + 4 reads per wave and step)");
    }
    return 0;
}
