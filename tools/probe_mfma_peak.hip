// Sustained MFMA rate probe (gfx950): register-only v_mfma_f32_16x16x32_bf16 loops, no memory traffic.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_peak.hip -o /tmp/probe_mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(float* out, int iters) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x & 7); b[j] = (__bf16)(float)(j + 1); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe32(float* out, int iters) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x & 7); b[j] = (__bf16)(float)(j + 1); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
}
template <int WAVES> void run32(float* out, int blocks, int iters) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe32<WAVES><<<blocks, WAVES * 64>>>(out, 1000); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); probe32<WAVES><<<blocks, WAVES * 64>>>(out, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double flops = (double)blocks * WAVES * iters * 8 * 32768.0;
    printf("32x32x16: %d waves/CU, %8d iters: %7.2f ms  %7.1f TFLOP/s\n", WAVES, iters, ms, flops / (ms * 1e-3) / 1e12);
}
// same loop with pseudo-random operand bits (16 different fragments per wave): data toggling costs power, and the sustained
// clock under a real-data MFMA load is what bounds GEMM kernels in practice.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe_rand(float* out, int iters) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ar[4], br[4];
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v[j] = (x & 0x807f807fu) | 0x3f003f00u; }   // bf16 pairs in [0.5, 2)
        ar[i] = (u32x4){v[0], v[1], v[2], v[3]}; br[i] = (u32x4){v[4], v[5], v[6], v[7]};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ar[i & 3]), __builtin_bit_cast(bf16x8, br[i >> 2]), acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
}
template <int WAVES> void run_rand(float* out, int blocks, int iters) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe_rand<WAVES><<<blocks, WAVES * 64>>>(out, 1000); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); probe_rand<WAVES><<<blocks, WAVES * 64>>>(out, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double flops = (double)blocks * WAVES * iters * 16 * 16384.0;
    printf("random operands: %d waves/CU, %8d iters: %7.2f ms  %7.1f TFLOP/s\n", WAVES, iters, ms, flops / (ms * 1e-3) / 1e12);
}
template <int WAVES> void run(float* out, int blocks, int iters) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe<WAVES><<<blocks, WAVES * 64>>>(out, 1000); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); probe<WAVES><<<blocks, WAVES * 64>>>(out, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double flops = (double)blocks * WAVES * iters * 16 * 16384.0;
    printf("%d waves/CU, %8d iters: %7.2f ms  %7.1f TFLOP/s  (=> %.2f clk per MFMA per SIMD at 2.4 GHz)\n", WAVES, iters, ms, flops / (ms * 1e-3) / 1e12,
           (ms * 1e-3 * 2.4e9) / ((double)iters * 16 * WAVES / 4));
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    run<4>(out, 256, 20000);      // ~ 1 ms
    run<8>(out, 256, 20000);
    run<8>(out, 256, 400000);     // ~ 40 ms: sustained clocks
    run<16>(out, 256, 200000);
    run_rand<8>(out, 256, 20000);
    run_rand<8>(out, 256, 400000);
    run_rand<16>(out, 256, 200000);
    run32<4>(out, 256, 200000);
    run32<8>(out, 256, 200000);
    run32<16>(out, 256, 100000);
    return 0;
}
