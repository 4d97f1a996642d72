// Sustained register-only MFMA rate by instruction shape and operand type, random operand bits in +-[0.5, 1), one and two waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_shapes.hip -o /tmp/probe_mfma_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// KIND 0: 32x32x16 f16, 1: 32x32x16 bf16, 2: 16x16x32 f16, 3: 16x16x32 bf16.  ZERO: operands all zero bits.
template <int KIND, int WAVES, bool ZERO>
__global__ __launch_bounds__(WAVES * 64) void probe(float* out, int iters) {
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    u32x4 ar[4], br[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            const unsigned f16v = (x & 0x83ff83ffu) | 0x38003800u, bf16v = (x & 0x807f807fu) | 0x3f003f00u;
            ar[i][j] = ZERO ? 0u : ((KIND & 1) ? bf16v : f16v);
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            const unsigned f16w = (x & 0x83ff83ffu) | 0x38003800u, bf16w = (x & 0x807f807fu) | 0x3f003f00u;
            br[i][j] = ZERO ? 0u : ((KIND & 1) ? bf16w : f16w);
        }
    float s = 0.f;
    if constexpr (KIND < 2) {
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ar[i & 3]), __builtin_bit_cast(f16x8, br[i >> 1]), acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ar[i & 3]), __builtin_bit_cast(bf16x8, br[i >> 1]), acc[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) s += acc[i][j];
    } else {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ar[i & 3]), __builtin_bit_cast(f16x8, br[i >> 2]), acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ar[i & 3]), __builtin_bit_cast(bf16x8, br[i >> 2]), acc[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    }
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
}
template <int KIND, int WAVES, bool ZERO> void run(float* out, int iters) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe<KIND, WAVES, ZERO><<<256, WAVES * 64>>>(out, 2000); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); probe<KIND, WAVES, ZERO><<<256, WAVES * 64>>>(out, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double flops = 256.0 * WAVES * iters * (KIND < 2 ? 8 * 32768.0 : 16 * 16384.0);
    const char* names[] = {"32x32x16 f16 ", "32x32x16 bf16", "16x16x32 f16 ", "16x16x32 bf16"};
    printf("%s %s %d waves/SIMD: %7.2f ms  %7.1f TFLOP/s\n", names[KIND], ZERO ? "zeros " : "random", WAVES / 4, ms, flops / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 4, false>(out, 100000); run<1, 4, false>(out, 100000); run<2, 4, false>(out, 100000); run<3, 4, false>(out, 100000);
        run<0, 8, false>(out, 50000); run<1, 8, false>(out, 50000); run<2, 8, false>(out, 50000); run<3, 8, false>(out, 50000);
        run<0, 4, true>(out, 100000); run<3, 4, true>(out, 100000);
    }
    return 0;
}
