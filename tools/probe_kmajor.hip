// LDS access-pattern probe for K-major weight-gradient tiles (gfx950): rate of ds_read_b128 / ds_read_b64 / ds_write_b64 for
// per-lane byte offsets computed on the host, so that candidate [channel][q] layouts can be checked for bank conflicts.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_kmajor.hip -o /tmp/probe_kmajor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int MODE>   // 0: read b128, 1: read b64, 2: write b64, 3: write b128
__global__ __launch_bounds__(256) void probe(const int* offs, unsigned* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[32768];             // 8 KB per wave: several workgroups per CU
    for (int i = threadIdx.x; i < 8192; i += 256) ((unsigned*)smem)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* base = smem + wave * 8192 + offs[lane];
    unsigned a0 = lane, a1 = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) { const u32x4 v = *(const u32x4*)(base); a0 ^= v.x ^ v.z; a1 ^= v.y ^ v.w; }
            else if (MODE == 1) { const u32x2 v = *(const u32x2*)(base); a0 ^= v.x; a1 ^= v.y; }
            else if (MODE == 2) { *(u32x2*)(base) = (u32x2){a0, a1}; a0 += 1; }
            else { *(u32x4*)(base) = (u32x4){a0, a1, a0, a1}; a0 += 1; }
            asm volatile("" ::: "memory");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1;
}
static int* d_offs; static unsigned* d_out;
template <int MODE> static double rate(const int* offs) {
    (void)hipMemcpy(d_offs, offs, 64 * 4, hipMemcpyHostToDevice);
    const int iters = 2000, blocks = 1024, bytes = (MODE == 0 || MODE == 3) ? 16 : 8;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe<MODE><<<blocks, 256>>>(d_offs, d_out, 10);
    (void)hipEventRecord(a); probe<MODE><<<blocks, 256>>>(d_offs, d_out, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return (double)iters * 4 * 256 * bytes * (blocks / 256.0) / (ms * 1e-3 * 2.4e9);
}
// candidate layouts: [channel][64 q] rows of 128 B (+ pad), 16-byte chunk swizzle f(c)
static int lay(int variant, int c, int q) {
    const int chunk = q >> 3, within = (q & 7) * 2;
    switch (variant) {
        case 0: return c * 128 + (((chunk ^ c ^ (c >> 3)) & 7) << 4) + within;           // the archived kernel
        case 1: return c * 144 + chunk * 16 + within;                                    // pad 16 B per row
        case 2: return c * 128 + (((chunk ^ c) & 7) << 4) + within;
        case 3: return c * 128 + (((chunk ^ (c >> 1)) & 7) << 4) + within;
        case 4: return c * 128 + (((chunk ^ (c >> 2)) & 7) << 4) + within;
        case 5: return c * 136 + chunk * 16 + within;                                    // pad 8 B per row (b128 unaligned? no: 8-byte aligned only)
        case 6: return c * 128 + (((chunk + c) & 7) << 4) + within;
        default: return c * 128 + chunk * 16 + within;
    }
}
int main() {
    (void)hipMalloc(&d_offs, 64 * 4); (void)hipMalloc(&d_out, 256 * 256 * 4);
    int offs[64];
    for (int l = 0; l < 64; ++l) offs[l] = l * 16;
    printf("reference: linear b128 read %.1f  b64 read %.1f  b64 write %.1f B/clk/CU\n", rate<0>(offs), ({for (int l = 0; l < 64; ++l) offs[l] = l * 8; rate<1>(offs);}), rate<2>(offs));
    for (int v = 0; v <= 7; ++v) {
        if (v == 5) continue;
        // fragment read (16x16x32): lane (n = l & 15, g = l >> 4): channel n (+ 16 mi), q = 8 g  -> b128
        for (int l = 0; l < 64; ++l) offs[l] = lay(v, l & 15, 8 * (l >> 4)) % 8192;
        const double r128 = rate<0>(offs);
        // fragment read (32x32x16): lane (n = l & 31, h = l >> 5): channel n, q = 8 h
        for (int l = 0; l < 64; ++l) offs[l] = lay(v, l & 31, 8 * (l >> 5)) % 8192;
        const double r128b = rate<0>(offs);
        // run tail: b64 at q = 8 g + 8
        for (int l = 0; l < 64; ++l) offs[l] = lay(v, l & 15, 8 * (l >> 4) + 8) % 8192;
        const double r64 = rate<1>(offs);
        // transposed write: lane -> channel quad cq = (l & 7) | ((l >> 4) << 3), row group rg = (l >> 3) & 1 (+ 2 wave): channel 4 cq + i, q = 4 rg
        double w64 = 1e9;
        for (int i = 0; i < 4; ++i) {
            for (int l = 0; l < 64; ++l) { const int cq = (l & 7) | ((l >> 4) << 3), rg = (l >> 3) & 1; offs[l] = lay(v, (4 * cq + i) & 63, 4 * rg) % 8192; }
            const double w = rate<2>(offs); if (w < w64) w64 = w;
        }
        // alternative write mapping: lanes along channels (cq = l & 31), rg = l >> 5
        double w64b = 1e9;
        for (int i = 0; i < 4; ++i) {
            for (int l = 0; l < 64; ++l) { const int cq = l & 31, rg = l >> 5; offs[l] = lay(v, (4 * cq + i) & 63, 4 * rg) % 8192; }
            const double w = rate<2>(offs); if (w < w64b) w64b = w;
        }
        printf("layout %d: frag16 b128 %.1f  frag32 b128 %.1f  tail b64 %.1f  write b64 (map A) %.1f  (map B) %.1f\n", v, r128, r128b, r64, w64, w64b);
    }
    // ---- search: chunk' = chunk ^ f(c), f bit j = parity(c & m_j), masks with <= 2 bits of the 6 channel bits
    {
        for (int l = 0; l < 64; ++l) offs[l] = l * 16;
        const double w128ref = rate<3>(offs), r128ref = rate<0>(offs);
        printf("reference b128 write %.1f read %.1f\n", w128ref, r128ref);
        int masks[22], nm = 0;
        for (int a = 0; a < 6; ++a) { masks[nm++] = 1 << a; for (int b = a + 1; b < 6; ++b) masks[nm++] = (1 << a) | (1 << b); }
        auto par = [](int x) { return __builtin_popcount(x) & 1; };
        struct Best { double score; int m0, m1, m2; double r16, r32, w1, w2; } best[8];
        int nb = 0;
        for (int i0 = 0; i0 < nm; ++i0) for (int i1 = 0; i1 < nm; ++i1) for (int i2 = 0; i2 < nm; ++i2) {
            const int m0 = masks[i0], m1 = masks[i1], m2 = masks[i2];
            auto L = [&](int c, int q) { const int f = par(c & m0) | (par(c & m1) << 1) | (par(c & m2) << 2); return (c * 128 + ((((q >> 3) ^ f) & 7) << 4) + (q & 7) * 2) % 8192; };
            for (int l = 0; l < 64; ++l) offs[l] = L(l & 15, 8 * (l >> 4));
            const double r16 = rate<0>(offs);
            if (r16 < 0.9 * r128ref) continue;
            double w1 = 1e9, w2 = 1e9;
            for (int c = 0; c < 8; c += 3) {      // channels 0, 3, 6 of the octet
                for (int l = 0; l < 64; ++l) offs[l] = L((8 * (l & 15) + c) & 63, 8 * (l >> 4));
                w1 = fmin(w1, rate<3>(offs));
                for (int l = 0; l < 64; ++l) offs[l] = L((8 * (l >> 2) + c) & 63, 8 * (l & 3));
                w2 = fmin(w2, rate<3>(offs));
            }
            const double w = fmax(w1, w2);
            if (w < 0.85 * w128ref) continue;
            for (int l = 0; l < 64; ++l) offs[l] = L(l & 31, 8 * (l >> 5));
            const double r32 = rate<0>(offs);
            if (nb < 8) best[nb++] = {r16 + w, m0, m1, m2, r16, r32, w1, w2};
            printf("f = (%02x, %02x, %02x): frag16 %.1f frag32 %.1f write-b128 map1 %.1f map2 %.1f\n", m0, m1, m2, r16, r32, w1, w2);
            fflush(stdout);
            if (nb >= 8) { i0 = i1 = i2 = nm; }
        }
    }
    return 0;
}
