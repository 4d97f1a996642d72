// LDS swizzle search (gfx950): measures the ds_read_b128 rate of arbitrary per-lane address patterns and searches the
// GF(2)-linear row -> chunk-XOR maps for 64-byte rows (4 chunks) and 128-byte rows (8 chunks) that keep the MFMA fragment
// reads of a conv/GEMM tile at full LDS rate for row shifts 0, 1, 2 (the three kx taps read one band at row offsets).
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_lds_swizzle.hip -o /tmp/probe_lds_swizzle
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void probe(const int* offs, unsigned* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    for (int i = threadIdx.x; i < 16384; i += 256) ((unsigned*)smem)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = smem + wave * 16384 + offs[lane];
    unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32x4 v = *(const u32x4*)(base + u * 2048);
            a0 ^= v.x; a1 ^= v.y; a2 ^= v.z; a3 ^= v.w;
        }
        asm volatile("" ::: "memory");
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
static int* d_offs; static unsigned* d_out;
static double rate(const int* offs) {
    (void)hipMemcpy(d_offs, offs, 64 * 4, hipMemcpyHostToDevice);
    const int iters = 4000, blocks = 256;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a); probe<<<blocks, 256>>>(d_offs, d_out, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return (double)iters * 4 * 256 * 16 / (ms * 1e-3 * 2.4e9);
}
static int parity(int x) { return __builtin_popcount(x) & 1; }
int main() {
    (void)hipMalloc(&d_offs, 64 * 4); (void)hipMalloc(&d_out, 256 * 256 * 4);
    int offs[64];
    for (int l = 0; l < 64; ++l) offs[l] = l * 16;
    rate(offs);
    printf("lane-linear: %.1f B/clk/CU\n", rate(offs));
    // ---- 64-byte rows: f(row) = (m1.row, m0.row) over row bits 0..4
    for (int frag = 0; frag < 2; ++frag) {           // 0: 16x16x32 (fr=l&15, c=l>>4); 1: 32x32x16 (fr=l&31, c=l>>5 of chunk pair kk=0)
        struct R { double worst; int m0, m1; double r[3]; };
        std::vector<R> res;
        for (int m1 = 0; m1 < 32; ++m1)
            for (int m0 = 0; m0 < 32; ++m0) {
                R r; r.m0 = m0; r.m1 = m1; r.worst = 1e9;
                for (int sh = 0; sh < 3; ++sh) {
                    for (int l = 0; l < 64; ++l) {
                        const int fr = frag == 0 ? (l & 15) : (l & 31), c = frag == 0 ? (l >> 4) : (l >> 5);
                        const int row = fr + sh, f = parity(row & m1) * 2 + parity(row & m0);
                        offs[l] = row * 64 + ((c ^ f) << 4);
                    }
                    r.r[sh] = rate(offs); r.worst = std::min(r.worst, r.r[sh]);
                }
                res.push_back(r);
            }
        std::sort(res.begin(), res.end(), [](const R& a, const R& b) { return a.worst > b.worst; });
        printf("64-B rows, %s fragment: best row->xor maps (m1, m0 = row-bit masks of xor bits 1, 0)\n", frag == 0 ? "16x16x32" : "32x32x16");
        for (int i = 0; i < 6; ++i) printf("  m1=0x%02x m0=0x%02x: shift0 %.1f shift1 %.1f shift2 %.1f\n", res[i].m1, res[i].m0, res[i].r[0], res[i].r[1], res[i].r[2]);
        printf("  worst map: %.1f\n", res.back().worst);
    }
    // ---- 128-byte rows (8 chunks): 16x16x32 fragment reads chunk kk*4 + g (kk = 0); f = 3-bit xor from row bits 0..3 (a few structured candidates)
    {
        const int cand[][3] = {{2, 4, 8}, {1, 2, 4}, {2, 4, 1}, {4, 8, 2}, {8, 4, 2}, {1, 4, 8}, {2, 8, 4}, {3, 4, 8}, {2, 5, 8}, {2, 4, 9}, {6, 4, 8}, {2, 12, 8}};
        for (auto& c : cand) {
            printf("128-B rows masks (b0,b1,b2)=(%d,%d,%d):", c[0], c[1], c[2]);
            for (int kk = 0; kk < 2; ++kk)
                for (int sh = 0; sh < 3; sh += 1) {
                    for (int l = 0; l < 64; ++l) {
                        const int row = (l & 15) + sh, g = l >> 4;
                        const int f = parity(row & c[0]) + 2 * parity(row & c[1]) + 4 * parity(row & c[2]);
                        offs[l] = row * 128 + (((kk * 4 + g) ^ f) << 4);
                    }
                    printf(" %.0f", rate(offs));
                }
            printf("\n");
        }
    }
    return 0;
}
