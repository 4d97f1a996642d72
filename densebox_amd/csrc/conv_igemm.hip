// Implicit-GEMM convolution on MFMA for gfx950 (framed NHWC, im2col-free).
//
//   D[co][m] = sum_k Wp[co][k] * A[m][k],   k = tap*cin_pad + ci,
//   A[m][k]  = x[pixel(m) + tapoff(tap)][ci]           (read straight from the zero-framed activation)
//
// One 256-thread workgroup (4 waves) owns a BM(pixels) x BN(couts) tile; each wave a 64x64 sub-tile as
// 4x4 MFMA 16x16 fragments.  The K loop advances 128 bytes of K per step (64 f16/bf16 or 32 f32):
// global -> registers (next step, in flight during the MFMAs) -> XOR-swizzled LDS (double buffered, one
// barrier per step) -> ds_read_b128 fragments -> MFMA.  The weight tile is the MFMA "A" operand and
// the pixel tile the "B" operand, so every lane ends up with 4 consecutive output channels of one pixel
// (an 8-byte NHWC store) instead of one channel of 4 pixels.
//
// f16/bf16 use v_mfma_f32_16x16x32_{f16,bf16}; f32 uses v_mfma_f32_16x16x4_f32 four times per 16-byte
// chunk (exact fp32; the parity anchor).  The same kernel serves the 3x3 backbone, the 1x1 heads, the
// un-padded 3x3/5x5 refine convs, the Cin=3/5 layers (SMALLC: one 16-byte chunk per tap) and -- with
// flipped/transposed packed weights -- every dgrad.
#include "common.hpp"
#include <vector>
#include <stdlib.h>
#include <string.h>

struct ConvArgs {
    const char* x;       // framed input, pointing at pixel (0,-pad,-pad) channel c_off
    const char* w;       // packed weights [cout_pad][ktot]
    const float* bias;
    char* y;             // framed output (or fp32 NCHW)
    const char* gate;    // framed, same n/h/w as y
    const unsigned char* dropmask;  // [M][dm_ld]
    int M, HoWo, Wo;     // output pixels
    int x_hp, x_wp, x_ld, x_org;    // framed input dims (pixels), elements per pixel, origin shift (x.pad - cpad)
    int y_hp, y_wp, y_ld, y_pad;
    int g_hp, g_wp, g_ld, g_pad;
    int kw;              // kernel width (taps = kh*kw)
    int ntaps;
    int cpt;             // 16-byte chunks per tap
    int ksteps;          // 128-byte K steps
    int ktot_bytes;      // bytes per packed weight row
    int cout_valid;      // channels actually stored
    int ntile_n;         // cout_pad / BN
    int nblocks;
    int epi;
    int dm_ld;
    unsigned drop_seed;
    // optional second destination: couts >= split_c (a multiple of the N tile) go to y2 with their own epilogue / gate
    char* y2; const char* gate2;
    int y2_hp, y2_wp, y2_ld, y2_pad, g2_hp, g2_wp, g2_ld, g2_pad;
    int split_c, epi2, cout_valid2;
    int rowskip;                    // band kernels: tiles over the frame WITHOUT its top / bottom halo rows (q' = (n H + y) Wp + fx)
    unsigned char* pool_idx;        // fused pooling (EPI2_POOL): arg-max nibbles of the pooled map (dbx_maxpool2x2_idx layout), or null
    // fused stage-2 head convs (ws kernel, EPIK 2: dbx_heads_forward_fused): fragment-order 512 nh -> k weights and the fp32 partial
    // sums [cout tile][output pixel][8] one workgroup tile contributes to its head's <= 8 outputs
    const char* w2f; float* part;
};

// XOR mask (in 16-byte chunks) of a tile row, applied on the LDS-DMA source side and on the fragment reads.
// Round 4: a ds_read_b128 is serviced in four NON-contiguous 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32:
// MI355X_MICROARCH.md, LDS), so a group of the 16x16x32 operand read (lane = 16 g + row) holds all sixteen rows, eight of them with
// chunk g and eight with g + 1.  The maps of rounds 1-3 -- (row >> 1) & 7 on 128-byte rows, {0,2,3,1}[(row >> 2) & 3] on 64-byte rows --
// were derived for contiguous groups and are 2-way conflicts on ~3/4 of those reads (SQ_LDS_BANK_CONFLICT = 47 % of SQ_LDS_IDX_ACTIVE in
// the 64 -> 64 halo kernel; enumeration over every row offset: 7.5 instead of 4 LDS cycles per read).  Exhaustive search over the
// GF(2)-linear row -> chunk maps under the real groups: `row & 6` (128-byte rows) and `(row >> 1) & 2` (64-byte rows) are conflict-free
// at EVERY row offset (the kx = 0, 1, 2 shifted band reads included), and both are still constant over the 2- / 4-row blocks of a
// 256-byte bank row, so the LDS-DMA source side keeps its access pattern.  (The 32x32x16 operand reads of conv3x3_ws.hpp have their own
// map: there (row >> 1) & 7 IS conflict-free.)
template <int BKB> __device__ __forceinline__ int dma_swz(int row) {
#ifdef DBX_OLD_SWZ                   // A/B builds (tools/build_variant.sh): the maps of rounds 1-3
    if (BKB == 128) return (row >> 1) & 7;
    else return (0x1320 >> (((row >> 2) & 3) * 4)) & 3;
#else
    if (BKB == 128) return row & 6;                                          // 2 rows per 256-B bank row
    else return (row >> 1) & 2;                                              // 4 rows per bank row
#endif
}

template <typename T> struct Mma;
template <> struct Mma<_Float16> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<__bf16> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // lane (i=l&15, g=l>>4) holds 4 consecutive k of its row; MFMA j consumes element j of both operands,
    // i.e. the hardware's k index g is mapped to logical k = 4g+j for A and B alike (a consistent permutation).
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
        // NB: bit-cast the whole vector; __builtin_bit_cast(float, a.y) on an element lvalue reads element 0 (hipcc 7.2)
        const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bf.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bf.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bf.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bf.w, c, 0, 0, 0);
    }
};

// Epilogue store helper (16-bit types): fragments ni and ni+1 of one pixel (lane (fr, g) holds couts 4g..4g+3 of each) are
// packed and exchanged across the 16-lane rows with v_permlane16_swap so that every lane owns EIGHT consecutive couts:
// rows g = 0, 2 end up with fragment ni, rows 1, 3 with fragment ni+1, couts (g>>1)*8 .. +7.  One 16-byte store per lane
// then writes 64 contiguous bytes per pixel instead of two 8-byte stores writing 32.  Must be called by all 64 lanes.
template <typename T>
__device__ __forceinline__ u32x4 pair_exchange(const f32x4& va, const f32x4& vb) {
    T pa[4] = {from_f32<T>(va.x), from_f32<T>(va.y), from_f32<T>(va.z), from_f32<T>(va.w)};
    T pb[4] = {from_f32<T>(vb.x), from_f32<T>(vb.y), from_f32<T>(vb.z), from_f32<T>(vb.w)};
    const u32x2 A = *(const u32x2*)pa, B = *(const u32x2*)pb;
    const auto r0 = __builtin_amdgcn_permlane16_swap(A.x, B.x, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(A.y, B.y, false, false);
    return (u32x4){r0[0], r1[0], r0[1], r1[1]};
}
__device__ __forceinline__ int pair_cout_off(int g, int ni) { return (ni + (g & 1)) * 16 + (g >> 1) * 8; }   // elements, ni even
// ReLU gate on the exchanged layout: keep an output where the 16-bit gate value is positive (one 16-byte gate load per lane).
__device__ __forceinline__ u32x4 gate_packed16(const u32x4& v, const u32x4& g) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned lo = ((g[i] & 0xffffu) - 1u) < 0x7fffu ? 0x0000ffffu : 0u;     // 0x0001..0x7fff: positive, non-zero
        const unsigned hi = ((g[i] >> 16) - 1u) < 0x7fffu ? 0xffff0000u : 0u;
        o[i] = v[i] & (lo | hi);
    }
    return o;
}

// HOIST: fetch the bias fragments once up front (a win for the one-tile-per-workgroup v1 kernel; in the persistent DMA
// kernel the early loads make the compiler drain vmcnt in front of the next tile's LDS-DMA issue, so it stays per-fragment).
template <typename T, int MI, int NI, bool HOIST = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[NI][MI], int m0, int n0, int wm_off, int wn_off, int lane) {
    // ---- epilogue: lane holds couts cb + ni*16 + (l>>4)*4 + {0..3} of pixel mb + mi*16 + (l&15)
    const int cb = n0 + wn_off + (lane >> 4) * 4;
    const int mb = m0 + wm_off + (lane & 15);
    const bool second = a.split_c > 0 && n0 >= a.split_c;               // uniform per workgroup
    const int epi = second ? a.epi2 : a.epi;
    char* const ybase = second ? a.y2 : a.y;
    const char* const gbase = second ? a.gate2 : a.gate;
    const int y_hp = second ? a.y2_hp : a.y_hp, y_wp = second ? a.y2_wp : a.y_wp, y_ld = second ? a.y2_ld : a.y_ld, y_pad = second ? a.y2_pad : a.y_pad;
    const int g_hp = second ? a.g2_hp : a.g_hp, g_wp = second ? a.g2_wp : a.g_wp, g_ld = second ? a.g2_ld : a.g_ld, g_pad = second ? a.g2_pad : a.g_pad;
    const int cshift = second ? a.split_c : 0, cvalid = second ? a.split_c + a.cout_valid2 : a.cout_valid;
    f32x4 hbias[HOIST ? NI : 1];
    if constexpr (HOIST) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
            hbias[ni] = ((epi & DBX_EPI_BIAS) && cb + ni * 16 < cvalid) ? *(const f32x4*)(a.bias + cb + ni * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // pixel coordinates: divided out once, then advanced 16 pixels per fragment row
    int n = mb / a.HoWo, r = mb - n * a.HoWo;
    int oy = r / a.Wo, ox = r - oy * a.Wo;
    const int Ho = a.HoWo / a.Wo;
    const bool stepwise = a.Wo >= 16;                       // one row wrap at most per 16-pixel advance
    if constexpr (sizeof(T) == 2 && NI % 2 == 0) {
        // framed 16-bit output with every fragment of this wave inside the valid couts (uniform): fragment pairs are
        // exchanged across the 16-lane rows and leave as 16-byte stores of eight consecutive couts (see pair_exchange)
        if (!(epi & DBX_EPI_F32_NCHW) && n0 + wn_off + NI * 16 <= cvalid) {
            const int g4 = lane >> 4;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = mb + mi * 16;
                if (mi > 0) {
                    if (stepwise) {
                        ox += 16;
                        if (ox >= a.Wo) { ox -= a.Wo; if (++oy == Ho) { oy = 0; ++n; } }
                    } else {
                        n = m / a.HoWo; const int rr = m - n * a.HoWo;
                        oy = rr / a.Wo; ox = rr - oy * a.Wo;
                    }
                }
                const bool ok = m < a.M;                     // per pixel: the four rows of a pixel agree
                T* ypix = (T*)ybase + ((size_t)(n * y_hp + oy + y_pad) * y_wp + (ox + y_pad)) * (size_t)y_ld + (n0 + wn_off - cshift);
                const size_t gpix = ((size_t)(n * g_hp + oy + g_pad) * g_wp + (ox + g_pad)) * (size_t)g_ld;
                auto fin = [&](int ni) {
                    const int c = cb + ni * 16;
                    f32x4 v = acc[ni][mi];
                    if constexpr (HOIST) v += hbias[ni];
                    else if (epi & DBX_EPI_BIAS) v += *(const f32x4*)(a.bias + c);
                    if (epi & DBX_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if ((epi & DBX_EPI_GATE) && ok) {
                        const T* gt = (const T*)gbase + gpix + (c - cshift);
                        v.x = to_f32(gt[0]) > 0.f ? v.x : 0.f; v.y = to_f32(gt[1]) > 0.f ? v.y : 0.f;
                        v.z = to_f32(gt[2]) > 0.f ? v.z : 0.f; v.w = to_f32(gt[3]) > 0.f ? v.w : 0.f;
                    }
                    if ((epi & DBX_EPI_DROPMASK) && ok) {
                        const unsigned int mk = *(const unsigned int*)(a.dropmask + (size_t)m * a.dm_ld + c);
                        v.x = (mk & 0xffu) ? v.x * 2.f : 0.f; v.y = (mk & 0xff00u) ? v.y * 2.f : 0.f;
                        v.z = (mk & 0xff0000u) ? v.z * 2.f : 0.f; v.w = (mk & 0xff000000u) ? v.w * 2.f : 0.f;
                    }
                    if (epi & DBX_EPI_DROPHASH) {
                        const unsigned kb = dbx_drop_bits4(a.drop_seed, (unsigned)m, (unsigned)c >> 2);
                        v.x = (kb & 1u) ? v.x * 2.f : 0.f; v.y = (kb & 2u) ? v.y * 2.f : 0.f;
                        v.z = (kb & 4u) ? v.z * 2.f : 0.f; v.w = (kb & 8u) ? v.w * 2.f : 0.f;
                    }
                    if ((epi & DBX_EPI_ACCUM) && ok) {
                        const T* o = ypix + g4 * 4 + ni * 16;
                        v.x += to_f32(o[0]); v.y += to_f32(o[1]); v.z += to_f32(o[2]); v.w += to_f32(o[3]);
                    }
                    return v;
                };
#pragma unroll
                for (int ni = 0; ni < NI; ni += 2) {
                    const f32x4 v0 = fin(ni), v1 = fin(ni + 1);
                    const u32x4 o = pair_exchange<T>(v0, v1);             // all lanes
                    if (ok) *(u32x4*)(ypix + pair_cout_off(g4, ni)) = o;
                }
            }
            return;
        }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = mb + mi * 16;
        if (mi > 0) {
            if (stepwise) {
                ox += 16; r += 16;
                if (ox >= a.Wo) { ox -= a.Wo; if (++oy == Ho) { oy = 0; ++n; r -= a.HoWo; } }
            } else {
                n = m / a.HoWo; r = m - n * a.HoWo;
                oy = r / a.Wo; ox = r - oy * a.Wo;
            }
        }
        if (m >= a.M) continue;
        const size_t ypix = ((size_t)(n * y_hp + oy + y_pad) * y_wp + (ox + y_pad)) * (size_t)y_ld;
        const size_t gpix = ((size_t)(n * g_hp + oy + g_pad) * g_wp + (ox + g_pad)) * (size_t)g_ld;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int c = cb + ni * 16;
            if (c >= cvalid) continue;
            f32x4 v = acc[ni][mi];
            if constexpr (HOIST) {
                v += hbias[ni];
            } else if (epi & DBX_EPI_BIAS) {
                const f32x4 b = *(const f32x4*)(a.bias + c);
                v += b;
            }
            if (epi & DBX_EPI_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if (epi & DBX_EPI_GATE) {
                const T* g = (const T*)gbase + gpix + (c - cshift);
                v.x = to_f32(g[0]) > 0.f ? v.x : 0.f; v.y = to_f32(g[1]) > 0.f ? v.y : 0.f;
                v.z = to_f32(g[2]) > 0.f ? v.z : 0.f; v.w = to_f32(g[3]) > 0.f ? v.w : 0.f;
            }
            if (epi & DBX_EPI_DROPMASK) {
                const unsigned int mk = *(const unsigned int*)(a.dropmask + (size_t)m * a.dm_ld + c);
                v.x = (mk & 0xffu) ? v.x * 2.f : 0.f; v.y = (mk & 0xff00u) ? v.y * 2.f : 0.f;
                v.z = (mk & 0xff0000u) ? v.z * 2.f : 0.f; v.w = (mk & 0xff000000u) ? v.w * 2.f : 0.f;
            }
            if (epi & DBX_EPI_DROPHASH) {
                const unsigned kb = dbx_drop_bits4(a.drop_seed, (unsigned)m, (unsigned)c >> 2);
                v.x = (kb & 1u) ? v.x * 2.f : 0.f; v.y = (kb & 2u) ? v.y * 2.f : 0.f;
                v.z = (kb & 4u) ? v.z * 2.f : 0.f; v.w = (kb & 8u) ? v.w * 2.f : 0.f;
            }
            if (epi & DBX_EPI_F32_NCHW) {
                float* o = (float*)ybase + ((size_t)n * a.cout_valid + c) * a.HoWo + r;
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < a.cout_valid) {
                        if (epi & DBX_EPI_ACCUM) o[(size_t)j * a.HoWo] += vv[j];
                        else o[(size_t)j * a.HoWo] = vv[j];
                    }
            } else {
                T* o = (T*)ybase + ypix + (c - cshift);
                if (epi & DBX_EPI_ACCUM) {
                    v.x += to_f32(o[0]); v.y += to_f32(o[1]); v.z += to_f32(o[2]); v.w += to_f32(o[3]);
                }
                if constexpr (sizeof(T) == 2) {
                    T p[4] = {from_f32<T>(v.x), from_f32<T>(v.y), from_f32<T>(v.z), from_f32<T>(v.w)};
                    *(u32x2*)o = *(const u32x2*)p;
                } else {
                    *(f32x4*)o = v;
                }
            }
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, bool SMALLC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int WTM = BM / WM, WTN = BN / WN;   // wave tile
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int A_LD = BM / 32, B_LD = BN / 32; // 16-byte chunks per thread per K step
    constexpr int ES = sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                       // [2][BM][128]
    char* Bs = smem + 2 * BM * 128;        // [2][BN][128]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: workgroup b runs on XCD b%8; give each XCD a contiguous run of tiles so that the
    // N-tiles of one pixel tile (same A panel) and neighbouring pixel tiles (shared halo rows) share an L2.
    int bid = blockIdx.x;
    {
        const int q = a.nblocks >> 3, r = a.nblocks & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tile_m = bid / a.ntile_n, tile_n = bid - tile_m * a.ntile_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread load assignment: chunk (tid&7) of rows (tid>>3) + 32*i
    const int lchunk = tid & 7, lrow = tid >> 3;
    const char* arow[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        int m = m0 + lrow + 32 * i;
        m = m < a.M ? m : a.M - 1;
        const int n = m / a.HoWo, r = m - n * a.HoWo;
        const int oy = r / a.Wo, ox = r - oy * a.Wo;
        arow[i] = a.x + ((size_t)(n * a.x_hp + oy + a.x_org) * a.x_wp + (ox + a.x_org)) * (size_t)a.x_ld * ES + lchunk * 16;
    }
    const char* brow[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) brow[i] = a.w + (size_t)(n0 + lrow + 32 * i) * a.ktot_bytes + lchunk * 16;

    const int pix_bytes = a.x_ld * ES;
    const int sw_w = dma_swz<128>(lrow);  // uses row bits 1..3: the same for rows lrow+32*i
    const int lds_w = lrow * 128 + ((lchunk ^ sw_w) << 4);

    u32x4 areg[A_LD], breg[B_LD];
    auto gload = [&](int ks) {
        int aoff;
        if constexpr (SMALLC) {
            int tap = ks * 8 + lchunk;                     // one 16-byte chunk per tap
            tap = tap < a.ntaps ? tap : 0;                 // K padding: weights there are zero
            const int ky = tap / a.kw, kx = tap - ky * a.kw;
            aoff = (ky * a.x_wp + kx) * pix_bytes - lchunk * 16;
        } else {
            const int c0 = ks * 8;                         // first chunk of this step (uniform)
            const int tap = c0 / a.cpt, within = c0 - tap * a.cpt;
            const int ky = tap / a.kw, kx = tap - ky * a.kw;
            aoff = (ky * a.x_wp + kx) * pix_bytes + within * 16;
        }
#pragma unroll
        for (int i = 0; i < A_LD; ++i) areg[i] = *(const u32x4*)(arow[i] + aoff);
#pragma unroll
        for (int i = 0; i < B_LD; ++i) breg[i] = *(const u32x4*)(brow[i] + ks * 128);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) *(u32x4*)(As + buf * BM * 128 + i * 32 * 128 + lds_w) = areg[i];
#pragma unroll
        for (int i = 0; i < B_LD; ++i) *(u32x4*)(Bs + buf * BN * 128 + i * 32 * 128 + lds_w) = breg[i];
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row (l&15) of a 16-row fragment, logical chunk kk*4 + (l>>4)
    const int fr = lane & 15;
    const int c0sw = ((lane >> 4) ^ dma_swz<128>(fr)) << 4;    // kk = 0; kk = 1 flips bit 6 (chunk ^ 4)
    const int rd_a = (wm * WTM + fr) * 128;
    const int rd_b = (wn * WTN + fr) * 128;

    gload(0);
    lstore(0);
    __syncthreads();
    for (int ks = 0; ks < a.ksteps; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < a.ksteps) gload(ks + 1);
        const char* Ab = As + buf * BM * 128 + rd_a;
        const char* Bb = Bs + buf * BN * 128 + rd_b;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int co = c0sw ^ (kk << 6);
            u32x4 wf[NI], xf[MI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const u32x4*)(Bb + ni * 16 * 128 + co);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) xf[mi] = *(const u32x4*)(Ab + mi * 16 * 128 + co);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) Mma<T>::run(wf[ni], xf[mi], acc[ni][mi]);
        }
        if (ks + 1 < a.ksteps) lstore(buf ^ 1);
        __syncthreads();
    }

    conv_epilogue<T, MI, NI, true>(a, acc, m0, n0, wm * WTM, wn * WTN, lane);
}

// ------------------------------------------------------------------------------------------------ v2: LDS-DMA ring
// 512 threads = 8 waves as 4(M) x (BN/64)(N)... each wave still owns a 64x64 output tile.  The A (pixel) and B (weight)
// tiles of one 128-byte K step go global -> LDS directly with global_load_lds_dwordx4 (no VGPR staging, no ds_write
// pass) into a 3-stage ring; loads run two K steps ahead of the MFMAs and are retired with a COUNTED vmcnt, one raw
// s_barrier per step.  Workgroups are PERSISTENT (one per CU) and the load stream runs across tile boundaries, so the
// ring never drains: while a tile's epilogue stores, the first two K steps of the next tile are already in flight:
//      wait(my loads of step ks) ; barrier ; issue loads of step ks+2 into the stage read at ks-1 ; MFMAs of step ks
// The LDS image of a wave-instruction is lane-linear (8 rows x 128 B), so the XOR swizzle that keeps ds_read_b128
// conflict-free is applied to the per-lane SOURCE address (logical chunk = physical chunk ^ ((row>>1)&7)).
// Template: BM x BN tile, BKB bytes of K per row per step (128 or 64), STAGES-deep ring, WM x WN waves (8 total).
//   <256,128,128,3,4,2> / <256,64,128,3,8,1>: 64x64 (32x64) per wave, loads 2 steps ahead
//   <256,256, 64,4,2,4>                     : 128x64 per wave, loads 3 steps ahead; 1.5x fewer staged bytes per FLOP

template <typename T, int BM, int BN, int BKB, int STAGES, int WM, int WN>
__global__ __launch_bounds__(512) void conv_igemm_dma_kernel(const ConvArgs a) {
    static_assert(WM * WN == 8, "8 waves");
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int ES = sizeof(T);
    constexpr int STAGE = (BM + BN) * BKB;
    constexpr int CPRW = BKB / 16;                         // 16-byte chunks per tile row
    constexpr int RPP = 64 / CPRW;                         // tile rows per 1-KiB piece (one wave-instruction)
    constexpr int A_PER_WAVE = BM / RPP / 8;
    constexpr int B_PER_WAVE = BN / RPP / 8;
    constexpr int LOADS = A_PER_WAVE + B_PER_WAVE;
    constexpr int DEPTH = STAGES - 1;                      // K steps in flight ahead of the MFMAs
    constexpr int KK = BKB / 64;                           // MFMA K-slices (64 bytes) per step
    static_assert(A_PER_WAVE >= 1 && B_PER_WAVE >= 1, "tile too small for 8 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane / CPRW, lc = lane % CPRW;
    const int pix_bytes = a.x_ld * ES;
    const int nk = a.ksteps * (128 / BKB);                 // a.ksteps counts 128-byte steps

    // ---- persistent tile schedule.  Workgroup b runs on XCD b%8: give every XCD one contiguous range of tiles and let
    // its workgroups walk that range with stride (workgroups per XCD), so concurrently running tiles are neighbours
    // (the N-tiles of one pixel tile share the A panel, adjacent pixel tiles share halo rows) in ONE L2.
    const int NT = a.nblocks, G = gridDim.x;
    int t_cur, t_end, t_stride;
    if ((G & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, q = NT >> 3, r = NT & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        t_cur = start + j; t_end = start + q + (xcd < r ? 1 : 0); t_stride = G >> 3;
    } else { t_cur = blockIdx.x; t_end = NT; t_stride = G; }
    const int my_tiles = t_cur < t_end ? (t_end - t_cur + t_stride - 1) / t_stride : 0;
    const int total_steps = my_tiles * nk;

    // ---- issue side: per-lane source pointers of the tile being fetched.  Piece p of this wave covers tile rows
    // RPP*(wave*PER_WAVE + p) + lr; the LDS image of a piece is lane-linear, so the swizzle goes on the source.
    const char* arow[A_PER_WAVE];
    const char* brow[B_PER_WAVE];
    int is_tile = t_cur, is_within = 0, is_kx = 0, is_tapoff = 0, is_k = 0, is_stage = 0;
    auto set_tile = [&](int tile) {
        const int tile_m = tile / a.ntile_n, tile_n = tile - tile_m * a.ntile_n;
#pragma unroll
        for (int p = 0; p < A_PER_WAVE; ++p) {
            const int row = RPP * (wave * A_PER_WAVE + p) + lr;
            int m = tile_m * BM + row;
            m = m < a.M ? m : a.M - 1;
            const int n = m / a.HoWo, r = m - n * a.HoWo;
            const int oy = r / a.Wo, ox = r - oy * a.Wo;
            const int chunk = lc ^ dma_swz<BKB>(row);
            arow[p] = a.x + ((size_t)(n * a.x_hp + oy + a.x_org) * a.x_wp + (ox + a.x_org)) * (size_t)a.x_ld * ES + chunk * 16;
        }
#pragma unroll
        for (int p = 0; p < B_PER_WAVE; ++p) {
            const int row = RPP * (wave * B_PER_WAVE + p) + lr;
            const int chunk = lc ^ dma_swz<BKB>(row);
            brow[p] = a.w + (size_t)(tile_n * BN + row) * a.ktot_bytes + chunk * 16;
        }
    };
    auto issue = [&]() {
        const int aoff = is_tapoff + is_within * 16;
        char* sbase = smem + is_stage * STAGE;
#pragma unroll
        for (int p = 0; p < A_PER_WAVE; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(arow[p] + aoff),
                                             (__attribute__((address_space(3))) void*)(sbase + (wave * A_PER_WAVE + p) * 1024), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < B_PER_WAVE; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(brow[p] + is_k * BKB),
                                             (__attribute__((address_space(3))) void*)(sbase + BM * BKB + (wave * B_PER_WAVE + p) * 1024), 16, 0, 0);
        is_stage = is_stage == STAGES - 1 ? 0 : is_stage + 1;
        is_within += CPRW;
        if (is_within == a.cpt) {                           // next tap
            is_within = 0;
            if (++is_kx == a.kw) { is_kx = 0; is_tapoff += (a.x_wp - a.kw + 1) * pix_bytes; }
            else is_tapoff += pix_bytes;
        }
        if (++is_k == nk) {                                 // roll over to this workgroup's next tile (prefetch across tiles)
            is_k = 0; is_within = 0; is_kx = 0; is_tapoff = 0;
            is_tile += t_stride;
            if (is_tile < t_end) set_tile(is_tile);
        }
    };

    // fragment reads: row (l&15) of a 16-row fragment, logical chunk kk*4 + (l>>4)
    const int fr = lane & 15;
    const int c0sw = ((lane >> 4) ^ dma_swz<BKB>(fr)) << 4;   // fragment bases are multiples of 16 rows: mask depends on fr only
    const int rd_a = (wm * WTM + fr) * BKB;
    const int rd_b = BM * BKB + (wn * WTN + fr) * BKB;

    if (my_tiles == 0) return;
    set_tile(t_cur);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < total_steps) issue();
    int stage = 0, gstep = 0;
    for (int tile = t_cur; tile < t_end; tile += t_stride) {
        f32x4 acc[NI][MI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < nk; ++ks, ++gstep) {
            // my loads of this step have landed once only the (up to DEPTH-1) younger steps are still outstanding
            const int younger = total_steps - 1 - gstep;
            if (younger >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * LOADS) : "memory");
            else if (DEPTH > 2 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // everyone's loads landed; everyone left the stage refilled next
            if (gstep + DEPTH < total_steps) issue();
            const char* Ab = smem + stage * STAGE + rd_a;
            const char* Bb = smem + stage * STAGE + rd_b;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int co = c0sw ^ (kk << 6);
                u32x4 wf[NI], xf[MI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const u32x4*)(Bb + ni * 16 * BKB + co);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) xf[mi] = *(const u32x4*)(Ab + mi * 16 * BKB + co);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) Mma<T>::run(wf[ni], xf[mi], acc[ni][mi]);
            }
            stage = stage == STAGES - 1 ? 0 : stage + 1;
        }
        const int tile_m = tile / a.ntile_n, tile_n = tile - tile_m * a.ntile_n;
        conv_epilogue<T, MI, NI>(a, acc, tile_m * BM, tile_n * BN, wm * WTM, wn * WTN, lane);
    }
}

// ------------------------------------------------------------------------------------------------ v3: 3x3 band kernel
// 3x3 / pad 1 layers (the whole backbone, forward and dgrad).  The output tile is BM (256 or 512) CONSECUTIVE pixels of the
// linearised frame (halo pixels included; their results are simply not stored), so the rows a tap needs are the tile's
// own rows shifted by a constant: for a fixed ky the three kx taps read rows r+0, r+1, r+2 of ONE band of BM+2 frame rows.
// One K stage therefore stages a single A band (BM+16 rows x 64 B) plus the weight tiles of the three kx taps and feeds
// 3 x 32 MFMAs per wave per barrier: 32-53 % fewer bytes through the L2 -> LDS path per FLOP than one-tap-per-stage
// (the measured limiter of v2), and a third of the barriers.  Pieces (1 KiB = 16 rows x 64 B) are dealt round-robin to
// the 8 waves; the source-side XOR swizzle and the counted-vmcnt ring are as in v2.
template <typename T, int BM, int BN, int STAGES, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, WM * WN == 4 ? 2 : 1) void conv3x3_band_kernel(const ConvArgs a) {
    constexpr int BKB = 64, AR = BM + 16;
    constexpr int NW = WM * WN;                                         // waves: 8 (two per SIMD) or 16 (four per SIMD, 128 registers each)
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int ES = sizeof(T);
    constexpr int AP = AR / 16, BP = 3 * BN / 16, TOT = AP + BP;      // 1-KiB pieces per stage
    constexpr int NP = (TOT + NW - 1) / NW;                             // per-wave slots
    constexpr int L_LO = TOT / NW, REM = TOT % NW;
    constexpr int STAGE = TOT * 1024;
    constexpr int DEPTH = STAGES - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const bool extra = wave < REM;                                       // this wave owns L_LO + 1 pieces

    int bid = blockIdx.x;
    {
        const int q = a.nblocks >> 3, r = a.nblocks & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tile_m = bid / a.ntile_n, tile_n = bid - tile_m * a.ntile_n;
    const long long q0 = (long long)tile_m * BM;
    const int n0 = tile_n * BN;
    const int pix_bytes = a.x_ld * ES;
    const int cin_bytes = a.cpt * 16;
    const int kc_steps = cin_bytes / BKB;
    const int nk = 3 * kc_steps;

    // ---- per-lane piece sources.  Slot i of this wave is piece pid = wave + 8 i: an A-band piece (rows 16 pid ..) when
    // pid < AP, else weight piece pid-AP = (kx tap, 16 cout rows).
    const int lr = lane >> 2, lc = lane & 3;
    // (uniform 64-bit bases + one 32-bit lane offset per slot: nine 64-bit lane pointers were 18 registers of a kernel at its limit)
    const char* const abase = a.x + (q0 - a.x_wp - 1) * (long long)pix_bytes;
    const char* const wbase = a.w + (size_t)n0 * a.ktot_bytes;
    unsigned src[NP];
    bool is_a[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int pid = wave + NW * i;
        is_a[i] = pid < AP;
        if (pid < AP) {
            const int row = 16 * pid + lr;
            const int chunk = lc ^ dma_swz<BKB>(row);
            // band ky=0 starts one frame row up and one pixel left of the tile (x frame == output frame: x.pad == cpad == 1).
            // rowskip: the tile runs over q' = (n H + y) Wp + fx (no halo rows: 6 % fewer MFMAs at 30x30, 3 % at 60x60); the frame
            // position of q' is q' + (2 img + 1) Wp -- a band row past an image seam simply starts 2 Wp further on, and the taps of
            // valid output pixels never cross a seam (only the dropped halo-column pixels' do)
            long long fq = q0 + row;
            if (a.rowskip) { const long long img = fq / ((long long)(a.x_hp - 2) * a.x_wp); fq += (2 * img + 1) * a.x_wp; }
            src[i] = (unsigned)((fq - q0) * (long long)pix_bytes) + chunk * 16;
        } else {
            const int pb = pid - AP;
            const int kx = pb / (BN / 16), row = 16 * (pb % (BN / 16)) + lr;
            const int chunk = lc ^ dma_swz<BKB>(row);
            src[i] = (unsigned)row * (unsigned)a.ktot_bytes + kx * cin_bytes + chunk * 16;
        }
    }
    int is_ky = 0, is_kc = 0, is_stage = 0;
    // LDS-DMA as inline asm: scalar 64-bit base + ONE 32-bit lane offset (the builtin took a 64-bit lane address per piece: a
    // v_lshl_add_u64 and a register pair each -- spilled once the stage loop was pipelined); every wait on these loads is explicit
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024;
    auto glds = [](const char* sbase64, unsigned voff, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase64), "s"(dst) : "memory");
    };
    auto issue = [&]() {
        const char* const ab = abase + (is_ky * a.x_wp * pix_bytes + is_kc * BKB);      // uniform
        const char* const wb = wbase + (is_ky * 3 * cin_bytes + is_kc * BKB);
        const unsigned sbase = lds0 + is_stage * STAGE;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (i < L_LO || extra) glds(is_a[i] ? ab : wb, src[i], sbase + NW * i * 1024);
        }
        is_stage = is_stage == STAGES - 1 ? 0 : is_stage + 1;
        // ky fastest: the three row bands of one channel chunk are consecutive stages, so the second and third find the rows the
        // first fetched still in L2 (ky-major order re-fetched every row three times from HBM / MALL: PMC traffic 2x)
        if (++is_ky == 3) { is_ky = 0; ++is_kc; }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets.  A: row fr + kx of a 16-row block (block bases are multiples of 16, so the swizzle depends on
    // fr + kx only); B: row fr.
    const int fr = lane & 15, g = lane >> 4;
    int offA[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) offA[kx] = (wm * WTM + fr + kx) * BKB + ((g ^ dma_swz<BKB>(fr + kx)) << 4);
    const int offB = AP * 1024 + (wn * WTN + fr) * BKB + ((g ^ dma_swz<BKB>(fr)) << 4);

    // sched_group_barrier pins the interleave (the scheduler otherwise bunches both taps' reads in front of one
    // lgkmcnt(0)): RPS reads of the next tap after each group of four MFMAs of the current one.
    constexpr int RD = MI + NI, SG = MI * NI / 4, RPS = (RD + SG - 1) / SG, NSGR = (RD + RPS - 1) / RPS;
    u32x4 wf[2][NI], xf[2][MI];
    auto rd = [&](const char* Sb, int kx, u32x4 (&w)[NI], u32x4 (&x)[MI]) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) w[ni] = *(const u32x4*)(Sb + offB + kx * BN * BKB + ni * 16 * BKB);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) x[mi] = *(const u32x4*)(Sb + offA[kx] + mi * 16 * BKB);
    };
    auto mm = [&](u32x4 (&w)[NI], u32x4 (&x)[MI]) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) Mma<T>::run(w[ni], x[mi], acc[ni][mi]);
    };
    // my loads of the awaited stage are done once only n younger stages of mine are outstanding (n <= STAGES - 1)
    auto wait_young = [&](int n) {
        if (extra) {
            switch (n) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * (L_LO + 1)) : "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (L_LO + 1) > 63 ? 63 : 2 * (L_LO + 1)) : "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (L_LO + 1) > 63 ? 63 : 3 * (L_LO + 1)) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (L_LO + 1) > 63 ? 63 : 4 * (L_LO + 1)) : "memory"); break;
            }
        } else {
            switch (n) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * L_LO) : "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L_LO > 63 ? 63 : 2 * L_LO) : "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * L_LO > 63 ? 63 : 3 * L_LO) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * L_LO > 63 ? 63 : 4 * L_LO) : "memory"); break;
            }
        }
    };
    static_assert(STAGES >= 2 && STAGES <= 5, "ring depth");
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < nk) issue();
    int stage = 0;
#ifdef BAND_TS                        // lab builds (tools/band_lab.hip): shader-clock stamps around the stage barrier, a.part = [wg < 64][wave][stage < 64][3] u64
    unsigned long long* const ts = (unsigned long long*)a.part + ((size_t)(blockIdx.x & 63) * NW + wave) * 64 * 3;
#endif
    for (int ks = 0; ks < nk; ++ks) {
        const int younger = nk - 1 - ks;
#ifdef BAND_TS
        const unsigned long long t_a = __builtin_readcyclecounter();
#endif
        wait_young(younger < DEPTH - 1 ? younger : DEPTH - 1);
#ifdef BAND_TS
        const unsigned long long t_w = __builtin_readcyclecounter();
#endif
        __builtin_amdgcn_s_barrier();
#ifdef BAND_TS
        const unsigned long long t_b = __builtin_readcyclecounter();
        if (lane == 0 && blockIdx.x < 64 && ks < 64) { ts[ks * 3] = t_a; ts[ks * 3 + 1] = t_w; ts[ks * 3 + 2] = t_b; }
#endif
        if (ks + DEPTH < nk) issue();
        const char* Sb = smem + stage * STAGE;
        // software pipeline over the three taps: the fragments of tap kx+1 are read while tap kx's MFMAs run
        rd(Sb, 0, wf[0], xf[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (t < 2) rd(Sb, t + 1, wf[(t + 1) & 1], xf[(t + 1) & 1]);
            mm(wf[t & 1], xf[t & 1]);
#pragma unroll
            for (int sg = 0; sg < SG; ++sg) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                if (t < 2 && sg < NSGR) __builtin_amdgcn_sched_group_barrier(0x100, RPS, 0);
            }
        }
        stage = stage == STAGES - 1 ? 0 : stage + 1;
    }

    // ---- epilogue over frame pixels: only interior pixels are stored (the frame of y stays zero).  The frame coordinates
    // are divided out once and then advanced 16 pixels per fragment row; bias is fetched once per column fragment.
    const int cb = n0 + wn * WTN + (lane >> 4) * 4;
    const int epi = a.epi;
    const int fy_lo = a.rowskip ? 1 : 0, fy_hi = a.rowskip ? a.x_hp - 1 : a.x_hp;       // frame rows a tile pixel can sit on: [fy_lo, fy_hi)
    const int fpix = (fy_hi - fy_lo) * a.x_wp, nimg = a.M / a.HoWo;
    f32x4 bias[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
        bias[ni] = ((epi & DBX_EPI_BIAS) && cb + ni * 16 < a.cout_valid) ? *(const f32x4*)(a.bias + cb + ni * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
    int n, fy, fx;
    {
        const long long q = q0 + wm * WTM + (lane & 15);
        n = (int)(q / fpix);
        const int rem = (int)(q - (long long)n * fpix);
        fy = rem / a.x_wp; fx = rem - fy * a.x_wp;
        fy += fy_lo;
    }
    static_assert(NI % 2 == 0, "fragment pairs");
    const int g4 = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        // interior test per pixel (= per fr): the four rows g of a pixel agree, so the row exchange below is safe
        const bool ok = n < nimg && fy >= 1 && fy <= a.x_hp - 2 && fx >= 1 && fx <= a.x_wp - 2;
        const int oy = fy - 1, ox = fx - 1;
        T* ypix = (T*)a.y + (size_t)((n * a.y_hp + oy + a.y_pad) * a.y_wp + (ox + a.y_pad)) * (size_t)a.y_ld + n0 + wn * WTN;
        const T* grow = (const T*)a.gate + (size_t)((n * a.g_hp + oy + a.g_pad) * a.g_wp + (ox + a.g_pad)) * (size_t)a.g_ld + cb;
        auto fin = [&](int ni) {
            f32x4 v = acc[ni][mi] + bias[ni];
            if (epi & DBX_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if constexpr (sizeof(T) == 4) {
                if ((epi & DBX_EPI_GATE) && ok) {
                    const T* gt = grow + ni * 16;
                    v.x = to_f32(gt[0]) > 0.f ? v.x : 0.f; v.y = to_f32(gt[1]) > 0.f ? v.y : 0.f;
                    v.z = to_f32(gt[2]) > 0.f ? v.z : 0.f; v.w = to_f32(gt[3]) > 0.f ? v.w : 0.f;
                }
            }
            if ((epi & DBX_EPI_ACCUM) && ok) {
                const T* o = ypix + g4 * 4 + ni * 16;
                v.x += to_f32(o[0]); v.y += to_f32(o[1]); v.z += to_f32(o[2]); v.w += to_f32(o[3]);
            }
            return v;
        };
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ni = 0; ni < NI; ni += 2) {
                const f32x4 v0 = fin(ni), v1 = fin(ni + 1);
                u32x4 o = pair_exchange<T>(v0, v1);                                // all lanes
                if (ok && cb + ni * 16 < a.cout_valid) {
                    if (epi & DBX_EPI_GATE) o = gate_packed16(o, *(const u32x4*)(grow - g4 * 4 + pair_cout_off(g4, ni)));
                    *(u32x4*)(ypix + pair_cout_off(g4, ni)) = o;
                }
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const f32x4 v = fin(ni);
                if (ok && cb + ni * 16 < a.cout_valid) *(f32x4*)(ypix + g4 * 4 + ni * 16) = v;
            }
        }
        if (a.x_wp >= 16) {                                  // one row wrap at most per 16-pixel advance
            fx += 16;
            if (fx >= a.x_wp) { fx -= a.x_wp; if (++fy == fy_hi) { fy = fy_lo; ++n; } }
        } else {
            const long long q = q0 + wm * WTM + (mi + 1) * 16 + (lane & 15);
            n = (int)(q / fpix);
            const int rem = (int)(q - (long long)n * fpix);
            fy = rem / a.x_wp; fx = rem - fy * a.x_wp;
            fy += fy_lo;
        }
    }
}

#include "conv3x3_ws.hpp"
#ifdef DBX_LAB                                   // superseded experiments, kept for tools/band_lab.hip
#include "../../tools/lab_kernels/conv3x3_pipe.hpp"
#include "../../tools/lab_kernels/conv3x3_pw4.hpp"
#endif

// ------------------------------------------------------------------------------------------------ v4: 64 -> 64 channels, 3x3
// conv1_2 (forward and dgrad) at 240x240: N = 64 couts is too narrow for the band kernel -- every 512-pixel tile re-stages
// its A band six times (3 ky x 2 K chunks) plus the weights: 270 KB of LDS fill per 512 pixels, fill-bound at 540 TFLOP/s.
// Here the weights of ALL nine taps (64 x 576 x 2 B = 72 KB) are loaded into LDS ONCE per persistent workgroup and the
// input is staged as 2-D halo tiles: an 8 x 32 pixel output tile needs 10 x 34 pixels x 128 B = 43.5 KB, read once for all
// taps and the whole K -- 87 KB per 512 pixels.  Two halo buffers: the next tile's LDS-DMA loads run during the whole
// compute of the current one; ONE barrier per tile, none inside its 18 K steps.  Wave w owns tile row w: 32 pixels x 64
// couts, 2 x 4 accumulator fragments, 144 MFMAs per tile.  LDS: 64 x 1184 B weights (rows padded by 32 B: conflict-free
// b128 column reads) + 2 x 43 520 B.  (Four fat waves on 32x32x16 MFMAs -- 1.5x fewer LDS fragment bytes -- measured 40 % slower.)
typedef short short4v __attribute__((ext_vector_type(4)));
struct C64Geo { int tiles_x, tiles_y, ntiles, H, W; const char* x0; int x0_ld; float* partial; float* bpartial; };   // x0 .. : WG1 variant

//
// POOL = true (dbx_conv_forward_pool: conv1_2 -> pool1, DenseBox.py:187): the 2x2/2 max pooling of the output happens in the
// epilogue.  A wave then owns TWO tile rows x 16 pixels (rows 2(w>>1), 2(w>>1)+1; columns 16(w&1)..+15) instead of one row
// x 32, so a pooling window is two accumulator fragments of one lane (vertical) and two neighbouring lanes (horizontal, one
// DPP quad permute); the pooled map costs a quarter of the output's store bytes and the separate pooling pass -- which
// re-read the whole 472 MB map at batch 64 -- disappears.  a.y2 is the pooled destination; EPI2_POOL_ONLY skips the full map.
constexpr int EPI2_POOL = 1 << 20, EPI2_POOL_ONLY = 1 << 21;
__device__ __forceinline__ float dpp_xor1(float v) {          // value of lane ^ 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

//
// WG1 = true (dbx_conv_dgrad_wgrad1: conv1_2's data gradient with conv1_1's WEIGHT gradient folded in, DenseBox.py:185-186
// backwards): the tile of d(conv1_1 output) this kernel has just computed (gated by conv1_1's ReLU) is the dz operand of
// conv1_1's weight gradient and of nothing else, so instead of writing the 472 MB map for a second kernel to re-read, every
// wave rounds its 32-pixel x 64-channel row to T, parks it in the halo buffer it has just finished with, and contracts it
// against the matching pixels of the 8-channel network input (a 10 x 34 halo tile of 16-byte pixels staged next to it) with
// 20 more MFMAs: dW1[co][(tap, c)] += sum_px d[px][co] * x0[px + tap][c], (tap, c) as five 16-column fragments exactly as in
// wgrad3x3_c8_kernel; the unused tenth tap reads a row holding 1.0 in channel 0, so its column IS the bias gradient.  The 80
// accumulator registers live across the persistent workgroup's tiles; at the end the eight waves are summed in a fixed
// order through LDS and each workgroup writes one [64][9][8] slab for wgrad_reduce_kernel.  Nothing is written to a.y.
#ifndef C64_ABL                      // lab builds only: 1 = no weight-fragment reads after the first step, 2 = no pixel-fragment reads, 4 = no stores, 8 = no halo DMA
#define C64_ABL 0
#endif
template <typename T, bool POOL, bool WG1 = false>
__global__ __launch_bounds__(512) void conv3x3_c64_kernel(const ConvArgs a, const C64Geo tg) {
    static_assert(sizeof(T) == 2, "16-bit types");
    static_assert(!(POOL && WG1), "one epilogue variant at a time");
    constexpr int TR = 8, TC = 32, HR = TR + 2, HC = TC + 2, HPX = HR * HC;      // 340 halo pixels of 128 B
    constexpr int WROW = 1152 + 32, W_BYTES = 64 * WROW, IN_BYTES = HPX * 128, IN_STRIDE = IN_BYTES + 512;     // (+32: conflict-free b128 column reads under the real lane groups)
    constexpr int PIECES = (HPX + 7) / 8;                                         // 1-KiB pieces (8 pixels): 43
    constexpr int NP = (PIECES + 7) / 8;                                          // per-wave slots: 6
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ws = smem;
    char* In = smem + W_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- weights: packed rows [cout][tap][cin] of 1152 B -> LDS rows of 1184 B (register-staged, once per workgroup)
    for (int c = tid; c < 64 * 72; c += 512) {
        const int row = c / 72, ch = c - row * 72;
        *(u32x4*)(Ws + row * WROW + ch * 16) = *(const u32x4*)(a.w + (size_t)row * a.ktot_bytes + ch * 16);
    }
    __syncthreads();

    // ---- halo-tile loads: piece = 8 consecutive halo pixels (lane: pixel l>>3, chunk l&7), swizzle on the source chunk
    const int lp = lane >> 3, lc = lane & 7;
    const int pix_bytes = a.x_ld * 2;
    int hr_[NP], hc_[NP];
    bool pv[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int piece = wave + 8 * i;
        const int p = piece * 8 + lp;                                             // halo pixel index
        pv[i] = piece < PIECES;                                                   // (the last piece's pixels 340..343 land in slack)
        const int pc = p < HPX ? p : HPX - 1;
        hr_[i] = pc / HC; hc_[i] = pc - hr_[i] * HC;
    }
    auto issue = [&](int tile, int buf) {
        const int n = tile / (tg.tiles_x * tg.tiles_y), r = tile - n * (tg.tiles_x * tg.tiles_y);
        const int ty = r / tg.tiles_x, tx = r - ty * tg.tiles_x;
        const int y0 = ty * TR, x0 = tx * TC;                                     // output origin == frame origin of the halo tile
        char* dst = In + buf * IN_STRIDE;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (pv[i] && !(C64_ABL & 8)) {
                int fy = y0 + hr_[i]; fy = fy < a.x_hp ? fy : a.x_hp - 1;
                const int p = (wave + 8 * i) * 8 + lp;
                const char* src = a.x + ((size_t)(n * a.x_hp + fy) * a.x_wp + (x0 + hc_[i])) * pix_bytes + ((lc ^ dma_swz<128>(p)) << 4);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + (wave + 8 * i) * 1024), 16, 0, 0);
            }
        }
    };

    // ---- fragment read offsets (tile-independent).  Pixels: halo pixel p = (wave + ky) * 34 + mi * 16 + fr + kx, chunk kc*4+g.
    const int fr = lane & 15, g = lane >> 4;
    int offX[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int p = POOL ? (2 * (wave >> 1) + mi + t / 3) * HC + (wave & 1) * 16 + fr + (t % 3)
                               : (wave + t / 3) * HC + mi * 16 + fr + (t % 3);
            offX[t][mi] = p * 128 + ((g ^ dma_swz<128>(p)) << 4);
        }
    const int offW = fr * WROW + g * 16;

    // WG1: thread t < 340 fetches halo pixel t of the network-input tile; 4 x 5 persistent weight-gradient fragments
    const int x0r = tid / HC, x0c = tid - (tid / HC) * HC;
    f32x4 wacc[WG1 ? 4 : 1][WG1 ? 5 : 1];
#pragma unroll
    for (int i = 0; i < (WG1 ? 4 : 1); ++i)
#pragma unroll
        for (int f = 0; f < (WG1 ? 5 : 1); ++f) wacc[i][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 x0reg = {0u, 0u, 0u, 0u};

    const int first = blockIdx.x, stride = gridDim.x;
    if (first >= tg.ntiles && !WG1) return;
    if (first < tg.ntiles) issue(first, 0);
    const int cb = (lane >> 4) * 4;
    const int epi = a.epi;
    f32x4 bias[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) bias[ni] = (epi & DBX_EPI_BIAS) ? *(const f32x4*)(a.bias + cb + ni * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
    int buf = 0;
    for (int tile = first; tile < tg.ntiles; tile += stride, buf ^= 1) {
        // this tile's halo (issued one tile ago) + older stores.  (Waiting only for the halo -- a counted vmcnt that leaves the previous
        // tile's output stores in flight, as conv3x3_c8_kernel does -- measured 2 % SLOWER here: 331 vs 324 us on the pooled forward.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                             // everyone's pieces landed; everyone left the other buffer
        if (tile + stride < tg.ntiles) issue(tile + stride, buf ^ 1);
        if constexpr (WG1) {
            if (tid < HPX) {
                const int n0 = tile / (tg.tiles_x * tg.tiles_y), r0 = tile - n0 * (tg.tiles_x * tg.tiles_y);
                const int ty0 = r0 / tg.tiles_x, tx0 = r0 - ty0 * tg.tiles_x;
                int fy = ty0 * TR + x0r; fy = fy < a.x_hp ? fy : a.x_hp - 1;
                x0reg = *(const u32x4*)(tg.x0 + ((size_t)(n0 * a.x_hp + fy) * a.x_wp + (tx0 * TC + x0c)) * (size_t)(tg.x0_ld * 2));
            }
        }
        const char* Xb = In + buf * IN_STRIDE;
        f32x4 acc[4][2];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // 18 K steps (tap, 32-channel half), software-pipelined: the six fragments of step s+1 are read while the eight
        // MFMAs of step s run; sched_group_barrier pins the interleave (2 MFMA : 2, 2, 1, 1 reads).
        u32x4 wf[2][4], xf[2][2];
        auto rd = [&](int st, u32x4 (&w)[4], u32x4 (&x)[2]) {
            const int t = st >> 1, kc = st & 1;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) if (!(C64_ABL & 1) || st < 2) w[ni] = *(const u32x4*)(Ws + offW + ni * 16 * WROW + t * 128 + kc * 64);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) if (!(C64_ABL & 2) || st < 2) x[mi] = *(const u32x4*)(Xb + (offX[t][mi] ^ (kc << 6)));
        };
        rd(0, wf[0], xf[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            if (st < 17) rd(st + 1, wf[(st + 1) & 1], xf[(st + 1) & 1]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) Mma<T>::run(wf[st & 1][ni], xf[st & 1][mi], acc[ni][mi]);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (st < 17) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (st < 17) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (st < 17) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if (st < 17) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        // ---- epilogue: wave's tile row, pixel x0 + mi*16 + fr; lane holds couts cb + ni*16 + {0..3}
        const int n = tile / (tg.tiles_x * tg.tiles_y), r = tile - n * (tg.tiles_x * tg.tiles_y);
        const int ty = r / tg.tiles_x, tx = r - ty * tg.tiles_x;
        if constexpr (WG1) {
            // ReLU gate of conv1_1's output in the exchanged layout (one 16-byte load per lane and fragment pair), issued before
            // the barrier so that its latency overlaps the other waves' last MFMAs
            const int oy = ty * TR + wave;
            u32x4 gch[2][2];
            bool okp[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int ox = tx * TC + mi * 16 + fr;
                okp[mi] = oy < tg.H && ox < tg.W;
                const T* gpix = (const T*)a.gate + (size_t)((n * a.g_hp + oy + a.g_pad) * a.g_wp + (ox + a.g_pad)) * (size_t)a.g_ld;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    gch[mi][q] = okp[mi] ? *(const u32x4*)(gpix + pair_cout_off(g, 2 * q)) : (u32x4){0u, 0u, 0u, 0u};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                        // every wave has left this halo buffer
            char* Dw = In + buf * IN_STRIDE + wave * 4096;               // this wave's row: [32 px][64 ch] of T, swizzled
            char* X0s = In + buf * IN_STRIDE + 32768;                    // [340 halo px][8 ch] + the "ones" row
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int px = mi * 16 + fr;
#pragma unroll
                for (int ni = 0; ni < 4; ni += 2) {
                    // all lanes: 8 consecutive channels of pixel fr; a zero gate chunk (pixels outside the image) clears them
                    const u32x4 o = gate_packed16(pair_exchange<T>(acc[ni][mi], acc[ni + 1][mi]), gch[mi][ni >> 1]);
                    *(u32x4*)(Dw + px * 128 + ((pair_cout_off(g, ni) * 2) ^ ((((px >> 1) & 1) << 5) | (((px >> 3) & 1) << 6)))) = o;
                }
            }
            if (tid < HPX) *(u32x4*)(X0s + tid * 16) = x0reg;
            if (tid == HPX) {                                                     // row 340: 1.0 in channel 0 (bias-gradient column)
                T one[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) one[i] = from_f32<T>(i == 0 ? 1.f : 0.f);
                *(u32x4*)(X0s + HPX * 16) = *(const u32x4*)one;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // transpose-read lane roles as in wgrad3x3_c8_kernel: lane L of a 16-lane group supplies row (L>>2), columns (L&3)*4..+3
            const int rsub = (lane & 15) >> 2, cq = lane & 3;
            auto trd = [&](const char* p) {
                return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p));
            };
            u32x4 af[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int cbyte = ni * 32 + cq * 8, ra = 8 * g + rsub, rb = ra + 4;
                const u32x2 lo = trd(Dw + ra * 128 + (cbyte ^ ((((ra >> 1) & 1) << 5) | (((ra >> 3) & 1) << 6))));
                const u32x2 hi = trd(Dw + rb * 128 + (cbyte ^ ((((rb >> 1) & 1) << 5) | (((rb >> 3) & 1) << 6))));
                af[ni] = (u32x4){lo.x, lo.y, hi.x, hi.y};
            }
#pragma unroll
            for (int f = 0; f < 5; ++f) {
                const int tap = 2 * f + (cq >> 1);
                const bool dead = tap >= 9;                                       // the tenth "tap": the ones row, every K
                const int ky = tap / 3, kx = tap - ky * 3;
                const int base = dead ? HPX * 16 + (cq & 1) * 8 : ((wave + ky) * HC + kx) * 16 + (cq & 1) * 8;
                const int r0 = dead ? 0 : (8 * g + rsub) * 16, r1 = dead ? 0 : (8 * g + 4 + rsub) * 16;
                const u32x2 lo = trd(X0s + base + r0), hi = trd(X0s + base + r1);
                const u32x4 bf = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) Mma<T>::run(af[ni], bf, wacc[ni][f]);
            }
            continue;
        }
        if constexpr (POOL) {
            const int oy = ty * TR + 2 * (wave >> 1), ox = tx * TC + (wave & 1) * 16 + fr;      // H, W even: rows oy, oy + 1 together
            const bool ok = ox < tg.W;
            if (oy < tg.H) {                                                     // wave-uniform
                f32x4 v[4][2];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        f32x4 t = acc[ni][mi] + bias[ni];
                        if (epi & DBX_EPI_RELU) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
                        v[ni][mi] = t;
                    }
                if (!(a.epi2 & EPI2_POOL_ONLY)) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        T* ypix = (T*)a.y + (size_t)((n * a.y_hp + oy + mi + a.y_pad) * a.y_wp + (ox + a.y_pad)) * (size_t)a.y_ld;
#pragma unroll
                        for (int ni = 0; ni < 4; ni += 2) {
                            const u32x4 o = pair_exchange<T>(v[ni][mi], v[ni + 1][mi]);
                            if (ok) *(u32x4*)(ypix + pair_cout_off(g, ni)) = o;
                        }
                    }
                }
                T* ppix = (T*)a.y2 + (size_t)((n * a.y2_hp + (oy >> 1) + a.y2_pad) * a.y2_wp + ((ox >> 1) + a.y2_pad)) * (size_t)a.y2_ld;
                if (a.pool_idx && (epi & DBX_EPI_RELU)) {
                    // Training: pooled map + arg-max nibbles for dbx_maxpool2x2_bwd_idx (layout of dbx_maxpool2x2_idx: channel c in byte
                    // c / 2), from the values ROUNDED to T and packed two channels per register.  After the ReLU they are >= +0, so their
                    // bit patterns order like the numbers and the window logic runs on packed unsigned 16-bit halves (v_pk_max_u16 /
                    // v_pk_min_u16): window order (0,0),(0,1),(1,0),(1,1) = (this lane, row 0), (lane ^ 1, row 0), (this lane, row 1),
                    // (lane ^ 1, row 1); row-wise first maxima, then the first of the two rows -- the first maximum in window order, as
                    // ATen; bitwise the nibbles dbx_maxpool2x2_idx takes from the stored map.  The packed maximum is also the pooled
                    // output (max of rounded == rounded max).  ~11 VALU operations per channel instead of 21 on the fp32 values.
                    // (inline asm: written with vector types the compiler turns the min / xor idiom back into per-half compares and
                    // selects, 20 operations per channel)
                    auto pk_max = [](unsigned x, unsigned y) { unsigned d; asm("v_pk_max_u16 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
                    auto pk_min = [](unsigned x, unsigned y) { unsigned d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
                    auto pk_sub = [](unsigned x, unsigned y) { unsigned d; asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
                    auto pk_mad = [](unsigned x, unsigned y, unsigned z) { unsigned d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z)); return d; };
                    const unsigned one = 0x00010001u, two = 0x00020002u, four = 0x00040004u;
                    unsigned char* ipix = a.pool_idx + ((size_t)(n * (tg.H >> 1) + (oy >> 1)) * (tg.W >> 1) + (ox >> 1)) * 32 + g * 2;
                    u32x2 M[4];
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        unsigned int nib[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            unsigned int pk[2];
#pragma unroll
                            for (int mi = 0; mi < 2; ++mi) {
                                const T p2[2] = {from_f32<T>(v[ni][mi][2 * q]), from_f32<T>(v[ni][mi][2 * q + 1])};
                                pk[mi] = *(const unsigned int*)p2 & 0x7fff7fffu;          // (-0 -> +0: keeps the unsigned order)
                            }
                            const unsigned A = pk[0], Cc = pk[1];
                            const unsigned Bn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk[0], 0xB1, 0xF, 0xF, true);
                            const unsigned Dn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk[1], 0xB1, 0xF, 0xF, true);
                            const unsigned t0 = pk_max(A, Bn), t1 = pk_max(Cc, Dn), m = pk_max(t0, t1);
                            const unsigned h0 = pk_min(t0 ^ A, one);            // 1: the neighbour column is strictly larger (row 0)
                            const unsigned h1 = pk_min(t1 ^ Cc, one);           //    ... (row 1)
                            const unsigned r = pk_min(m ^ t0, one);             // 1: row 1 is strictly larger
                            const unsigned pos = pk_min(m, one);
                            const unsigned b0 = pk_mad(r, pk_sub(h1, h0), h0);  // r ? h1 : h0  (mod 2^16)
                            nib[q] = pk_mad(pos, four, pk_mad(r, two, b0));     // halves: channel 2q (low), 2q + 1 (high)
                            M[ni][q] = m;
                        }
                        const unsigned t = nib[0] | (nib[1] << 8);              // x | z << 8   ..   y | w << 8 in the high half
                        const unsigned w16 = (t | (t >> 12)) & 0xffffu;         // x | y << 4 | z << 8 | w << 12
                        if (ok && !(fr & 1)) *(unsigned short*)(ipix + ni * 8) = (unsigned short)w16;
                    }
#pragma unroll
                    for (int ni = 0; ni < 4; ni += 2) {
                        const auto r0 = __builtin_amdgcn_permlane16_swap(M[ni].x, M[ni + 1].x, false, false);      // as pair_exchange
                        const auto r1 = __builtin_amdgcn_permlane16_swap(M[ni].y, M[ni + 1].y, false, false);
                        const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                        if (ok && !(fr & 1)) *(u32x4*)(ppix + pair_cout_off(g, ni)) = o;
                    }
                    continue;
                }
                if (a.pool_idx) {
                    // no ReLU in the epilogue (values of either sign): the same window logic on the fp32 values before rounding
                    unsigned char* ipix = a.pool_idx + ((size_t)(n * (tg.H >> 1) + (oy >> 1)) * (tg.W >> 1) + (ox >> 1)) * 32 + g * 2;
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        unsigned int w16 = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float a0 = v[ni][0][j], c0 = v[ni][1][j];
                            const float b0 = dpp_xor1(a0), d0 = dpp_xor1(c0);
                            int arg = 0;
                            float m = a0;
                            if (b0 > m) { m = b0; arg = 1; }
                            if (c0 > m) { m = c0; arg = 2; }
                            if (d0 > m) { m = d0; arg = 3; }
                            w16 |= (unsigned int)(arg | (m > 0.f ? 4 : 0)) << (4 * j);
                        }
                        if (ok && !(fr & 1)) *(unsigned short*)(ipix + ni * 8) = (unsigned short)w16;
                    }
                }
                // rounding to T is monotonic: max of the f32 values, then rounded == max of the rounded values (dbx_maxpool2x2)
#pragma unroll
                for (int ni = 0; ni < 4; ni += 2) {
                    f32x4 m[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x4 t;
                        t.x = fmaxf(v[ni + q][0].x, v[ni + q][1].x); t.y = fmaxf(v[ni + q][0].y, v[ni + q][1].y);
                        t.z = fmaxf(v[ni + q][0].z, v[ni + q][1].z); t.w = fmaxf(v[ni + q][0].w, v[ni + q][1].w);
                        t.x = fmaxf(t.x, dpp_xor1(t.x)); t.y = fmaxf(t.y, dpp_xor1(t.y));
                        t.z = fmaxf(t.z, dpp_xor1(t.z)); t.w = fmaxf(t.w, dpp_xor1(t.w));
                        m[q] = t;
                    }
                    const u32x4 o = pair_exchange<T>(m[0], m[1]);
                    if (ok && !(fr & 1)) *(u32x4*)(ppix + pair_cout_off(g, ni)) = o;
                }
            }
            continue;
        }
        const int oy = ty * TR + wave;
        if (oy < tg.H) {                                                         // wave-uniform
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int ox = tx * TC + mi * 16 + fr;
                const bool ok = ox < tg.W;
                T* ypix = (T*)a.y + (size_t)((n * a.y_hp + oy + a.y_pad) * a.y_wp + (ox + a.y_pad)) * (size_t)a.y_ld;
                const T* grow = (const T*)a.gate + (size_t)((n * a.g_hp + oy + a.g_pad) * a.g_wp + (ox + a.g_pad)) * (size_t)a.g_ld + cb;
                auto fin = [&](int ni) {
                    f32x4 v = acc[ni][mi] + bias[ni];
                    if (epi & DBX_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if ((epi & DBX_EPI_GATE) && ok) {
                        const T* gt = grow + ni * 16;
                        v.x = to_f32(gt[0]) > 0.f ? v.x : 0.f; v.y = to_f32(gt[1]) > 0.f ? v.y : 0.f;
                        v.z = to_f32(gt[2]) > 0.f ? v.z : 0.f; v.w = to_f32(gt[3]) > 0.f ? v.w : 0.f;
                    }
                    if ((epi & DBX_EPI_ACCUM) && ok) {
                        const T* o = ypix + cb + ni * 16;
                        v.x += to_f32(o[0]); v.y += to_f32(o[1]); v.z += to_f32(o[2]); v.w += to_f32(o[3]);
                    }
                    return v;
                };
#pragma unroll
                for (int ni = 0; ni < 4; ni += 2) {
                    const f32x4 v0 = fin(ni), v1 = fin(ni + 1);
                    const u32x4 o = pair_exchange<T>(v0, v1);                     // all lanes
                    if (ok && !((C64_ABL & 4) && o[0] != 0x12345678u)) *(u32x4*)(ypix + pair_cout_off(g, ni)) = o;
                }
            }
        }
    }
    if constexpr (WG1) {
        // fixed-order sum of the eight waves' fragments through LDS (80 KB at a time), then one slab per workgroup:
        // [64 co][9 taps][8 ci] + the bias column; D layout: lane holds rows (couts) 4 g + r of column lane & 15
        __syncthreads();
        float* red = (float*)smem;
#pragma unroll
        for (int half = 4; half >= 1; half >>= 1) {
            if (wave >= half && wave < 2 * half) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int f = 0; f < 5; ++f) *(f32x4*)(red + (((wave - half) * 20 + ni * 5 + f) * 64 + lane) * 4) = wacc[ni][f];
            }
            __syncthreads();
            if (wave < half) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int f = 0; f < 5; ++f) wacc[ni][f] += *(const f32x4*)(red + ((wave * 20 + ni * 5 + f) * 64 + lane) * 4);
            }
            __syncthreads();
        }
        if (wave == 0) {
            float* P = tg.partial + (size_t)blockIdx.x * 64 * 9 * 8;
            const int col = lane & 15;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int f = 0; f < 5; ++f) {
                    const int tap = 2 * f + (col >> 3), co_b = ni * 16 + (lane >> 4) * 4;
                    const float v[4] = {wacc[ni][f].x, wacc[ni][f].y, wacc[ni][f].z, wacc[ni][f].w};
                    if (tap < 9) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) P[((size_t)(co_b + r) * 9 + tap) * 8 + (col & 7)] = v[r];
                    } else if (col == 8 && tg.bpartial) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) tg.bpartial[(size_t)blockIdx.x * 64 + co_b + r] = v[r];
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------ v5: 3 (8) -> 64 channels, 3x3
// conv1_1 forward: 26 FLOP per byte, bound by writing its 64-channel output (472 MB at batch 64).  Same 8 x 32 halo tiles
// as v4 with one 16-byte chunk per pixel (5.4 KB per tile); a K step of 32 elements covers four taps, so a lane's B
// fragment is the pixel shifted by ITS tap (three K steps for nine taps; taps 9..11 have zero weights and re-read tap 8).
// The wave's weight fragments (4 x 3) live in registers for the whole kernel; two workgroups per CU.
template <typename T>
__global__ __launch_bounds__(512, 2) void conv3x3_c8_kernel(const ConvArgs a, const C64Geo tg) {
    static_assert(sizeof(T) == 2, "16-bit types");
    constexpr int TR = 8, TC = 32, HR = TR + 2, HC = TC + 2, HPX = HR * HC;      // 340 halo pixels of 16 B
    constexpr int IN_BYTES = 6 * 1024;                                            // six 1-KiB pieces (64 pixels each)
    __shared__ __attribute__((aligned(16))) char In[2 * IN_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int pix_bytes = a.x_ld * 2;

    u32x4 wf[3][4];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) wf[ks][ni] = *(const u32x4*)(a.w + (size_t)(ni * 16 + fr) * a.ktot_bytes + (ks * 4 + g) * 16);
    // halo pixel of (tile row `wave`, pixel block mi, lane's tap of K step ks)
    int offX[3][2];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        int tap = ks * 4 + g; tap = tap < 9 ? tap : 8;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) offX[ks][mi] = ((wave + tap / 3) * HC + mi * 16 + fr + tap % 3) * 16;
    }
    auto issue = [&](int tile, int buf) {
        if (wave < 6) {
            const int n = tile / (tg.tiles_x * tg.tiles_y), r = tile - n * (tg.tiles_x * tg.tiles_y);
            const int ty = r / tg.tiles_x, tx = r - ty * tg.tiles_x;
            const int p = wave * 64 + lane, pc = p < HPX ? p : HPX - 1;
            const int hr = pc / HC, hc = pc - hr * HC;
            int fy = ty * TR + hr; fy = fy < a.x_hp ? fy : a.x_hp - 1;
            const char* src = a.x + ((size_t)(n * a.x_hp + fy) * a.x_wp + (tx * TC + hc)) * pix_bytes;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(In + buf * IN_BYTES + wave * 1024), 16, 0, 0);
        }
    };
    const int first = blockIdx.x, stride = gridDim.x;
    if (first >= tg.ntiles) return;
    issue(first, 0);
    const int cb = g * 4;
    const int epi = a.epi;
    f32x4 bias[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) bias[ni] = (epi & DBX_EPI_BIAS) ? *(const f32x4*)(a.bias + cb + ni * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
    int buf = 0;
    bool drained = true;                                   // false: the previous tile left exactly four stores behind this tile's halo load
    for (int tile = first; tile < tg.ntiles; tile += stride, buf ^= 1) {
        // This tile's halo piece (issued one tile ago) must have landed; the four output stores issued AFTER it need not have: vmcnt
        // retires in order, so "at most four outstanding" means the older load is done.  The count is only trusted where it is a
        // lower bound -- a tile whose every lane stores (interior tile, row inside the map: no store is skipped on an empty exec
        // mask); everywhere else the loop waits for everything, as it always did (the write stream no longer stalls every tile)
        if (drained) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tile + stride < tg.ntiles) issue(tile + stride, buf ^ 1);
        const char* Xb = In + buf * IN_BYTES;
        f32x4 acc[4][2];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            u32x4 xf[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) xf[mi] = *(const u32x4*)(Xb + offX[ks][mi]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) Mma<T>::run(wf[ks][ni], xf[mi], acc[ni][mi]);
        }
        const int n = tile / (tg.tiles_x * tg.tiles_y), r = tile - n * (tg.tiles_x * tg.tiles_y);
        const int ty = r / tg.tiles_x, tx = r - ty * tg.tiles_x;
        const int oy = ty * TR + wave;
        if (oy < tg.H) {                                                         // wave-uniform
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int ox = tx * TC + mi * 16 + fr;
                T* ypix = (T*)a.y + (size_t)((n * a.y_hp + oy + a.y_pad) * a.y_wp + (ox + a.y_pad)) * (size_t)a.y_ld;
#pragma unroll
                for (int ni = 0; ni < 4; ni += 2) {
                    f32x4 v0 = acc[ni][mi] + bias[ni], v1 = acc[ni + 1][mi] + bias[ni + 1];
                    if (epi & DBX_EPI_RELU) {
                        v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
                        v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
                    }
                    const u32x4 o = pair_exchange<T>(v0, v1);                     // all lanes
                    if (ox < tg.W) *(u32x4*)(ypix + pair_cout_off(g, ni)) = o;
                }
            }
        }
        drained = !(oy < tg.H && tx * TC + TC <= tg.W);
    }
}

// ------------------------------------------------------------------------------------------------ host
static int64_t packed_k_elems(const dbx_conv_desc* d) {
    const int es = dbx_esize(d->dtype);
    const int64_t kbytes = (int64_t)d->kh * d->kw * d->cin_pad * es;
    return ((kbytes + 127) / 128) * 128 / es;
}

extern "C" int64_t dbx_conv_packed_elems(const dbx_conv_desc* d) {
    // 3x3 layers that may be packed in fragment order: the ws kernel's weight stream over-runs a tile's image by D steps
    const int64_t slack = (d->cout_pad % 128 == 0 && d->cin_pad % 64 == 0 && (d->kh == 3 || d->kh == 1)) ? (int64_t)ws::DMAX * 8192 / dbx_esize(d->dtype) : 0;
    return (int64_t)d->cout_pad * packed_k_elems(d) + slack;
}

template <typename T, int BM, int BN, int WM, int WN, bool SMALLC>
static int launch_conv(const ConvArgs& a, hipStream_t s) {
    constexpr int smem = 2 * (BM + BN) * 128;
    static DbxDevOnce attr_once; int attr_dev = 0;
    if (attr_once.pending(&attr_dev)) {
        DBX_HIP(hipFuncSetAttribute((const void*)conv_igemm_kernel<T, BM, BN, WM, WN, SMALLC>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_once.mark(attr_dev);
    }
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN, SMALLC>), dim3(a.nblocks), dim3(256), smem, s, a);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

template <typename T, int BM, int BN, int BKB, int STAGES, int WM, int WN>
static int launch_conv_dma(const ConvArgs& a, hipStream_t s) {
    constexpr int smem = STAGES * (BM + BN) * BKB;
    static DbxDevOnce attr_once; int attr_dev = 0;
    if (attr_once.pending(&attr_dev)) {
        DBX_HIP(hipFuncSetAttribute((const void*)conv_igemm_dma_kernel<T, BM, BN, BKB, STAGES, WM, WN>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_once.mark(attr_dev);
    }
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        DBX_HIP(hipGetDevice(&dev));
        DBX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int grid = a.nblocks < ncu ? a.nblocks : ncu;            // one persistent workgroup per CU
    hipLaunchKernelGGL((conv_igemm_dma_kernel<T, BM, BN, BKB, STAGES, WM, WN>), dim3(grid), dim3(512), smem, s, a);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

template <typename T, int BM, int BN, int STAGES, int WM, int WN>
static int launch_conv_band(const ConvArgs& a, hipStream_t s) {
    constexpr int smem = STAGES * ((BM + 16) / 16 + 3 * BN / 16) * 1024;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static DbxDevOnce attr_once; int attr_dev = 0;
    if (attr_once.pending(&attr_dev)) {
        DBX_HIP(hipFuncSetAttribute((const void*)conv3x3_band_kernel<T, BM, BN, STAGES, WM, WN>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_once.mark(attr_dev);
    }
    hipLaunchKernelGGL((conv3x3_band_kernel<T, BM, BN, STAGES, WM, WN>), dim3(a.nblocks), dim3(64 * WM * WN), smem, s, a);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

template <typename T>
static int launch_conv_c8(const ConvArgs& a, int n, int h, int w, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        static int ncu = 0;
        if (!ncu) {
            int dev = 0;
            DBX_HIP(hipGetDevice(&dev));
            DBX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        }
        C64Geo tg;
        tg.tiles_x = (w + 31) / 32; tg.tiles_y = (h + 7) / 8; tg.ntiles = n * tg.tiles_x * tg.tiles_y; tg.H = h; tg.W = w;
        const int grid = tg.ntiles < 2 * ncu ? tg.ntiles : 2 * ncu;         // two persistent workgroups per CU
        hipLaunchKernelGGL((conv3x3_c8_kernel<T>), dim3(grid), dim3(512), 0, s, a, tg);
        DBX_LAUNCH_CHECK();
    }
    return DBX_OK;
}

static int c64_wg1_slabs() {         // workgroups (= partial slabs) of the fused dgrad + weight-gradient launch: one per CU
    static int n = 0;
    if (!n) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; }
    return n;
}
template <typename T, bool POOL = false, bool WG1 = false>
static int launch_conv_c64(const ConvArgs& a, int n, int h, int w, hipStream_t s, const C64Geo* extra = nullptr) {
    if constexpr (sizeof(T) == 2) {
        constexpr int smem = 64 * 1184 + 2 * (340 * 128 + 512);
        static_assert(smem <= 160 * 1024, "LDS budget");
        static DbxDevOnce attr_once; int attr_dev = 0;
        if (attr_once.pending(&attr_dev)) {
            DBX_HIP(hipFuncSetAttribute((const void*)conv3x3_c64_kernel<T, POOL, WG1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr_once.mark(attr_dev);
        }
        static int ncu = 0;
        if (!ncu) {
            int dev = 0;
            DBX_HIP(hipGetDevice(&dev));
            DBX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        }
        C64Geo tg;
        tg.x0 = nullptr; tg.x0_ld = 0; tg.partial = nullptr; tg.bpartial = nullptr;
        if (extra) tg = *extra;
        tg.tiles_x = (w + 31) / 32; tg.tiles_y = (h + 7) / 8; tg.ntiles = n * tg.tiles_x * tg.tiles_y; tg.H = h; tg.W = w;
        // one persistent workgroup per CU (the weight-gradient variant: always all of them, every slab gets written)
        const int grid = WG1 ? c64_wg1_slabs() : (tg.ntiles < ncu ? tg.ntiles : ncu);
        hipLaunchKernelGGL((conv3x3_c64_kernel<T, POOL, WG1>), dim3(grid), dim3(512), smem, s, a, tg);
        DBX_LAUNCH_CHECK();
    }
    return DBX_OK;
}

#include "conv3x3_c64p.hpp"
#include "conv3x3_p8.hpp"
#include "heads_gen.hpp"

static int ws_level() {              // DBX_WS=0 keeps the LDS band kernels on every layer, 2 plans ws wherever it can run (A/B testing)
    static int v = -1;
    if (v < 0) { const char* e = getenv("DBX_WS"); v = e ? atoi(e) : 1; }
    return v;
}
static bool ws_enabled() { return ws_level() != 0; }
// problems the pooled epilogue exists for: exactly the halo-tile kernel's (conv3x3_c64_kernel) with even output sizes
template <typename T>
static bool c64_pool_ok(const dbx_conv_desc* d, const dbx_view* x, const dbx_view* y) {
    return sizeof(T) == 2 && d->kh == 3 && d->kw == 3 && d->cpad == 1 && x->pad == 1 && d->cin_pad == 64 && d->cout_pad == 64 &&
           y->c == 64 && x->c >= 64 && !(d->epilogue & ~(DBX_EPI_BIAS | DBX_EPI_RELU)) && y->h % 2 == 0 && y->w % 2 == 0 &&
           x->h == y->h && x->w == y->w && x->n == y->n;
}
static int g_conv_variant_override = -1;   // set by in-tree lab programs that include this file (tools/band_lab.hip)
static int conv_variant() {          // DBX_CONV_VARIANT=1 forces the register-staged v1 kernel everywhere (A/B testing)
    static int v = -1;
    if (g_conv_variant_override >= 0) return g_conv_variant_override;
    if (v < 0) { const char* e = getenv("DBX_CONV_VARIANT"); v = e ? atoi(e) : 0; }
    return v;
}

template <typename T>
static int conv_forward_t(const dbx_conv_desc* d, const dbx_view* x, const void* w, const float* bias,
                          const dbx_view* y, const dbx_view* gate, const uint8_t* dropmask, int dm_ld, hipStream_t s,
                          const dbx_view* y2 = nullptr, const dbx_view* gate2 = nullptr, int split_c = 0, int epi2 = 0,
                          dbx_conv_plan_t* plan = nullptr, void* pool_idx = nullptr, const void* w2_frag = nullptr, float* part = nullptr) {
    constexpr int ES = sizeof(T);
    // One selection path for launching and for dbx_conv_plan(): with `plan` set, the chosen kernel is reported instead of launched.
    static const char* const tname = sizeof(T) == 4 ? "f32" : (DType<T>::id == DBX_F16 ? "f16" : "bf16");
#define DBX_SELECT(ID, TM, TN, NAME, CALL)                                                                   \
    do {                                                                                                     \
        if (plan) {                                                                                          \
            plan->kernel = ID; plan->tile_m = TM; plan->tile_n = TN; plan->w_frag = (ID) == DBX_K_WS ? 1 : 0; \
            snprintf(plan->name, sizeof plan->name, NAME "<%s,%d,%d>", tname, TM, TN);                      \
            return DBX_OK;                                                                                   \
        }                                                                                                    \
        return CALL;                                                                                         \
    } while (0)
    // the ws kernels: the name carries the instantiation's own template arguments <T, WM, KS, EPIK> (a profile symbol is then
    // matched exactly); the tile is in tile_m / tile_n
#define DBX_SELECT_WS(TM, TN, WMV, KSV, EPIKV, CALL)                                                         \
    do {                                                                                                     \
        if (plan) {                                                                                          \
            plan->kernel = DBX_K_WS; plan->tile_m = TM; plan->tile_n = TN; plan->w_frag = 1;                 \
            snprintf(plan->name, sizeof plan->name, "conv3x3_ws_kernel<%s,%d,%d,%d>", tname, WMV, KSV, EPIKV); \
            return DBX_OK;                                                                                   \
        }                                                                                                    \
        return CALL;                                                                                         \
    } while (0)
    const int ho = x->h + 2 * d->cpad - d->kh + 1, wo = x->w + 2 * d->cpad - d->kw + 1;
    DBX_REQUIRE(ho == y->h && wo == y->w && x->n == y->n, "conv: output %dx%d does not match %dx%d", y->h, y->w, ho, wo);
    DBX_REQUIRE(x->pad >= d->cpad, "conv: input frame %d < conv padding %d", x->pad, d->cpad);
    DBX_REQUIRE((d->cin_pad * ES) % 16 == 0 && d->cout_pad % 64 == 0, "conv: cin_pad/cout_pad alignment");
    DBX_REQUIRE(((size_t)x->ptr % 16) == 0 && (x->ld * ES) % 16 == 0 && (x->c_off * ES) % 16 == 0, "conv: x alignment");
    const int cpt = d->cin_pad * ES / 16;
    const bool smallc = cpt < 8;
    DBX_REQUIRE(smallc ? cpt == 1 : cpt % 8 == 0, "conv: cin_pad*esize must be 16 or a multiple of 128 bytes (got %d)", d->cin_pad * ES);
    DBX_REQUIRE(x->c >= (smallc ? 1 : d->cin_pad) && x->c_off + d->cin_pad <= x->ld, "conv: x view too narrow");
    const bool nchw = d->epilogue & DBX_EPI_F32_NCHW;
    if (!nchw) DBX_REQUIRE(y->c % 4 == 0 && ((y->c_off * ES) % 8) == 0 && (y->ld * ES) % 8 == 0, "conv: y alignment");
    DBX_REQUIRE(y->c <= d->cout_pad, "conv: y has more channels than the packed weight");
    if (!plan) {
        if (d->epilogue & DBX_EPI_GATE) DBX_REQUIRE(gate && gate->h == y->h && gate->w == y->w && gate->c >= y->c, "conv: bad gate view");
        if (d->epilogue & DBX_EPI_DROPMASK) DBX_REQUIRE(dropmask && dm_ld % 4 == 0, "conv: bad dropout mask");
        if (d->epilogue & DBX_EPI_BIAS) DBX_REQUIRE(bias != nullptr, "conv: bias missing");
    }

    ConvArgs a;
    a.x = (const char*)x->ptr + (size_t)x->c_off * ES;
    a.w = (const char*)w;
    a.bias = bias;
    a.y = nchw ? (char*)y->ptr : (char*)y->ptr + (size_t)y->c_off * ES;
    a.gate = gate ? (const char*)gate->ptr + (size_t)gate->c_off * ES : nullptr;
    a.dropmask = dropmask;
    a.M = x->n * ho * wo; a.HoWo = ho * wo; a.Wo = wo;
    a.x_hp = x->h + 2 * x->pad; a.x_wp = x->w + 2 * x->pad; a.x_ld = x->ld; a.x_org = x->pad - d->cpad;
    a.y_hp = y->h + 2 * y->pad; a.y_wp = y->w + 2 * y->pad; a.y_ld = y->ld; a.y_pad = y->pad;
    if (gate) { a.g_hp = gate->h + 2 * gate->pad; a.g_wp = gate->w + 2 * gate->pad; a.g_ld = gate->ld; a.g_pad = gate->pad; }
    else { a.g_hp = a.g_wp = a.g_ld = a.g_pad = 0; }
    a.kw = d->kw; a.ntaps = d->kh * d->kw; a.cpt = cpt;
    a.ktot_bytes = (int)(packed_k_elems(d) * ES);
    a.ksteps = a.ktot_bytes / 128;
    a.cout_valid = y->c;
    a.epi = d->epilogue & ~DBX_CONV_WFRAG; a.dm_ld = dm_ld; a.drop_seed = d->drop_seed;
    a.y2 = nullptr; a.gate2 = nullptr; a.split_c = 0; a.epi2 = 0; a.cout_valid2 = 0;
    a.y2_hp = a.y2_wp = a.y2_ld = a.y2_pad = a.g2_hp = a.g2_wp = a.g2_ld = a.g2_pad = 0;
    a.pool_idx = (unsigned char*)pool_idx;
    a.w2f = (const char*)w2_frag; a.part = part;
    a.rowskip = 0;
    bool pool_p8 = false;       // pooled second destination on the 8-phase kernels (round 5: conv2_2 -> pool2, conv3_4 -> pool3; EPIK 3)
    if (y2 && (epi2 & EPI2_POOL) && !c64_pool_ok<T>(d, x, y)) {
        DBX_REQUIRE(sizeof(T) == 2 && d->kh == 3 && d->kw == 3 && d->cpad == 1 && x->pad == 1 && (d->epilogue & DBX_EPI_RELU) &&
                    !(d->epilogue & ~(DBX_EPI_BIAS | DBX_EPI_RELU)) && y->h % 2 == 0 && y->w % 2 == 0 && x->h == y->h && x->w == y->w,
                    "conv pool: needs a 16-bit 3x3/pad 1 layer on congruent frames with even H, W and a (bias +) ReLU epilogue");
        DBX_REQUIRE(y2->n == y->n && y2->h == y->h / 2 && y2->w == y->w / 2 && y2->c == y->c && ((y2->c_off * ES) % 16) == 0 && (y2->ld * ES) % 16 == 0 &&
                    (int64_t)y2->n * (y2->h + 2 * y2->pad) * (y2->w + 2 * y2->pad) * y2->ld < ((int64_t)1 << 32), "conv pool: pooled view must be N x H/2 x W/2 x C, 16-byte aligned");
        a.y2 = (char*)y2->ptr + (size_t)y2->c_off * ES;
        a.y2_hp = y2->h + 2 * y2->pad; a.y2_wp = y2->w + 2 * y2->pad; a.y2_ld = y2->ld; a.y2_pad = y2->pad;
        a.epi2 = epi2;
        pool_p8 = true;
    } else if (y2 && (epi2 & EPI2_POOL)) {
        // pooled second destination of the 64 -> 64 halo-tile kernel
        DBX_REQUIRE(c64_pool_ok<T>(d, x, y), "conv pool: needs a 16-bit 3x3/pad 1 64 -> 64 layer on congruent frames with even H, W and a bias/ReLU epilogue");
        DBX_REQUIRE(y2->n == y->n && y2->h == y->h / 2 && y2->w == y->w / 2 && y2->c == 64 && ((y2->c_off * ES) % 16) == 0 && (y2->ld * ES) % 16 == 0 &&
                    ((y->c_off * ES) % 16) == 0 && (y->ld * ES) % 16 == 0, "conv pool: pooled view must be N x H/2 x W/2 x 64, 16-byte aligned");
        a.y2 = (char*)y2->ptr + (size_t)y2->c_off * ES;
        a.y2_hp = y2->h + 2 * y2->pad; a.y2_wp = y2->w + 2 * y2->pad; a.y2_ld = y2->ld; a.y2_pad = y2->pad;
        a.epi2 = epi2;
        a.ntile_n = 1;
        // the tile loop software-pipelined (conv3x3_c64p.hpp); DBX_CONV_VARIANT=9 keeps the one-episode-per-tile kernel (A/B, tests)
        if (conv_variant() == 9) { if (plan) DBX_SELECT(DBX_K_C64, 256, 64, "conv3x3_c64_kernel", 0); return launch_conv_c64<T, true>(a, x->n, x->h, x->w, s); }
        if (plan) DBX_SELECT(DBX_K_C64, 256, 64, "conv3x3_c64p_kernel", 0);
        return launch_conv_c64p<T, true>(a, x->n, x->h, x->w, s);
    } else if (y2) {
        // split destination: 1x1 GEMM on the 256-wide DMA tiles only
        DBX_REQUIRE(sizeof(T) == 2 && d->kh == 1 && d->kw == 1 && !smallc && split_c > 0 && split_c % 256 == 0 && y->c == split_c &&
                    d->cout_pad % 256 == 0 && split_c + y2->c <= d->cout_pad && y2->c % 4 == 0 && y2->h == y->h && y2->w == y->w && y2->n == y->n &&
                    !(d->epilogue & DBX_EPI_F32_NCHW) && !(epi2 & (DBX_EPI_F32_NCHW | DBX_EPI_DROPMASK | DBX_EPI_DROPHASH | DBX_EPI_BIAS)),
                    "conv split: needs a 1x1 16-bit GEMM, split_c = y.c a multiple of 256, plain/gate/accumulate second epilogue");
        DBX_REQUIRE(((y2->c_off * ES) % 8) == 0 && (y2->ld * ES) % 8 == 0, "conv split: y2 alignment");
        if (epi2 & DBX_EPI_GATE) DBX_REQUIRE(gate2 && gate2->h == y2->h && gate2->w == y2->w && gate2->c >= y2->c, "conv split: bad gate2 view");
        a.y2 = (char*)y2->ptr + (size_t)y2->c_off * ES;
        a.y2_hp = y2->h + 2 * y2->pad; a.y2_wp = y2->w + 2 * y2->pad; a.y2_ld = y2->ld; a.y2_pad = y2->pad;
        if (gate2) {
            a.gate2 = (const char*)gate2->ptr + (size_t)gate2->c_off * ES;
            a.g2_hp = gate2->h + 2 * gate2->pad; a.g2_wp = gate2->w + 2 * gate2->pad; a.g2_ld = gate2->ld; a.g2_pad = gate2->pad;
        }
        a.split_c = split_c; a.epi2 = epi2; a.cout_valid2 = y2->c;
    }
    DBX_REQUIRE(a.M > 0 && (int64_t)x->n * a.x_hp * a.x_wp * x->ld * ES < (int64_t)1 << 40, "conv: empty or oversized input");

    // Round 5: the wide 3x3 layers (couts a multiple of 256, >= 128 input channels, plain / ReLU / ReLU-gate epilogue) on the 8-phase
    // MFMA core (conv3x3_p8.hpp; plain packed weights).  DBX_P8=0: off (the ws / band kernels of rounds 2-4 keep them); 2: every eligible problem.
    {
        static int p8_level = -1;
        if (p8_level < 0) { const char* e = getenv("DBX_P8"); p8_level = e ? atoi(e) : 1; }
        const bool k3 = d->kh == 3 && d->kw == 3 && d->cpad == 1 && x->pad == 1;
        const bool k1 = d->kh == 1 && d->kw == 1 && d->cpad == 0;
        const int kk = d->epilogue & ~(DBX_EPI_BIAS | DBX_CONV_WFRAG);
        const int taps = d->kh * d->kw;
        // the heads' forward GEMM: bias + hash dropout (+ the second 1x1 convs on the tile when the caller hands over their weights)
        const bool heads = k1 && (d->epilogue & ~DBX_CONV_WFRAG) == (DBX_EPI_BIAS | DBX_EPI_DROPHASH);
        bool p8_ok = p8_level != 0 && !smallc && sizeof(T) == 2 && (k3 || k1) && !(d->epilogue & DBX_CONV_WFRAG) && (!y2 || (pool_p8 && k3)) && (!pool_idx || pool_p8) && (heads || !w2_frag) &&
                     (kk == 0 || kk == DBX_EPI_RELU || (kk == DBX_EPI_GATE && !(d->epilogue & DBX_EPI_BIAS)) || heads) && d->cin_pad % 64 == 0 && (taps * (d->cin_pad / 64)) % 2 == 0 &&
                     taps * (d->cin_pad / 64) >= 4 && taps * (d->cin_pad / 64) < 7000 && d->cout_pad % 256 == 0 && y->c == d->cout_pad &&
                     (y->c_off * ES) % 16 == 0 && (y->ld * ES) % 16 == 0 && a.M < (1 << 24) &&
                     (int64_t)x->n * a.x_hp * a.x_wp * x->ld * ES < ((int64_t)1 << 32) && (int64_t)64 * a.ktot_bytes < ((int64_t)1 << 31) &&
                     (int64_t)y->n * a.y_hp * a.y_wp * y->ld < ((int64_t)1 << 32);            // 32-bit byte offsets into x, element offsets into y
        if (p8_ok && (d->epilogue & DBX_EPI_GATE) && gate) p8_ok = (gate->c_off * ES) % 16 == 0 && (gate->ld * ES) % 16 == 0;
        if (p8_ok) {
            static int ncu = 0;
            if (!ncu) { int dev = 0; DBX_HIP(hipGetDevice(&dev)); DBX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev)); }
            P8Args ts;
            p8_ok = p8_schedule(a.M, y->c / 256, ncu, ts) && ts.items >= (ncu * 3) / 4;                 // 7- / 8-unit tiles that fill the chip
        }
        // preferred: every eligible 3x3 layer with >= 128 input channels (same-box A/B per layer: tools/gpu_conv_plan_bench.py), and the long-K
        // 1x1 GEMMs (the heads' 2048 -> 512 data gradient at 30 x 30: 150 -> 108 us; tools/gpu_conv1x1_bench.py)
        // ... and the heads' forward GEMM with its fused epilogue (DBX_P8_HEADS=0: the ws kernel keeps it)
        static int p8_heads = -1;
        if (p8_heads < 0) { const char* e = getenv("DBX_P8_HEADS"); p8_heads = e ? atoi(e) : 1; }
        const bool p8_pref = (k3 && d->cin_pad >= 128) || (k1 && !heads && d->cin_pad >= 1024) || (heads && p8_heads != 0);
        if (p8_ok && (p8_pref || (p8_level >= 2 && !heads))) {
            a.ntile_n = y->c / 256;
            if (plan) {
                plan->kernel = DBX_K_P8; plan->tile_m = 256; plan->tile_n = 256; plan->w_frag = 0;
                if (heads) snprintf(plan->name, sizeof plan->name, "conv3x3_p8_kernel<%s,1,1>", tname);      // <T, KS, EPIK>
                else snprintf(plan->name, sizeof plan->name, "conv3x3_p8_kernel<%s,%d>", tname, k3 ? 3 : 1);
                return DBX_OK;
            }
            if (heads) return launch_conv_p8<T, 1, 1>(a, s);
            if (pool_p8) return launch_conv_p8<T, 3, 3>(a, s);
            if (kk == DBX_EPI_GATE) return k3 ? launch_conv_p8<T, 3, 2>(a, s) : launch_conv_p8<T, 1, 2>(a, s);
            return k3 ? launch_conv_p8<T, 3>(a, s) : launch_conv_p8<T, 1>(a, s);
        }
        // 128-cout layers (conv2_2 forward / data gradient, conv3_1's data gradient): 512-pixel x 128-cout tiles on the p8w core (DBX_P8W=0: off)
        static int p8w_on = -1;
        if (p8w_on < 0) { const char* e = getenv("DBX_P8W"); p8w_on = e ? atoi(e) : 1; }
        bool p8w_ok = p8_level != 0 && p8w_on != 0 && !smallc && sizeof(T) == 2 && k3 && !(d->epilogue & DBX_CONV_WFRAG) && (!y2 || pool_p8) && (!pool_idx || pool_p8) && !w2_frag &&
                      (kk == 0 || kk == DBX_EPI_RELU || (kk == DBX_EPI_GATE && !(d->epilogue & DBX_EPI_BIAS))) && d->cin_pad % 128 == 0 &&
                      9 * (d->cin_pad / 64) < 7000 && d->cout_pad % 128 == 0 && d->cout_pad % 256 != 0 && y->c == d->cout_pad &&
                      (y->c_off * ES) % 16 == 0 && (y->ld * ES) % 16 == 0 && a.M < (1 << 24) &&
                      (int64_t)x->n * a.x_hp * a.x_wp * x->ld * ES < ((int64_t)1 << 32) && (int64_t)64 * a.ktot_bytes < ((int64_t)1 << 31) &&
                      (int64_t)y->n * a.y_hp * a.y_wp * y->ld < ((int64_t)1 << 32);
        if (p8w_ok && (d->epilogue & DBX_EPI_GATE) && gate) p8w_ok = (gate->c_off * ES) % 16 == 0 && (gate->ld * ES) % 16 == 0;
        if (p8w_ok) {
            static int ncu = 0;
            if (!ncu) { int dev = 0; DBX_HIP(hipGetDevice(&dev)); DBX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev)); }
            P8Args ts;
            p8w_ok = p8w_schedule(a.M, y->c / 128, ncu, ts) && ts.items >= (ncu * 3) / 4;
        }
        if (p8w_ok) {
            a.ntile_n = y->c / 128;
            if (plan) {
                plan->kernel = DBX_K_P8; plan->tile_m = 512; plan->tile_n = 128; plan->w_frag = 0;
                snprintf(plan->name, sizeof plan->name, "conv3x3_p8w_kernel<%s,3>", tname);
                return DBX_OK;
            }
            if (pool_p8) return launch_conv_p8w<T, 3, 3>(a, s);
            return kk == DBX_EPI_GATE ? launch_conv_p8w<T, 3, 2>(a, s) : launch_conv_p8w<T, 3>(a, s);
        }
        DBX_REQUIRE(!pool_p8, "conv pool: the problem does not qualify for a pooling epilogue (ask dbx_conv_pool_fusable)");
        DBX_REQUIRE(!(w2_frag && !(d->epilogue & DBX_CONV_WFRAG)), "heads forward fused: plain-layout weights, but the problem does not qualify for the 8-phase kernel (ask dbx_heads_forward_fusable)");
    }
    // Wide 16-bit layers with enough tiles to fill the chip: register-streamed weights (conv3x3_ws.hpp) -- the 3x3 / pad 1
    // backbone layers on congruent frames and the 1x1 head GEMMs (768 -> 512 heads forward; its split-destination data
    // gradient).  The kernel needs the weights in fragment order: a launch takes it iff the caller says so (DBX_CONV_WFRAG);
    // dbx_conv_plan() reports it whenever the problem qualifies.
    {
        const bool wfrag = d->epilogue & DBX_CONV_WFRAG;
        const bool k3 = d->kh == 3 && d->kw == 3 && d->cpad == 1 && x->pad == 1;
        const bool k1 = d->kh == 1 && d->kw == 1 && d->cpad == 0 && x->pad <= 1 && d->cin_pad % 128 == 0 && d->cout_pad % 256 == 0;
        const int ctot = y2 ? split_c + y2->c : y->c;                                   // couts over both destinations
        bool ws_ok = !smallc && sizeof(T) == 2 && (k3 || k1) && ws_enabled() &&
                     !(d->epilogue & (DBX_EPI_F32_NCHW | DBX_EPI_DROPMASK | DBX_EPI_ACCUM)) && (k1 || !(d->epilogue & DBX_EPI_DROPHASH)) &&
                     (k1 || !y2) && d->cin_pad % 64 == 0 && d->cin_pad >= 128 && ctot == d->cout_pad && d->cout_pad % 128 == 0 &&
                     (x->c_off * ES) % 128 == 0 && (y->c_off * ES) % 16 == 0 && (y->ld * ES) % 16 == 0;
        {   // the kernel's epilogue kinds: plain / ReLU / ReLU-gate (3x3), plain / hash dropout / ReLU-gate / plain + gated second destination (1x1)
            const int kk = d->epilogue & (DBX_EPI_RELU | DBX_EPI_GATE | DBX_EPI_DROPHASH);
            if (y2) ws_ok = ws_ok && kk == 0 && (epi2 & (DBX_EPI_RELU | DBX_EPI_GATE | DBX_EPI_DROPHASH | DBX_EPI_BIAS)) == DBX_EPI_GATE;
            else if (k3) ws_ok = ws_ok && (kk == 0 || kk == DBX_EPI_RELU || kk == DBX_EPI_GATE);
            else ws_ok = ws_ok && (kk == 0 || kk == DBX_EPI_DROPHASH || kk == DBX_EPI_GATE);
        }
        if (ws_ok && y2) ws_ok = split_c % 256 == 0 && y2->c % 256 == 0 && !(epi2 & DBX_EPI_ACCUM) && (y2->c_off * ES) % 16 == 0 && (y2->ld * ES) % 16 == 0;
        const int wm = d->cout_pad % 256 == 0 ? 1 : 2;
        const long long qtot = (long long)x->n * x->h * a.x_wp;
        if (ws_ok) {
            ws_ok = (long long)x->h * a.x_wp >= 256 * wm + 8 && qtot < (1ll << 30) &&                   // one image seam per tile
                    (qtot / (256 * wm)) * (ctot / (256 / wm)) >= 192;                                       // fills the chip
            if ((d->epilogue & DBX_EPI_GATE) && gate) ws_ok = ws_ok && (gate->c_off * ES) % 16 == 0 && (gate->ld * ES) % 16 == 0;
            if (y2 && (epi2 & DBX_EPI_GATE) && gate2) ws_ok = ws_ok && (gate2->c_off * ES) % 16 == 0 && (gate2->ld * ES) % 16 == 0;
        }
        if (wfrag) DBX_REQUIRE(ws_ok, "conv: DBX_CONV_WFRAG weights, but the problem does not qualify for the ws kernel (ask dbx_conv_plan)");
        // Which eligible problems the plan PREFERS on it (same-box A/B inside the training step, profiles/r02_ws_vs_band.txt): the kernel's
        // fixed cost per tile (prologue, one-wave-per-SIMD epilogue with nothing to overlap it) is amortised over the K loop: it wins on
        // the un-gated 512 -> 512 and 256 -> 256 layers, on 128-cout layers with >= 256 input channels, and on the 1x1 heads; the LDS band
        // kernels keep the rest.  The GATED 512 -> 512 / 256 -> 256 data gradients (gate chunks prefetched four fragments ahead in its
        // epilogue): round 3 kept them on the band kernel on the strength of three same-box pairs (step 1.2 % slower with ws, read as power
        // coupling).  Round 4 repeated it with five alternating pairs, per-kernel durations AND clocks for both arms
        // (profiles/r04_power_ab.txt): the step is 0.2 % FASTER with ws on them (9.219 vs 9.236 ms; kernel time of the two families
        // -48 us per step), no other kernel's clock moves -- the round-3 reading is retired and ws takes them.  DBX_WS_GATED=0: band.
        static int ws_gated = -1;
        if (ws_gated < 0) { const char* e = getenv("DBX_WS_GATED"); ws_gated = e ? atoi(e) : 1; }
        const bool nogate = !(d->epilogue & DBX_EPI_GATE) || ws_gated != 0;
        const bool ws_pref = k1 || (wm == 2 && d->cin_pad >= 256) ||
                             (nogate && wm == 1 && ((d->cin_pad >= 512 && d->cout_pad >= 512) || (d->cin_pad == 256 && d->cout_pad == 256)));
        if (ws_ok && (wfrag || (plan && (ws_pref || ws_level() >= 2)))) {
            a.ntile_n = ctot / (256 / wm);
            if (k1 && !y2 && (a.epi & ~DBX_EPI_ACCUM) == (DBX_EPI_BIAS | DBX_EPI_DROPHASH) && ws_level() != 3) {    // heads forward: fixed epilogue
                // DBX_WS_NFX=4: 128-pixel tiles, TWO workgroups per CU (one wave of each per SIMD: one's epilogue and store drain under the
                // other's K loop)
                static int nfx = -1;
                if (nfx < 0) { const char* e = getenv("DBX_WS_NFX"); nfx = e ? atoi(e) : 4; }
                if (a.w2f) DBX_SELECT_WS(256, 256, 1, 1, 2, (nfx == 4 ? launch_conv_ws<T, 1, 1, 2, 4>(a, x->n, x->h, x->pad, s) : launch_conv_ws<T, 1, 1, 2>(a, x->n, x->h, x->pad, s)));       // + the second 1x1 convs
                DBX_SELECT_WS(256, 256, 1, 1, 1, (nfx == 4 ? launch_conv_ws<T, 1, 1, 1, 4>(a, x->n, x->h, x->pad, s) : launch_conv_ws<T, 1, 1, 1>(a, x->n, x->h, x->pad, s)));
            }
            DBX_REQUIRE(!a.w2f, "heads forward fused: needs the fixed bias + hash-dropout epilogue of the 1x1 ws kernel");
            // (128-pixel tiles with two workgroups per CU for the 3x3 instantiation: measured 3-6 % SLOWER in the lab -- twice the weight and
            //  band traffic per MFMA, and its K loop is 97 % of the tile: there is no epilogue worth overlapping)
            if (k1) DBX_SELECT_WS(256, 256, 1, 1, 0, (launch_conv_ws<T, 1, 1>(a, x->n, x->h, x->pad, s)));
            if (wm == 1) DBX_SELECT_WS(256, 256, 1, 3, 0, (launch_conv_ws<T, 1, 3>(a, x->n, x->h, 1, s)));
            DBX_SELECT_WS(512, 128, 2, 3, 0, (launch_conv_ws<T, 2, 3>(a, x->n, x->h, 1, s)));
        }
    }
    // 3x3 / pad 1 on congruent frames (x.pad == 1), 16-bit, plain NHWC epilogue: the band kernel over the linearised frame
    if (!smallc && sizeof(T) == 2 && (conv_variant() == 0 || conv_variant() >= 4) && d->kh == 3 && d->kw == 3 && d->cpad == 1 && x->pad == 1 &&
        !(d->epilogue & (DBX_EPI_F32_NCHW | DBX_EPI_DROPMASK | DBX_EPI_DROPHASH)) && (d->cin_pad * ES) % 64 == 0 && y->c % 64 == 0) {
        // tiles over the frame without its top / bottom halo rows (DBX_BAND_ROWSKIP=0: over the whole frame, as rounds 1-2 did)
        static int rowskip = -1;
        if (rowskip < 0) { const char* e = getenv("DBX_BAND_ROWSKIP"); rowskip = e ? atoi(e) : 1; }
        a.rowskip = rowskip && (long long)x->h * a.x_wp >= 16 ? 1 : 0;
        const long long Q = (long long)x->n * (a.rowskip ? x->h : a.x_hp) * a.x_wp;
        const int tiles256 = (int)((Q + 255) / 256), tiles512 = (int)((Q + 511) / 512);
        const bool tall = conv_variant() != 4 && tiles512 >= 1024;      // enough work for >= 4 tall tiles per CU
        // small problems (single-image inference: 64x64 or 128x128 maps): a wide tile would leave most CUs without a workgroup,
        // so the N tile narrows until there are ~200 workgroups (the A band is then re-read by more N tiles, from L2)
        const int want = 200;
        const bool few256 = (long long)tiles256 * (y->c / 256) < want, few128 = (long long)tiles256 * (y->c / 128) < want;
        // 64 input channels, 128 couts on big maps (conv2_1): the pipelined halo-tile kernel, one 64-cout slice per workgroup (DBX_C64P_WIDE=0: band)
        {
            static int c64p_wide = -1;
            if (c64p_wide < 0) { const char* e = getenv("DBX_C64P_WIDE"); c64p_wide = e ? atoi(e) : 1; }
            if (c64p_wide && d->cin_pad == 64 && d->cout_pad == 128 && y->c == 128 && x->c >= 64 && a.ktot_bytes == 1152 && conv_variant() == 0 &&
                !(a.epi & DBX_EPI_ACCUM) && (y->c_off * ES) % 16 == 0 && (y->ld * ES) % 16 == 0 &&
                (long long)x->n * ((x->h + 7) / 8) * ((x->w + 31) / 32) >= 512) {        // (>= 4 tiles per workgroup and slice)
                a.ntile_n = 2;
                DBX_SELECT(DBX_K_C64, 256, 64, "conv3x3_c64p_kernel", (launch_conv_c64p<T, false>(a, x->n, x->h, x->w, s)));
            }
        }
        if (y->c % 256 == 0 && d->cout_pad % 256 == 0 && !few256) {
            a.ntile_n = y->c / 256; a.nblocks = tiles256 * a.ntile_n;
#ifdef DBX_LAB
            if (conv_variant() == 5 && (d->cin_pad * ES) % 128 == 0) return launch_conv_pipe<T, 5>(a, s);
            if (conv_variant() == 7 && (d->cin_pad * ES) % 128 == 0) return launch_conv_pw4<T, 0>(a, s);
#endif
#ifdef DBX_LAB
            if (conv_variant() == 30) {            // two 4-wave workgroups per CU (independent barrier domains): 240 x 128 tiles, 80 KB of LDS each
                a.ntile_n = y->c / 128; a.nblocks = (int)((Q + 239) / 240) * a.ntile_n;
                return launch_conv_band<T, 240, 128, 2, 1, 4>(a, s);
            }
#endif
            DBX_SELECT(DBX_K_BAND, 256, 256, "conv3x3_band_kernel", (launch_conv_band<T, 256, 256, 2, 2, 4>(a, s)));
        }
        // Single-image maps whose 192-pixel tiles fit ONE round of one workgroup per CU (512 x 512 input: conv3 at 128 x 128, conv4 at
        // 64 x 64): the tile's 72-144 K stages are each a 64-ns MFMA burst behind a ~0.5-us L2 / MALL round trip, so what counts is
        // how many stages are in flight -- with the whole LDS of a CU to itself a workgroup runs a 4- or 5-deep ring instead of the
        // 3 stages that two co-resident 128-pixel workgroups can afford (DBX_CONV_VARIANT=8: off).
        const int tiles192 = (int)((Q + 191) / 192);
        // round 6: 144-pixel tiles on SIX waves (3 x 2: 48 pixels per wave) where they still fit one round but fill more of the 256 CUs than
        // the 192-pixel ones -- a 512 x 512 image's conv4 (64 x 64 map): 30 x 8 = 240 workgroups instead of 22 x 8 = 176, each with 3/4 of the
        // work; conv3 (128 x 128): 116 x 2 = 232 instead of 87 x 2 = 174 (DBX_BAND144=0: the 192-pixel tiles)
        {
            static int b144 = -1;
            if (b144 < 0) { const char* e = getenv("DBX_BAND144"); b144 = e ? atoi(e) : 1; }
            const int tiles144 = (int)((Q + 143) / 144);
            if (b144 && conv_variant() != 8 && x->n == 1) {
                if (y->c % 128 == 0 && d->cout_pad % 128 == 0 && y->c <= 256 && tiles144 * (y->c / 128) <= 256 && tiles192 * (y->c / 128) >= 128) {
                    a.ntile_n = y->c / 128; a.nblocks = tiles144 * a.ntile_n;
                    DBX_SELECT(DBX_K_BAND, 144, 128, "conv3x3_band_kernel", (launch_conv_band<T, 144, 128, 4, 3, 2>(a, s)));
                }
                if (tiles144 * (y->c / 64) <= 256 && tiles192 * (y->c / 64) >= 128 && !(y->c % 128 == 0 && y->c <= 256 && tiles192 * (y->c / 128) >= 128)) {
                    a.ntile_n = y->c / 64; a.nblocks = tiles144 * a.ntile_n;
                    DBX_SELECT(DBX_K_BAND, 144, 64, "conv3x3_band_kernel", (launch_conv_band<T, 144, 64, 5, 3, 2>(a, s)));
                }
            }
        }
        if (conv_variant() != 8 && x->n == 1 && y->c % 128 == 0 && d->cout_pad % 128 == 0 && y->c <= 256 && tiles192 * (y->c / 128) <= 256 &&
            tiles192 * (y->c / 128) >= 128) {
            a.ntile_n = y->c / 128; a.nblocks = tiles192 * a.ntile_n;
            DBX_SELECT(DBX_K_BAND, 192, 128, "conv3x3_band_kernel", (launch_conv_band<T, 192, 128, 4, 4, 2>(a, s)));
        }
        if (conv_variant() != 8 && x->n == 1 && tiles192 * (y->c / 64) <= 256 && tiles192 * (y->c / 64) >= 128) {
            a.ntile_n = y->c / 64; a.nblocks = tiles192 * a.ntile_n;
            DBX_SELECT(DBX_K_BAND, 192, 64, "conv3x3_band_kernel", (launch_conv_band<T, 192, 64, 5, 4, 2>(a, s)));
        }
        // ... and a 128-cout layer whose 256-pixel tiles number just over one round (conv2 of a 512 x 512 image: 261 workgroups of one
        // per CU = two rounds, the second with five workgroups): 288-pixel tiles (232) finish in one
        const int tiles288 = (int)((Q + 287) / 288);
        if (conv_variant() != 8 && x->n == 1 && y->c == 128 && d->cout_pad == 128 && tiles256 > 256 && tiles288 <= 256) {
            a.ntile_n = 1; a.nblocks = tiles288;
            DBX_SELECT(DBX_K_BAND, 288, 128, "conv3x3_band_kernel", (launch_conv_band<T, 288, 128, 3, 2, 4>(a, s)));
        }
        if (y->c % 128 == 0 && d->cout_pad % 128 == 0 && (!few128 || y->c == 128)) {
            a.ntile_n = y->c / 128;
            if (tall) { a.nblocks = tiles512 * a.ntile_n; DBX_SELECT(DBX_K_BAND, 512, 128, "conv3x3_band_kernel", (launch_conv_band<T, 512, 128, 2, 4, 2>(a, s))); }
            a.nblocks = tiles256 * a.ntile_n;
            DBX_SELECT(DBX_K_BAND, 256, 128, "conv3x3_band_kernel", (launch_conv_band<T, 256, 128, 3, 4, 2>(a, s)));
        }
        // 64 -> 64 channels on big maps: weights-stationary halo-tile kernel (DBX_CONV_VARIANT=4 keeps the band kernel, 9 the
        // one-episode-per-tile halo kernel instead of the software-pipelined one)
        if (d->cin_pad == 64 && d->cout_pad == 64 && y->c == 64 && x->c >= 64 && a.ktot_bytes == 1152 && conv_variant() != 4 &&
            (long long)x->n * ((x->h + 7) / 8) * ((x->w + 31) / 32) >= 256) {           // at least one 8x32 tile per CU
            a.ntile_n = 1;
            if (conv_variant() == 9 || (a.epi & DBX_EPI_ACCUM)) DBX_SELECT(DBX_K_C64, 256, 64, "conv3x3_c64_kernel", (launch_conv_c64<T>(a, x->n, x->h, x->w, s)));
            DBX_SELECT(DBX_K_C64, 256, 64, "conv3x3_c64p_kernel", (launch_conv_c64p<T, false>(a, x->n, x->h, x->w, s)));
        }
        a.ntile_n = y->c / 64;
        if (tall) { a.nblocks = tiles512 * a.ntile_n; DBX_SELECT(DBX_K_BAND, 512, 64, "conv3x3_band_kernel", (launch_conv_band<T, 512, 64, 3, 8, 1>(a, s))); }
        // single-image maps (64x64, 128x128): 256-pixel tiles give 0.6 or 1.05 rounds of workgroups on 256 CUs; 128-pixel tiles
        // (three stages = 63 KB: two workgroups per CU) put more than one workgroup on every CU instead (DBX_CONV_VARIANT=8: off)
        if ((long long)tiles256 * a.ntile_n <= 320 && conv_variant() != 8) {
            a.nblocks = (int)((Q + 127) / 128) * a.ntile_n;
            DBX_SELECT(DBX_K_BAND, 128, 64, "conv3x3_band_kernel", (launch_conv_band<T, 128, 64, 3, 8, 1>(a, s)));
        }
        a.nblocks = tiles256 * a.ntile_n;
        DBX_SELECT(DBX_K_BAND, 256, 64, "conv3x3_band_kernel", (launch_conv_band<T, 256, 64, 4, 8, 1>(a, s)));
    }
    // LDS-DMA ring kernel: any non-small-Cin layer (f16/bf16/f32 alike); 256x128 tiles, 256x64 when the layer has 64 couts
    if (!smallc && conv_variant() != 1 && a.ksteps >= 2) {
        const bool n64 = (d->cout_pad % 128 != 0) || y->c <= 64;
        if (n64) {
            a.ntile_n = y->c <= 64 ? 1 : d->cout_pad / 64;
            a.nblocks = ((a.M + 255) / 256) * a.ntile_n;
            DBX_SELECT(DBX_K_DMA, 256, 64, "conv_igemm_dma_kernel", (launch_conv_dma<T, 256, 64, 128, 3, 8, 1>(a, s)));
        }
        if (y2) {                                                                    // split destination: 256x256 tiles over both
            a.ntile_n = (split_c + y2->c + 255) / 256;
            a.nblocks = ((a.M + 255) / 256) * a.ntile_n;
            DBX_SELECT(DBX_K_DMA, 256, 256, "conv_igemm_dma_kernel", (launch_conv_dma<T, 256, 256, 64, 4, 2, 4>(a, s)));
        }
        if (d->cout_pad % 256 == 0 && y->c % 256 == 0 && conv_variant() != 2) {      // wide layers: 256x256 tile
            a.ntile_n = y->c / 256;
            a.nblocks = ((a.M + 255) / 256) * a.ntile_n;
            DBX_SELECT(DBX_K_DMA, 256, 256, "conv_igemm_dma_kernel", (launch_conv_dma<T, 256, 256, 64, 4, 2, 4>(a, s)));
        }
        a.ntile_n = (y->c + 127) / 128;
        a.nblocks = ((a.M + 255) / 256) * a.ntile_n;
        DBX_SELECT(DBX_K_DMA, 256, 128, "conv_igemm_dma_kernel", (launch_conv_dma<T, 256, 128, 128, 3, 4, 2>(a, s)));
    }
    // conv1_1: one chunk per pixel, 64 couts, plain bias/ReLU epilogue, big maps: halo-tile kernel
    if (smallc && sizeof(T) == 2 && d->kh == 3 && d->kw == 3 && d->cpad == 1 && x->pad == 1 && d->cout_pad == 64 && y->c == 64 &&
        x->ld * ES == 16 && !(d->epilogue & ~(DBX_EPI_BIAS | DBX_EPI_RELU)) && a.ktot_bytes == 256 && conv_variant() == 0 &&
        (long long)x->n * ((x->h + 7) / 8) * ((x->w + 31) / 32) >= 256)
        DBX_SELECT(DBX_K_C8, 256, 64, "conv3x3_c8_kernel", (launch_conv_c8<T>(a, x->n, x->h, x->w, s)));
    // tile choice: couts are tiled by 128 unless the layer has 64 (or the result is tiny, e.g. the 512->k heads)
    const bool narrow = (d->cout_pad % 128 != 0) || y->c <= 64;
    if (narrow) {
        a.ntile_n = d->cout_pad / 64;
        if (y->c <= 64) a.ntile_n = 1;
        a.nblocks = ((a.M + 255) / 256) * a.ntile_n;
        DBX_SELECT(DBX_K_IGEMM, 256, 64, "conv_igemm_kernel", (smallc ? launch_conv<T, 256, 64, 4, 1, true>(a, s) : launch_conv<T, 256, 64, 4, 1, false>(a, s)));
    }
    a.ntile_n = (y->c + 127) / 128;
    a.nblocks = ((a.M + 127) / 128) * a.ntile_n;
    DBX_SELECT(DBX_K_IGEMM, 128, 128, "conv_igemm_kernel", (smallc ? launch_conv<T, 128, 128, 2, 2, true>(a, s) : launch_conv<T, 128, 128, 2, 2, false>(a, s)));
}

#undef DBX_SELECT
#undef DBX_SELECT_WS

extern "C" int dbx_conv_forward(const dbx_conv_desc* d, const dbx_view* x, const void* w_packed, const float* bias,
                                const dbx_view* y, const dbx_view* gate, const uint8_t* dropmask, int32_t dropmask_ld,
                                void* stream) {
    if (!d || !x || !y || !w_packed) { dbx_set_error("conv: null argument"); return DBX_ERR_ARG; }
    DBX_DISPATCH_DTYPE(d->dtype, conv_forward_t, d, x, w_packed, bias, y, gate, dropmask, dropmask_ld, (hipStream_t)stream);
}

template <typename T>
static int conv_plan_t(const dbx_conv_desc* d, const dbx_view* x, const dbx_view* y, dbx_conv_plan_t* out) {
    return conv_forward_t<T>(d, x, nullptr, nullptr, y, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, out);
}
extern "C" int dbx_conv_plan(const dbx_conv_desc* d, const dbx_view* x, const dbx_view* y, dbx_conv_plan_t* out) {
    if (!d || !x || !y || !out) { dbx_set_error("conv plan: null argument"); return DBX_ERR_ARG; }
    memset(out, 0, sizeof *out);
    DBX_DISPATCH_DTYPE(d->dtype, conv_plan_t, d, x, y, out);
}

extern "C" int dbx_conv_forward_split(const dbx_conv_desc* d, const dbx_view* x, const void* w_packed, const float* bias,
                                      const dbx_view* y, const dbx_view* gate, const dbx_view* y2, const dbx_view* gate2,
                                      int32_t split_c, int32_t epilogue2, void* stream) {
    if (!d || !x || !y || !y2 || !w_packed) { dbx_set_error("conv split: null argument"); return DBX_ERR_ARG; }
    if (d->dtype == DBX_F32) { dbx_set_error("conv split: 16-bit types only"); return DBX_ERR_DTYPE; }
    DBX_DISPATCH_DTYPE(d->dtype, conv_forward_t, d, x, w_packed, bias, y, gate, nullptr, 0, (hipStream_t)stream, y2, gate2, split_c, epilogue2);
}

// ---- heads forward, both 1x1 convs in one pass over the pixels (conv3x3_ws_kernel<T, 1, 1, 2>, see conv3x3_ws.hpp)
struct Heads2Fin { int k[4], off[4], nh, ktot; float* outp[4]; };   // outp[i] != NULL: head i's own [N][k_i][H][W] tensor
__global__ void heads2_finish_kernel(const float* __restrict__ part, const float* __restrict__ bias2, float* __restrict__ out, int M, int HW,
                                     const Heads2Fin hf) {
    // out[n][off_i + m][r] = bias2[off_i + m] + part[2 i][p][m] + part[2 i + 1][p][m]   (a head's two 256-cout tiles, fixed order)
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < M; p += gridDim.x * blockDim.x) {
        const int n = p / HW, r = p - n * HW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= hf.nh) break;
            const float* p0 = part + ((size_t)(2 * i) * M + p) * 8;
            const float* p1 = part + ((size_t)(2 * i + 1) * M + p) * 8;
            const f32x4 a0 = *(const f32x4*)p0, b0 = *(const f32x4*)p1;
            f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, b1 = a1;
            if (hf.k[i] > 4) { a1 = *(const f32x4*)(p0 + 4); b1 = *(const f32x4*)(p1 + 4); }
            const float s[8] = {a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w, a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w};
#pragma unroll
            for (int m = 0; m < 8; ++m)
                if (m < hf.k[i]) {
                    float* o = hf.outp[i] ? hf.outp[i] + ((size_t)n * hf.k[i] + m) * HW + r : out + ((size_t)n * hf.ktot + hf.off[i] + m) * HW + r;
                    *o = bias2[hf.off[i] + m] + s[m];
                }
        }
    }
}
static bool heads_fused_shape_ok(const dbx_conv_desc* d, const dbx_view* x, const dbx_view* hid, const int32_t* k, int nh) {
    if (!d || !x || !hid || !k || nh < 1 || nh > 4) return false;
    if (d->dtype == DBX_F32 || d->kh != 1 || d->kw != 1 || d->cpad != 0 || d->cout_pad != 512 * nh || hid->c != 512 * nh) return false;
    if ((d->epilogue & ~DBX_CONV_WFRAG) != (DBX_EPI_BIAS | DBX_EPI_DROPHASH)) return false;
    for (int i = 0; i < nh; ++i) if (k[i] < 1 || k[i] > 8) return false;
    return true;
}
// which kernel takes the heads' forward GEMM with the second convs fused: 1 = the 1x1 ws kernel (fragment-order weights, DBX_CONV_WFRAG),
// 2 = the 8-phase kernel (plain packed weights; second weights as a plain [64][512 nh] image with every head's rows at 0..k-1), 0 = neither
static int heads_fused_kind(const dbx_conv_desc* d, const dbx_view* x, const dbx_view* hid, const int32_t* k, int nh) {
    if (!heads_fused_shape_ok(d, x, hid, k, nh)) return 0;
    dbx_conv_plan_t pl;
    dbx_conv_desc dd = *d; dd.epilogue &= ~DBX_CONV_WFRAG;
    if (dbx_conv_plan(&dd, x, hid, &pl) != DBX_OK) return 0;
    if (pl.kernel == DBX_K_WS && strstr(pl.name, ",1,1,1>") != nullptr) return 1;       // the 1x1 ws kernel with the fixed epilogue takes this problem
    if (pl.kernel == DBX_K_P8 && strstr(pl.name, ",1,1>") != nullptr) return 2;
    return 0;
}
extern "C" int dbx_heads_forward_fusable(const dbx_conv_desc* d, const dbx_view* x, const dbx_view* hid, const int32_t* k, int32_t nh) {
    return heads_fused_kind(d, x, hid, k, nh);
}
extern "C" int64_t dbx_heads_forward_fused_scratch_bytes(int32_t nh, int64_t pixels) { return (int64_t)2 * nh * pixels * 8 * 4 + 256; }
static int heads_forward_fused_impl(const dbx_conv_desc* d, const dbx_view* x, const void* w1_frag, const float* bias1, const dbx_view* hid,
                                    const void* w2_frag, const float* bias2, const int32_t* k, int32_t nh, float* out_nchw, float* const* outs,
                                    void* scratch, void* stream) {
    if (!d || !x || !hid || !w1_frag || !bias1 || !w2_frag || !bias2 || !k || (!out_nchw && !outs) || !scratch) { dbx_set_error("heads forward fused: null argument"); return DBX_ERR_ARG; }
    {
        // fragment-order weights (DBX_CONV_WFRAG): the ws kernel, wherever it can run the problem; plain weights: the 8-phase kernel, where the plan gives it the problem
        const int kind = heads_fused_kind(d, x, hid, k, nh);
        DBX_REQUIRE((d->epilogue & DBX_CONV_WFRAG) ? heads_fused_shape_ok(d, x, hid, k, nh) : kind == 2,
                    "heads forward fused: needs the 16-bit 1x1 768 -> 512 nh GEMM with bias + hash dropout, nh <= 4 heads of <= 8 outputs, and the weight layout of the kernel dbx_heads_forward_fusable names");
    }
    float* part = (float*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    int rc;
    if (d->dtype == DBX_F16) rc = conv_forward_t<_Float16>(d, x, w1_frag, bias1, hid, nullptr, nullptr, 0, (hipStream_t)stream, nullptr, nullptr, 0, 0, nullptr, nullptr, w2_frag, part);
    else rc = conv_forward_t<__bf16>(d, x, w1_frag, bias1, hid, nullptr, nullptr, 0, (hipStream_t)stream, nullptr, nullptr, 0, 0, nullptr, nullptr, w2_frag, part);
    if (rc != DBX_OK) return rc;
    Heads2Fin hf; hf.nh = nh; hf.ktot = 0;
    for (int i = 0; i < 4; ++i) {
        hf.k[i] = i < nh ? k[i] : 0; hf.off[i] = hf.ktot; hf.ktot += hf.k[i];
        hf.outp[i] = (outs && i < nh) ? outs[i] : nullptr;
        if (outs && i < nh && !outs[i]) { dbx_set_error("heads forward fused: null head output"); return DBX_ERR_ARG; }
    }
    const int M = hid->n * hid->h * hid->w, HW = hid->h * hid->w;
    hipLaunchKernelGGL(heads2_finish_kernel, dim3((M + 255) / 256 < 2048 ? (M + 255) / 256 : 2048), dim3(256), 0, (hipStream_t)stream, part, bias2, out_nchw, M, HW, hf);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_heads_forward_fused(const dbx_conv_desc* d, const dbx_view* x, const void* w1_frag, const float* bias1, const dbx_view* hid,
                                       const void* w2_frag, const float* bias2, const int32_t* k, int32_t nh, float* out_nchw,
                                       void* scratch, void* stream) {
    return heads_forward_fused_impl(d, x, w1_frag, bias1, hid, w2_frag, bias2, k, nh, out_nchw, nullptr, scratch, stream);
}
extern "C" int dbx_heads_forward_fused_heads(const dbx_conv_desc* d, const dbx_view* x, const void* w1_frag, const float* bias1, const dbx_view* hid,
                                             const void* w2_frag, const float* bias2, const int32_t* k, int32_t nh, float* const* outs,
                                             void* scratch, void* stream) {
    return heads_forward_fused_impl(d, x, w1_frag, bias1, hid, w2_frag, bias2, k, nh, nullptr, outs, scratch, stream);
}

// conv1_2 data gradient + conv1_1 weight gradient in one launch (conv3x3_c64_kernel<T, false, true>)
int dbx_internal_wgrad_reduce(const float* partial, const float* bpartial, int splits, int co, int ci, int taps, int co_pad, int ci_pad,
                              float* dw, float* db, int accumulate, hipStream_t s);           // conv_wgrad.hip
template <typename T>
static bool c64_wg1_ok(const dbx_conv_desc* d, const dbx_view* dz, const dbx_view* gate, const dbx_view* x0) {
    return sizeof(T) == 2 && d->kh == 3 && d->kw == 3 && d->cpad == 1 && dz->pad == 1 && d->cin_pad == 64 && d->cout_pad == 64 &&
           dz->c >= 64 && (d->epilogue & ~DBX_CONV_WFRAG) == DBX_EPI_GATE && gate && gate->n == dz->n && gate->h == dz->h && gate->w == dz->w &&
           gate->c >= 64 && x0 && x0->n == dz->n && x0->h == dz->h && x0->w == dz->w && x0->pad == 1 && x0->c * sizeof(T) == 16 &&
           x0->ld == x0->c && x0->c_off == 0 && ((size_t)x0->ptr % 16) == 0;
}
template <typename T>
static int conv_dgrad_wgrad1_t(const dbx_conv_desc* d, const dbx_view* dz, const void* w, const dbx_view* gate, const dbx_view* x0, int ci,
                               float* dw, float* db, void* scratch, int accumulate, hipStream_t s) {
    constexpr int ES = sizeof(T);
    DBX_REQUIRE(c64_wg1_ok<T>(d, dz, gate, x0), "dgrad+wgrad1: needs a 16-bit 3x3/pad 1 64 -> 64 gated data gradient on congruent frames and an 8-channel framed input");
    DBX_REQUIRE(ci >= 1 && ci <= 8 && dw && scratch, "dgrad+wgrad1: 1..8 real input channels, dw and scratch required");
    DBX_REQUIRE(((size_t)dz->ptr % 16) == 0 && (dz->ld * ES) % 16 == 0 && (dz->c_off * ES) % 16 == 0 && (gate->c_off * ES) % 16 == 0 && (gate->ld * ES) % 16 == 0 && ((size_t)gate->ptr % 16) == 0,
                "dgrad+wgrad1: 16-byte alignment of dz and gate");
    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.x = (const char*)dz->ptr + (size_t)dz->c_off * ES;
    a.w = (const char*)w;
    a.gate = (const char*)gate->ptr + (size_t)gate->c_off * ES;
    a.M = dz->n * dz->h * dz->w; a.HoWo = dz->h * dz->w; a.Wo = dz->w;
    a.x_hp = dz->h + 2; a.x_wp = dz->w + 2; a.x_ld = dz->ld; a.x_org = 0;
    a.g_hp = gate->h + 2 * gate->pad; a.g_wp = gate->w + 2 * gate->pad; a.g_ld = gate->ld; a.g_pad = gate->pad;
    a.kw = 3; a.ntaps = 9; a.cpt = 8; a.ktot_bytes = 1152; a.ksteps = 9; a.cout_valid = 64; a.ntile_n = 1; a.epi = DBX_EPI_GATE;
    C64Geo tg;
    memset(&tg, 0, sizeof tg);
    const int slabs = c64_wg1_slabs();
    tg.x0 = (const char*)x0->ptr; tg.x0_ld = x0->ld;
    tg.partial = (float*)scratch;
    tg.bpartial = db ? (float*)scratch + (size_t)slabs * 64 * 9 * 8 : nullptr;
    const int rc = launch_conv_c64<T, false, true>(a, dz->n, dz->h, dz->w, s, &tg);
    if (rc != DBX_OK) return rc;
    return dbx_internal_wgrad_reduce(tg.partial, tg.bpartial, slabs, 64, ci, 9, 64, 8, dw, db, accumulate, s);
}
extern "C" int64_t dbx_conv_dgrad_wgrad1_scratch_bytes(void) { return ((int64_t)c64_wg1_slabs() * (64 * 9 * 8 + 64)) * 4 + 256; }
extern "C" int dbx_conv_dgrad_wgrad1_fusable(const dbx_conv_desc* d, const dbx_view* dz, const dbx_view* gate, const dbx_view* x0) {
    if (!d || !dz || !gate || !x0) return 0;
    switch (d->dtype) {
        case DBX_F16: return c64_wg1_ok<_Float16>(d, dz, gate, x0) ? 1 : 0;
        case DBX_BF16: return c64_wg1_ok<__bf16>(d, dz, gate, x0) ? 1 : 0;
        default: return 0;
    }
}
extern "C" int dbx_conv_dgrad_wgrad1(const dbx_conv_desc* d, const dbx_view* dz, const void* w_packed, const dbx_view* gate, const dbx_view* x0,
                                     int32_t ci, float* dw_oihw, float* db, void* scratch, int32_t accumulate, void* stream) {
    if (!d || !dz || !w_packed || !gate || !x0) { dbx_set_error("dgrad+wgrad1: null argument"); return DBX_ERR_ARG; }
    if (d->dtype == DBX_F32) { dbx_set_error("dgrad+wgrad1: 16-bit types only"); return DBX_ERR_DTYPE; }
    DBX_DISPATCH_DTYPE(d->dtype, conv_dgrad_wgrad1_t, d, dz, w_packed, gate, x0, ci, dw_oihw, db, scratch, accumulate, (hipStream_t)stream);
}

template <typename T>
static int conv_pool_ok_t(const dbx_conv_desc* d, const dbx_view* x, const dbx_view* y) {
    if (c64_pool_ok<T>(d, x, y)) return 1;
    // the 8-phase kernels' pooling epilogue: whatever the one selection path gives a pooled call (asked in plan mode with a stand-in pooled view)
    if (sizeof(T) != 2 || y->h % 2 || y->w % 2 || y->h < 2 || y->w < 2 || (y->c * (int)sizeof(T)) % 16) return 0;
    static int on = -1;
    if (on < 0) { const char* e = getenv("DBX_P8_POOL"); on = e ? atoi(e) : 1; }
    if (!on) return 0;
    dbx_view yp = *y;
    yp.h = y->h / 2; yp.w = y->w / 2; yp.pad = 1; yp.ld = y->c; yp.c_off = 0;
    dbx_conv_plan_t pl;
    // (a probe is not a failed call: a negative answer must not replace the caller's dbx_last_error() line)
    char saved[512];
    strncpy(saved, dbx_last_error(), sizeof saved - 1); saved[sizeof saved - 1] = 0;
    const int rc = conv_forward_t<T>(d, x, nullptr, nullptr, y, nullptr, nullptr, 0, nullptr, &yp, nullptr, 0, EPI2_POOL, &pl);
    if (rc != DBX_OK) dbx_set_error("%s", saved);
    return rc == DBX_OK && pl.kernel == DBX_K_P8 ? 1 : 0;
}
extern "C" int dbx_conv_pool_fusable(const dbx_conv_desc* d, const dbx_view* x, const dbx_view* y) {
    if (!d || !x || !y) return 0;
    switch (d->dtype) {
        case DBX_F16: return conv_pool_ok_t<_Float16>(d, x, y);
        case DBX_BF16: return conv_pool_ok_t<__bf16>(d, x, y);
        default: return 0;
    }
}
extern "C" int dbx_conv_forward_pool_idx(const dbx_conv_desc* d, const dbx_view* x, const void* w_packed, const float* bias,
                                         const dbx_view* y, const dbx_view* ypool, int32_t write_full, void* idx, void* stream) {
    if (!d || !x || !y || !ypool || !w_packed) { dbx_set_error("conv pool: null argument"); return DBX_ERR_ARG; }
    if (d->dtype == DBX_F32) { dbx_set_error("conv pool: 16-bit types only"); return DBX_ERR_DTYPE; }
    if (idx && ((size_t)idx % 4) != 0) { dbx_set_error("conv pool: idx must be 4-byte aligned"); return DBX_ERR_ARG; }
    DBX_DISPATCH_DTYPE(d->dtype, conv_forward_t, d, x, w_packed, bias, y, nullptr, nullptr, 0, (hipStream_t)stream, ypool, nullptr, 0,
                       EPI2_POOL | (write_full ? 0 : EPI2_POOL_ONLY), nullptr, idx);
}
extern "C" int dbx_conv_forward_pool(const dbx_conv_desc* d, const dbx_view* x, const void* w_packed, const float* bias,
                                     const dbx_view* y, const dbx_view* ypool, int32_t write_full, void* stream) {
    return dbx_conv_forward_pool_idx(d, x, w_packed, bias, y, ypool, write_full, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------ weight packing
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, int co, int ci, int taps, int mode,
                                   T* __restrict__ wp, int64_t ktot, int cin_pad, int row_off, int k_off, int rows_pad) {
    const int64_t total = (int64_t)co * ci * taps;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % taps);
        const int c = (int)((i / taps) % ci);
        const int o = (int)(i / ((int64_t)taps * ci));
        const float v = w[i];
        {   // an element whose destination lies outside the packed matrix is skipped: negative offsets cut a channel range out of a wider tensor
            const bool fwd = mode == 0 || mode == 4;
            const int row = row_off + (fwd ? o : c), col = k_off + (fwd ? c : o);
            if (row < 0 || row >= rows_pad || col < 0 || col >= cin_pad) continue;
        }
        if (mode == 0) wp[(int64_t)(row_off + o) * ktot + (int64_t)t * cin_pad + k_off + c] = from_f32<T>(v);
        else if (mode == 1) wp[(int64_t)(row_off + c) * ktot + (int64_t)(taps - 1 - t) * cin_pad + k_off + o] = from_f32<T>(v);
        else if (mode == 4) wp[dbx_frag_index(row_off + o, t, k_off + c, cin_pad, rows_pad, taps)] = from_f32<T>(v);
        else wp[dbx_frag_index(row_off + c, taps - 1 - t, k_off + o, cin_pad, rows_pad, taps)] = from_f32<T>(v);
    }
}

template <typename T>
static int pack_weight_t(int mode, const float* w, int co, int ci, int kh, int kw, void* wp, int rows_pad, int cin_pad,
                         int row_off, int k_off, hipStream_t s) {
    dbx_conv_desc d; d.dtype = DType<T>::id; d.kh = kh; d.kw = kw; d.cin_pad = cin_pad; d.cout_pad = rows_pad; d.drop_seed = 0;
    const int64_t ktot = packed_k_elems(&d);
    DBX_REQUIRE(mode == 0 || mode == 1 || mode == 4 || mode == 5, "pack_weight: mode %d", mode);
    // (rows / columns that fall outside [0, rows_pad) x [0, cin_pad) are skipped: a negative row_off / k_off packs a channel range)
    if (mode >= 4) DBX_REQUIRE(sizeof(T) == 2 && ((kh == 3 && kw == 3 && rows_pad % 128 == 0 && cin_pad % 64 == 0) ||
                                                 (kh == 1 && kw == 1 && rows_pad % 256 == 0 && cin_pad % 128 == 0)),
                               "pack_weight: fragment order needs a 16-bit 3x3 layer (rows_pad %% 128, cin_pad %% 64) or 1x1 layer (rows_pad %% 256, cin_pad %% 128)");
    const int64_t total = (int64_t)co * ci * kh * kw;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_weight_kernel<T>, dim3(blocks), dim3(256), 0, s, w, co, ci, kh * kw, mode, (T*)wp, ktot,
                       cin_pad, row_off, k_off, rows_pad);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

extern "C" int dbx_pack_weight(int32_t dtype, int32_t mode, const float* w_oihw, int32_t co, int32_t ci, int32_t kh,
                               int32_t kw, void* w_packed, int32_t rows_pad, int32_t cin_pad, int32_t row_off,
                               int32_t k_off, void* stream) {
    DBX_DISPATCH_DTYPE(dtype, pack_weight_t, mode, w_oihw, co, ci, kh, kw, w_packed, rows_pad, cin_pad, row_off, k_off,
                       (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- heads: d_x with the hidden gradient generated
// dbx_heads1_dgrad_gen: the ReLU-gated data gradient of the heads' first 1x1 convs, d_x = gate(x) * (W1^T d_hid), with d_hid = keep *
// (d_out W2) generated in registers (heads_gen.hpp) -- dbx_conv_forward(d_hid, W1^T image, DBX_EPI_GATE) without d_hid in memory.
// d_out / w2 / k / use_hash / drop_seed as for dbx_heads1_wgrad_gen; w1t_frag: dbx_pack_weight mode 5 image of W1 restricted to the
// 256 input channels of y (rows_pad 256, cin_pad 512 nh); y, gate: 256-channel views of the same pixels (any padding).
int dbx_internal_heads1_dgrad_gen_f32(const dbx_view* d_out, const float* const* w2, const int32_t* k, int nh, int use_hash, unsigned seed,
                                      const void* w1t_plain, const dbx_view* y, const dbx_view* gate, hipStream_t s);
template <typename T>
static int heads1_dgrad_gen_t(const dbx_view* d_out, const float* const* w2, const int32_t* k, int nh, int use_hash, unsigned seed, const void* w1t_frag,
                              const dbx_view* y, const dbx_view* gate, hipStream_t s) {
    // fp32: the parity suite's reference instantiation (heads_ref_f32.hip); w1t_frag is then the PLAIN data-gradient image (dbx_pack_weight
    // mode 1, rows_pad = y's channels, cin_pad = 512 nh: rows of 512 nh floats)
    if constexpr (sizeof(T) != 2) return dbx_internal_heads1_dgrad_gen_f32(d_out, w2, k, nh, use_hash, seed, w1t_frag, y, gate, s);
    else {
        DBX_REQUIRE(nh >= 1 && nh <= 4 && d_out->pad == 0 && d_out->n == y->n && d_out->h == y->h && d_out->w == y->w && gate->n == y->n &&
                    gate->h == y->h && gate->w == y->w, "heads1_dgrad_gen: d_out (compact), y and gate are maps of the same pixels");
        DBX_REQUIRE(y->c == 256 && gate->c == 256 && (y->ld * 2) % 16 == 0 && (gate->ld * 2) % 16 == 0 && (y->c_off * 2) % 16 == 0 && (gate->c_off * 2) % 16 == 0 &&
                    ((size_t)y->ptr % 16) == 0 && ((size_t)gate->ptr % 16) == 0 && ((size_t)w1t_frag % 16) == 0, "heads1_dgrad_gen: 256 output channels, 16-byte aligned");
        DBX_REQUIRE(d_out->c % nh == 0 && d_out->c / nh >= 8 && ((size_t)d_out->ptr % 16) == 0 && (d_out->ld * 2) % 16 == 0 && (d_out->c_off * 2) % 16 == 0 &&
                    ((d_out->c / nh) * 2) % 16 == 0, "heads1_dgrad_gen: d_out slots of >= 8 channels, 16-byte aligned");
        const int64_t npix = (int64_t)y->n * y->h * y->w;
        DBX_REQUIRE(npix > 0 && npix < (1 << 24) && npix * d_out->ld * 2 < ((int64_t)1 << 32), "heads1_dgrad_gen: pixel count");
        HGenArgs a;
        a.g.dout = (const char*)d_out->ptr + (size_t)d_out->c_off * 2;
        for (int i = 0; i < 4; ++i) {
            a.g.w2[i] = i < nh ? w2[i] : nullptr; a.g.k[i] = i < nh ? k[i] : 0;
            if (i < nh) DBX_REQUIRE(w2[i] && k[i] >= 1 && k[i] <= 8, "heads1_dgrad_gen: k in 1..8");
        }
        a.g.ld = d_out->ld; a.g.slot = d_out->c / nh; a.g.nh = nh; a.g.H = y->h; a.g.W = y->w; a.g.pad = 0; a.g.seed = seed; a.g.use_hash = use_hash ? 1 : 0;
        a.w1f = (const char*)w1t_frag;
        a.y = (char*)y->ptr + (size_t)y->c_off * 2; a.gate = (const char*)gate->ptr + (size_t)gate->c_off * 2;
        a.y_hp = y->h + 2 * y->pad; a.y_wp = y->w + 2 * y->pad; a.y_ld = y->ld; a.y_pad = y->pad;
        a.g_hp = gate->h + 2 * gate->pad; a.g_wp = gate->w + 2 * gate->pad; a.g_ld = gate->ld; a.g_pad = gate->pad;
        a.npix = (int)npix; a.ntiles = (int)((npix + 255) / 256); a.nhb = 16 * nh;
        return launch_heads1_dgrad_gen<T>(a, s);
    }
}
extern "C" int dbx_heads1_dgrad_gen(int32_t dtype, const dbx_view* d_out, const float* const* w2, const int32_t* k, int32_t nh, int32_t use_hash,
                                    uint32_t drop_seed, const void* w1t_frag, const dbx_view* y, const dbx_view* gate, void* stream) {
    if (!d_out || !w2 || !k || !w1t_frag || !y || !gate) { dbx_set_error("heads1_dgrad_gen: null argument"); return DBX_ERR_ARG; }
    DBX_DISPATCH_DTYPE(dtype, heads1_dgrad_gen_t, d_out, w2, k, nh, use_hash, (unsigned)drop_seed, w1t_frag, y, gate, (hipStream_t)stream);
}
