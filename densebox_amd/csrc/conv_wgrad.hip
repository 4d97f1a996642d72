// Convolution weight gradient on MFMA for gfx950.
//
//   dW[co][tap][ci] = sum_q dz[q][co] * x[q + shift(tap)][ci]          q = linear index over the WHOLE frame
//
// dz and x live in frames of identical geometry (N x Hp x Wp pixels); the frame of dz is zero, so walking the
// frame linearly needs no 2-D index math at all: every tap is the same GEMM against x shifted by a constant
// number of pixels ("im2col-free" in the literal sense).  The reduction dimension K' = q is the slow (row) index
// of both operands in memory, so the MFMA fragments need a transpose: f16/bf16 tiles are staged [32 q][channels]
// in LDS exactly as they sit in HBM and read with ds_read_b64_tr_b16 (each lane receives 4 consecutive q of its
// channel); f32 uses v_mfma_f32_16x16x4_f32 whose operands are one element per lane (plain ds_read_b32).
//
// Grid: (co-tile, ci-tile, tap) x split-K.  Each workgroup writes an fp32 partial slab; a second kernel reduces the
// slabs in fixed order (deterministic) and scatters into the fp32 OIHW gradient.  Workgroups of ci-tile 0 / tap 0
// also accumulate the bias gradient db[co] = sum_q dz[q][co] from the tiles they stream anyway.
#include "common.hpp"
#include <stdlib.h>
#include <stdio.h>

typedef short short4v __attribute__((ext_vector_type(4)));
template <int N> struct IC2 { static constexpr int value = N; };

struct WgradArgs {
    const char* dz; const char* x;      // pointing at frame origin, channel c_off
    float* partial;                     // [splits][co_pad][taps][ci_pad]
    float* bpartial;                    // [splits][co_pad]
    long long Q;                        // frame pixels N*Hp*Wp
    int dz_ld, x_ld;                    // elements per pixel
    int dz_c, x_c;                      // valid channels in the views
    int co_pad, ci_pad;                 // padded to tile multiples
    int taps, kw, wp;                   // taps = kh*kw, frame width
    int shift0;                         // (pad_x - pad_dz - cpad) applied to both ky and kx
    int tiles_co, tiles_ci;
    int rows_per_split;                 // multiple of 32
    int nsplit;
    int hp, nstrips, units, units_per_split;   // strip walk (wgrad3x3_strip_kernel): frame height, 32-pixel column strips per row, N * nstrips
    int spi, img_rows, row0, steps_total, steps_per_split;   // compact walk (wgrad_row3_kernel): 64-row steps per image over its valid rows only
    // round 6, wgrad3x3_strip_kernel<T, true>: dz is the backward of a 2x2 / stride 2 max pooling that is NOT in memory -- dz[y][x][c] =
    // (arg-max nibble of window (y / 2, x / 2), channel c) == (4 | 2 (y & 1) | (x & 1)) ? pdy[y / 2][x / 2][c] : 0 (dbx_maxpool2x2_bwd_idx with relu_gate)
    const char* pdy; const unsigned char* pidx;   // pooled gradient (channel offset applied), arg-max nibbles (dbx_maxpool2x2_idx layout, all channels)
    int p_hp, p_wp, p_ld, p_pad, p_h, p_w;        // pdy's frame (framed rows / columns, elements per pixel, pad) and pooled extent
    int z_h, z_w, z_pad, z_ctot, z_coff;          // the virtual dz: image extent, frame pad, channels of the pooled layer (nibble row = z_ctot / 2 bytes), view offset
    char* zout;                                   // optional: the framed dz map itself is ALSO written (channel offset applied) -- the pooling backward as a by-product
};

// Workgroup -> (tile, split).  All tiles of one split stream the same frame rows of dz / x, so they should share an L2:
// workgroup L runs on XCD L%8; when the split count is a multiple of 8, XCD x owns splits x, x+8, ... and walks
// (split, tile) with tile fastest.  Pure speed: any mapping is correct.
__device__ __forceinline__ void wg_tile_split(int ntile, int nsplit, int& tile, int& split) {
    const int L = blockIdx.x;
    if ((nsplit & 7) == 0) {
        const int xcd = L & 7, idx = L >> 3;
        tile = idx % ntile;
        split = (idx / ntile) * 8 + xcd;
    } else { tile = L % ntile; split = L / ntile; }
}

template <int W> __device__ __forceinline__ int swz16(int r, int b) {   // 2-byte tiles, W channels per row
    if (W == 128) return r * 256 + (b ^ (((r & 3) << 5) | (((r >> 3) & 1) << 7)));
    else return r * 128 + (b ^ ((((r >> 1) & 1) << 5) | (((r >> 3) & 1) << 6)));
}

template <typename T, int BMC, int BNC>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    constexpr int ES = sizeof(T);
    // frame rows per K step: 64 for the 16-bit 128x128 tile (32 MFMAs per wave per barrier), else 32
    constexpr int R = (ES == 2 && BMC == 128) ? 64 : 32;
    constexpr int WTM = BMC / 2, WTN = BNC / 2;             // 2x2 waves over (co, ci)
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int CPR_A = BMC * ES / 16, CPR_B = BNC * ES / 16;   // 16-byte chunks per tile row
    constexpr int LD_A = R * CPR_A / 256, LD_B = R * CPR_B / 256;    // chunks per thread per step
    constexpr int RSTEP_A = 256 / CPR_A, RSTEP_B = 256 / CPR_B;
    constexpr int A_BYTES = R * BMC * ES, B_BYTES = R * BNC * ES;
    __shared__ __attribute__((aligned(16))) char smem[2 * (A_BYTES + B_BYTES)];
    char* As = smem;
    char* Bs = smem + 2 * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int t, split;
    wg_tile_split(a.tiles_co * a.tiles_ci * a.taps, a.nsplit, t, split);
    const int tile_ci = t % a.tiles_ci; t /= a.tiles_ci;
    const int tile_co = t % a.tiles_co; t /= a.tiles_co;
    const int tap = t;
    const int ky = tap / a.kw, kx = tap - ky * a.kw;
    const long long shift = (long long)(ky + a.shift0) * a.wp + (kx + a.shift0);
    const long long q0 = (long long)split * a.rows_per_split;
    long long q1 = q0 + a.rows_per_split; if (q1 > a.Q) q1 = a.Q;
    const int nsteps = q1 > q0 ? (int)((q1 - q0 + R - 1) / R) : 0;
    const bool do_bias = (tile_ci == 0 && tap == 0 && a.bpartial != nullptr);

    // ---- loaders
    const int ca = tid % CPR_A, ra = tid / CPR_A, cb = tid % CPR_B, rb = tid / CPR_B;
    const bool a_ok = (tile_co * BMC * ES + ca * 16) < a.dz_c * ES;      // chunk inside the view's channels
    const bool b_ok = (tile_ci * BNC * ES + cb * 16) < a.x_c * ES;
    const char* ap = a.dz + ((q0 + ra) * a.dz_ld + tile_co * BMC) * (long long)ES + ca * 16;
    const char* bp = a.x + ((q0 + shift + rb) * a.x_ld + tile_ci * BNC) * (long long)ES + cb * 16;
    const long long a_row = (long long)a.dz_ld * ES, b_row = (long long)a.x_ld * ES;
    u32x4 areg[LD_A], breg[LD_B];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    auto gload = [&](int s) {
#pragma unroll
        for (int i = 0; i < LD_A; ++i) areg[i] = a_ok ? *(const u32x4*)(ap + ((long long)s * R + i * RSTEP_A) * a_row) : zero4;
#pragma unroll
        for (int i = 0; i < LD_B; ++i) breg[i] = b_ok ? *(const u32x4*)(bp + ((long long)s * R + i * RSTEP_B) * b_row) : zero4;
    };
    auto lds_off_a = [&](int r, int b) { if constexpr (ES == 2) return swz16<BMC>(r, b); else return r * BMC * 4 + b; };
    auto lds_off_b = [&](int r, int b) { if constexpr (ES == 2) return swz16<BNC>(r, b); else return r * BNC * 4 + b; };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LD_A; ++i) *(u32x4*)(As + buf * A_BYTES + lds_off_a(ra + i * RSTEP_A, ca * 16)) = areg[i];
#pragma unroll
        for (int i = 0; i < LD_B; ++i) *(u32x4*)(Bs + buf * B_BYTES + lds_off_b(rb + i * RSTEP_B, cb * 16)) = breg[i];
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[16 / ES];
#pragma unroll
    for (int j = 0; j < 16 / ES; ++j) bsum[j] = 0.f;
    auto bias_acc = [&]() {
#pragma unroll
        for (int i = 0; i < LD_A; ++i) {
            const T* e = (const T*)&areg[i];
#pragma unroll
            for (int j = 0; j < 16 / ES; ++j) bsum[j] += to_f32(e[j]);
        }
    };

    if (nsteps > 0) {
        gload(0);
        if (do_bias) bias_acc();
        lstore(0);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
        const char* Ab = As + buf * A_BYTES;
        const char* Bb = Bs + buf * B_BYTES;
        if constexpr (ES == 2) {
            // lane l of 16-lane group g: tr-read of the 4x16 block [q = 8g+4h .. +3][c0 .. c0+15] delivers, to lane i,
            // the 4 consecutive q of channel c0+i: out[i][j] = in[lane 4j + (i>>2)][i&3]
            const int g = lane >> 4, rsub = (lane & 15) >> 2, csub = (lane & 3) * 8;
#pragma unroll
            for (int hh = 0; hh < R / 32; ++hh) {
                const int r0 = hh * 32 + 8 * g + rsub;
                u32x4 af[MI], bf[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int cbyte = (wm * WTM + mi * 16) * 2 + csub;
                    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) short4v*)(Ab + swz16<BMC>(r0, cbyte)));
                    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) short4v*)(Ab + swz16<BMC>(r0 + 4, cbyte)));
                    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                    af[mi] = (u32x4){l2.x, l2.y, h2.x, h2.y};
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int cbyte = (wn * WTN + ni * 16) * 2 + csub;
                    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) short4v*)(Bb + swz16<BNC>(r0, cbyte)));
                    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) short4v*)(Bb + swz16<BNC>(r0 + 4, cbyte)));
                    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                    bf[ni] = (u32x4){l2.x, l2.y, h2.x, h2.y};
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        if constexpr (sizeof(T) == 2 && DType<T>::id == DBX_F16)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[mi]),
                                                                                 __builtin_bit_cast(f16x8, bf[ni]), acc[mi][ni], 0, 0, 0);
                        else
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[mi]),
                                                                                  __builtin_bit_cast(bf16x8, bf[ni]), acc[mi][ni], 0, 0, 0);
                    }
            }
        } else {
            // f32: A[i][k=g] = dz[q=4kk+g][co0+i], B[k=g][j] = x[q=4kk+g][ci0+j]
            const int g = lane >> 4, i16 = lane & 15;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                float af[MI], bf[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) af[mi] = *(const float*)(Ab + ((4 * kk + g) * BMC + wm * WTM + mi * 16 + i16) * 4);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bf[ni] = *(const float*)(Bb + ((4 * kk + g) * BNC + wn * WTN + ni * 16 + i16) * 4);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
            }
        }
        if (s + 1 < nsteps) {
            if (do_bias) bias_acc();
            lstore(buf ^ 1);
        }
        __syncthreads();
    }

    // ---- partial slab: [split][co][tap][ci]; lane holds co = (l>>4)*4 + r, ci = l&15 of each fragment
    {
        float* P = a.partial + (((long long)split * a.co_pad) * a.taps) * a.ci_pad;
        const int co_b = tile_co * BMC + wm * WTM + (lane >> 4) * 4;
        const int ci_b = tile_ci * BNC + wn * WTN + (lane & 15);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const float v[4] = {acc[mi][ni].x, acc[mi][ni].y, acc[mi][ni].z, acc[mi][ni].w};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    P[((long long)(co_b + mi * 16 + r) * a.taps + tap) * a.ci_pad + ci_b + ni * 16] = v[r];
            }
    }
    if (do_bias) {
        // reduce the per-thread column sums over the row groups (threads with equal chunk column ca)
        __syncthreads();
        float* red = (float*)smem;                       // [RSTEP_A][BMC]
#pragma unroll
        for (int j = 0; j < 16 / ES; ++j) red[ra * BMC + ca * (16 / ES) + j] = bsum[j];
        __syncthreads();
        if (tid < BMC) {
            float s = 0.f;
            for (int r = 0; r < RSTEP_A; ++r) s += red[r * BMC + tid];
            a.bpartial[(long long)split * a.co_pad + tile_co * BMC + tid] = s;
        }
    }
}





// ------------------------------------------------------------------------------------------------ wide 1x1 layers: 256 x 256 tiles
// The stage-1 head GEMM (768 -> 2048 over every pixel) has no taps to share, so the LDS-read relief comes from the wave
// tile alone: 256(co) x 256(ci) per workgroup, 8 waves as 4(co) x 2(ci) with 64 x 128 per wave -- 0.75 transpose reads
// per MFMA instead of 1.0 and half the staged bytes per FLOP of the 128 x 128 tile.  64-row K steps (64 MFMAs per wave
// per barrier) cover the ~0.85 us global-load latency with the single register-staged prefetch.  128 KB of LDS, one
// workgroup per CU.  Requires dz_c % 256 == 0 and x_c % 256 == 0.  Grid: (co-tile, ci-tile, tap) x split-K.
__device__ __forceinline__ int swz16w(int r, int b) {                   // 256 channels (512 B) per row, same XOR mask as swz16<128>
    return r * 512 + (b ^ (((r & 3) << 5) | (((r >> 3) & 1) << 7)));
}

template <typename T>
__global__ __launch_bounds__(512) void wgrad_wide_kernel(const WgradArgs a) {
    static_assert(sizeof(T) == 2, "16-bit tiles");
    constexpr int R = 64;
    constexpr int T_BYTES = R * 512;                                    // one operand tile: 64 rows x 256 channels
    extern __shared__ __attribute__((aligned(16))) char smem[];        // 2 buffers x (A, B) = 128 KB
    char* As = smem;
    char* Bs = smem + 2 * T_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                           // wave tile 64(co) x 128(ci)
    int t, split;
    wg_tile_split(a.tiles_co * a.tiles_ci * a.taps, a.nsplit, t, split);
    const int tile_ci = t % a.tiles_ci; t /= a.tiles_ci;
    const int tile_co = t % a.tiles_co; t /= a.tiles_co;
    const int tap = t, ky = tap / a.kw, kx = tap - ky * a.kw;
    const long long q0 = (long long)split * a.rows_per_split;
    long long q1 = q0 + a.rows_per_split; if (q1 > a.Q) q1 = a.Q;
    const int nsteps = q1 > q0 ? (int)((q1 - q0 + R - 1) / R) : 0;
    const bool do_bias = (tile_ci == 0 && tap == 0 && a.bpartial != nullptr);

    // loaders: thread t owns chunk t%32 of rows t/32 + 16 i (i = 0..3) of both tiles
    const int ca = tid & 31, ra = tid >> 5;
    const char* ap = a.dz + ((q0 + ra) * a.dz_ld + tile_co * 256) * 2LL + ca * 16;
    const long long xrow0 = q0 + (long long)(ky + a.shift0) * a.wp + kx + a.shift0;
    const char* bp = a.x + ((xrow0 + ra) * a.x_ld + tile_ci * 256) * 2LL + ca * 16;
    const long long a16 = 16LL * a.dz_ld * 2, b16 = 16LL * a.x_ld * 2;
    u32x4 areg[4], breg[4];
    auto gload = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            areg[i] = *(const u32x4*)(ap + (s * 4 + i) * a16);
            breg[i] = *(const u32x4*)(bp + (s * 4 + i) * b16);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(u32x4*)(As + buf * T_BYTES + swz16w(ra + 16 * i, ca * 16)) = areg[i];
            *(u32x4*)(Bs + buf * T_BYTES + swz16w(ra + 16 * i, ca * 16)) = breg[i];
        }
    };
    f32x4 acc[4][8];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    auto bias_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const T* e = (const T*)&areg[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum[j] += to_f32(e[j]);
        }
    };
    if (nsteps > 0) { gload(0); if (do_bias) bias_acc(); lstore(0); }
    __syncthreads();
    const int g = lane >> 4, rsub = (lane & 15) >> 2, csub = (lane & 3) * 8;
    auto trd = [&](const char* p) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p));
    };
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
        const char* Ab = As + buf * T_BYTES;
        const char* Bb = Bs + buf * T_BYTES;
        // software pipeline over the 16 (kk, ni) groups of four MFMAs: the B fragment of the next group and one of the
        // next kk's A fragments are in flight while the current group computes; sched_group_barrier pins the interleave.
        auto rdA = [&](int kk, int mi) {
            const int r0 = 32 * kk + 8 * g + rsub, cbyte = (wm * 64 + mi * 16) * 2 + csub;
            const u32x2 lo = trd(Ab + swz16w(r0, cbyte)), hi = trd(Ab + swz16w(r0 + 4, cbyte));
            return (u32x4){lo.x, lo.y, hi.x, hi.y};
        };
        auto rdB = [&](int kk, int ni) {
            const int r0 = 32 * kk + 8 * g + rsub, cbyte = (wn * 128 + ni * 16) * 2 + csub;
            const u32x2 lo = trd(Bb + swz16w(r0, cbyte)), hi = trd(Bb + swz16w(r0 + 4, cbyte));
            return (u32x4){lo.x, lo.y, hi.x, hi.y};
        };
        u32x4 af[2][4], bf[2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[0][mi] = rdA(0, mi);
        bf[0] = rdB(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#pragma unroll
        for (int grp = 0; grp < 16; ++grp) {
            const int kk = grp >> 3, ni = grp & 7;
            if (grp < 15) bf[(grp + 1) & 1] = rdB((grp + 1) >> 3, (grp + 1) & 7);
            if (grp < 4) af[1][grp] = rdA(1, grp);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                if constexpr (DType<T>::id == DBX_F16)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bf[grp & 1]), __builtin_bit_cast(f16x8, af[kk][mi]), acc[mi][ni], 0, 0, 0);
                else
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[grp & 1]), __builtin_bit_cast(bf16x8, af[kk][mi]), acc[mi][ni], 0, 0, 0);
            }
            if (grp < 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            else if (grp < 15) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        if (s + 1 < nsteps) { if (do_bias) bias_acc(); lstore(buf ^ 1); }
        __syncthreads();
    }
    {
        float* P = a.partial + (((long long)split * a.co_pad) * a.taps) * a.ci_pad;
        // x is the first MFMA operand: a lane holds four consecutive ci of one co -> one 16-byte store per fragment
        const int co_b = tile_co * 256 + wm * 64 + (lane & 15);
        const int ci_b = tile_ci * 256 + wn * 128 + (lane >> 4) * 4;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni)
                *(f32x4*)(P + ((long long)(co_b + mi * 16) * a.taps + tap) * a.ci_pad + ci_b + ni * 16) = acc[mi][ni];
    }
    if (do_bias) {
        __syncthreads();
        float* red = (float*)smem;                       // [16 row groups][256]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[ra * 256 + ca * 8 + j] = bsum[j];
        __syncthreads();
        if (tid < 256) {
            float sum = 0.f;
            for (int r = 0; r < 16; ++r) sum += red[r * 256 + tid];
            a.bpartial[(long long)split * a.co_pad + tile_co * 256 + tid] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------ 3x3 wide layers: one ky row per workgroup
// The per-tap kernel above is bound by its LDS operand reads: ds_read_b64_tr_b16 runs at ~62 B/clk/CU (half the b128
// rate, tools/probe_lds_rate.hip) and a 64x64 wave tile needs one such read per MFMA.  Here a workgroup owns a
// 128(co) x 128(ci) tile for the THREE kx taps of one ky:
//   * the dz fragments are read once and used for all three taps;
//   * per x channel column a lane reads ONE run of 12 band rows (three transpose reads); the fragment of tap kx is rows
//     kx..kx+7 of that run: kx=0 and kx=2 are register subsets, kx=1 is four v_alignbit -- 14 reads per 24 MFMAs;
//   * global traffic per MFLOP drops 2.9x (dz tile staged once, one 34-row x band for the three taps).
// 512 threads = 8 waves as 2(co) x 4(ci), 64x32 per wave, 3 x 8 accumulator fragments; register-staged double-buffered
// LDS as in wgrad_kernel.  Requires dz_c % 128 == 0 and x_c % 128 == 0.  Grid: (co-tile, ci-tile, ky) x split-K.
template <typename T>
__global__ __launch_bounds__(512) void wgrad_row3_kernel(const WgradArgs a) {
    static_assert(sizeof(T) == 2, "16-bit tiles");
    constexpr int R = 64, BAND = R + 2;                                 // frame rows per K step (one global-latency period)
    constexpr int A_BYTES = R * 256, B_BYTES = (BAND + 2) * 256;        // 128 channels x 2 B per row (+2 rows read, unused)
    extern __shared__ __attribute__((aligned(16))) char smem[];        // 2 x (A_BYTES + B_BYTES) = 66 KB
    char* As = smem;
    char* Bs = smem + 2 * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                           // wave tile 64(co) x 32(ci)
    int t, split;
    wg_tile_split(a.tiles_co * a.tiles_ci * 3, a.nsplit, t, split);
    const int tile_ci = t % a.tiles_ci; t /= a.tiles_ci;
    const int tile_co = t % a.tiles_co; t /= a.tiles_co;
    const int ky = t;
    // K range.  Linear walk: rows [split * rows_per_split, ...) of the whole frame.  Compact walk (a.spi > 0): a step is 64 rows of
    // ONE image's valid rows (frame rows pad .. pad + H - 1: dz is zero in the halo rows, so they need no MFMAs: 15 instead of
    // 16 steps per 32 x 32 frame); the last step of an image runs over into its bottom halo row and the next image's top one
    // (zeros: Wp >= 32).
    const bool compact = a.spi > 0;
    const int gs0 = compact ? split * a.steps_per_split : 0;
    const long long q0 = compact ? 0 : (long long)split * a.rows_per_split;
    long long q1 = q0 + a.rows_per_split; if (q1 > a.Q) q1 = a.Q;
    int nsteps = q1 > q0 ? (int)((q1 - q0 + R - 1) / R) : 0;
    if (compact) { const int gs1 = min(gs0 + a.steps_per_split, a.steps_total); nsteps = gs1 > gs0 ? gs1 - gs0 : 0; }
    auto qof = [&](int s) -> long long {
        if (!compact) return q0 + 64LL * s;
        const int gsx = gs0 + s, img = gsx / a.spi, j = gsx - img * a.spi;
        return (long long)img * a.img_rows + a.row0 + 64LL * j;
    };
    const bool do_bias = (tile_ci == 0 && ky == 0 && a.bpartial != nullptr);

    // loaders: thread t owns chunk t%16 of dz rows t/16 and t/16+32, of band rows t/16 and t/16+32, and (t < 32) of row t/16+64
    const int ca = tid & 15, ra = tid >> 4;
    const char* apb = a.dz + ((long long)ra * a.dz_ld + tile_co * 128) * 2LL + ca * 16;
    const long long xshift = (long long)(ky + a.shift0) * a.wp + a.shift0;
    const char* bpb = a.x + ((xshift + ra) * a.x_ld + tile_ci * 128) * 2LL + ca * 16;
    const long long a32 = 32LL * a.dz_ld * 2, b32 = 32LL * a.x_ld * 2;
    const bool has2 = tid < 32;
    u32x4 areg[2], breg[3];
    auto gload = [&](int s) {
        const long long qs = qof(s);
        const char* ap = apb + qs * a.dz_ld * 2LL;
        const char* bp = bpb + qs * a.x_ld * 2LL;
        areg[0] = *(const u32x4*)(ap);
        areg[1] = *(const u32x4*)(ap + a32);
        breg[0] = *(const u32x4*)(bp);
        breg[1] = *(const u32x4*)(bp + b32);
        if (has2) breg[2] = *(const u32x4*)(bp + 2 * b32);
    };
    auto lstore = [&](int buf) {
        *(u32x4*)(As + buf * A_BYTES + swz16<128>(ra, ca * 16)) = areg[0];
        *(u32x4*)(As + buf * A_BYTES + swz16<128>(32 + ra, ca * 16)) = areg[1];
        *(u32x4*)(Bs + buf * B_BYTES + swz16<128>(ra, ca * 16)) = breg[0];
        *(u32x4*)(Bs + buf * B_BYTES + swz16<128>(32 + ra, ca * 16)) = breg[1];
        if (has2) *(u32x4*)(Bs + buf * B_BYTES + swz16<128>(64 + ra, ca * 16)) = breg[2];
    };
    f32x4 acc[3][4][2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[k][mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    auto bias_acc = [&]() {
        const T* e0 = (const T*)&areg[0];
        const T* e1 = (const T*)&areg[1];
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] += to_f32(e0[j]) + to_f32(e1[j]);
    };
    if (nsteps > 0) { gload(0); if (do_bias) bias_acc(); lstore(0); }
    __syncthreads();
    const int g = lane >> 4, rsub = (lane & 15) >> 2, csub = (lane & 3) * 8;
    auto trd = [&](const char* p) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p));
    };
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
        const char* Ab = As + buf * A_BYTES;
        const char* Bb = Bs + buf * B_BYTES;
        // software pipeline over the four (kk, ni) units of 3 x 4 MFMAs: the 12-row run of the next unit and half of the
        // next kk's dz fragments are in flight while the current unit computes; sched_group_barrier pins the interleave.
        auto rdA = [&](int kk, int mi) {
            const int r0 = 32 * kk + 8 * g + rsub, cbyte = (wm * 64 + mi * 16) * 2 + csub;
            const u32x2 lo = trd(Ab + swz16<128>(r0, cbyte)), hi = trd(Ab + swz16<128>(r0 + 4, cbyte));
            return (u32x4){lo.x, lo.y, hi.x, hi.y};
        };
        struct Run { u32x2 r0, r1, r2; };      // rows r..r+11 of one channel column: row pairs (0,1)(2,3) | (4,5)(6,7) | (8,9)(10,11)
        auto rdRun = [&](int u) {
            const int r0 = 32 * (u >> 1) + 8 * g + rsub, cbyte = (wn * 32 + (u & 1) * 16) * 2 + csub;
            Run r;
            r.r0 = trd(Bb + swz16<128>(r0, cbyte)); r.r1 = trd(Bb + swz16<128>(r0 + 4, cbyte)); r.r2 = trd(Bb + swz16<128>(r0 + 8, cbyte));
            return r;
        };
        u32x4 af[2][4];
        Run run[2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[0][mi] = rdA(0, mi);
        run[0] = rdRun(0);
        __builtin_amdgcn_sched_group_barrier(0x100, 11, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = u >> 1, ni = u & 1;
            const Run c = run[u & 1];
            if (u < 3) run[(u + 1) & 1] = rdRun(u + 1);
            if (u < 2) { af[1][2 * u] = rdA(1, 2 * u); af[1][2 * u + 1] = rdA(1, 2 * u + 1); }
            u32x4 bf[3];
            bf[0] = (u32x4){c.r0.x, c.r0.y, c.r1.x, c.r1.y};
            bf[1] = (u32x4){__builtin_amdgcn_alignbit(c.r0.y, c.r0.x, 16), __builtin_amdgcn_alignbit(c.r1.x, c.r0.y, 16),
                            __builtin_amdgcn_alignbit(c.r1.y, c.r1.x, 16), __builtin_amdgcn_alignbit(c.r2.x, c.r1.y, 16)};
            bf[2] = (u32x4){c.r0.y, c.r1.x, c.r1.y, c.r2.x};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    if constexpr (DType<T>::id == DBX_F16)
                        acc[kx][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bf[kx]), __builtin_bit_cast(f16x8, af[kk][mi]), acc[kx][mi][ni], 0, 0, 0);
                    else
                        acc[kx][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[kx]), __builtin_bit_cast(bf16x8, af[kk][mi]), acc[kx][mi][ni], 0, 0, 0);
                }
                if (u < 2 && kx == 0) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);          // 3 run reads + 4 dz reads
                else if (u < 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else if (u < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);              // 3 run reads
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
        if (s + 1 < nsteps) { if (do_bias) bias_acc(); lstore(buf ^ 1); }
        __syncthreads();
    }
    {
        float* P = a.partial + (((long long)split * a.co_pad) * 9) * a.ci_pad;
        // x is the first MFMA operand: a lane holds four consecutive ci of one co -> one 16-byte store per fragment
        const int co_b = tile_co * 128 + wm * 64 + (lane & 15);
        const int ci_b = tile_ci * 128 + wn * 32 + (lane >> 4) * 4;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    *(f32x4*)(P + ((long long)(co_b + mi * 16) * 9 + ky * 3 + kx) * a.ci_pad + ci_b + ni * 16) = acc[kx][mi][ni];
    }
    if (do_bias) {
        __syncthreads();
        float* red = (float*)smem;                       // [32 rows][128]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[ra * 128 + ca * 8 + j] = bsum[j];
        __syncthreads();
        if (tid < 128) {
            float sum = 0.f;
            for (int r = 0; r < 32; ++r) sum += red[r * 128 + tid];
            a.bpartial[(long long)split * a.co_pad + tile_co * 128 + tid] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------ 3x3 wide layers, round 3: all nine taps, LDS-DMA ring
// wgrad_row3_kernel above splits a (co, ci) tile over three workgroups (one per ky) and 16 K ranges: 768 workgroups write
// 151 MB of fp32 slabs per launch for <= 9.4 MB of dW, its register-staged double buffer puts five ds_write_b128 and the
// vmcnt(0) they need between the last MFMA of a step and the barrier, and the first transpose reads of the next step start
// from an idle pipe.  This kernel:
//   * tile = 128(co) x 64(ci) x ALL NINE taps: 256 workgroups (one per CU, one round) -- half the slab bytes and half the splits
//     to reduce, dz fragments read once for nine taps (17 transpose reads per 36 MFMAs instead of 14 per 24);
//   * operands go global -> LDS by LDS-DMA (no staging registers, no ds_write, no vmcnt(0) in the loop): a stage is the dz
//     tile (64 rows x 256 B, 16 pieces) + three x bands (ky = 0..2: 72 rows x 128 B each, 27 pieces); the XOR swizzles of
//     the transpose reads are applied on the DMA source addresses;
//   * three-stage ring, ONE barrier per 64-row step placed in front of the step's last unit: tile s+2 is issued at the top of
//     step s (its stage held tile s-1, whose last read precedes the previous barrier), tile s+1 is waited for (counted vmcnt)
//     in front of the barrier, and the first fragments of tile s+1 are read behind it, under the last 12 MFMAs of step s --
//     the barrier never waits for data and no step starts with an empty pipe;
//   * the bias gradient comes from an MFMA against a ones fragment (each wave one of its four co fragments, in the steps
//     assigned to its ci tile): no extra LDS reads, no VALU reduction.
// 8 waves as 2(co) x 4(ci), 64 x 16 per wave, 9 x 4 accumulator fragments.  Requires dz_c % 128 == 0, x_c % 64 == 0.
// Grid: (co-tile, ci-tile) x split-K; bias partial rows = splits * tiles_ci.
template <typename T>
__global__ __launch_bounds__(512) void wgrad_all9_kernel(const WgradArgs a) {
    static_assert(sizeof(T) == 2, "16-bit tiles");
    constexpr int R = 64, BROWS = 72;
    constexpr int A_BYTES = R * 256, BAND_BYTES = BROWS * 128;
    constexpr int STAGE = A_BYTES + 3 * BAND_BYTES;                     // 44032 B
    constexpr int AP = A_BYTES / 1024, BP = 3 * BAND_BYTES / 1024, NPIECE = AP + BP, SLOTS = (NPIECE + 7) / 8;   // 16 + 27 pieces, 6 slots
    extern __shared__ __attribute__((aligned(16))) char smem[];        // 3 x STAGE = 129 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                           // wave tile 64(co) x 16(ci)
    int t, split;
    wg_tile_split(a.tiles_co * a.tiles_ci, a.nsplit, t, split);
    const int tile_ci = t % a.tiles_ci, tile_co = t / a.tiles_ci;
    // K range: as wgrad_row3_kernel (compact walk over the valid rows of each image when a.spi > 0)
    const bool compact = a.spi > 0;
    const int gs0 = compact ? split * a.steps_per_split : (int)(((long long)split * a.rows_per_split) / R);
    const long long q0 = compact ? 0 : (long long)split * a.rows_per_split;
    long long q1 = q0 + a.rows_per_split; if (q1 > a.Q) q1 = a.Q;
    int nsteps = q1 > q0 ? (int)((q1 - q0 + R - 1) / R) : 0;
    if (compact) { const int gs1 = min(gs0 + a.steps_per_split, a.steps_total); nsteps = gs1 > gs0 ? gs1 - gs0 : 0; }

    // ---- LDS-DMA sources.  A piece = 4 dz rows (lane: row l>>4, chunk position l&15), a B piece = 8 band rows (row l>>3, chunk
    // position l&7); position p of LDS row r receives source chunk p ^ mask(r) (swz16<128> / swz16<64>): the row bits that come
    // from the lane are folded into the lane offset, the one that comes from the piece index is a uniform XOR.
    const long long dzrow = (long long)a.dz_ld * 2, xrow = (long long)a.x_ld * 2;
    const unsigned laA = (unsigned)(lane >> 4) * (unsigned)dzrow + (unsigned)(((lane & 15) ^ ((lane >> 4) << 1)) << 4);
    const unsigned laB = (unsigned)(lane >> 3) * (unsigned)xrow + (unsigned)(((lane & 7) ^ (((lane >> 4) & 1) << 1)) << 4);
    // The tiles are issued in order, so the frame row of the next one is carried as two uniform pointers (dz, x) that advance by
    // 64 rows, or -- compact walk, last step of an image -- by the jump to the next image's first valid row: no division and no
    // 64-bit multiply per step.  A slot's piece is fixed for the kernel (piece = wave + 8 slot): its uniform byte offset from
    // those pointers is computed once.
    int ij = 0;
    long long iq0;
    if (compact) { const int img = gs0 / a.spi; ij = gs0 - img * a.spi; iq0 = (long long)img * a.img_rows + a.row0 + 64LL * ij; }
    else iq0 = q0;
    const char* ap = a.dz + tile_co * 256 + iq0 * dzrow;
    const char* bp = a.x + tile_ci * 128 + ((long long)a.shift0 * (a.wp + 1) + iq0) * xrow;
    const long long jump = compact ? a.img_rows - 64LL * (a.spi - 1) : 64;
    const long long a64 = 64 * dzrow, b64 = 64 * xrow, ajmp = jump * dzrow, bjmp = jump * xrow;
    // piece of slot i = wave + 8 i: slots 0, 1 are dz pieces, 2 .. 5 band pieces (slot 5 of waves 3 .. 7 repeats their slot 4:
    // every wave issues SLOTS loads, the waits are counted).  The swizzle bit that comes from the piece index depends on the wave
    // only -> folded into the lane offsets; the row offsets of the slots are six uniform constants.
    static_assert(AP == 16 && NPIECE == 43 && SLOTS == 6, "slot layout");
    const unsigned vA = laA ^ (unsigned)(((wave >> 1) & 1) << 7), vB = laB ^ (unsigned)((wave & 1) << 6);
    const int pb5 = wave < 3 ? wave + 24 : wave + 16;
    auto boff = [&](int pb) { const int band = pb / 9, pr = pb - band * 9; return (unsigned)(band * a.wp + 8 * pr) * (unsigned)xrow; };
    const unsigned sA0 = (unsigned)(4 * wave) * (unsigned)dzrow, sA1 = sA0 + 32u * (unsigned)dzrow;
    const unsigned sB2 = boff(wave), sB3 = boff(wave + 8), sB4 = boff(wave + 16), sB5 = boff(pb5);
    // The DMA goes out as inline asm (saddr form: uniform base + one lane offset register): for the builtin the compiler's
    // waitcnt pass puts s_waitcnt vmcnt(0) in front of the next transpose read (it cannot prove that the ds_read_tr intrinsic
    // does not alias the LDS-DMA destination), which drains the ring every step.  Invisible to that pass, the loads are counted by
    // hand: the only other VMEM operations of the kernel are the slab stores behind the final vmcnt(0).
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ldsw = lds0 + wave * 1024, lds5 = lds0 + (AP + pb5) * 1024;
    int issued = 0;
    auto glds = [](const char* src, unsigned voff, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
    };
    // issue the next tile into the stage at byte offset sb (the last tile again past the end: keeps the counts), then advance
#define ALL9_ISSUE(sb)                                                                      \
    do {                                                                                    \
        glds(ap + sA0, vA, ldsw + (sb));                                                    \
        glds(ap + sA1, vA, ldsw + (sb) + 8 * 1024);                                         \
        glds(bp + sB2, vB, ldsw + (sb) + 16 * 1024);                                        \
        glds(bp + sB3, vB, ldsw + (sb) + 24 * 1024);                                        \
        glds(bp + sB4, vB, ldsw + (sb) + 32 * 1024);                                        \
        glds(bp + sB5, vB, lds5 + (sb));                                                    \
        if (++issued < nsteps) {                                                            \
            if (compact && ++ij == a.spi) { ij = 0; ap += ajmp; bp += bjmp; }               \
            else { ap += a64; bp += b64; }                                                  \
        }                                                                                   \
    } while (0)

    f32x4 acc[9][4];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[tp][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bacc = {0.f, 0.f, 0.f, 0.f};
    const unsigned one2 = DType<T>::id == DBX_F16 ? 0x3C003C00u : 0x3F803F80u;
    const u32x4 ones = {one2, one2, one2, one2};
    const bool do_bias = a.bpartial != nullptr;

    const int g = lane >> 4, rsub = (lane & 15) >> 2, csub = (lane & 3) * 8;
    auto trd = [&](const char* p) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p));
    };
    auto rdA = [&](const char* Sb, int kk, int mi) {
        const int r0 = 32 * kk + 8 * g + rsub, cbyte = (wm * 64 + mi * 16) * 2 + csub;
        const u32x2 lo = trd(Sb + swz16<128>(r0, cbyte)), hi = trd(Sb + swz16<128>(r0 + 4, cbyte));
        return (u32x4){lo.x, lo.y, hi.x, hi.y};
    };
    struct Run { u32x2 r0, r1, r2; };          // band rows r .. r+11 of one channel column (see wgrad_row3_kernel)
    auto rdRun = [&](const char* Sb, int kk, int ky) {
        const int r0 = ky * BROWS + 32 * kk + 8 * g + rsub, cbyte = (wn * 16) * 2 + csub;
        const char* Bb = Sb + A_BYTES;
        Run r;
        r.r0 = trd(Bb + swz16<64>(r0, cbyte)); r.r1 = trd(Bb + swz16<64>(r0 + 4, cbyte)); r.r2 = trd(Bb + swz16<64>(r0 + 8, cbyte));
        return r;
    };
    auto mma = [&](const u32x4& xf, const u32x4& zf, f32x4& c) {
        if constexpr (DType<T>::id == DBX_F16)
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xf), __builtin_bit_cast(f16x8, zf), c, 0, 0, 0);
        else
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xf), __builtin_bit_cast(bf16x8, zf), c, 0, 0, 0);
    };

    u32x4 af[2][4];
    Run run[2];
    if (nsteps > 0) {
        ALL9_ISSUE(0u);
        ALL9_ISSUE((unsigned)STAGE);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLOTS) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[0][mi] = rdA(smem, 0, mi);
        run[0] = rdRun(smem, 0, 0);
    }
    // one 64-row step per iteration: six units u = (kk, ky) of 3 (kx) x 4 (mi) MFMAs; the stages rotate as uniform byte offsets
    unsigned so = 0, no = STAGE, po = 2 * STAGE;                        // this tile's stage, the next tile's, the one tile s + 2 goes to
    int bctr = gs0 % a.tiles_ci;                                        // the ci tile that owns this step's bias sums
    for (int s = 0; s < nsteps; ++s) {
        const char* Sb = smem + so;
        const char* Nb = smem + no;
        ALL9_ISSUE(po);                                                 // tile s + 2
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int kk = u / 3, ky = u - 3 * kk;
            if (u == 5) {
                // tile s+1 has landed (this wave's pieces; the barrier covers the others') and every wave is done with tile s's
                // LDS reads except the ones already in its registers
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLOTS) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            const Run c = run[u & 1];
            if (u < 5) run[(u + 1) & 1] = rdRun(Sb, (u + 1) / 3, (u + 1) % 3);
            else run[0] = rdRun(Nb, 0, 0);
            if (u == 1) { af[1][0] = rdA(Sb, 1, 0); af[1][1] = rdA(Sb, 1, 1); }
            if (u == 2) { af[1][2] = rdA(Sb, 1, 2); af[1][3] = rdA(Sb, 1, 3); }
            u32x4 bf[3];
            bf[0] = (u32x4){c.r0.x, c.r0.y, c.r1.x, c.r1.y};
            bf[1] = (u32x4){__builtin_amdgcn_alignbit(c.r0.y, c.r0.x, 16), __builtin_amdgcn_alignbit(c.r1.x, c.r0.y, 16),
                            __builtin_amdgcn_alignbit(c.r1.y, c.r1.x, 16), __builtin_amdgcn_alignbit(c.r2.x, c.r1.y, 16)};
            bf[2] = (u32x4){c.r0.y, c.r1.x, c.r1.y, c.r2.x};
            if (u == 5) {
                // (the next tile's dz fragments overwrite af[0]: kk = 1 computes from af[1])
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) af[0][mi] = rdA(Nb, 0, mi);
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) mma(bf[kx], af[kk][mi], acc[ky * 3 + kx][mi]);
                // transpose reads in flight behind each group of four MFMAs: 3 per plain unit, 7 with two dz fragments, 11 in the prefetch unit
                if (u == 1 || u == 2) { if (kx == 0) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0); else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
                else if (u == 5) { if (kx == 2) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0); else __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); }
                else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            if (u == 3 && do_bias && bctr == tile_ci) {
                // db partial: sum_q dz[q][co] of this wave's fragment wn, both K halves of the step
                __builtin_amdgcn_sched_barrier(0);
                if (wn == 0) { mma(ones, af[0][0], bacc); mma(ones, af[1][0], bacc); }
                else if (wn == 1) { mma(ones, af[0][1], bacc); mma(ones, af[1][1], bacc); }
                else if (wn == 2) { mma(ones, af[0][2], bacc); mma(ones, af[1][2], bacc); }
                else { mma(ones, af[0][3], bacc); mma(ones, af[1][3], bacc); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (++bctr == a.tiles_ci) bctr = 0;
        { const unsigned t3 = so; so = no; no = po; po = t3; }
    }
#undef ALL9_ISSUE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the over-run tiles land before the workgroup ends
    {
        // slab = the accumulator registers as they are: [split][tile][wave][tap][mi][lane][4 ci] -- every store instruction writes
        // 1 KiB of consecutive bytes; wgrad_reduce_all9_kernel maps (wave, tap, mi, lane) back to (co, tap, ci)
        float* P = a.partial + (((long long)split * (a.tiles_co * a.tiles_ci) + t) * 8 + wave) * (36 * 256) + lane * 4;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                *(f32x4*)(P + (tp * 4 + mi) * 256) = acc[tp][mi];
    }
    if (do_bias && lane < 16)       // every row of the ones-product holds the column sum
        a.bpartial[((long long)split * a.tiles_ci + tile_ci) * a.co_pad + tile_co * 128 + wm * 64 + wn * 16 + lane] = bacc.x;
}

// ------------------------------------------------------------------------------------------------ 3x3 wide layers, round 5: wgrad_all9_kernel's tile with the wave groups one barrier apart
template <typename T>
__global__ __launch_bounds__(512) void wgrad_all9s_kernel(const WgradArgs a) {
    static_assert(sizeof(T) == 2, "16-bit tiles");
    constexpr int R = 64, BROWS = 72;
    constexpr int A_BYTES = R * 256, BAND_BYTES = BROWS * 128;
    constexpr int STAGE = A_BYTES + 3 * BAND_BYTES;                     // 44032 B
    constexpr int AP = A_BYTES / 1024, BP = 3 * BAND_BYTES / 1024, NPIECE = AP + BP, SLOTS = (NPIECE + 7) / 8;   // 16 + 27 pieces, 6 slots
    extern __shared__ __attribute__((aligned(16))) char smem[];        // 3 x STAGE = 129 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                           // wave tile 64(co) x 16(ci)
    int t, split;
    wg_tile_split(a.tiles_co * a.tiles_ci, a.nsplit, t, split);
    const int tile_ci = t % a.tiles_ci, tile_co = t / a.tiles_ci;
    // K range: as wgrad_row3_kernel (compact walk over the valid rows of each image when a.spi > 0)
    const bool compact = a.spi > 0;
    const int gs0 = compact ? split * a.steps_per_split : (int)(((long long)split * a.rows_per_split) / R);
    const long long q0 = compact ? 0 : (long long)split * a.rows_per_split;
    long long q1 = q0 + a.rows_per_split; if (q1 > a.Q) q1 = a.Q;
    int nsteps = q1 > q0 ? (int)((q1 - q0 + R - 1) / R) : 0;
    if (compact) { const int gs1 = min(gs0 + a.steps_per_split, a.steps_total); nsteps = gs1 > gs0 ? gs1 - gs0 : 0; }

    // ---- LDS-DMA sources.  A piece = 4 dz rows (lane: row l>>4, chunk position l&15), a B piece = 8 band rows (row l>>3, chunk
    // position l&7); position p of LDS row r receives source chunk p ^ mask(r) (swz16<128> / swz16<64>): the row bits that come
    // from the lane are folded into the lane offset, the one that comes from the piece index is a uniform XOR.
    const long long dzrow = (long long)a.dz_ld * 2, xrow = (long long)a.x_ld * 2;
    const unsigned laA = (unsigned)(lane >> 4) * (unsigned)dzrow + (unsigned)(((lane & 15) ^ ((lane >> 4) << 1)) << 4);
    const unsigned laB = (unsigned)(lane >> 3) * (unsigned)xrow + (unsigned)(((lane & 7) ^ (((lane >> 4) & 1) << 1)) << 4);
    // The tiles are issued in order, so the frame row of the next one is carried as two uniform pointers (dz, x) that advance by
    // 64 rows, or -- compact walk, last step of an image -- by the jump to the next image's first valid row: no division and no
    // 64-bit multiply per step.  A slot's piece is fixed for the kernel (piece = wave + 8 slot): its uniform byte offset from
    // those pointers is computed once.
    int ij = 0;
    long long iq0;
    if (compact) { const int img = gs0 / a.spi; ij = gs0 - img * a.spi; iq0 = (long long)img * a.img_rows + a.row0 + 64LL * ij; }
    else iq0 = q0;
    const char* ap = a.dz + tile_co * 256 + iq0 * dzrow;
    const char* bp = a.x + tile_ci * 128 + ((long long)a.shift0 * (a.wp + 1) + iq0) * xrow;
    const long long jump = compact ? a.img_rows - 64LL * (a.spi - 1) : 64;
    const long long a64 = 64 * dzrow, b64 = 64 * xrow, ajmp = jump * dzrow, bjmp = jump * xrow;
    // piece of slot i = wave + 8 i: slots 0, 1 are dz pieces, 2 .. 5 band pieces (slot 5 of waves 3 .. 7 repeats their slot 4:
    // every wave issues SLOTS loads, the waits are counted).  The swizzle bit that comes from the piece index depends on the wave
    // only -> folded into the lane offsets; the row offsets of the slots are six uniform constants.
    static_assert(AP == 16 && NPIECE == 43 && SLOTS == 6, "slot layout");
    const unsigned vA = laA ^ (unsigned)(((wave >> 1) & 1) << 7), vB = laB ^ (unsigned)((wave & 1) << 6);
    const int pb5 = wave < 3 ? wave + 24 : wave + 16;
    auto boff = [&](int pb) { const int band = pb / 9, pr = pb - band * 9; return (unsigned)(band * a.wp + 8 * pr) * (unsigned)xrow; };
    const unsigned sA0 = (unsigned)(4 * wave) * (unsigned)dzrow, sA1 = sA0 + 32u * (unsigned)dzrow;
    const unsigned sB2 = boff(wave), sB3 = boff(wave + 8), sB4 = boff(wave + 16), sB5 = boff(pb5);
    // The DMA goes out as inline asm (saddr form: uniform base + one lane offset register): for the builtin the compiler's
    // waitcnt pass puts s_waitcnt vmcnt(0) in front of the next transpose read (it cannot prove that the ds_read_tr intrinsic
    // does not alias the LDS-DMA destination), which drains the ring every step.  Invisible to that pass, the loads are counted by
    // hand: the only other VMEM operations of the kernel are the slab stores behind the final vmcnt(0).
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ldsw = lds0 + wave * 1024, lds5 = lds0 + (AP + pb5) * 1024;
    int issued = 0;
    auto glds = [](const char* src, unsigned voff, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
    };
    // A tile's six LDS-DMA go out in three parts of two (one part per phase: an LDS-DMA costs 100+ issue cycles next to a dozen reads); the
    // pointers advance behind the last part (the last tile again past the end: keeps the counted waits uniform)
#define ALL9S_PART0(sb) do { glds(ap + sA0, vA, ldsw + (sb)); glds(ap + sA1, vA, ldsw + (sb) + 8 * 1024); } while (0)
#define ALL9S_PART1(sb) do { glds(bp + sB2, vB, ldsw + (sb) + 16 * 1024); glds(bp + sB3, vB, ldsw + (sb) + 24 * 1024); } while (0)
#define ALL9S_PART2(sb)                                                                     \
    do {                                                                                    \
        glds(bp + sB4, vB, ldsw + (sb) + 32 * 1024); glds(bp + sB5, vB, lds5 + (sb));      \
        if (++issued < nsteps) {                                                            \
            if (compact && ++ij == a.spi) { ij = 0; ap += ajmp; bp += bjmp; }               \
            else { ap += a64; bp += b64; }                                                  \
        }                                                                                   \
    } while (0)
    f32x4 acc[9][4];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[tp][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bacc = {0.f, 0.f, 0.f, 0.f};
    const unsigned one2 = DType<T>::id == DBX_F16 ? 0x3C003C00u : 0x3F803F80u;
    const u32x4 ones = {one2, one2, one2, one2};
    const bool do_bias = a.bpartial != nullptr;

    const int g = lane >> 4, rsub = (lane & 15) >> 2, csub = (lane & 3) * 8;
    auto trd = [&](const char* p) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p));
    };
    auto rdA = [&](const char* Sb, int kk, int mi) {
        const int r0 = 32 * kk + 8 * g + rsub, cbyte = (wm * 64 + mi * 16) * 2 + csub;
        const u32x2 lo = trd(Sb + swz16<128>(r0, cbyte)), hi = trd(Sb + swz16<128>(r0 + 4, cbyte));
        return (u32x4){lo.x, lo.y, hi.x, hi.y};
    };
    struct Run { u32x2 r0, r1, r2; };          // band rows r .. r+11 of one channel column (see wgrad_row3_kernel)
    auto rdRun = [&](const char* Sb, int kk, int ky) {
        const int r0 = ky * BROWS + 32 * kk + 8 * g + rsub, cbyte = (wn * 16) * 2 + csub;
        const char* Bb = Sb + A_BYTES;
        Run r;
        r.r0 = trd(Bb + swz16<64>(r0, cbyte)); r.r1 = trd(Bb + swz16<64>(r0 + 4, cbyte)); r.r2 = trd(Bb + swz16<64>(r0 + 8, cbyte));
        return r;
    };
    auto mma = [&](const u32x4& xf, const u32x4& zf, f32x4& c) {
        if constexpr (DType<T>::id == DBX_F16)
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xf), __builtin_bit_cast(f16x8, zf), c, 0, 0, 0);
        else
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xf), __builtin_bit_cast(bf16x8, zf), c, 0, 0, 0);
    };

    // ---- the 8-phase idea on this tile (round 5): the 8 waves are two groups of four (wm = 0 / 1: one wave of each per SIMD) that run the
    // same phase program ONE BARRIER APART, so that one group's pure cluster of 24 MFMAs runs while the other issues its transpose
    // reads, its LDS-DMA and the fragment shuffles of its next cluster.  A 64-row step = three phases of two units (kk, ky):
    //      A: (0,0) (0,1)    reads dz fragments of K half 0 + two band runs, LDS-DMA part 0 of tile s + 2
    //      B: (0,2) (1,0)    reads dz fragments of K half 1 + two band runs, part 1
    //      C: (1,1) (1,2)    reads two band runs, part 2, s_waitcnt vmcnt(6): tile s + 1 has landed
    // phase = [reads ; LDS-DMA ; lgkmcnt(0) ; fragment shuffles ; barrier ; 24 MFMAs ; barrier].  Tile s + 2 goes into the stage tile s - 1
    // left (its last reads, phase C of step s - 1, were retired in front of that phase's first barrier by BOTH groups before the first
    // group reaches phase A of step s); tile s + 1 is read one phase after the wait that retires it.
    // The transpose reads go out as inline asm with immediate offsets from SIX lane address registers (the swizzles of swz16 are XORs
    // of lane-constant bits: four dz bases -- one per co fragment, the fragment index is XORed with the row -- and two band bases --
    // the parity of the 8-row block flips with ky and with the third 4-row piece of a run); the compiler hoisted 34 loop-invariant
    // addresses out of the intrinsic form and spilled 100 registers.
    unsigned aA[4], aR[2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) aA[mi] = lds0 + (8 * g + rsub) * 256 + csub + (((mi ^ rsub) & 3) << 5) + ((wm ^ (g & 1)) << 7);
#pragma unroll
    for (int v = 0; v < 2; ++v) aR[v] = lds0 + (8 * g + rsub) * 128 + csub + (((wn & 1) ^ ((rsub >> 1) & 1)) << 5) + (((wn >> 1) ^ ((g ^ v) & 1)) << 6);
    auto tr = [](u32x2& d, unsigned addr, auto OFF_) {
        constexpr int OFF = decltype(OFF_)::value;
        static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
    };
    // dz fragments of K half kk (stage byte offset sb is added to the bases once per step)
    // (the halves stay the asm's own output registers until the hand-written lgkmcnt(0) has retired the reads -- wait_a4 below takes them
    //  as in/out operands, so no register move or spill of theirs can be scheduled in front of it: the compiler's waitcnt pass does not
    //  see LDS reads issued from inline asm -- and only then become the MFMA operands)
    struct RawA4 { u32x2 lo[4], hi[4]; };
    auto rdA4 = [&](RawA4& ra, unsigned sb, auto KK_) {
        constexpr int kk = decltype(KK_)::value;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { tr(ra.lo[mi], aA[mi] + sb, pipe::IC<kk * 8192>{}); tr(ra.hi[mi], aA[mi] + sb, pipe::IC<kk * 8192 + 1024>{}); }
    };
    auto wait_a4 = [](RawA4& ra) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra.lo[0]), "+v"(ra.lo[1]), "+v"(ra.lo[2]), "+v"(ra.lo[3]), "+v"(ra.hi[0]), "+v"(ra.hi[1]), "+v"(ra.hi[2]),
                     "+v"(ra.hi[3]) :: "memory");
    };
    auto asmA4 = [](u32x4 (&af)[4], const RawA4& ra) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[mi] = (u32x4){ra.lo[mi].x, ra.lo[mi].y, ra.hi[mi].x, ra.hi[mi].y};
    };
    auto rdRunA = [&](Run& r, unsigned sb, auto KK_, auto KY_) {
        constexpr int kk = decltype(KK_)::value, ky = decltype(KY_)::value;
        constexpr int base = A_BYTES + (ky * BROWS + 32 * kk) * 128;
        tr(r.r0, aR[ky & 1] + sb, pipe::IC<base>{}); tr(r.r1, aR[ky & 1] + sb, pipe::IC<base + 512>{}); tr(r.r2, aR[(ky + 1) & 1] + sb, pipe::IC<base + 1024>{});
    };
    struct Frag3 { u32x4 b[3]; };
    auto shuffle = [&](const Run& c) {
        Frag3 f;
        f.b[0] = (u32x4){c.r0.x, c.r0.y, c.r1.x, c.r1.y};
        f.b[1] = (u32x4){__builtin_amdgcn_alignbit(c.r0.y, c.r0.x, 16), __builtin_amdgcn_alignbit(c.r1.x, c.r0.y, 16),
                         __builtin_amdgcn_alignbit(c.r1.y, c.r1.x, 16), __builtin_amdgcn_alignbit(c.r2.x, c.r1.y, 16)};
        f.b[2] = (u32x4){c.r0.y, c.r1.x, c.r1.y, c.r2.x};
        return f;
    };
    auto unit = [&](const Frag3& f, const u32x4 (&afk)[4], int ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) mma(f.b[kx], afk[mi], acc[ky * 3 + kx][mi]);
    };
    auto sbar = []() { asm volatile("s_barrier" ::: "memory"); };
    u32x4 af0[4], af1[4];
    if (nsteps > 0) {
        ALL9S_PART0(0u); ALL9S_PART1(0u); ALL9S_PART2(0u);
        ALL9S_PART0((unsigned)STAGE); ALL9S_PART1((unsigned)STAGE); ALL9S_PART2((unsigned)STAGE);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLOTS) : "memory");
        sbar();
        if (wm == 1) sbar();                                            // the groups part
    }
    unsigned so = 0, no = STAGE, po = 2 * STAGE;
    int bctr = gs0 % a.tiles_ci;
    for (int s = 0; s < nsteps; ++s) {
        const bool bias_step = do_bias && bctr == tile_ci;
        // ---- phase A
        {
            Run r0, r1;
            RawA4 ra;
            rdA4(ra, so, pipe::IC<0>{}); rdRunA(r0, so, pipe::IC<0>{}, pipe::IC<0>{}); rdRunA(r1, so, pipe::IC<0>{}, pipe::IC<1>{});
            __builtin_amdgcn_sched_barrier(0);
            ALL9S_PART0(po);
            wait_a4(ra);
            __builtin_amdgcn_sched_barrier(0);
            asmA4(af0, ra);
            const Frag3 f0 = shuffle(r0), f1 = shuffle(r1);
            __builtin_amdgcn_sched_barrier(0);
            sbar();
            __builtin_amdgcn_sched_barrier(0);
            unit(f0, af0, 0); unit(f1, af0, 1);
            __builtin_amdgcn_sched_barrier(0);
            sbar();
        }
        // ---- phase B
        {
            Run r0, r1;
            RawA4 ra;
            rdA4(ra, so, pipe::IC<1>{}); rdRunA(r0, so, pipe::IC<0>{}, pipe::IC<2>{}); rdRunA(r1, so, pipe::IC<1>{}, pipe::IC<0>{});
            __builtin_amdgcn_sched_barrier(0);
            ALL9S_PART1(po);
            wait_a4(ra);
            __builtin_amdgcn_sched_barrier(0);
            asmA4(af1, ra);
            const Frag3 f0 = shuffle(r0), f1 = shuffle(r1);
            __builtin_amdgcn_sched_barrier(0);
            sbar();
            __builtin_amdgcn_sched_barrier(0);
            unit(f0, af0, 2); unit(f1, af1, 0);
            if (bias_step) {                                            // db partial: sum_q dz[q][co] of this wave's fragment wn, K half 0
                if (wn == 0) mma(ones, af0[0], bacc); else if (wn == 1) mma(ones, af0[1], bacc); else if (wn == 2) mma(ones, af0[2], bacc); else mma(ones, af0[3], bacc);
            }
            __builtin_amdgcn_sched_barrier(0);
            sbar();
        }
        // ---- phase C
        {
            Run r0, r1;
            rdRunA(r0, so, pipe::IC<1>{}, pipe::IC<1>{}); rdRunA(r1, so, pipe::IC<1>{}, pipe::IC<2>{});
            __builtin_amdgcn_sched_barrier(0);
            ALL9S_PART2(po);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const Frag3 f0 = shuffle(r0), f1 = shuffle(r1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLOTS) : "memory");   // all but tile s + 2's six: tile s + 1 has landed (this wave's pieces)
            sbar();
            __builtin_amdgcn_sched_barrier(0);
            unit(f0, af1, 1); unit(f1, af1, 2);
            if (bias_step) {
                if (wn == 0) mma(ones, af1[0], bacc); else if (wn == 1) mma(ones, af1[1], bacc); else if (wn == 2) mma(ones, af1[2], bacc); else mma(ones, af1[3], bacc);
            }
            __builtin_amdgcn_sched_barrier(0);
            sbar();
        }
        if (++bctr == a.tiles_ci) bctr = 0;
        { const unsigned t3 = so; so = no; no = po; po = t3; }
    }
#undef ALL9S_PART0
#undef ALL9S_PART1
#undef ALL9S_PART2
    if (nsteps > 0 && wm == 0) sbar();                                  // the groups meet again
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the over-run tiles land before the workgroup ends
    {
        float* P = a.partial + (((long long)split * (a.tiles_co * a.tiles_ci) + t) * 8 + wave) * (36 * 256) + lane * 4;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                *(f32x4*)(P + (tp * 4 + mi) * 256) = acc[tp][mi];
    }
    if (do_bias && lane < 16)       // every row of the ones-product holds the column sum
        a.bpartial[((long long)split * a.tiles_ci + tile_ci) * a.co_pad + tile_co * 128 + wm * 64 + wn * 16 + lane] = bacc.x;
}

// ------------------------------------------------------------------------------------------------ wide 1x1 layers, round 3: LDS-DMA ring
// The heads' weight gradients (2048 couts x 512 / 256 input channels) with wgrad_all9_kernel's pipeline: 256(co) x 256(ci)
// tile, 8 waves as 4(co) x 2(ci) (64 x 128 per wave, 4 x 8 accumulator fragments), 32-row K steps, FOUR 32-KB stages (dz tile +
// x tile, 32 rows x 512 B each) filled by inline-asm LDS-DMA: tile s+3 is issued behind the barrier of step s (into the stage
// tile s-1 has left), tile s+1 is waited for (counted vmcnt) in front of it, and the barrier sits in front of the step's last
// three units, under which the next step's first fragments are read.  One workgroup per CU (256 workgroups, one round): the
// slab is the register dump (67 MB instead of 201 MB for the 768-workgroup wgrad_wide_kernel).  Compact walk over the valid
// rows of each image (a 1x1 tap needs no halo rows).  The bias partial comes from ones-MFMAs of the waves' dz fragments (wave
// column wn takes fragments 2 wn, 2 wn + 1) in the steps assigned to the workgroup's ci tile.
// Requires dz_c % 256 == 0, x_c % 256 == 0.  Grid: (co-tile, ci-tile) x split-K; bias partial rows = splits * tiles_ci.
// GEN (round 4, dbx_heads1_wgrad_gen): the dz tile is not loaded but GENERATED -- dz = the heads' hidden gradient d_hid = keep * (d_out W2)
// (common.hpp: GenHid): per step wave w owns the 32 hidden channels 32 w .. 32 w + 31 of the tile's 32 pixel rows: one
// v_mfma_f32_32x32x16 on the pixels' d_out slots (loaded two steps ahead: 16 bytes per lane, after the step's two x pieces in the vmcnt
// order), one hash per lane, 16 selects, four ds_write_b64 into the stage the DMA version fills.  The generated tile's LDS map XORs the
// row bits 2 and 4 into address bits 3 and 4 on top of swz16w's (swzg): the 32 lanes of a write hold 32 different rows of ONE column --
// 32 different 8-byte bank pairs instead of 8 -- and the transposing reads keep their two-pass pattern.  The 944 MB hidden gradient is
// never read (nor written, once the data-gradient GEMM generates it too).
__device__ __forceinline__ int swzg(int r, int b) {
    return r * 512 + (b ^ (((r & 3) << 5) | (((r >> 3) & 1) << 7) | (((r >> 2) & 1) << 3) | (((r >> 4) & 1) << 4)));
}
#ifndef GEN_ABL
#define GEN_ABL 0                                                       // lab builds: 1 no mask + store, 2 no generating MFMA, 4 no d_out fetch
#endif
template <typename T, bool GEN = false>
__global__ __launch_bounds__(512) void wgrad_wide2_kernel(const WgradArgs a, const GenHid gh) {
    static_assert(sizeof(T) == 2, "16-bit tiles");
    constexpr int R = 32, T_BYTES = R * 512, STAGE = 2 * T_BYTES, NST = 4;
    constexpr int SLOTS = GEN ? 3 : 4;                                  // loads per wave and step: 32 pieces of 1 KiB (2 rows) per stage over 8 waves; GEN: two x pieces + one d_out slot
    extern __shared__ __attribute__((aligned(16))) char smem[];        // 4 x 32 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                           // wave tile 64(co) x 128(ci)
    int t, split;
    wg_tile_split(a.tiles_co * a.tiles_ci, a.nsplit, t, split);
    const int tile_ci = t % a.tiles_ci, tile_co = t / a.tiles_ci;
    const bool compact = a.spi > 0;
    const int gs0 = compact ? split * a.steps_per_split : (int)(((long long)split * a.rows_per_split) / R);
    const long long q0 = compact ? 0 : (long long)split * a.rows_per_split;
    long long q1 = q0 + a.rows_per_split; if (q1 > a.Q) q1 = a.Q;
    int nsteps = q1 > q0 ? (int)((q1 - q0 + R - 1) / R) : 0;
    if (compact) { const int gs1 = min(gs0 + a.steps_per_split, a.steps_total); nsteps = gs1 > gs0 ? gs1 - gs0 : 0; }

    // ---- LDS-DMA sources: a piece = 2 rows x 512 B (lane: row l>>5, chunk position l&31); position p of LDS row r receives source
    // chunk p ^ ((r & 3) << 1 | ((r >> 3) & 1) << 3) (swz16w); with piece = wave + 8 slot the piece-dependent bits depend on the wave only
    const long long dzrow = (long long)a.dz_ld * 2, xrow = (long long)a.x_ld * 2;
    const unsigned chunk = (unsigned)((lane & 31) ^ ((lane >> 5) << 1) ^ ((wave & 1) << 2) ^ (((wave >> 2) & 1) << 3));
    const unsigned vA = (unsigned)(lane >> 5) * (unsigned)dzrow + (chunk << 4), vB = (unsigned)(lane >> 5) * (unsigned)xrow + (chunk << 4);
    int ij = 0;
    long long iq0;
    if (compact) { const int img = gs0 / a.spi; ij = gs0 - img * a.spi; iq0 = (long long)img * a.img_rows + a.row0 + (long long)R * ij; }
    else iq0 = q0;
    const long long xshift = (long long)a.shift0 * (a.wp + 1);         // 1x1: tap (0, 0)
    const char* ap = a.dz + tile_co * 512 + iq0 * dzrow;
    const char* bp = a.x + tile_ci * 512 + (xshift + iq0) * xrow;
    const long long jump = compact ? a.img_rows - (long long)R * (a.spi - 1) : R;
    const long long aR = R * dzrow, bR = R * xrow, ajmp = jump * dzrow, bjmp = jump * xrow;
    const unsigned sA0 = (unsigned)(2 * wave) * (unsigned)dzrow, sA1 = sA0 + 16u * (unsigned)dzrow;
    const unsigned sB0 = (unsigned)(2 * wave) * (unsigned)xrow, sB1 = sB0 + 16u * (unsigned)xrow;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ldsw = lds0 + wave * 1024;
    int issued = 0;
    auto glds = [](const char* src, unsigned voff, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
    };
#define WIDE2_ISSUE(sb)                                                                     \
    do {                                                                                    \
        if (!GEN) {                                                                         \
            glds(ap + sA0, vA, ldsw + (sb));                                                \
            glds(ap + sA1, vA, ldsw + (sb) + 8 * 1024);                                     \
        }                                                                                   \
        glds(bp + sB0, vB, ldsw + (sb) + T_BYTES);                                          \
        glds(bp + sB1, vB, ldsw + (sb) + T_BYTES + 8 * 1024);                               \
        if (++issued < nsteps) {                                                            \
            if (compact && ++ij == a.spi) { ij = 0; ap += ajmp; bp += bjmp; }               \
            else { ap += aR; bp += bR; }                                                    \
        }                                                                                   \
    } while (0)

    // ---- GEN: the d_out slots of the tiles two steps ahead, the generating MFMA's A operand, the walk of the fetches.  The walk is
    // incremental (a frame position 32 further on is the same row or the next one: wp >= 32): no division, no 32-bit multiply per step
    // (this kernel's VALU instructions are not hidden by the other wave's MFMAs: both waves of a SIMD generate behind the same barrier --
    // 190 of them per step cost 0.6 us of a 1.0 us step in the first version)
    __shared__ u32x2 glut[16];                                          // keep nibble -> masks of two packed 16-bit pairs
    const int ghd = (tile_co * 256) / 512, gc32 = tile_co * 8 + wave;  // head of the tile's hidden channels; the wave's 32-channel block (hash index)
    u32x4 gw = {0u, 0u, 0u, 0u}, dq[2];
    int dm[2] = {-1, -1};
    unsigned dh[2] = {0u, 0u};                                          // m * 0x9E3779B1 of the slots' pixels
    int fimg = 0, fij = 0, fissued = 0;
    int gfx0 = 0, gy0 = 0, gfx = 0, gy = 0, gm = 0;                     // frame column / map row of the lane's pixel in the next tile to fetch, its compact index
    unsigned ghm = 0u;
    const int gdw = 32 - a.wp + gh.W;                                   // compact-index step when the walk wraps into the next frame row
    const unsigned gh32 = 32u * 0x9E3779B1u, ghw = (unsigned)gdw * 0x9E3779B1u, gcc = gh.seed ^ ((unsigned)gc32 * 0x85EBCA77u);
    const unsigned ld2 = (unsigned)gh.ld * 2u, goff = (unsigned)(ghd * gh.slot) * 2u;
    int glc = 0;                                                        // lane constant: compact index of the lane's pixel in image 0's first tile
    unsigned glh = 0u;
    const unsigned gimg_h = (unsigned)(gh.H * gh.W) * 0x9E3779B1u;
    if constexpr (GEN) {
        if (tid < 16) glut[tid] = (u32x2){((tid & 1) ? 0xffffu : 0u) | ((tid & 2) ? 0xffff0000u : 0u), ((tid & 4) ? 0xffffu : 0u) | ((tid & 8) ? 0xffff0000u : 0u)};
        gw = genhid_wfrag<T>(gh, ghd, (tile_co * 256) % 512 + 32 * wave + (lane & 31), lane);
        fimg = gs0 / a.spi; fij = gs0 - fimg * a.spi;
        const int q0l = a.row0 + (lane & 31);                           // the lane's frame position in an image's first tile
        const int fy0 = q0l / a.wp;
        gfx0 = q0l - fy0 * a.wp; gy0 = fy0 - gh.pad;
        glc = gy0 * gh.W + gfx0 - gh.pad; glh = (unsigned)glc * 0x9E3779B1u;
        gfx = gfx0; gy = gy0; gm = fimg * gh.H * gh.W + glc; ghm = (unsigned)fimg * gimg_h + glh;
        for (int t = 0; t < fij; ++t) {                                 // (a split that starts inside an image: once per workgroup)
            gfx += 32; gm += 32; ghm += gh32;
            if (gfx >= a.wp) { gfx -= a.wp; ++gy; gm += gdw - 32; ghm += ghw - gh32; }
        }
        __syncthreads();
    }
    // the slot load is inline asm like the LDS-DMA pieces: the compiler's own wait for a load it tracks is vmcnt(0) in front of the
    // generating MFMA -- behind the two x pieces just issued, a full memory latency per step
    auto gfetch = [&](int j) {
        const bool valid = (unsigned)gy < (unsigned)gh.H && (unsigned)(gfx - gh.pad) < (unsigned)gh.W;
        const unsigned voff = __umul24((unsigned)(valid ? gm : 0), ld2) + goff;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dq[j]) : "v"(voff), "s"(gh.dout) : "memory");
        dm[j] = valid ? gm : -1; dh[j] = ghm;
        // advance to the next tile (branch-free: uniform selects; the last tiles of a split re-fetch its last tile)
        ++fissued;
        const bool adv = fissued < nsteps, newimg = adv && fij + 1 == a.spi;
        fij = newimg ? 0 : (adv ? fij + 1 : fij);
        fimg += newimg ? 1 : 0;
        const int nx = gfx + 32;
        const bool wrap = nx >= a.wp;
        const int sx = wrap ? nx - a.wp : nx, sy = gy + (wrap ? 1 : 0), sm = gm + (wrap ? gdw : 32);
        const unsigned sh = ghm + (wrap ? ghw : gh32);
        gfx = newimg ? gfx0 : (adv ? sx : gfx);
        gy = newimg ? gy0 : (adv ? sy : gy);
        gm = newimg ? fimg * gh.H * gh.W + glc : (adv ? sm : gm);
        ghm = newimg ? (unsigned)fimg * gimg_h + glh : (adv ? sh : ghm);
    };
    f32x16 gd;
    auto ggen_mma = [&](int j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gd[r] = 0.f;
        Mma32<T>::run(gw, genhid_bfrag(dq[j], dm[j], lane), gd);           // (dq[j] is tied to the counted wait in front of the barrier)
    };
    auto ggen_store = [&](unsigned sb, int j) {
        const f32x16 d = gd;
        unsigned x = gcc ^ dh[j];                                       // dbx_drop_hash32 with its two products precomputed (branch-free: no dropout = all bits)
        x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
        const unsigned bits = (x >> (4 * (lane >> 5))) | (gh.use_hash ? 0u : 0xffffffffu);
        char* row = smem + sb;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const u32x2 mk = glut[(bits >> (8 * b)) & 15u];
            typedef float f32x2v __attribute__((ext_vector_type(2)));
            typedef T t2v __attribute__((ext_vector_type(2)));
            const unsigned p0 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v){d[4 * b], d[4 * b + 1]}, t2v));
            const unsigned p1 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v){d[4 * b + 2], d[4 * b + 3]}, t2v));
            *(u32x2*)(row + swzg(lane & 31, (32 * wave + 8 * b + 4 * (lane >> 5)) * 2)) = (u32x2){p0 & mk.x, p1 & mk.y};
        }
    };

    f32x4 acc[4][8];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const unsigned one2 = DType<T>::id == DBX_F16 ? 0x3C003C00u : 0x3F803F80u;
    const u32x4 ones = {one2, one2, one2, one2};
    const bool do_bias = a.bpartial != nullptr;

    const int g = lane >> 4, rsub = (lane & 15) >> 2, csub = (lane & 3) * 8;
    auto trd = [&](const char* p) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p));
    };
    auto rdA = [&](const char* Sb, int mi) {
        const int r0 = 8 * g + rsub, cbyte = (wm * 64 + mi * 16) * 2 + csub;
        const u32x2 lo = trd(Sb + (GEN ? swzg(r0, cbyte) : swz16w(r0, cbyte))), hi = trd(Sb + (GEN ? swzg(r0 + 4, cbyte) : swz16w(r0 + 4, cbyte)));
        return (u32x4){lo.x, lo.y, hi.x, hi.y};
    };
    auto rdB = [&](const char* Sb, int ni) {
        const int r0 = 8 * g + rsub, cbyte = (wn * 128 + ni * 16) * 2 + csub;
        const u32x2 lo = trd(Sb + T_BYTES + swz16w(r0, cbyte)), hi = trd(Sb + T_BYTES + swz16w(r0 + 4, cbyte));
        return (u32x4){lo.x, lo.y, hi.x, hi.y};
    };
    auto mma = [&](const u32x4& xf, const u32x4& zf, f32x4& c) {
        if constexpr (DType<T>::id == DBX_F16)
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xf), __builtin_bit_cast(f16x8, zf), c, 0, 0, 0);
        else
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xf), __builtin_bit_cast(bf16x8, zf), c, 0, 0, 0);
    };

    u32x4 af[2][4], bf[2];
    if (nsteps > 0) {
        WIDE2_ISSUE(0u);
        WIDE2_ISSUE((unsigned)STAGE);
        WIDE2_ISSUE((unsigned)(2 * STAGE));
        if constexpr (GEN) {                                            // tiles 0..2 (cold start: their d_out loads are waited for one by one)
            gfetch(0); gfetch(1);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(dq[0]), "+v"(dq[1]) :: "memory");
            ggen_mma(0); ggen_store(0u, 0); ggen_mma(1); ggen_store((unsigned)STAGE, 1);
            gfetch(0);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(dq[0]) :: "memory");
            ggen_mma(0); ggen_store((unsigned)(2 * STAGE), 0);
            gfetch(1); gfetch(0);                                       // tiles 3 and 4
            asm volatile("s_waitcnt vmcnt(1)" : "+v"(dq[1]) :: "memory");   // (tile 3's slot: the loop's counted wait assumes x pieces behind it)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SLOTS) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[0][mi] = rdA(smem, mi);
        bf[0] = rdB(smem, 0);
    }
    // one 32-row step per iteration: eight units ni of 4 (mi) MFMAs; stages rotate as uniform byte offsets
    unsigned so = 0, no = STAGE, n2 = 2 * STAGE, po = 3 * STAGE;        // tile s, s+1, s+2, and the stage tile s+3 goes to
    int bctr = gs0 % a.tiles_ci;
    // af[0] holds this step's dz fragments, af[1] receives the next step's (the roles swap every step: two unrolled bodies)
    auto body = [&](auto CUR_) {
        constexpr int CUR = decltype(CUR_)::value, NXT = CUR ^ 1;
        const char* Sb = smem + so;
        const char* Nb = smem + no;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u == 5) {
                // tile s+1 has landed (this wave's pieces; the barrier covers the others'), every wave is past tile s-1
                if constexpr (GEN) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(dq[NXT]) : "n"(SLOTS) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLOTS) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                WIDE2_ISSUE(po);                                        // tile s + 3
                if constexpr (GEN && !(GEN_ABL & 2)) ggen_mma(NXT);     // its dz half from the slot fetched two steps ago ...
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (GEN) {
                if (u == 7) { if (!(GEN_ABL & 1)) ggen_store(po, NXT); if (!(GEN_ABL & 4)) gfetch(NXT); }       // ... masked and written under the step's last MFMAs; tile s + 5's slot
            }
            const u32x4 b = bf[u & 1];
            if (u < 7) bf[(u + 1) & 1] = rdB(Sb, u + 1);
            else bf[0] = rdB(Nb, 0);
            if (u == 5) { af[NXT][0] = rdA(Nb, 0); af[NXT][1] = rdA(Nb, 1); }
            if (u == 6) { af[NXT][2] = rdA(Nb, 2); af[NXT][3] = rdA(Nb, 3); }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) mma(b, af[CUR][mi], acc[mi][u]);
            if (u == 5 || u == 6) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        if (do_bias && bctr == tile_ci) {
            __builtin_amdgcn_sched_barrier(0);
            if (wn == 0) { mma(ones, af[CUR][0], bacc[0]); mma(ones, af[CUR][1], bacc[1]); }
            else { mma(ones, af[CUR][2], bacc[0]); mma(ones, af[CUR][3], bacc[1]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (++bctr == a.tiles_ci) bctr = 0;
        { const unsigned t4 = so; so = no; no = n2; n2 = po; po = t4; }
    };
    for (int s = 0; s < nsteps; s += 2) {
        body(IC2<0>{});
        if (s + 1 < nsteps) body(IC2<1>{});
    }
#undef WIDE2_ISSUE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the over-run tiles land before the workgroup ends
    {
        // slab = register dump [split][tile][wave][mi * 8 + ni][lane][4 ci]: 1-KiB store instructions (wgrad_reduce_wide2_kernel)
        float* P = a.partial + (((long long)split * (a.tiles_co * a.tiles_ci) + t) * 8 + wave) * (32 * 256) + lane * 4;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni)
                *(f32x4*)(P + (mi * 8 + ni) * 256) = acc[mi][ni];
    }
    if (do_bias && lane < 16) {
        float* bpt = a.bpartial + ((long long)split * a.tiles_ci + tile_ci) * a.co_pad + tile_co * 256 + wm * 64 + wn * 32 + lane;
        bpt[0] = bacc[0].x; bpt[16] = bacc[1].x;
    }
}

// ------------------------------------------------------------------------------------------------ 3x3, all taps per workgroup
// For layers with few channels (Cout, Cin <= 128 at 240x240 / 120x120) the per-tap tiling above is bound by refilling
// LDS: every tap re-reads the same dz rows and a shifted copy of the same x rows (32 FLOP per byte filled).  Here one
// workgroup owns a 64(co) x 64(ci) tile for ALL nine taps: per K step it stages 32 rows of dz once and three row bands
// of x (one per ky, 32+2 rows so that the three kx shifts are just +0/+1/+2 row offsets of the transpose reads), and
// issues 9 x 4 MFMAs per wave per barrier.  4.3x fewer bytes through L2/LDS per FLOP; conv1_1's dz is streamed once
// instead of nine times.  16-bit types only (the f32 parity path keeps the generic kernel).
template <typename T>
__global__ __launch_bounds__(256, 2) void wgrad3x3_kernel(const WgradArgs a) {
    static_assert(sizeof(T) == 2, "16-bit tiles");
    // Per ky band the three kx fragments of a channel column come from ONE 12-row register run (see wgrad_row3_kernel):
    // 22 transpose reads per 36 MFMAs instead of 40.  (64-row K steps do not fit: 144 accumulator registers + staging.)
    constexpr int R = 32, BAND = R + 2, BROWS = BAND + 2;              // +2 rows read by the runs, never used
    constexpr int A_BYTES = R * 128, B_BYTES = 3 * BROWS * 128;        // 64 channels x 2 B per row
    constexpr int A_CHUNKS = R * 8, B_CHUNKS = 3 * BAND * 8;           // 256 and 816 16-byte chunks
    constexpr int LD_A = A_CHUNKS / 256, LD_B = (B_CHUNKS + 255) / 256;   // 1 and 4 (last one partial) per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];        // 2 x (A_BYTES + B_BYTES) = 35 KB
    char* As = smem;
    char* Bs = smem + 2 * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                           // wave tile 32(co) x 32(ci)
    int t, split;
    wg_tile_split(a.tiles_co * a.tiles_ci, a.nsplit, t, split);
    const int tile_ci = t % a.tiles_ci, tile_co = t / a.tiles_ci;
    const long long q0 = (long long)split * a.rows_per_split;
    long long q1 = q0 + a.rows_per_split; if (q1 > a.Q) q1 = a.Q;
    const int nsteps = q1 > q0 ? (int)((q1 - q0 + R - 1) / R) : 0;
    const bool do_bias = (tile_ci == 0 && a.bpartial != nullptr);

    // ---- loaders: dz R rows x 8 chunks; x 3 bands x (R+2) rows x 8 chunks
    const int ca = tid & 7, ra = tid >> 3;
    const bool a_ok = (tile_co * 128 + ca * 16) < a.dz_c * 2;
    const char* ap = a.dz + ((q0 + ra) * a.dz_ld + tile_co * 64) * 2LL + ca * 16;
    const long long a_row = (long long)a.dz_ld * 2, b_row = (long long)a.x_ld * 2;
    const char* bp[LD_B];
    int b_lds[LD_B];
    bool b_ok[LD_B];
#pragma unroll
    for (int i = 0; i < LD_B; ++i) {
        const int c = tid + 256 * i;                                  // chunk id
        const int row = c >> 3, cb = c & 7;                           // (band, j) = (row / BAND, row % BAND), chunk in row
        const int band = row / BAND, j = row - band * BAND;
        b_ok[i] = c < B_CHUNKS && (tile_ci * 128 + cb * 16) < a.x_c * 2;
        const long long grow = q0 + (long long)(band + a.shift0) * a.wp + a.shift0 + j;
        bp[i] = a.x + (grow * a.x_ld + tile_ci * 64) * 2LL + cb * 16;
        b_lds[i] = c < B_CHUNKS ? swz16<64>(band * BROWS + j, cb * 16) : 0;
    }
    u32x4 areg[LD_A], breg[LD_B];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    auto gload = [&](int s) {
#pragma unroll
        for (int i = 0; i < LD_A; ++i) areg[i] = a_ok ? *(const u32x4*)(ap + ((long long)s * R + 32 * i) * a_row) : zero4;
#pragma unroll
        for (int i = 0; i < LD_B; ++i) breg[i] = b_ok[i] ? *(const u32x4*)(bp[i] + (long long)s * R * b_row) : zero4;
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LD_A; ++i) *(u32x4*)(As + buf * A_BYTES + swz16<64>(ra + 32 * i, ca * 16)) = areg[i];
#pragma unroll
        for (int i = 0; i < LD_B; ++i)
            if (tid + 256 * i < B_CHUNKS) *(u32x4*)(Bs + buf * B_BYTES + b_lds[i]) = breg[i];
    };

    f32x4 acc[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[t][mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    auto bias_acc = [&]() {
#pragma unroll
        for (int i = 0; i < LD_A; ++i) {
            const T* e = (const T*)&areg[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum[j] += to_f32(e[j]);
        }
    };

    if (nsteps > 0) { gload(0); if (do_bias) bias_acc(); lstore(0); }
    __syncthreads();
    const int g = lane >> 4, rsub = (lane & 15) >> 2, csub = (lane & 3) * 8;
    auto trd = [&](const char* p) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p));
    };
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
        const char* Ab = As + buf * A_BYTES;
        const char* Bb = Bs + buf * B_BYTES;
        // software pipeline over the six (ky, ni) units of 3 x 2 MFMAs: the 12-row run of the next unit is in flight while
        // the current unit computes; sched_group_barrier pins the interleave.
        const int r0 = 8 * g + rsub;
        u32x4 af[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int cbyte = (wm * 32 + mi * 16) * 2 + csub;
            const u32x2 lo = trd(Ab + swz16<64>(r0, cbyte)), hi = trd(Ab + swz16<64>(r0 + 4, cbyte));
            af[mi] = (u32x4){lo.x, lo.y, hi.x, hi.y};
        }
        struct Run { u32x2 c0, c1, c2; };      // rows rb..rb+11 of one channel column: row pairs (0,1)(2,3) | (4,5)(6,7) | (8,9)(10,11)
        auto rdRun = [&](int u) {
            const int cbyte = (wn * 32 + (u & 1) * 16) * 2 + csub, rb = (u >> 1) * BROWS + r0;
            Run r;
            r.c0 = trd(Bb + swz16<64>(rb, cbyte)); r.c1 = trd(Bb + swz16<64>(rb + 4, cbyte)); r.c2 = trd(Bb + swz16<64>(rb + 8, cbyte));
            return r;
        };
        Run run[2];
        run[0] = rdRun(0);
        __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int ky = u >> 1, ni = u & 1;
            const Run c = run[u & 1];
            if (u < 5) run[(u + 1) & 1] = rdRun(u + 1);
            u32x4 bf[3];
            bf[0] = (u32x4){c.c0.x, c.c0.y, c.c1.x, c.c1.y};
            bf[1] = (u32x4){__builtin_amdgcn_alignbit(c.c0.y, c.c0.x, 16), __builtin_amdgcn_alignbit(c.c1.x, c.c0.y, 16),
                            __builtin_amdgcn_alignbit(c.c1.y, c.c1.x, 16), __builtin_amdgcn_alignbit(c.c2.x, c.c1.y, 16)};
            bf[2] = (u32x4){c.c0.y, c.c1.x, c.c1.y, c.c2.x};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    f32x4& cc = acc[ky * 3 + kx][mi][ni];
                    if constexpr (DType<T>::id == DBX_F16)
                        cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bf[kx]), __builtin_bit_cast(f16x8, af[mi]), cc, 0, 0, 0);
                    else
                        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[kx]), __builtin_bit_cast(bf16x8, af[mi]), cc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (u < 5) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        if (s + 1 < nsteps) { if (do_bias) bias_acc(); lstore(buf ^ 1); }
        __syncthreads();
    }

    {
        float* P = a.partial + (((long long)split * a.co_pad) * 9) * a.ci_pad;
        // x is the first MFMA operand: a lane holds four consecutive ci of one co -> one 16-byte store per fragment
        const int co_b = tile_co * 64 + wm * 32 + (lane & 15);
        const int ci_b = tile_ci * 64 + wn * 32 + (lane >> 4) * 4;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    *(f32x4*)(P + ((long long)(co_b + mi * 16) * 9 + t) * a.ci_pad + ci_b + ni * 16) = acc[t][mi][ni];
    }
    if (do_bias) {
        __syncthreads();
        float* red = (float*)smem;                       // [32 rows][64]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[ra * 64 + ca * 8 + j] = bsum[j];
        __syncthreads();
        if (tid < 64) {
            float sum = 0.f;
            for (int r = 0; r < 32; ++r) sum += red[r * 64 + tid];
            a.bpartial[(long long)split * a.co_pad + tile_co * 64 + tid] = sum;
        }
    }
}


// ------------------------------------------------------------------------------------------------ 3x3, all taps, column-strip walk
// wgrad3x3_kernel walks the frame linearly: the x row it stages as band ky = 2 is needed again as ky = 1 one frame row (7.6
// steps at 240x240) later and as ky = 0 after another -- by then 512 workgroups have streamed 66 MB through the 32 MB of L2, so
// every x row is fetched three times from HBM / MALL (conv1_2: 1.9 GB for 0.96 GB of operands; the kernel ran at the same
// speed with ONE workgroup per CU: bandwidth-bound, not occupancy-bound).  Here a workgroup walks DOWN a 32-pixel column
// strip of one image: step y needs x rows y-1, y, y+1 of the strip (34 pixels with the kx halo), two of which the previous step
// already staged -- a four-slot ring of x rows in LDS, ONE new row (4.3 KB) and one dz row (4 KB) per step instead of 17 KB.
// dz pixels of the last strip that lie past the frame width are zeroed (they alias the next frame row).  Same tile (64 x 64,
// all nine taps), fragment reads, MFMA schedule, slab layout and bias sums as wgrad3x3_kernel; the K range of a workgroup is
// units_per_split consecutive (image, strip) units.
// POOLDZ (round 6): conv1_2's weight gradient reads pool1's backward (d_a12: 472 MB at batch 64, written by dbx_maxpool2x2_bwd_idx and re-read
// here) straight from its sources instead: the pooled gradient d_p1 (118 MB) and the arg-max nibbles (30 MB).  The dz row of a step is
// register-staged already; a thread's chunk (eight channels of one pixel) becomes the pooled pixel's chunk ANDed with the eight 16-bit
// masks "this pixel is the window's arg-max and the maximum was positive" decoded from one dword of nibbles (22 VALU instructions).
#ifndef STRIP_MI
#define STRIP_MI 4
#endif
template <typename T, bool POOLDZ = false>
__global__ __launch_bounds__(256, 2) void wgrad3x3_strip_kernel(const WgradArgs a) {
    static_assert(sizeof(T) == 2, "16-bit tiles");
    constexpr int R = 32, BAND = R + 2, BROWS = BAND + 2, SLOTS = 4;
    constexpr int A_BYTES = R * 128, B_BYTES = SLOTS * BROWS * 128;   // 64 channels x 2 B per row
    extern __shared__ __attribute__((aligned(16))) char smem[];        // 2 x A_BYTES + B_BYTES = 26 KB
    char* As = smem;
    char* Bs = smem + 2 * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave tile: (16 MI) co x (16 NI) ci with MI NI = 4.  Round 6: 64 x 16 (MI 4: every wave reads all four dz fragments and ONE band run per ky:
    // 17 transposed LDS reads per 36 MFMAs) instead of 32 x 32 (MI 2: 22 reads) -- the kernel is bound by those reads (STRIP_MI=2: the old tiling)
    constexpr int MI = STRIP_MI, NI = 4 / MI;
    static_assert(MI == 2 || MI == 4, "wave tiling");
    const int wm = MI == 4 ? 0 : wave >> 1, wn = MI == 4 ? wave : wave & 1;
    int t, split;
    wg_tile_split(a.tiles_co * a.tiles_ci, a.nsplit, t, split);
    const int tile_ci = t % a.tiles_ci, tile_co = t / a.tiles_ci;
    const int u0 = split * a.units_per_split, u1 = min(u0 + a.units_per_split, a.units);
    const bool do_bias = (tile_ci == 0 && a.bpartial != nullptr);

    // ---- loaders: dz 32 pixels x 8 chunks (one per thread); x one row of 34 pixels x 8 chunks (threads 0..255 + 0..15)
    const int ca = tid & 7, ra = tid >> 3;
    const bool a_ch_ok = (tile_co * 128 + ca * 16) < a.dz_c * 2;
    const bool b_ch_ok = (tile_ci * 128 + ca * 16) < a.x_c * 2;
    const long long a_row = (long long)a.dz_ld * 2, b_row = (long long)a.x_ld * 2;
    const bool has2 = tid < (BAND * 8 - 256);                          // second x chunk: pixels 32, 33
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    u32x4 areg[2], breg[2][2];                                         // two register sets: loads run TWO steps ahead
    unsigned nreg[2] = {0u, 0u};                                        // POOLDZ: the chunk's eight arg-max nibbles

    f32x4 acc[9][MI][NI];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[t9][mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    auto bias_acc = [&](int set) {
        const T* e = (const T*)&areg[set];
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] += to_f32(e[j]);
    };
    const int g = lane >> 4, rsub = (lane & 15) >> 2, csub = (lane & 3) * 8;
    auto trd = [&](const char* p) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p));
    };

    for (int u = u0; u < u1; ++u) {
        const int n = u / a.nstrips, strip = u - n * a.nstrips;
        const int cx = strip * R;
        const long long qimg = (long long)n * a.hp * a.wp;
        bool a_ok = a_ch_ok && (cx + ra < a.wp);                        // dz pixels past the frame width alias the next row
        const char* ap = POOLDZ ? nullptr : a.dz + ((qimg + cx + ra) * a.dz_ld + tile_co * 64) * 2LL + ca * 16;
        // POOLDZ: this thread's pixel column x (image coordinates) is fixed for the unit: its pooled column, the parity that selects the
        // window position, and whether it lies inside the image at all (halo columns and the strip's overhang are zero)
        const int zx = cx + ra - a.z_pad;
        const int zpx = (zx >> 1) < 0 ? 0 : ((zx >> 1) < a.p_w ? (zx >> 1) : a.p_w - 1);
        if (POOLDZ) a_ok = a_ok && zx >= 0 && zx < a.z_w;
        const char* pp0 = nullptr; const unsigned char* ip0 = nullptr;
        char* zp0 = nullptr;
        if constexpr (POOLDZ) {
            if (a.zout && tile_ci == 0) zp0 = a.zout + ((qimg + cx + ra) * a.dz_ld + tile_co * 64) * 2LL + ca * 16;
            pp0 = a.pdy + (((long long)n * a.p_hp + a.p_pad) * a.p_wp + zpx + a.p_pad) * a.p_ld * 2LL + tile_co * 128 + ca * 16;
            ip0 = a.pidx + ((long long)n * a.p_h * a.p_w + zpx) * (a.z_ctot >> 1) + ((a.z_coff + tile_co * 64 + ca * 8) >> 1);
        }
        const char* bp = a.x + ((qimg + cx + a.shift0 + ra) * a.x_ld + tile_ci * 64) * 2LL + ca * 16;   // pixel ra of the 34
        // frame row fy of dz; x row index xr (may be -1 or hp: the neighbouring image's halo row / the zero guard, as in the linear walk)
        // Every lane ALWAYS loads (masked lanes read valid neighbouring memory and are zeroed at the LDS store; lanes without a
        // second x chunk re-read their first): loads inside divergent or uniform branches make the compiler's counter
        // bookkeeping fall back to s_waitcnt vmcnt(0) in front of the next issue, which serialises the two-step prefetch.
        const long long b2 = has2 ? 32 * b_row : 0;
        auto gload_a = [&](int set, int fy) {
            if constexpr (POOLDZ) {
                int py = (fy - a.z_pad) >> 1;                            // (uniform) halo rows read a valid row, zeroed at the LDS store
                py = py < 0 ? 0 : (py < a.p_h ? py : a.p_h - 1);
                areg[set] = *(const u32x4*)(pp0 + (long long)py * a.p_wp * a.p_ld * 2LL);
                nreg[set] = *(const unsigned*)(ip0 + (long long)py * a.p_w * (a.z_ctot >> 1));
            } else areg[set] = *(const u32x4*)(ap + (long long)fy * a.wp * a_row);
        };
        auto gload_b = [&](int set, int xr) {
            breg[set][0] = *(const u32x4*)(bp + (long long)xr * a.wp * b_row);
            breg[set][1] = *(const u32x4*)(bp + (long long)xr * a.wp * b_row + b2);
        };
        auto lstore_a = [&](int set, int buf, int fy) {
            if constexpr (POOLDZ) {
                const int zy = fy - a.z_pad;
                // nibble == 4 | position (bit 2: the window's maximum was > 0, the ReLU gate) -> an all-ones / zero mask per 16-bit element
                const unsigned want = (4u | (unsigned)((zy & 1) << 1) | (unsigned)(zx & 1)) * 0x11111111u;
                unsigned t = nreg[set] ^ want;
                t |= t >> 1; t |= t >> 2;
                const unsigned hit = ~t & 0x11111111u;                  // bit 4 c: channel c of the chunk routes its gradient to this pixel
                u32x4 v = areg[set];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int lo = (int)(hit << (31 - 8 * d)) >> 31, hi = (int)(hit << (27 - 8 * d)) >> 31;    // (v_bfe_i32: the bit, sign-extended)
                    v[d] &= ((unsigned)lo & 0xffffu) | ((unsigned)hi & 0xffff0000u);
                }
                areg[set] = (a_ok && zy >= 0 && zy < a.z_h) ? v : zero4;
            } else if (!a_ok) areg[set] = zero4;
            *(u32x4*)(As + buf * A_BYTES + swz16<64>(ra, ca * 16)) = areg[set];
            if constexpr (POOLDZ) {
                // the un-pooled gradient map as a by-product (its other consumer, conv1_2's data gradient, reads it): every frame pixel of the
                // strip's real columns exactly once over the grid, halo rows / columns as zeros -- dbx_maxpool2x2_bwd_idx's launch goes away
                if (zp0 && a_ch_ok && cx + ra < a.wp) *(u32x4*)(zp0 + (long long)fy * a.wp * a_row) = areg[set];
            }
        };
        auto lstore_b = [&](int set, int xr) {
            const int sb = ((xr + 4) & 3) * BROWS;
            *(u32x4*)(Bs + swz16<64>(sb + ra, ca * 16)) = b_ch_ok ? breg[set][0] : zero4;
            if (has2) *(u32x4*)(Bs + swz16<64>(sb + 32 + ra, ca * 16)) = b_ch_ok ? breg[set][1] : zero4;
        };
        // prologue of the unit: x rows shift0 + 0..2 and dz row 0
        __syncthreads();                                                 // the previous unit's last reads are done
#pragma unroll 1
        for (int i = 0; i < 3; ++i) { gload_b(0, a.shift0 + i); lstore_b(0, a.shift0 + i); }
        gload_a(0, 0); lstore_a(0, 0, 0); if (do_bias) bias_acc(0);
        const int nsteps = a.hp;
        { const int r1 = nsteps > 1 ? 1 : 0; gload_a(1, r1); gload_b(1, r1 + 2 + a.shift0); }   // step 1's operands: set 1
        __syncthreads();
        // step s: the loads of step s + 2 go to register set s & 1 (free since its contents were stored at the end of step s - 1);
        // set (s + 1) & 1 -- issued a whole step ago -- is stored at the end of this step.  The ~1.5 us global latency is
        // covered by two steps instead of one (one step of 36 MFMAs per wave is 0.3 us).
        auto step = [&](int s, auto SET_) {
            constexpr int SET = decltype(SET_)::value;
            const int buf = s & 1;
            { const int r2 = s + 2 < nsteps ? s + 2 : nsteps - 1; gload_a(SET, r2); gload_b(SET, r2 + 2 + a.shift0); }   // (clamped: no branch)
            const char* Ab = As + buf * A_BYTES;
            const int r0 = 8 * g + rsub;
            u32x4 af[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int cbyte = (wm * 16 * MI + mi * 16) * 2 + csub;
                const u32x2 lo = trd(Ab + swz16<64>(r0, cbyte)), hi = trd(Ab + swz16<64>(r0 + 4, cbyte));
                af[mi] = (u32x4){lo.x, lo.y, hi.x, hi.y};
            }
            struct Run { u32x2 c0, c1, c2; };
            auto rdRun = [&](int v) {
                const int cbyte = (wn * 16 * NI + (v % NI) * 16) * 2 + csub, rb = ((s + a.shift0 + (v / NI) + 4) & 3) * BROWS + r0;
                Run r;
                r.c0 = trd(Bs + swz16<64>(rb, cbyte)); r.c1 = trd(Bs + swz16<64>(rb + 4, cbyte)); r.c2 = trd(Bs + swz16<64>(rb + 8, cbyte));
                return r;
            };
            Run run[2];
            run[0] = rdRun(0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * MI + 3, 0);
            constexpr int NV = 3 * NI;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int ky = v / NI, ni = v % NI;
                const Run c = run[v & 1];
                if (v < NV - 1) run[(v + 1) & 1] = rdRun(v + 1);
                u32x4 bf[3];
                bf[0] = (u32x4){c.c0.x, c.c0.y, c.c1.x, c.c1.y};
                bf[1] = (u32x4){__builtin_amdgcn_alignbit(c.c0.y, c.c0.x, 16), __builtin_amdgcn_alignbit(c.c1.x, c.c0.y, 16),
                                __builtin_amdgcn_alignbit(c.c1.y, c.c1.x, 16), __builtin_amdgcn_alignbit(c.c2.x, c.c1.y, 16)};
                bf[2] = (u32x4){c.c0.y, c.c1.x, c.c1.y, c.c2.x};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        f32x4& cc = acc[ky * 3 + kx][mi][ni];
                        if constexpr (DType<T>::id == DBX_F16)
                            cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bf[kx]), __builtin_bit_cast(f16x8, af[mi]), cc, 0, 0, 0);
                        else
                            cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[kx]), __builtin_bit_cast(bf16x8, af[mi]), cc, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, MI, 0);
                    if (v < NV - 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            // the new x row goes to the slot of row s + shift0 - 1 (dead since the previous step's barrier), dz to the other buffer
            if (s + 1 < nsteps) { lstore_a(SET ^ 1, buf ^ 1, s + 1); if (do_bias) bias_acc(SET ^ 1); lstore_b(SET ^ 1, s + 3 + a.shift0); }
            __syncthreads();
        };
        for (int s = 0; s < nsteps; s += 2) {
            step(s, IC2<0>{});
            if (s + 1 < nsteps) step(s + 1, IC2<1>{});
        }
    }

    {
        float* P = a.partial + (((long long)split * a.co_pad) * 9) * a.ci_pad;
        const int co_b = tile_co * 64 + wm * 16 * MI + (lane & 15);
        const int ci_b = tile_ci * 64 + wn * 16 * NI + (lane >> 4) * 4;
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    *(f32x4*)(P + ((long long)(co_b + mi * 16) * 9 + t9) * a.ci_pad + ci_b + ni * 16) = acc[t9][mi][ni];
    }
    if (do_bias) {
        __syncthreads();
        float* red = (float*)smem;                       // [32 rows][64]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[ra * 64 + ca * 8 + j] = bsum[j];
        __syncthreads();
        if (tid < 64) {
            float sum = 0.f;
            for (int r = 0; r < 32; ++r) sum += red[r * 64 + tid];
            a.bpartial[(long long)split * a.co_pad + tile_co * 64 + tid] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------ 3x3 with Cin <= 8 (conv1_1)
// dW[co][tap][c] for an 8-channel (one 16-byte chunk per pixel) input: the GEMM's N dimension is (tap, channel) = 72
// columns -> five 16-column fragments (two taps x 8 channels each) instead of nine taps x a 64-channel tile that is 7/8
// padding.  Each transpose-read lane chooses its own LDS address, so a fragment gathers its two taps straight from the
// row bands.  The kernel streams dz once (HBM-bound: 472 MB for conv1_1 at batch 64) with up to 8 workgroups per CU.
template <typename T>
__global__ __launch_bounds__(256) void wgrad3x3_c8_kernel(const WgradArgs a) {
    static_assert(sizeof(T) == 2, "16-bit tiles");
    constexpr int R = 32, BAND = R + 2, XROWS = 3 * BAND + 2;           // +2: a zero row for the unused 10th tap, padding
    constexpr int A_BYTES = R * 128, X_BYTES = XROWS * 16;
    __shared__ __attribute__((aligned(16))) char smem[2 * (A_BYTES + X_BYTES)];
    char* As = smem;
    char* Xs = smem + 2 * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave w owns couts 16w .. 16w+15
    int tile, split;
    wg_tile_split(1, a.nsplit, tile, split);
    const long long q0 = (long long)split * a.rows_per_split;
    long long q1 = q0 + a.rows_per_split; if (q1 > a.Q) q1 = a.Q;
    const int nsteps = q1 > q0 ? (int)((q1 - q0 + R - 1) / R) : 0;
    const bool do_bias = a.bpartial != nullptr;

    const int ca = tid & 7, ra = tid >> 3;
    const bool a_ok = ca * 16 < a.dz_c * 2;
    const char* ap = a.dz + (q0 + ra) * (long long)a.dz_ld * 2 + ca * 16;
    const long long a_row = (long long)a.dz_ld * 2, b_row = (long long)a.x_ld * 2;
    const bool x_ld_ok = tid < 3 * BAND;
    const int xband = tid / BAND, xj = tid - xband * BAND;
    const char* xp = a.x + (q0 + (long long)(xband + a.shift0) * a.wp + a.shift0 + xj) * b_row;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    if (tid < 2) { *(u32x4*)(Xs + (3 * BAND + tid) * 16) = zero4; *(u32x4*)(Xs + X_BYTES + (3 * BAND + tid) * 16) = zero4; }
    u32x4 areg, xreg;
    auto gload = [&](int s) {
        areg = a_ok ? *(const u32x4*)(ap + (long long)s * R * a_row) : zero4;
        xreg = x_ld_ok ? *(const u32x4*)(xp + (long long)s * R * b_row) : zero4;
    };
    auto lstore = [&](int buf) {
        *(u32x4*)(As + buf * A_BYTES + swz16<64>(ra, ca * 16)) = areg;
        if (x_ld_ok) *(u32x4*)(Xs + buf * X_BYTES + tid * 16) = xreg;
    };
    f32x4 acc[5];
#pragma unroll
    for (int f = 0; f < 5; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    auto bias_acc = [&]() {
        const T* e = (const T*)&areg;
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] += to_f32(e[j]);
    };
    // transpose-read lane roles: lane L of a 16-lane group supplies row (L>>2) of the 4-row block, 4 columns (L&3)*4..+3
    const int g = lane >> 4, rsub = (lane & 15) >> 2, cq = lane & 3;
    int xoff[5];                                                        // byte offset of this lane's source for fragment f
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        const int tap = 2 * f + (cq >> 1);
        const int ky = tap / 3, kx = tap - ky * 3;
        xoff[f] = (ky * BAND + kx) * 16 + (cq & 1) * 8;                 // (the 10th tap is redirected to the zero row below)
    }
    if (nsteps > 0) { gload(0); if (do_bias) bias_acc(); lstore(0); }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
        const char* Ab = As + buf * A_BYTES;
        const char* Xb = Xs + buf * X_BYTES;
        const int cbyte = wave * 32 + cq * 8;
        const short4v alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(Ab + swz16<64>(8 * g + rsub, cbyte)));
        const short4v ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(Ab + swz16<64>(8 * g + 4 + rsub, cbyte)));
        const u32x2 al = __builtin_bit_cast(u32x2, alo), ah = __builtin_bit_cast(u32x2, ahi);
        const u32x4 af = {al.x, al.y, ah.x, ah.y};
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            const bool dead = (2 * f + (cq >> 1)) >= 9;                 // the 10th "tap": read the zero row
            const int r0 = dead ? 0 : (8 * g + rsub) * 16, r1 = dead ? 0 : (8 * g + 4 + rsub) * 16;
            const int base = dead ? (3 * BAND) * 16 : xoff[f];
            const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(Xb + base + r0));
            const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(Xb + base + r1));
            const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
            const u32x4 bf = {l2.x, l2.y, h2.x, h2.y};
            if constexpr (DType<T>::id == DBX_F16)
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af), __builtin_bit_cast(f16x8, bf), acc[f], 0, 0, 0);
            else
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc[f], 0, 0, 0);
        }
        if (s + 1 < nsteps) { if (do_bias) bias_acc(); lstore(buf ^ 1); }
        __syncthreads();
    }
    {   // slab [split][co (64)][tap (9)][ci_pad (8)]: fragment f column i = tap 2f + (i>>3), channel i&7
        float* P = a.partial + ((long long)split * a.co_pad) * 9 * a.ci_pad;
        const int co_b = wave * 16 + (lane >> 4) * 4, col = lane & 15;
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            const int tap = 2 * f + (col >> 3);
            if (tap >= 9) continue;
            const float v[4] = {acc[f].x, acc[f].y, acc[f].z, acc[f].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) P[((long long)(co_b + r) * 9 + tap) * a.ci_pad + (col & 7)] = v[r];
        }
    }
    if (do_bias) {
        __syncthreads();
        float* red = (float*)smem;                       // [32 rows][64]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[ra * 64 + ca * 8 + j] = bsum[j];
        __syncthreads();
        if (tid < 64) {
            float sum = 0.f;
            for (int r = 0; r < 32; ++r) sum += red[r * 64 + tid];
            a.bpartial[(long long)split * a.co_pad + tid] = sum;
        }
    }
}

// dw[co][ci][tap] = sum_s partial[s][co][tap][ci], db[co] = sum_s bpartial[s][co].  Deterministic: 4 lanes own one output
// element, lane g sums splits g, g+4, ... in ascending order, and the four partial sums are combined in the fixed order
// ((s0+s1)+(s2+s3)).  Consecutive elements of a slab row map to consecutive 4-lane groups (coalesced 16-element reads).
// Fixed-order split-K reduction: partial [splits][co_pad][taps][ci_pad] -> dw [co][ci][taps] (OIHW), bpartial -> db.
// A lane owns four consecutive ci (one 16-byte load per split); the four waves of a workgroup each sum every fourth split
// with independent loads in flight and are combined in a fixed order through LDS, so results are bitwise repeatable.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bpartial, int splits, int bsplits,
                                                           int co, int ci, int taps, int co_pad, int ci_pad, float* __restrict__ dw,
                                                           float* __restrict__ db, int accumulate, int ci_total, int ci_off) {
    const int c4n = ci_pad >> 2;                                       // 4-float groups per (co, tap) row
    const long long total4 = (long long)co * taps * c4n;
    const long long nb = (co + 3) / 4;                                 // bias groups (four couts each)
    const long long slab = (long long)co_pad * taps * ci_pad;
    const int g = threadIdx.x >> 6;                                   // split group 0..3 (one wave each)
    const int e = threadIdx.x & 63;
    __shared__ f32x4 red[4][64];
    for (long long base = (long long)blockIdx.x * 64; base < total4 + nb; base += (long long)gridDim.x * 64) {
        const long long i = base + e;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int o = 0, t = 0, c = 0;
        const bool is_w = i < total4, is_b = !is_w && i < total4 + nb && db != nullptr;
        if (is_w) {
            const int cg = (int)(i % c4n);
            t = (int)((i / c4n) % taps);
            o = (int)(i / ((long long)c4n * taps));
            c = cg * 4;
            const float* src = partial + ((long long)o * taps + t) * ci_pad + c;
            int k = g;
            for (; k + 12 < splits; k += 16) {                         // four independent 16-byte loads in flight
                const f32x4 v0 = *(const f32x4*)(src + (long long)k * slab), v1 = *(const f32x4*)(src + (long long)(k + 4) * slab);
                const f32x4 v2 = *(const f32x4*)(src + (long long)(k + 8) * slab), v3 = *(const f32x4*)(src + (long long)(k + 12) * slab);
                s += (v0 + v1) + (v2 + v3);
            }
            for (; k < splits; k += 4) s += *(const f32x4*)(src + (long long)k * slab);
        } else if (is_b) {
            o = (int)(i - total4) * 4;
            for (int k = g; k < bsplits; k += 4) {
                const float* bp = bpartial + (long long)k * co_pad + o;      // co_pad is a multiple of 64: in bounds
                s += (f32x4){bp[0], bp[1], bp[2], bp[3]};
            }
        }
        red[g][e] = s;
        __syncthreads();
        if (g == 0 && (is_w || is_b)) {
            const f32x4 v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
            const float vv[4] = {v.x, v.y, v.z, v.w};
            if (is_w) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < ci) {
                        float* out = dw + ((long long)o * ci_total + ci_off + c + j) * taps + t;
                        *out = accumulate ? *out + vv[j] : vv[j];
                    }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (o + j < co) db[o + j] = accumulate ? db[o + j] + vv[j] : vv[j];
            }
        }
        __syncthreads();
    }
}

// Split reduction for wgrad_all9_kernel's slabs (register dumps [split][tile][wave][tap * 4 + mi][lane][4]): a workgroup owns 64
// consecutive 16-byte elements of a slab (one fragment of one wave), its four waves sum every fourth split with coalesced 1-KiB
// loads and are combined in a fixed order through LDS (bitwise repeatable); element -> (co, tap, ci .. ci+3) -> fp32 OIHW.
// Bias: bpartial [bsplits][co_pad], summed the same way by the workgroups past the last slab element.
// WIDE: wgrad_wide2_kernel's dumps instead ([split][tile][wave][mi * 8 + ni][lane][4], 256 x 256 tiles, one tap).
template <bool WIDE>
__global__ __launch_bounds__(256) void wgrad_reduce_all9_kernel(const float* __restrict__ partial, const float* __restrict__ bpartial, int splits,
                                                                int bsplits, int co, int ci, int tiles_co, int tiles_ci, int co_pad,
                                                                float* __restrict__ dw, float* __restrict__ db, int accumulate, int ci_total, int ci_off) {
    constexpr int NF = WIDE ? 32 : 36;
    const int ntile = tiles_co * tiles_ci;
    const long long total4 = (long long)ntile * 8 * NF * 64;            // 16-byte elements per slab
    const long long nb = (co + 3) / 4;
    const long long slab4 = total4;
    const int g = threadIdx.x >> 6, e = threadIdx.x & 63;
    __shared__ f32x4 red[4][64];
    for (long long base = (long long)blockIdx.x * 64; base < total4 + nb; base += (long long)gridDim.x * 64) {
        const long long i = base + e;
        const bool is_w = i < total4, is_b = !is_w && i < total4 + nb && db != nullptr;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (is_w) {
            const f32x4* src = (const f32x4*)partial + i;
            int k = g;
            for (; k + 12 < splits; k += 16) {
                const f32x4 v0 = src[(long long)k * slab4], v1 = src[(long long)(k + 4) * slab4];
                const f32x4 v2 = src[(long long)(k + 8) * slab4], v3 = src[(long long)(k + 12) * slab4];
                s += (v0 + v1) + (v2 + v3);
            }
            for (; k < splits; k += 4) s += src[(long long)k * slab4];
        } else if (is_b) {
            const int o = (int)(i - total4) * 4;
            for (int k = g; k < bsplits; k += 4) {
                const float* bq = bpartial + (long long)k * co_pad + o;  // co_pad is a multiple of 128: in bounds
                s += (f32x4){bq[0], bq[1], bq[2], bq[3]};
            }
        }
        red[g][e] = s;
        __syncthreads();
        if (g == 0 && (is_w || is_b)) {
            const f32x4 v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
            const float vv[4] = {v.x, v.y, v.z, v.w};
            if (is_w) {
                const int lane = (int)(i & 63);
                long long r = i >> 6;
                const int f = (int)(r % NF); r /= NF;
                const int wave = (int)(r & 7); r >>= 3;
                const int tile = (int)r, tile_ci = tile % tiles_ci, tile_co = tile / tiles_ci;
                const int tp = WIDE ? 0 : f >> 2, mi = WIDE ? f >> 3 : f & 3, taps = WIDE ? 1 : 9;
                const int o = WIDE ? tile_co * 256 + (wave >> 1) * 64 + mi * 16 + (lane & 15) : tile_co * 128 + (wave >> 2) * 64 + mi * 16 + (lane & 15);
                const int c = WIDE ? tile_ci * 256 + (wave & 1) * 128 + (f & 7) * 16 + (lane >> 4) * 4 : tile_ci * 64 + (wave & 3) * 16 + (lane >> 4) * 4;
                if (o < co) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c + j < ci) {
                            float* out = dw + ((long long)o * ci_total + ci_off + c + j) * taps + tp;
                            *out = accumulate ? *out + vv[j] : vv[j];
                        }
                }
            } else {
                const int o = (int)(i - total4) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (o + j < co) db[o + j] = accumulate ? db[o + j] + vv[j] : vv[j];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ host
struct WgradPlan { int bmc, bnc, co_pad, ci_pad, tiles_co, tiles_ci, taps, splits, rows_per_split, alltaps, c8, row3, wide, all9, wide2, bsplits; long long Q;
                   int strip, nstrips, units, units_per_split, spi, steps_total, steps_per_split; };

static int device_cus() {            // CUs of the current device (workgroup targets of the one-round kernels)
    static int n = 0;
    if (!n) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; }
    return n;
}

static int wgrad_variant() {          // DBX_WGRAD_VARIANT=1: generic per-tap kernel everywhere (A/B testing)
    static int v = -1;
    if (v < 0) { const char* e = getenv("DBX_WGRAD_VARIANT"); v = e ? atoi(e) : 0; }
    return v;
}

static WgradPlan wgrad_plan(int dtype, const dbx_view* dz, const dbx_view* x, int kh, int kw) {
    WgradPlan p;
    p.bmc = dz->c > 64 ? 128 : 64;
    p.bnc = x->c > 64 ? 128 : 64;
    if (p.bmc != p.bnc) { p.bmc = 64; p.bnc = 64; }        // compiled tile shapes: 128x128 and 64x64
    // few-channel 3x3 layers (Cout or Cin < 128): one workgroup accumulates all nine taps of a 64x64 tile
    p.alltaps = (dtype != DBX_F32 && kh == 3 && kw == 3 && wgrad_variant() != 1 && ((dz->c <= 128 && x->c <= 128) || wgrad_variant() == 2)) ? 1 : 0;
    // 128 -> 128 (conv2_2): the row3 kernel measured 10 % faster than the all-taps tile (374 vs 417 us at batch 64; DBX_WGRAD_VARIANT=10: all-taps)
    if (dz->c == 128 && x->c == 128 && wgrad_variant() != 10 && wgrad_variant() != 2) p.alltaps = 0;
    if (p.alltaps) { p.bmc = 64; p.bnc = 64; }
    // conv1_1: 8-channel (one chunk per pixel) input, 64 couts: (tap, channel) pairs form the GEMM N dimension
    p.c8 = (p.alltaps && x->c * dbx_esize(dtype) == 16 && x->ld == x->c && dz->c == 64 && wgrad_variant() != 4) ? 1 : 0;
    // wide 3x3 layers: one ky row (three kx taps) per workgroup, 128x128 tiles (DBX_WGRAD_VARIANT=6: per-tap kernel)
    p.row3 = (!p.alltaps && dtype != DBX_F32 && kh == 3 && kw == 3 && dz->c % 128 == 0 && x->c % 128 == 0 && wgrad_variant() != 6) ? 1 : 0;
    // wide 1x1 layers: 256x256 tiles (DBX_WGRAD_VARIANT=7: 128x128 per-tap kernel)
    p.wide = (dtype != DBX_F32 && kh == 1 && kw == 1 && dz->c % 256 == 0 && x->c % 256 == 0 && wgrad_variant() != 7) ? 1 : 0;
    if (p.wide) { p.bmc = 256; p.bnc = 256; }
    p.co_pad = (dz->c + p.bmc - 1) / p.bmc * p.bmc;
    p.ci_pad = p.c8 ? 8 : (x->c + p.bnc - 1) / p.bnc * p.bnc;
    p.tiles_co = p.co_pad / p.bmc; p.tiles_ci = p.c8 ? 1 : p.ci_pad / p.bnc;
    p.taps = kh * kw;
    p.Q = (long long)dz->n * (dz->h + 2 * dz->pad) * (dz->w + 2 * dz->pad);
    const long long tiles = (long long)p.tiles_co * p.tiles_ci * (p.alltaps ? 1 : (p.row3 ? 3 : p.taps));
    const long long steps = (p.Q + 31) / 32;
    // aim for ~4 workgroups per CU (2 resident per CU); the c8 kernel streams dz once: 2 workgroups per CU suffice
    // row3: one 512-thread workgroup per CU, three full rounds of 256 workgroups
    static int wgs_big = 768;                                  // target workgroup count of the row3 / wide kernels
#ifdef DBX_LAB
    { static bool rd = false; if (!rd) { const char* e = getenv("DBX_WGRAD_WGS"); if (e) wgs_big = atoi(e); rd = true; } }   // lab builds: A/B
#endif
    long long splits = ((p.row3 || p.wide ? wgs_big : p.c8 ? 512 : 1024) + tiles - 1) / tiles;
    const long long max_by_steps = steps / 16 > 0 ? steps / 16 : 1;   // at least 16 K-steps (512 rows) per split
    if (splits > max_by_steps) splits = max_by_steps;
    if (splits > (p.c8 ? 512 : 256)) splits = p.c8 ? 512 : 256;
    if (splits < 1) splits = 1;
#ifdef DBX_LAB
    { static int ov = -1; if (ov < 0) { const char* e = getenv("DBX_WGRAD_SPLITS"); ov = e ? atoi(e) : 0; } if (ov > 0 && !p.alltaps) splits = ov; }
#endif
    if (splits >= 8) splits = (splits + 7) / 8 * 8;           // XCD-aware workgroup mapping wants a multiple of 8
    long long sps = (steps + splits - 1) / splits;
    sps = (sps + 1) / 2 * 2;                                   // whole 64-row K steps for the R=64 kernel
    p.rows_per_split = (int)(sps * 32);
    p.splits = (int)splits;                                    // trailing splits may be empty: they write zero slabs
    // all-taps layers on wide frames: column-strip walk (every x row staged once instead of three times; DBX_WGRAD_VARIANT=11: linear walk)
    // row3: 64-row steps over the valid rows of each image only (DBX_WGRAD_VARIANT=14: linear walk over the whole frame)
    p.spi = p.steps_total = p.steps_per_split = 0;
    if (p.row3 && dz->pad == 1 && dz->w + 2 >= 32 && wgrad_variant() != 14) {
        p.spi = (dz->h * (dz->w + 2) + 63) / 64;
        p.steps_total = dz->n * p.spi;
        p.steps_per_split = (p.steps_total + p.splits - 1) / p.splits;
    }
    p.all9 = 0;
    // wide 3x3 layers, round 3: 128(co) x 64(ci) tiles over all nine taps, one workgroup per CU (DBX_WGRAD_VARIANT=20: the row3 /
    // all-taps kernels above instead)
    if (dtype != DBX_F32 && kh == 3 && kw == 3 && dz->c % 128 == 0 && x->c % 64 == 0 && x->c * dbx_esize(dtype) >= 128 && wgrad_variant() == 0) {
        p.all9 = 1; p.alltaps = p.c8 = p.row3 = p.wide = 0;
        p.bmc = 128; p.bnc = 64; p.co_pad = dz->c; p.ci_pad = x->c; p.tiles_co = dz->c / 128; p.tiles_ci = x->c / 64;
        const int ntile = p.tiles_co * p.tiles_ci;
        const bool compact = dz->pad == 1 && dz->w + 2 >= 32;
        p.spi = compact ? (dz->h * (dz->w + 2) + 63) / 64 : 0;
        const long long steps_tot = compact ? (long long)dz->n * p.spi : (p.Q + 63) / 64;
        long long sp = device_cus() / ntile; if (sp < 1) sp = 1;
        if (sp > steps_tot / 4) sp = steps_tot / 4 > 0 ? steps_tot / 4 : 1;       // at least four 64-row steps per split
        if (sp >= 8) sp = sp / 8 * 8;                                              // XCD-aware workgroup mapping
        const long long sps = (steps_tot + sp - 1) / sp;
        p.splits = (int)((steps_tot + sps - 1) / sps);                              // no empty trailing splits
        if (sp >= 8 && p.splits % 8) p.splits = (int)sp;                            // (keep the multiple of 8: trailing splits may be short / empty)
        p.steps_total = compact ? (int)steps_tot : 0; p.steps_per_split = compact ? (int)sps : 0;
        p.rows_per_split = (int)(sps * 64);
        p.bsplits = p.splits * p.tiles_ci;
    }
    p.wide2 = 0;
    // wide 1x1 layers, round 3: the same pipeline on 256 x 256 tiles (DBX_WGRAD_VARIANT=20: wgrad_wide_kernel)
    if (p.wide && wgrad_variant() == 0) {
        p.wide2 = 1; p.all9 = 1; p.wide = 0;                                          // (all9: shares the one-round split rule and the dump reduce)
        const int ntile = p.tiles_co * p.tiles_ci;
        const bool compact = dz->pad >= 1 && (dz->w + 2 * dz->pad) >= 32;
        p.spi = compact ? (dz->h * (dz->w + 2 * dz->pad) + 31) / 32 : 0;
        const long long steps_tot = compact ? (long long)dz->n * p.spi : (p.Q + 31) / 32;
        long long sp = device_cus() / ntile; if (sp < 1) sp = 1;
        if (sp > steps_tot / 8) sp = steps_tot / 8 > 0 ? steps_tot / 8 : 1;
        if (sp >= 8) sp = sp / 8 * 8;
        const long long sps = (steps_tot + sp - 1) / sp;
        p.splits = (int)((steps_tot + sps - 1) / sps);
        if (sp >= 8 && p.splits % 8) p.splits = (int)sp;
        p.steps_total = compact ? (int)steps_tot : 0; p.steps_per_split = compact ? (int)sps : 0;
        p.rows_per_split = (int)(sps * 32);
        p.bsplits = p.splits * p.tiles_ci;
    }
    p.strip = (p.alltaps && !p.c8 && dz->w + 2 * dz->pad >= 64 && wgrad_variant() != 11) ? 1 : 0;
    p.nstrips = p.units = p.units_per_split = 0;
    if (p.strip) {
        const int wp = dz->w + 2 * dz->pad;
        p.nstrips = (wp + 31) / 32;
        p.units = dz->n * p.nstrips;
        long long want = (1024 + tiles - 1) / tiles;            // ~4 workgroups per CU over all tiles
        if (want > p.units) want = p.units;
        if (want < 1) want = 1;
        p.units_per_split = (int)((p.units + want - 1) / want);
        p.splits = (p.units + p.units_per_split - 1) / p.units_per_split;
    }
    if (!p.all9) p.bsplits = p.splits;
    return p;
}

extern "C" int64_t dbx_conv_wgrad_scratch_bytes(int32_t dtype, const dbx_view* dz, const dbx_view* x, int32_t kh, int32_t kw) {
    const WgradPlan p = wgrad_plan(dtype, dz, x, kh, kw);
    return ((int64_t)p.splits * p.co_pad * p.taps * p.ci_pad + (int64_t)p.bsplits * p.co_pad) * 4 + 256;
}

// The kernel wgrad_t launches for this problem (one selection rule: wgrad_plan), for callers that label profiles.
extern "C" int dbx_conv_wgrad_plan(int32_t dtype, const dbx_view* dz, const dbx_view* x, int32_t kh, int32_t kw, char* name, int32_t name_len,
                                   int32_t* splits) {
    if (!dz || !x || !name || name_len < 32) { dbx_set_error("wgrad plan: null argument / short name buffer"); return DBX_ERR_ARG; }
    if (dtype != DBX_F16 && dtype != DBX_BF16 && dtype != DBX_F32) { dbx_set_error("bad dtype %d", (int)dtype); return DBX_ERR_DTYPE; }
    const WgradPlan p = wgrad_plan(dtype, dz, x, kh, kw);
    const char* tn = dtype == DBX_F32 ? "f32" : (dtype == DBX_F16 ? "f16" : "bf16");
    if (p.wide2) snprintf(name, name_len, "wgrad_wide2_kernel<%s>", tn);
    else if (p.all9) snprintf(name, name_len, "wgrad_all9_kernel<%s>", tn);
    else if (p.c8) snprintf(name, name_len, "wgrad3x3_c8_kernel<%s>", tn);
    else if (p.alltaps && p.strip) snprintf(name, name_len, "wgrad3x3_strip_kernel<%s>", tn);
    else if (p.alltaps) snprintf(name, name_len, "wgrad3x3_kernel<%s>", tn);
    else if (p.row3) snprintf(name, name_len, "wgrad_row3_kernel<%s>", tn);
    else if (p.wide) snprintf(name, name_len, "wgrad_wide_kernel<%s>", tn);
    else snprintf(name, name_len, "wgrad_kernel<%s,%d,%d>", tn, p.bmc, p.bnc);
    if (splits) *splits = p.splits;
    return DBX_OK;
}

// dz as the backward of a max pooling that is not in memory (dbx_conv_wgrad_pool_dz): the pooled gradient and the arg-max nibbles
struct PoolDz { const dbx_view* dy; const unsigned char* idx; int ctot; int write_dz; };
template <typename T>
static int wgrad_t(const dbx_view* dz, const dbx_view* x, int kh, int kw, int cpad, int co, int ci, float* dw, float* db,
                   void* scratch, int accumulate, hipStream_t s, int ci_total, int ci_off, const GenHid* gen = nullptr, const PoolDz* pz = nullptr) {
    DBX_REQUIRE(ci_off >= 0 && ci_off + ci <= ci_total, "wgrad: column slice [%d, %d) outside the %d input channels of dw", ci_off, ci_off + ci, ci_total);
    constexpr int ES = sizeof(T);
    DBX_REQUIRE(dz->n == x->n && dz->h + 2 * dz->pad == x->h + 2 * x->pad && dz->w + 2 * dz->pad == x->w + 2 * x->pad,
                "wgrad: dz frame %dx%d(+%d) and x frame %dx%d(+%d) are not congruent", dz->h, dz->w, dz->pad, x->h, x->w, x->pad);
    DBX_REQUIRE(x->h + 2 * cpad - kh + 1 == dz->h && x->w + 2 * cpad - kw + 1 == dz->w, "wgrad: dz is not the conv output shape");
    DBX_REQUIRE(co <= dz->c && ci <= x->c, "wgrad: real channel counts exceed the views");
    DBX_REQUIRE((pz || ((size_t)dz->ptr % 16) == 0) && ((size_t)x->ptr % 16) == 0 && (dz->ld * ES) % 16 == 0 && (x->ld * ES) % 16 == 0 &&
                    (dz->c_off * ES) % 16 == 0 && (x->c_off * ES) % 16 == 0 && (dz->c * ES) % 16 == 0 && (x->c * ES) % 16 == 0,
                "wgrad: 16-byte alignment");
    const WgradPlan p = wgrad_plan(DType<T>::id, dz, x, kh, kw);
    WgradArgs a;
    a.dz = (const char*)dz->ptr + (size_t)dz->c_off * ES;
    a.x = (const char*)x->ptr + (size_t)x->c_off * ES;
    a.partial = (float*)scratch;
    a.bpartial = db ? (float*)scratch + (size_t)p.splits * p.co_pad * p.taps * p.ci_pad : nullptr;
    a.Q = p.Q; a.dz_ld = dz->ld; a.x_ld = x->ld; a.dz_c = dz->c; a.x_c = x->c;
    a.co_pad = p.co_pad; a.ci_pad = p.ci_pad; a.taps = p.taps; a.kw = kw; a.wp = x->w + 2 * x->pad;
    a.shift0 = x->pad - dz->pad - cpad;
    a.tiles_co = p.tiles_co; a.tiles_ci = p.tiles_ci; a.rows_per_split = p.rows_per_split; a.nsplit = p.splits;
    a.hp = a.nstrips = a.units = a.units_per_split = 0;
    a.spi = p.spi; a.img_rows = (dz->h + 2 * dz->pad) * (dz->w + 2 * dz->pad); a.row0 = (dz->w + 2 * dz->pad) * dz->pad;
    a.steps_total = p.steps_total; a.steps_per_split = p.steps_per_split;
    DBX_REQUIRE(!gen || (p.wide2 && p.spi > 0 && sizeof(T) == 2), "wgrad with a generated hidden gradient: needs the wide 1x1 kernel on padded frames of >= 32 columns");
    DBX_REQUIRE(!pz || (p.alltaps && p.strip && !p.wide2 && !p.all9 && !p.c8 && sizeof(T) == 2), "wgrad with a pooling backward as dz: needs the 3x3 column-strip kernel (dbx_conv_wgrad_pool_dz_ok)");
    a.zout = nullptr;
    a.pdy = nullptr; a.pidx = nullptr; a.p_hp = a.p_wp = a.p_ld = a.p_pad = a.p_h = a.p_w = a.z_h = a.z_w = a.z_pad = a.z_ctot = a.z_coff = 0;
    if (pz) {
        const dbx_view* dy = pz->dy;
        DBX_REQUIRE(dy->n == dz->n && dy->h == dz->h / 2 && dy->w == dz->w / 2 && dz->h % 2 == 0 && dz->w % 2 == 0 && dy->c == dz->c && dy->c_off == dz->c_off &&
                        ((size_t)dy->ptr % 16) == 0 && (dy->ld * ES) % 16 == 0 && (dy->c_off * ES) % 16 == 0 && ((size_t)pz->idx % 4) == 0 && dz->c_off % 8 == 0 &&
                        pz->ctot % 8 == 0 && dz->c_off + dz->c <= pz->ctot,
                    "wgrad pool dz: dy is the pooled map of dz (even extent, same channels / channel offset), 16-byte aligned, idx 4-byte aligned");
        a.pdy = (const char*)dy->ptr + (size_t)dy->c_off * ES; a.pidx = pz->idx;
        a.p_hp = dy->h + 2 * dy->pad; a.p_wp = dy->w + 2 * dy->pad; a.p_ld = dy->ld; a.p_pad = dy->pad; a.p_h = dy->h; a.p_w = dy->w;
        a.z_h = dz->h; a.z_w = dz->w; a.z_pad = dz->pad; a.z_ctot = pz->ctot; a.z_coff = dz->c_off;
        if (pz->write_dz) {
            DBX_REQUIRE(dz->ptr && ((size_t)dz->ptr % 16) == 0, "wgrad pool dz: write_dz needs the dz map's memory");
            a.zout = (char*)dz->ptr + (size_t)dz->c_off * ES;
        }
    }
    if (p.wide2) {
        if constexpr (sizeof(T) == 2) {
            constexpr int smem = 4 * 2 * 32 * 512;
            static DbxDevOnce attr_once, attr_once_g; int attr_dev = 0;
            if (gen) {
                if (attr_once_g.pending(&attr_dev)) {
                    DBX_HIP(hipFuncSetAttribute((const void*)wgrad_wide2_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                    attr_once_g.mark(attr_dev);
                }
                hipLaunchKernelGGL((wgrad_wide2_kernel<T, true>), dim3(p.tiles_co * p.tiles_ci * p.splits), dim3(512), smem, s, a, *gen);
            } else {
                if (attr_once.pending(&attr_dev)) {
                    DBX_HIP(hipFuncSetAttribute((const void*)wgrad_wide2_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                    attr_once.mark(attr_dev);
                }
                hipLaunchKernelGGL((wgrad_wide2_kernel<T, false>), dim3(p.tiles_co * p.tiles_ci * p.splits), dim3(512), smem, s, a, GenHid{});
            }
        }
    } else if (p.all9) {
        if constexpr (sizeof(T) == 2) {
            constexpr int smem = 3 * (64 * 256 + 3 * 72 * 128);
            static DbxDevOnce attr_once; int attr_dev = 0;
            if (attr_once.pending(&attr_dev)) {
                DBX_HIP(hipFuncSetAttribute((const void*)wgrad_all9_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                DBX_HIP(hipFuncSetAttribute((const void*)wgrad_all9s_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                attr_once.mark(attr_dev);
            }
            // round 5: the same tile with the two wave groups one barrier apart (DBX_WGRAD_STAG=0: the lock-step kernel of round 3)
            static int stag = -1;
            if (stag < 0) { const char* e = getenv("DBX_WGRAD_STAG"); stag = e ? atoi(e) : 1; }
            if (stag) hipLaunchKernelGGL((wgrad_all9s_kernel<T>), dim3(p.tiles_co * p.tiles_ci * p.splits), dim3(512), smem, s, a);
            else hipLaunchKernelGGL((wgrad_all9_kernel<T>), dim3(p.tiles_co * p.tiles_ci * p.splits), dim3(512), smem, s, a);
        }
    } else if (p.c8) {
        if constexpr (sizeof(T) == 2) hipLaunchKernelGGL((wgrad3x3_c8_kernel<T>), dim3(p.splits), dim3(256), 0, s, a);
    } else if (p.alltaps) {
        if constexpr (sizeof(T) == 2)
            {
                constexpr int smem = 2 * (32 * 128 + 3 * 36 * 128);
                static DbxDevOnce attr_once; int attr_dev = 0;
                static int pad = 0;                                   // lab builds: DBX_WGRAD_LDSPAD = extra dynamic LDS (occupancy experiments)
                if (attr_once.pending(&attr_dev)) {
#ifdef DBX_LAB
                    const char* e = getenv("DBX_WGRAD_LDSPAD"); pad = e ? atoi(e) : 0;
#endif
                    DBX_HIP(hipFuncSetAttribute((const void*)wgrad3x3_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, smem + pad));
                    attr_once.mark(attr_dev);
                }
                if (p.strip) {
                    constexpr int smem2 = 2 * 32 * 128 + 4 * 36 * 128;
                    a.hp = dz->h + 2 * dz->pad; a.nstrips = p.nstrips; a.units = p.units; a.units_per_split = p.units_per_split;
                    if (pz) hipLaunchKernelGGL((wgrad3x3_strip_kernel<T, true>), dim3(p.tiles_co * p.tiles_ci * p.splits), dim3(256), smem2, s, a);
                    else hipLaunchKernelGGL((wgrad3x3_strip_kernel<T, false>), dim3(p.tiles_co * p.tiles_ci * p.splits), dim3(256), smem2, s, a);
                } else
                hipLaunchKernelGGL((wgrad3x3_kernel<T>), dim3(p.tiles_co * p.tiles_ci * p.splits), dim3(256), smem + pad, s, a);
            }
    } else if (p.row3) {
        if constexpr (sizeof(T) == 2) {
            constexpr int smem = 2 * (64 * 256 + 68 * 256);
            static DbxDevOnce attr_once; int attr_dev = 0;
            if (attr_once.pending(&attr_dev)) {
                DBX_HIP(hipFuncSetAttribute((const void*)wgrad_row3_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                attr_once.mark(attr_dev);
            }
            hipLaunchKernelGGL((wgrad_row3_kernel<T>), dim3(p.tiles_co * p.tiles_ci * 3 * p.splits), dim3(512), smem, s, a);
        }
    } else if (p.wide) {
        if constexpr (sizeof(T) == 2) {
            constexpr int smem = 4 * 64 * 512;
            static DbxDevOnce attr_once; int attr_dev = 0;
            if (attr_once.pending(&attr_dev)) {
                DBX_HIP(hipFuncSetAttribute((const void*)wgrad_wide_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                attr_once.mark(attr_dev);
            }
            hipLaunchKernelGGL((wgrad_wide_kernel<T>), dim3(p.tiles_co * p.tiles_ci * p.taps * p.splits), dim3(512), smem, s, a);
        }
    } else {
        const dim3 grid(p.tiles_co * p.tiles_ci * p.taps * p.splits);
        if (p.bmc == 128) hipLaunchKernelGGL((wgrad_kernel<T, 128, 128>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_kernel<T, 64, 64>), grid, dim3(256), 0, s, a);
    }
    DBX_LAUNCH_CHECK();
    if (p.all9) {
        const long long total = (long long)p.tiles_co * p.tiles_ci * 8 * (p.wide2 ? 32 : 36) * 64 + (co + 3) / 4;
        int blocks = (int)((total + 63) / 64); blocks = blocks > 8192 ? 8192 : blocks;
        if (p.wide2)
            hipLaunchKernelGGL(wgrad_reduce_all9_kernel<true>, dim3(blocks), dim3(256), 0, s, a.partial, a.bpartial, p.splits, p.bsplits, co, ci, p.tiles_co,
                               p.tiles_ci, p.co_pad, dw, db, accumulate, ci_total, ci_off);
        else
            hipLaunchKernelGGL(wgrad_reduce_all9_kernel<false>, dim3(blocks), dim3(256), 0, s, a.partial, a.bpartial, p.splits, p.bsplits, co, ci, p.tiles_co,
                               p.tiles_ci, p.co_pad, dw, db, accumulate, ci_total, ci_off);
        DBX_LAUNCH_CHECK();
        return DBX_OK;
    }
    const long long total = (long long)co * p.taps * (p.ci_pad / 4) + (co + 3) / 4;
    int blocks = (int)((total + 63) / 64); blocks = blocks > 8192 ? 8192 : blocks;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, a.partial, a.bpartial, p.splits, p.bsplits, co, ci, p.taps, p.co_pad,
                       p.ci_pad, dw, db, accumulate, ci_total, ci_off);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

// the split reduction as a host helper for other translation units (conv_igemm.hip: fused dgrad + conv1_1 weight gradient)
int dbx_internal_wgrad_reduce(const float* partial, const float* bpartial, int splits, int co, int ci, int taps, int co_pad, int ci_pad,
                              float* dw, float* db, int accumulate, hipStream_t s) {
    const long long total = (long long)co * taps * (ci_pad / 4) + (co + 3) / 4;
    int blocks = (int)((total + 63) / 64); blocks = blocks > 8192 ? 8192 : blocks;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, partial, bpartial, splits, splits, co, ci, taps, co_pad, ci_pad, dw, db, accumulate, ci, 0);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

extern "C" int dbx_conv_wgrad(int32_t dtype, const dbx_view* dz, const dbx_view* x, int32_t kh, int32_t kw, int32_t cpad,
                              int32_t co, int32_t ci, float* dw_oihw, float* db, void* scratch, int32_t accumulate, void* stream) {
    if (!dz || !x || !dw_oihw || !scratch) { dbx_set_error("wgrad: null argument"); return DBX_ERR_ARG; }
    DBX_DISPATCH_DTYPE(dtype, wgrad_t, dz, x, kh, kw, cpad, co, ci, dw_oihw, db, scratch, accumulate, (hipStream_t)stream, ci, 0);
}
extern "C" int dbx_conv_wgrad_slice(int32_t dtype, const dbx_view* dz, const dbx_view* x, int32_t kh, int32_t kw, int32_t cpad,
                                    int32_t co, int32_t ci, float* dw_oihw, int32_t dw_ci_total, int32_t dw_ci_off, float* db, void* scratch,
                                    int32_t accumulate, void* stream) {
    if (!dz || !x || !dw_oihw || !scratch) { dbx_set_error("wgrad: null argument"); return DBX_ERR_ARG; }
    DBX_DISPATCH_DTYPE(dtype, wgrad_t, dz, x, kh, kw, cpad, co, ci, dw_oihw, db, scratch, accumulate, (hipStream_t)stream, dw_ci_total, dw_ci_off);
}

// ---------------------------------------------------------------------------------------------- dz = a max pooling's backward, not in memory
// dbx_conv_wgrad_pool_dz (round 6): dbx_maxpool2x2_bwd_idx(idx, dy, dz, relu_gate) + dbx_conv_wgrad(dz, x, 3x3) without dz in memory
// (wgrad3x3_strip_kernel<T, true>).  dz: the SHAPE of the un-pooled gradient (frame congruent with x; ptr is not dereferenced).
template <typename T>
static int wgrad_pool_dz_t(const dbx_view* dy, const void* idx, int idx_ctot, const dbx_view* dz, const dbx_view* x, int kh, int kw, int cpad, int co, int ci,
                           float* dw, float* db, void* scratch, int accumulate, int write_dz, hipStream_t s) {
    if constexpr (sizeof(T) != 2) { dbx_set_error("wgrad pool dz: 16-bit compute types only"); return DBX_ERR_DTYPE; }
    else {
        PoolDz pz{dy, (const unsigned char*)idx, idx_ctot, write_dz};
        return wgrad_t<T>(dz, x, kh, kw, cpad, co, ci, dw, db, scratch, accumulate, s, ci, 0, nullptr, &pz);
    }
}
extern "C" int dbx_conv_wgrad_pool_dz_ok(int32_t dtype, const dbx_view* dz, const dbx_view* x, int32_t kh, int32_t kw) {
    if (!dz || !x || (dtype != DBX_F16 && dtype != DBX_BF16) || dz->h % 2 || dz->w % 2) return 0;
    const WgradPlan p = wgrad_plan(dtype, dz, x, kh, kw);
    return (p.alltaps && p.strip && !p.wide2 && !p.all9 && !p.c8) ? 1 : 0;
}
extern "C" int dbx_conv_wgrad_pool_dz(int32_t dtype, const dbx_view* dy, const void* idx, int32_t idx_channels, const dbx_view* dz, const dbx_view* x,
                                      int32_t kh, int32_t kw, int32_t cpad, int32_t co, int32_t ci, float* dw_oihw, float* db, void* scratch,
                                      int32_t accumulate, int32_t write_dz, void* stream) {
    if (!dy || !idx || !dz || !x || !dw_oihw || !scratch) { dbx_set_error("wgrad pool dz: null argument"); return DBX_ERR_ARG; }
    DBX_DISPATCH_DTYPE(dtype, wgrad_pool_dz_t, dy, idx, idx_channels, dz, x, kh, kw, cpad, co, ci, dw_oihw, db, scratch, accumulate, write_dz, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- heads: dW1 with the hidden gradient generated
// dbx_heads1_wgrad_gen: dbx_conv_wgrad_slice(d_hid, x, 1x1) for the heads' first convs WITHOUT d_hid in memory: d_hid = keep * (d_out W2)
// is generated tile by tile inside wgrad_wide2_kernel<T, true> (common.hpp: GenHid).  d_out: the compact [N, H, W] map of the heads'
// output gradients (pad 0, one slot of slot_c >= 8 channels per head, channels >= k[i] zero); x: the 1x1 convs' input on its padded
// frame (pad >= 1, >= 32 columns); w2[i]: fp32 [k[i]][512].  use_hash / drop_seed: the forward's hash dropout (0: no dropout).
// Scratch: dbx_conv_wgrad_scratch_bytes for a dz view of 512 * nh channels congruent with x.
int dbx_internal_heads1_wgrad_gen_f32(const dbx_view* d_out, const dbx_view* x, const float* const* w2, const int32_t* k, int nh, int use_hash,
                                      unsigned seed, int ci, float* dw, int ci_total, int ci_off, float* db, hipStream_t s);
template <typename T>
static int heads1_wgrad_gen_t(const dbx_view* d_out, const dbx_view* x, const float* const* w2, const int32_t* k, int nh, int use_hash, unsigned seed,
                              int ci, float* dw, int ci_total, int ci_off, float* db, void* scratch, hipStream_t s) {
    // fp32: the parity suite's reference instantiation (heads_ref_f32.hip; scratch unused)
    if constexpr (sizeof(T) != 2) return dbx_internal_heads1_wgrad_gen_f32(d_out, x, w2, k, nh, use_hash, seed, ci, dw, ci_total, ci_off, db, s);
    else {
        DBX_REQUIRE(nh >= 1 && nh <= 4 && d_out->pad == 0 && d_out->n == x->n && d_out->h == x->h && d_out->w == x->w && x->pad >= 1,
                    "heads1_wgrad_gen: d_out is the compact map of x's pixels, x a padded frame");
        DBX_REQUIRE(d_out->c % nh == 0 && d_out->c / nh >= 8 && ((size_t)d_out->ptr % 16) == 0 && (d_out->ld * 2) % 16 == 0 && (d_out->c_off * 2) % 16 == 0 &&
                    ((d_out->c / nh) * 2) % 16 == 0, "heads1_wgrad_gen: d_out slots of >= 8 channels, 16-byte aligned");
        DBX_REQUIRE((int64_t)d_out->n * d_out->h * d_out->w * d_out->ld * 2 < ((int64_t)1 << 31), "heads1_wgrad_gen: d_out beyond 2 GiB");
        GenHid g;
        g.dout = (const char*)d_out->ptr + (size_t)d_out->c_off * 2;
        for (int i = 0; i < 4; ++i) {
            g.w2[i] = i < nh ? w2[i] : nullptr; g.k[i] = i < nh ? k[i] : 0;
            if (i < nh) DBX_REQUIRE(w2[i] && k[i] >= 1 && k[i] <= 8, "heads1_wgrad_gen: k in 1..8");
        }
        g.ld = d_out->ld; g.slot = d_out->c / nh; g.nh = nh; g.H = x->h; g.W = x->w; g.pad = x->pad; g.seed = seed; g.use_hash = use_hash ? 1 : 0;
        dbx_view dz = *x;                                   // the virtual hidden gradient: congruent with x, 512 * nh channels, never dereferenced
        dz.c = dz.ld = 512 * nh; dz.c_off = 0;
        return wgrad_t<T>(&dz, x, 1, 1, 0, 512 * nh, ci, dw, db, scratch, 0, s, ci_total, ci_off, &g);
    }
}
// 1 when dbx_heads1_wgrad_gen can run for x (the wide 1x1 kernel on a padded frame of >= 32 columns)
extern "C" int dbx_heads1_wgrad_gen_ok(int32_t dtype, const dbx_view* x, int32_t nh) {
    if (dtype == DBX_F32) return (x && nh >= 1 && nh <= 4) ? 1 : 0;        // (the fp32 reference instantiation takes any frame)
    if (!x || (dtype != DBX_F16 && dtype != DBX_BF16) || nh < 1 || nh > 4 || x->pad < 1) return 0;
    dbx_view dz = *x;
    dz.c = dz.ld = 512 * nh; dz.c_off = 0;
    const WgradPlan p = wgrad_plan(dtype, &dz, x, 1, 1);
    return (p.wide2 && p.spi > 0) ? 1 : 0;
}
extern "C" int dbx_heads1_wgrad_gen(int32_t dtype, const dbx_view* d_out, const dbx_view* x, const float* const* w2, const int32_t* k, int32_t nh,
                                    int32_t use_hash, uint32_t drop_seed, int32_t ci, float* dw_oihw, int32_t dw_ci_total, int32_t dw_ci_off,
                                    float* db, void* scratch, void* stream) {
    if (!d_out || !x || !w2 || !k || !dw_oihw || !scratch) { dbx_set_error("heads1_wgrad_gen: null argument"); return DBX_ERR_ARG; }
    DBX_DISPATCH_DTYPE(dtype, heads1_wgrad_gen_t, d_out, x, w2, k, nh, use_hash, (unsigned)drop_seed, ci, dw_oihw, dw_ci_total, dw_ci_off, db, scratch,
                       (hipStream_t)stream);
}
