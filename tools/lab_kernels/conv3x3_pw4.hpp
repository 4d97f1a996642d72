// LAB ONLY (tools/band_lab.hip, -DDBX_LAB) -- measured and superseded by conv3x3_ws.hpp (each weight fragment is fetched by
// two waves here: 37 KB of vector-memory traffic per 1024 MFMA-cycles against 22 KB, 482 k clocks against 390 k).
// v6: 3x3 / pad-1 convolution, four fat waves (one per SIMD, the whole 512-register file each), weights streamed straight
// into the MFMA operand registers.  Included by conv_igemm.hip (needs ConvArgs, gate_packed16, Mma32 from conv3x3_pipe.hpp).
//
// Why: the 8-wave kernels are LDS-bound, not MFMA-bound.  On the 256 x 256 tile with 128 x 64 per wave every tap-step moves
// 96 KB of fragment reads + 24 KB of LDS-DMA writes through an LDS that delivers ~128 B/clk; taking the reads out of the loop
// (lab ablation, tools/band_lab.hip) speeds the kernel up by 25 %, the DMA by another 13 %.  Here
//   * a wave owns 128 pixels x 128 couts (16 accumulator tiles of 32x32 = 256 registers): 1.5x fewer fragment bytes per FLOP;
//   * the weights never touch the LDS: they are packed in MFMA-fragment order (dbx_pack_weight modes 2/3: one 1-KiB block =
//     the A operand of one v_mfma_f32_32x32x16 for all 64 lanes), so a wave's four fragments of a K=16 step are FOUR fully
//     coalesced global_load_dwordx4 from one 4-KiB run, issued two steps ahead of their MFMAs (L2-resident: every
//     workgroup of an XCD streams the same 2.4 MB at the same time);
//   * only the pixel band goes through the LDS (33 KB per (ky, 64-channel chunk) period, double buffered, LDS-DMA), read by
//     four ds_read_b128 per 16 MFMAs: 40 B/clk instead of 117;
//   * ONE barrier per period (12 steps = 192 MFMAs per wave) instead of one per tap.
// Inline-asm loads are invisible to the compiler's s_waitcnt bookkeeping: every VMEM operation in the loop is counted by
// hand (pw4::allowed) and each wait is followed by sched_barrier(0) so that no MFMA is hoisted above it.
#pragma once

namespace pw4 {
constexpr int BM = 256, BN = 256;
constexpr int AP = 33;                          // A-band pieces (8 rows x 128 B) per period
constexpr int ABUF = AP * 1024;
constexpr int D = 2;                            // weight loads run D steps ahead
constexpr int NSTEP = 12;                       // K=16 steps per period: kx * 4 + 16-channel chunk
constexpr int WSTEP = 2 * 4 * 1024;             // packed weight bytes per step of one 256-cout tile: [wn][ni][lane][16 B]
// LDS-DMA loads a wave issues at step i of a period (behind the step's four weight loads): the NEXT period's band, 9 slots
constexpr int g(int i, bool last) { return (i >= 0 && i <= 8 && !last) ? 1 : 0; }
// VMEM operations issued after the weight loads of step j (which go out at step j - D): may still be in flight at its wait
constexpr int allowed(int j, bool last) {
    int n = g(j - D, last);
    for (int i = j - D + 1; i <= j - 1; ++i) n += 4 + g(i, last);
    return n;
}
}  // namespace pw4

template <typename T, int ABL = 0>
__global__ __launch_bounds__(256) void conv3x3_pw4_kernel(const ConvArgs a) {
    using namespace pw4;
    constexpr int ES = sizeof(T);
    static_assert(ES == 2, "16-bit types");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

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
    const int KC = cin_bytes / 128;                                    // 64-channel chunks
    const int P = 3 * KC;                                              // periods

    // ---- weight stream: scalar base walks the packed image step by step, one lane offset for the whole kernel
    const char* wptr = a.w + (size_t)tile_n * P * NSTEP * WSTEP;        // uniform
    const unsigned wvoff = wn * 4096 + lane * 16;
    u32x4 wr[D + 1][4];
    auto wload = [&](int set) {
        if (!(ABL & 8)) {
            asm volatile("global_load_dwordx4 %0, %4, %5\n\t"
                         "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                         "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
                         "global_load_dwordx4 %3, %4, %5 offset:3072"
                         : "=&v"(wr[set][0]), "=&v"(wr[set][1]), "=&v"(wr[set][2]), "=&v"(wr[set][3])
                         : "v"(wvoff), "s"(wptr)
                         : "memory");
        }
        wptr += WSTEP;
    };

    // ---- band pieces by LDS-DMA: piece q = wave + 4 i (i = 0..8); only wave 0 has a ninth piece (q = 32), the others repeat
    // their eighth so that every wave issues the same number of loads (uniform counted waits)
    const char* asrc;                                                   // piece wave + 4 i: + 32 i pixels
    {
        const int lr8 = lane >> 3, lc8 = lane & 7;
        const int sw = (4 * wave + (lr8 >> 1)) & 7;                     // (row >> 1) & 7 of row = 8 (wave + 4 i) + lr8
        asrc = a.x + ((q0 + 8 * wave + lr8) - a.x_wp - 1) * (long long)pix_bytes + ((lc8 ^ sw) << 4);
    }
    auto issue_a = [&](int ab, int buf, int i) {
        const int ii = (i == 8 && wave != 0) ? 7 : i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc + ab + (long long)ii * 32 * pix_bytes),
                                         (__attribute__((address_space(3))) void*)(smem + buf * ABUF + (wave + 4 * ii) * 1024), 16, 0, 0);
    };
    int p_ky = 0, p_kc = 0, ab_nxt = 0;
    auto advance = [&]() {
        if (++p_kc == KC) { p_kc = 0; ++p_ky; }
        ab_nxt = p_ky * a.x_wp * pix_bytes + p_kc * 128;
    };

    // ---- fragment read addresses: 128-byte rows, chunk 2 c + h of row l31 + kx (+ 32 mi), swizzle (row >> 1) & 7
    int xlane[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xlane[kx] = (wm * 128 + l31 + kx) * 128 + ((h ^ (((l31 + kx) >> 1) & 7)) << 4);

    f32x16 acc[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
    u32x4 xf[2][4];

    // ---- prologue
#pragma unroll
    for (int i = 0; i < 9; ++i) issue_a(0, 0, i);
#pragma unroll
    for (int d = 0; d < D; ++d) wload(d);
    advance();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (D - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    {
        const char* xp = smem + xlane[0];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) xf[0][mi] = *(const u32x4*)(xp + mi * 4096);
    }

    int buf = 0;
    auto period = [&](auto LAST_) {
        constexpr bool LAST = decltype(LAST_)::value != 0;
        auto step = [&](auto J_) {
            constexpr int j = decltype(J_)::value;
            constexpr int ws = j % (D + 1), wnx = (j + D) % (D + 1), xs = j & 1;
            // weights of this step have landed once only the younger loads are outstanding
            if (!(ABL & 4)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allowed(j, LAST)) : "memory");
            if (j == NSTEP - 1 && !LAST && !(ABL & 4)) {
                // period seam: the band of the next period landed (its loads are older than the weights just waited for);
                // every wave has finished reading the band of this one (reads of step 11 were issued a step ago)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            const bool rd = !(LAST && j == NSTEP - 1) && !(ABL & 2);
            const int rbuf = j == NSTEP - 1 ? buf ^ 1 : buf;
            constexpr int jn = (j + 1) % NSTEP;
            const char* xp = smem + rbuf * ABUF + (xlane[jn >> 2] ^ ((jn & 3) << 5));
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                Mma32<T>::run(wr[ws][k >> 2], xf[xs][k & 3], acc[k >> 2][k & 3]);
                if (rd && k < 4) xf[xs ^ 1][k] = *(const u32x4*)(xp + k * 4096);
                if (k == 4) wload(wnx);
                if (k == 8 && g(j, LAST) && !(ABL & 1)) issue_a(ab_nxt, buf ^ 1, j);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        step(pipe::IC<0>{}); step(pipe::IC<1>{}); step(pipe::IC<2>{}); step(pipe::IC<3>{}); step(pipe::IC<4>{}); step(pipe::IC<5>{});
        step(pipe::IC<6>{}); step(pipe::IC<7>{}); step(pipe::IC<8>{}); step(pipe::IC<9>{}); step(pipe::IC<10>{}); step(pipe::IC<11>{});
    };
    if (!(ABL & 16)) {
    for (int m = 0; m < P - 1; ++m) {
        period(pipe::IC<0>{});
        advance();
        buf ^= 1;
    }
    period(pipe::IC<1>{});
    }
    // the weight loads of the last D steps ran past the end of the stream: nobody consumes them, so the compiler would hand
    // their destination registers to the epilogue while the loads are still in flight.  Drain, and keep the registers
    // "used" up to here.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int d = 0; d <= D; ++d)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(wr[d][i]));
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue over frame pixels (see conv3x3_pipe.hpp): acc[ni][mi][r]: pixel = q0 + 128 wm + 32 mi + l31,
    // cout = n0 + 128 wn + 32 ni + 8 (r >> 2) + 4 h + (r & 3)
    const int epi = a.epi;
    const int fpix = a.x_hp * a.x_wp, nimg = a.M / a.HoWo;
    const int cw = n0 + wn * 128;
    int n, fy, fx;
    {
        const long long q = q0 + wm * 128 + l31;
        n = (int)(q / fpix);
        const int rem = (int)(q - (long long)n * fpix);
        fy = rem / a.x_wp; fx = rem - fy * a.x_wp;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const bool ok = n < nimg && fy >= 1 && fy <= a.x_hp - 2 && fx >= 1 && fx <= a.x_wp - 2;
        const int oy = fy - 1, ox = fx - 1;
        T* ypix = (T*)a.y + (size_t)((n * a.y_hp + oy + a.y_pad) * a.y_wp + (ox + a.y_pad)) * (size_t)a.y_ld + cw;
        const T* gpix = (const T*)a.gate + (size_t)((n * a.g_hp + oy + a.g_pad) * a.g_wp + (ox + a.g_pad)) * (size_t)a.g_ld + cw;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                u32x2 pk[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * jp + jj;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = acc[ni][mi][4 * j + i];
                    if (epi & DBX_EPI_BIAS) {
                        const f32x4 b = *(const f32x4*)(a.bias + cw + ni * 32 + 8 * j + 4 * h);
                        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                    }
                    if (epi & DBX_EPI_RELU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                    }
                    if ((epi & DBX_EPI_ACCUM) && ok) {
                        const T* o = ypix + ni * 32 + 8 * j + 4 * h;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += to_f32(o[i]);
                    }
                    T p[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
                    pk[jj] = *(const u32x2*)p;
                }
                const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                u32x4 o = (u32x4){r0[0], r1[0], r0[1], r1[1]};
                const int coff = ni * 32 + 16 * jp + 8 * h;
                if (ok) {
                    if (epi & DBX_EPI_GATE) o = gate_packed16(o, *(const u32x4*)(gpix + coff));
                    if (!(ABL & 32) || o.x == 0x12345u) *(u32x4*)(ypix + coff) = o;
                }
            }
        }
        if (a.x_wp >= 32) {
            fx += 32;
            if (fx >= a.x_wp) { fx -= a.x_wp; if (++fy == a.x_hp) { fy = 0; ++n; } }
        } else {
            const long long q = q0 + wm * 128 + (mi + 1) * 32 + l31;
            n = (int)(q / fpix);
            const int rem = (int)(q - (long long)n * fpix);
            fy = rem / a.x_wp; fx = rem - fy * a.x_wp;
        }
    }
}

// fragment-order weight image of a 3x3 layer for conv3x3_pw4_kernel: element (cout, tap = 3 ky + kx, ci) lives at
//   [cout / 256][period = ky * KC + ci / 64][step = kx * 4 + (ci % 64) / 16][(cout % 256) / 128][(cout % 128) / 32]
//   [lane = 32 * ((ci % 16) / 8) + cout % 32][ci % 8]
__host__ __device__ inline size_t pw4_weight_index(int co, int tap, int ci, int cin_pad) {
    const int KC = cin_pad / 64, ky = tap / 3, kx = tap - 3 * ky;
    const int period = ky * KC + ci / 64, step = kx * 4 + (ci % 64) / 16;
    const int lane = 32 * ((ci % 16) / 8) + co % 32;
    size_t blk = (((size_t)(co / 256) * (3 * KC) + period) * 12 + step) * 2 + (co % 256) / 128;
    blk = blk * 4 + (co % 128) / 32;
    return (blk * 64 + lane) * 8 + ci % 8;
}
constexpr size_t PW4_SLACK_BYTES = (size_t)pw4::D * pw4::WSTEP;        // the last D steps' loads run past the image

template <typename T, int ABL = 0>
static int launch_conv_pw4(const ConvArgs& a, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        constexpr int smem = 2 * pw4::ABUF;
        static bool attr_set = false;
        if (!attr_set) {
            DBX_HIP(hipFuncSetAttribute((const void*)conv3x3_pw4_kernel<T, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr_set = true;
        }
        hipLaunchKernelGGL((conv3x3_pw4_kernel<T, ABL>), dim3(a.nblocks), dim3(256), smem, s, a);
        DBX_LAUNCH_CHECK();
    }
    return DBX_OK;
}
