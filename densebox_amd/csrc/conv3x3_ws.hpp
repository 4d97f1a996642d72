// v7: 3x3 / pad-1 convolution with register-streamed weights ("ws"): four fat waves (one per SIMD, the whole 512-register
// file each), persistent workgroups.  Included by conv_igemm.hip (needs ConvArgs, gate_packed16, Mma32 / pipe::IC).
// Forward and -- with flipped/transposed weights -- dgrad of every wide backbone layer (16-bit types).
//
// Measured background (tools/band_lab.hip on MI355X, conv4_2 at batch 64, uniform random operands): the 8-wave LDS kernels
// are bound by what moves through the LDS (~128 B/clk) and the per-CU vector-memory front end (~64 B/clk), and by power
// (1.8-2.0 GHz under load):
//   8 waves, A band + W tiles via LDS (v3 / v5): 120 KB LDS + 24 KB VMEM per 1024 MFMA-cycles   472 k / 440 k clocks
//   4 waves 128x128, W straight to registers (v6): 37 KB LDS + 37 KB VMEM (W fetched by two waves)  482 k
//   4 waves 256x64,  W straight to registers (this): 70 KB LDS + 22 KB VMEM                          390 k (MFMA alone: 383 k)
//
// * Weights are packed in MFMA-fragment order (dbx_pack_weight modes 2/3: one 1-KiB block = the A operand of one
//   v_mfma_f32_32x32x16 for all 64 lanes), so the two fragments of a K=16 step are two fully coalesced
//   global_load_dwordx4 from one 2-KiB run, issued D steps ahead of their MFMAs.  They never touch the LDS.
// * Only the pixel band goes through the LDS: BM + 2 rows x 128 B (64 channels) per (ky, 64-channel chunk) period, double
//   buffered by LDS-DMA; the three kx taps read it at row shifts 0/1/2.  One barrier per period (192 MFMAs per wave).
// * Pixels: the linearised frame WITHOUT its top/bottom halo rows (q' = (n H + fy - 1) Wp + fx).  A tile is NF = 7 or 8
//   fragments of 32 consecutive q' per wave row; a tile that runs into the next image skips the two halo rows between them
//   on the LDS-DMA source side (rows past the switch point come from 2 Wp further on), so the fragment rows stay
//   consecutive in the LDS.  Only the left/right halo columns are computed and dropped (conv4: 6 % instead of 12 %), and
//   7.5-fragment average tiles let 2 (conv4) / 4 (conv3) tiles per CU divide the work evenly: 15 instead of 16 fragment
//   times per CU at 30x30, 29-30 instead of 32 at 60x60.
// * Persistent workgroups (one per CU) walk their tiles; the first band and weight fragments of the NEXT tile are issued
//   during the last period of the current one, so the cold-start latency hides under MFMAs and the epilogue.
// * Inline-asm loads are invisible to the compiler's s_waitcnt bookkeeping: every VMEM operation in the loop is counted by
//   hand (ws::allowed), each wait is followed by sched_barrier(0), and nothing asynchronous is in flight across the
//   compiler-scheduled epilogue (vmcnt(0) on both sides).
#pragma once

namespace ws {
constexpr int D = 2;                            // weight loads run D steps ahead
constexpr int NSTEP = 12;                       // K=16 steps per period: kx * 4 + 16-channel chunk
constexpr int WL = 2;                           // weight loads per wave and step (64 couts = two 32-row fragments)
// band pieces (8 rows x 128 B) per period for WM wave rows of 256 pixels, and per-wave LDS-DMA slots (4 waves)
constexpr int pieces(int WM) { return (256 * WM + 2 + 7) / 8; }
constexpr int slots(int WM) { return (pieces(WM) + 3) / 4; }
// LDS-DMA loads a wave issues at step i of a period (behind the step's weight loads): the next period's band
// -- only in steps 0 .. NSTEP-D-2: the seam wait at step NSTEP-1 (for the weights issued at step NSTEP-1-D) must cover them all
constexpr int g(int i, int WM) {
    constexpr int NA = NSTEP - D - 1;                                   // 9 issue steps
    if (i < 0 || i >= NA) return 0;
    const int s = slots(WM), lo = s / NA, rem = s % NA;                // WM 1: 9 = one per step; WM 2: 17 = 2 in steps 0..7, 1 in step 8
    return lo + (i < rem ? 1 : 0);
}
constexpr int gsum(int i, int WM) { int n = 0; for (int k = 0; k < i; ++k) n += g(k, WM); return n; }   // slots before step i
// VMEM operations issued after the weight loads of step j (which go out at step j - D): may still be in flight at its wait.
// Steps before 0 belong to the previous period (same schedule) -- or to the tile start, where everything was drained.
// The last period of a tile issues no weight loads in its last D steps (the next tile starts its own stream).
constexpr int allowed(int j, int WM, bool first, bool last) {
    int n = 0;
    for (int i = j - D; i <= j - 1; ++i) {
        const bool prev = i < 0;
        if (prev && first) continue;
        const int ii = prev ? i + NSTEP : i;
        n += g(ii, WM) + ((i > j - D && !(last && !prev && i >= NSTEP - D)) ? WL : 0);
    }
    return n;
}
}  // namespace ws

struct WsArgs {
    int mt;              // M tiles
    int base, extra;     // tile t covers base + (t < extra) units of WM fragments (32 q' each), starting at unit t base + min(t, extra)
    int items;           // mt * ntile_n work items
    int hwp;             // H * Wp: q' per image
    int qtot;            // N * H * Wp
};

template <typename T, int WM>
__global__ __launch_bounds__(256) void conv3x3_ws_kernel(const ConvArgs a, const WsArgs t) {
    using namespace ws;
    constexpr int ES = sizeof(T);
    static_assert(ES == 2, "16-bit types");
    constexpr int WN = 4 / WM;                                          // waves along the couts
    constexpr int BN = 64 * WN;
    constexpr int AP = pieces(WM), SL = slots(WM), ABUF = AP * 1024;
    constexpr int WSTEP = BN * 32;                                      // packed weight bytes per step of one BN-cout tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;
    const int pix_bytes = a.x_ld * ES;
    const int cin_bytes = a.cpt * 16;
    const int KC = cin_bytes / 128;                                    // 64-channel chunks
    const int P = 3 * KC;                                              // periods
    const int epi = a.epi;
    const int wp = a.x_wp;

    // ---- persistent schedule: workgroup g runs on XCD g % 8; item = round * G + (xcd-contiguous index), so the 32 workgroups of
    // an XCD work on 32 consecutive items (both cout tiles of 16 neighbouring pixel tiles) at any time
    const int G = gridDim.x;
    int item = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;

    // ---- per-tile state
    struct Tile { int q0, nf, n0, sw; const char* wbase; long long src0; };
    auto tile_of = [&](int it) {
        Tile r;
        const int tm = it / a.ntile_n, tn = it - tm * a.ntile_n;
        const int u0 = tm * t.base + (tm < t.extra ? tm : t.extra);
        r.nf = t.base + (tm < t.extra ? 1 : 0);
        r.q0 = u0 * (32 * WM);
        r.n0 = tn * BN;
        const int img = r.q0 / t.hwp;
        // first band row that belongs to the next image: pixel index of its first q' in this tile, + 1 (see header)
        r.sw = (img + 1) * t.hwp - r.q0 + 1;
        r.wbase = a.w + (size_t)tn * P * NSTEP * WSTEP;
        // frame position of band row 0 for ky = 0: pos(q0) - 1 - Wp, pos(q') = q' + (2 img + 1) Wp
        r.src0 = (long long)r.q0 + (long long)(2 * img + 1) * wp - 1 - wp;
        return r;
    };

    // ---- weight stream: scalar base walks the packed image step by step, one lane offset for the whole kernel
    const unsigned wvoff = wn * 2048 + lane * 16;
    const char* wptr;
    u32x4 wr[D + 1][2];
    auto wload = [&](int set) {
        // s_nop 4: the base may have just been restored from a spill by v_readlane (VALU-written SGPR -> VMEM address needs five
        // wait states; the compiler pads its own instructions, not the inside of an asm statement)
        asm volatile("s_nop 4\n\t"
                     "global_load_dwordx4 %0, %2, %3\n\t"
                     "global_load_dwordx4 %1, %2, %3 offset:1024"
                     : "=&v"(wr[set][0]), "=&v"(wr[set][1])
                     : "v"(wvoff), "s"(wptr)
                     : "memory");
        wptr += WSTEP;
    };

    // ---- band pieces by LDS-DMA: slot i of this wave is piece wave + 4 i; slots past the last piece repeat the wave's last real
    // piece (same bytes to the same place) so that every wave issues the same number of loads (uniform counted waits).
    // The chunk swizzle goes on the source address: chunk ^ ((row >> 1) & 7), row = 8 (wave + 4 i) + lr8.
    // Address = uniform 64-bit base of the piece (scalar arithmetic) + one 32-bit lane offset; the halo-row skip is a per-lane
    // select.  (Per-slot 64-bit lane pointers would be loop invariants the compiler keeps -- and spills.)
    const int lr8 = lane >> 3, lc8 = lane & 7;
    const unsigned a_lane = lr8 * pix_bytes + ((lc8 ^ ((4 * wave + (lr8 >> 1)) & 7)) << 4);
    const unsigned skip_bytes = 2 * wp * pix_bytes;
    auto issue_a = [&](const Tile& tl, int ab, int buf, int i) {
        int ii = i;
        if (wave + 4 * i >= AP) ii = i - 1;                             // uniform per wave
        const int row0 = 8 * (wave + 4 * ii);                           // uniform
        const char* sbase = a.x + (tl.src0 + row0) * (long long)pix_bytes + ab;
        const unsigned voff = a_lane + ((row0 + lr8 >= tl.sw) ? skip_bytes : 0u);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + voff),
                                         (__attribute__((address_space(3))) void*)(smem + buf * ABUF + (wave + 4 * ii) * 1024), 16, 0, 0);
    };

    // ---- fragment read addresses: 128-byte rows, chunk 2 c + h of row l31 + kx (+ 32 mi), swizzle (row >> 1) & 7
    int xlane[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xlane[kx] = (l31 + kx) * 128 + ((h ^ (((l31 + kx) >> 1) & 7)) << 4);   // + wave row: 32 NF wm rows

    u32x4 xf[2][8];
    int buf = 0;
    if (item >= t.items) return;
    Tile cur = tile_of(item);

    // ---- prologue of the first tile: its first band (the later tiles' first bands are issued by their predecessors)
#pragma unroll
    for (int i = 0; i < SL; ++i) issue_a(cur, 0, 0, i);

    for (;;) {
        const int nxt_item = item + G;
        const Tile nxt = tile_of(nxt_item < t.items ? nxt_item : item);     // last tile: a harmless re-fetch of its own start
        // ---- tile start: band 0 has been issued (prologue or the previous tile's last period: LDS-DMA involves no registers, so it
        // may fly across the compiler-scheduled epilogue).  The first D weight steps are fetched HERE, behind the epilogue's
        // stores, and waited for at once: an asm load whose destination the compiler believes to be ready must not be in
        // flight across code that may spill or move it.
        wptr = cur.wbase;
#pragma unroll
        for (int d = 0; d < D; ++d) wload(d);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);

        // One tile, specialised on its fragment count.  The accumulators live and die inside: the two instantiations assign them
        // to different registers, and nothing but scalars and the prefetched operands crosses the merge behind the branch.
        auto body = [&](auto NF_) {
            constexpr int NF = decltype(NF_)::value;
            constexpr int NM = 2 * NF;                                  // MFMAs per step
            f32x16 acc[2][NF];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < NF; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
            const int xrow = wm * NF * 32 * 128;                        // this wave row's first band row (bytes)
            {
                const char* xp = smem + buf * ABUF + xrow + xlane[0];
#pragma unroll
                for (int mi = 0; mi < NF; ++mi) xf[0][mi] = *(const u32x4*)(xp + mi * 4096);
            }
            int p_ky = 0, p_kc = 0;
            // FIRST: nothing was in flight at the tile start; LAST: the band / weights issued are the next tile's
            auto period = [&](auto FIRST_, auto LAST_) {
                constexpr bool FIRST = decltype(FIRST_)::value != 0, LAST = decltype(LAST_)::value != 0;
                int ab_nxt = 0;
                if (!LAST) {
                    if (++p_kc == KC) { p_kc = 0; ++p_ky; }
                    ab_nxt = p_ky * wp * pix_bytes + p_kc * 128;
                }
                const Tile& atile = LAST ? nxt : cur;
                auto step = [&](auto J_) {
                    constexpr int j = decltype(J_)::value;
                    constexpr int wsx = j % (D + 1), wnx = (j + D) % (D + 1), xs = j & 1;
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allowed(j, WM, FIRST, LAST)) : "memory");
                    if (j == NSTEP - 1 && !LAST) {
                        // period seam: the next band landed (its loads are older than the weights just waited for) and every
                        // wave is done with this one (its last reads were issued a step ago)
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    constexpr bool rd = !(LAST && j == NSTEP - 1);
                    const int rbuf = j == NSTEP - 1 ? buf ^ 1 : buf;
                    constexpr int jn = (j + 1) % NSTEP;
                    int xb = xlane[jn >> 2];
                    asm volatile("" : "+v"(xb));                        // recompute per step: twelve hoisted address registers spill
                    const char* xp = smem + rbuf * ABUF + xrow + (xb ^ ((jn & 3) << 5));
                    constexpr int GA = g(j, WM), G0 = gsum(j, WM);
#pragma unroll
                    for (int k = 0; k < NM; ++k) {
                        Mma32<T>::run(wr[wsx][k / NF], xf[xs][k % NF], acc[k / NF][k % NF]);
                        if (rd && k < NF) xf[xs ^ 1][k] = *(const u32x4*)(xp + k * 4096);
                        if (k == NF + 1 && !(LAST && j >= NSTEP - D)) wload(wnx);
                        if (k == NF + 3 && GA > 0) issue_a(atile, ab_nxt, buf ^ 1, G0);
                        if (k == NF + 5 && GA > 1) issue_a(atile, ab_nxt, buf ^ 1, G0 + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                step(pipe::IC<0>{}); step(pipe::IC<1>{}); step(pipe::IC<2>{}); step(pipe::IC<3>{}); step(pipe::IC<4>{}); step(pipe::IC<5>{});
                step(pipe::IC<6>{}); step(pipe::IC<7>{}); step(pipe::IC<8>{}); step(pipe::IC<9>{}); step(pipe::IC<10>{}); step(pipe::IC<11>{});
                buf ^= 1;
            };
            period(pipe::IC<1>{}, pipe::IC<0>{});
            for (int m = 1; m < P - 1; ++m) period(pipe::IC<0>{}, pipe::IC<0>{});
            period(pipe::IC<0>{}, pipe::IC<1>{});

        // ---- epilogue.  Everything asynchronous (next tile's band, its first weight fragments) lands first: the compiler owns
        // the schedule and the register allocation from here to the vmcnt(0) at the next tile start.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // acc[ni][mi][r]: pixel q' = q0 + 256 wm + 32 mi + l31, cout = cw + 32 ni + 8 (r >> 2) + 4 h + (r & 3).  Group pairs (0,1)
        // and (2,3) are exchanged between the lane halves (v_permlane32_swap): every lane stores 16 bytes = eight consecutive
        // couts; the four stores of a pixel fragment complete one 128-byte line per pixel.
        {
            const int cw = cur.n0 + wn * 64;
            f32x4 bias[2][4];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bias[ni][j] = (epi & DBX_EPI_BIAS) ? *(const f32x4*)(a.bias + cw + ni * 32 + 8 * j + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
            const int q = cur.q0 + wm * NF * 32 + l31;
            int n = q / t.hwp;
            const int rem = q - n * t.hwp;
            int oy = rem / wp, fx = rem - oy * wp;
            const int H = t.hwp / wp;
            int qq = q;
#pragma unroll
            for (int mi = 0; mi < NF; ++mi) {
                const bool ok = qq < t.qtot && fx >= 1 && fx <= wp - 2;
                const int ox = fx - 1;
                T* ypix = (T*)a.y + (size_t)((n * a.y_hp + oy + a.y_pad) * a.y_wp + (ox + a.y_pad)) * (size_t)a.y_ld + cw + 8 * h;
                u32x4 gt[2][2];
                if ((epi & DBX_EPI_GATE) && ok) {
                    const T* gpix = (const T*)a.gate + (size_t)((n * a.g_hp + oy + a.g_pad) * a.g_wp + (ox + a.g_pad)) * (size_t)a.g_ld + cw + 8 * h;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int jp = 0; jp < 2; ++jp) gt[ni][jp] = *(const u32x4*)(gpix + ni * 32 + 16 * jp);
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        u32x2 pk[2];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = 2 * jp + jj;
                            float v[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = acc[ni][mi][4 * j + i] + bias[ni][j][i];
                            if (epi & DBX_EPI_RELU) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                            }
                            if ((epi & DBX_EPI_ACCUM) && ok) {
                                const T* o = ypix - 8 * h + ni * 32 + 8 * j + 4 * h;
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[i] += to_f32(o[i]);
                            }
                            T p[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
                            pk[jj] = *(const u32x2*)p;
                        }
                        // lower half keeps its group 2jp and receives the upper half's; upper half receives the lower's 2jp+1
                        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                        u32x4 o = (u32x4){r0[0], r1[0], r0[1], r1[1]};
                        if (ok) {
                            if (epi & DBX_EPI_GATE) o = gate_packed16(o, gt[ni][jp]);
                            *(u32x4*)(ypix + ni * 32 + 16 * jp) = o;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);                      // one fragment at a time: bounds the live accumulator copies
                // advance 32 q': at most one row wrap (Wp >= 32) or a division
                qq += 32;
                if (wp >= 32) {
                    fx += 32;
                    if (fx >= wp) { fx -= wp; if (++oy == H) { oy = 0; ++n; } }
                } else {
                    n = qq / t.hwp;
                    const int rr = qq - n * t.hwp;
                    oy = rr / wp; fx = rr - oy * wp;
                }
            }
        }
        };
        if (cur.nf == 8) body(pipe::IC<8>{});
        else body(pipe::IC<7>{});
        if (nxt_item >= t.items) break;
        item = nxt_item;
        cur = nxt;
    }
}

// tile schedule: units of WM fragments; tiles of 7..8 units, their number rounded up to fill whole rounds of CUs
static inline WsArgs ws_schedule(long long qtot, int hwp, int wm, int ntile_n, int ncu) {
    WsArgs t;
    const long long units = (qtot + 32 * wm - 1) / (32 * wm);
    long long mt = (units + 7) / 8;
    const long long wgs = mt * ntile_n;
    if (wgs > ncu) {
        const long long up = (wgs + ncu - 1) / ncu * ncu / ntile_n;     // tiles that fill the last round
        if (up > mt && units / up >= 6) mt = up;
    }
    t.mt = (int)mt; t.base = (int)(units / mt); t.extra = (int)(units % mt);
    t.items = (int)(mt * ntile_n); t.hwp = hwp; t.qtot = (int)qtot;
    return t;
}

template <typename T, int WM>
static int launch_conv_ws(const ConvArgs& a, int n, int h, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        constexpr int smem = 2 * ws::pieces(WM) * 1024;
        static_assert(smem <= 160 * 1024, "LDS budget");
        static bool attr_set = false;
        if (!attr_set) {
            DBX_HIP(hipFuncSetAttribute((const void*)conv3x3_ws_kernel<T, WM>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr_set = true;
        }
        static int ncu = 0;
        if (!ncu) {
            int dev = 0;
            DBX_HIP(hipGetDevice(&dev));
            DBX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        }
        const WsArgs t = ws_schedule((long long)n * h * a.x_wp, h * a.x_wp, WM, a.ntile_n, ncu);
        const int grid = t.items < ncu ? t.items : ncu;
        hipLaunchKernelGGL((conv3x3_ws_kernel<T, WM>), dim3(grid), dim3(256), smem, s, a, t);
        DBX_LAUNCH_CHECK();
    }
    return DBX_OK;
}
