// LAB ONLY (tools/band_lab.hip, -DDBX_LAB) -- measured and superseded by conv3x3_ws.hpp: v5, 3x3 / pad-1 convolution as ONE
// continuous tap pipeline, 8 waves, A band + W tiles through the LDS (440 k clocks on conv4_2 at batch 64 vs 472 k for the
// band kernel and 390 k for the ws kernel).  Included by conv_igemm.hip behind DBX_LAB (needs ConvArgs, dma_swz, Mma32).
//
// Tile: 256 consecutive pixels of the linearised frame x 256 couts, 8 waves as 2 (pixel halves) x 4 (cout quarters), each wave
// 128 px x 64 couts on v_mfma_f32_32x32x16 (weights = A operand / accumulator rows, pixels = B operand / accumulator
// columns: a lane ends up with runs of four consecutive couts of ONE pixel).
//
// K order: period m = (ky, 64-channel chunk); inside a period six tap-steps s = 2 kx + half, each one 3x3 tap x 32 channels
// = 16 MFMAs per wave (two k16 sub-steps of 8).
//   * A band of period m: rows q0 - Wp - 1 + ky Wp .. + 263 of the frame, 128 B (64 channels) per row, 33 pieces of
//     8 rows (full 128-byte line requests), double buffered per period; the three kx taps read it at row shifts 0/1/2.
//   * W tile of a tap-step: 256 couts x 64 B, 16 pieces, in a ring of NW slots (loads run NW-2 tap-steps ahead).
// The fragment reads of sub-step u+1 are issued while the MFMAs of sub-step u run (two register sets), ACROSS tap-step and
// period boundaries; the one barrier per tap-step sits between its two sub-steps, so there is no stage-boundary bubble:
// the barrier of step t publishes the data of step t+1 (every wave waited its own counted vmcnt first) and frees the slot
// of step t-1 (whose last reads were issued a full step earlier), which is refilled right behind it.
#pragma once

namespace pipe {
constexpr int BM = 256, BN = 256;
constexpr int AP = 33;                          // A-band pieces (8 rows x 128 B) per period
constexpr int ABUF = AP * 1024;
constexpr int WSLOT = 16 * 1024;
constexpr int W0 = 2 * ABUF;
// loads every wave issues at step s of a period (minimum over the waves): two W pieces + the next period's A pieces
constexpr int issued(int s, bool last, int NW) { return last ? (s < 7 - NW ? 2 : 0) : 2 + (s < 2 ? 2 : (s == 2 ? 1 : 0)); }
// loads that may still be in flight at the wait of step t: everything issued in the NW-3 steps before it
constexpr int allowed(int t, bool last, int NW) {
    int n = 0;
    for (int k = 1; k <= NW - 3; ++k) {
        const int s = t - k;
        n += s >= 0 ? issued(s, last, NW) : issued(s + 6, false, NW);
    }
    return n;
}
}  // namespace pipe

// ABL (lab only): timing ablations, results are wrong -- 1: no LDS-DMA in the loop, 2: no fragment reads in the loop, 4: no waits/barriers
template <typename T, int NW, int ABL = 0>
__global__ __launch_bounds__(512) void conv3x3_pipe_kernel(const ConvArgs a) {
    using namespace pipe;
    constexpr int ES = sizeof(T);
    static_assert(ES == 2, "16-bit types");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
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

    // ---- LDS-DMA sources (the LDS image of a piece is lane-linear: the chunk swizzle goes on the source address)
    const char* wsrc[2];
    {
        const int lr = lane >> 2, lc = lane & 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 16 * (2 * wave + j) + lr;
            wsrc[j] = a.w + (size_t)(n0 + row) * a.ktot_bytes + ((lc ^ dma_swz<64>(row)) << 4);
        }
    }
    const char* asrc;                                                   // piece q = wave + 8 j: + 64 j pixels
    {
        const int lr8 = lane >> 3, lc8 = lane & 7;
        const int sw = (4 * (wave & 1) + (lr8 >> 1)) & 7;               // (row >> 1) & 7 of row = 8 (wave + 8 j) + lr8
        asrc = a.x + ((q0 + 8 * wave + lr8) - a.x_wp - 1) * (long long)pix_bytes + ((lc8 ^ sw) << 4);
    }
    auto glds = [&](const char* src, int lds_off) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + lds_off), 16, 0, 0);
    };
    // byte offsets of a period: ky-major, then the 64-channel chunk
    int wb_cur = 0, ab_cur = 0, p_ky = 0, p_kc = 0;                     // period m
    int wb_nxt, ab_nxt;                                                 // period m + 1
    auto advance = [&](int& wb, int& ab) {
        if (++p_kc == KC) { p_kc = 0; ++p_ky; }
        wb = p_ky * 3 * cin_bytes + p_kc * 128;
        ab = p_ky * a.x_wp * pix_bytes + p_kc * 128;
    };
    int slot_w = 0;                                                     // ring slot the next W tile goes to
    auto issue_w = [&](int wb, int s, int j) {                          // piece j of the W tile of tap-step s of the period at wb
        const int off = wb + (s >> 1) * cin_bytes + (s & 1) * 64;
        glds(wsrc[j] + off, W0 + slot_w * WSLOT + (2 * wave + j) * 1024);
    };
    // A piece wave + 8 j of the period at ab.  The band has 33 pieces: the fifth slot (j = 4) exists for wave 0 only; the other
    // waves repeat their fourth piece there (same bytes to the same place) so that every wave issues the same number of
    // loads per step -- the counted vmcnt waits and the pinned instruction order stay uniform and branch-free.
    auto issue_a = [&](int ab, int buf, int j) {
        const int jj = (j == 4 && wave != 0) ? 3 : j;
        glds(asrc + ab + (long long)jj * 64 * pix_bytes, buf * ABUF + (wave + 8 * jj) * 1024);
    };

    // ---- fragment read addresses.  W slot: 64-byte rows, lane (l31, h) reads 16-byte chunk 2 i + h of row l31 (+ 32 ni) at
    // physical chunk (2 i + h) ^ swz(row) = ((h ^ swz) ^ 2 i): one base per lane, sub-step i flips address bit 5.
    // A band: 128-byte rows, chunk 2 i + h (i = 0..3) of row l31 + kx (+ 32 mi), swizzle (row >> 1) & 7.
    const int wlane = (wn * 64 + l31) * 64 + ((h ^ dma_swz<64>(l31)) << 4);
    int xlane[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xlane[kx] = (wm * 128 + l31 + kx) * 128 + ((h ^ (((l31 + kx) >> 1) & 7)) << 4);

    f32x16 acc[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
    u32x4 wf[2][2], xf[2][4];
    int slot_r = 0;                                                     // ring slot of the tap-step being read
    // One sub-step: the 8 MFMAs of register set `cur`, with the six fragment reads of the NEXT sub-step (set cur ^ 1; W chunk
    // pair e of the slot at slot_r, band chunk pair 2 (s & 1) + e at row shift s >> 1 of buffer rbuf) and the step's G
    // LDS-DMA loads placed by hand: one read behind each of the first six MFMAs, one load behind MFMAs 2..5.  The scheduler
    // otherwise parks the reads behind the MFMAs, right in front of the next sub-step's lgkmcnt(0); sched_barrier(0) keeps
    // the source order.
    auto substep = [&](int cur, bool do_read, int rbuf, int s, int e, int G, auto&& ld) {
        const char* wp = smem + W0 + slot_r * WSLOT + (wlane ^ (e << 5));
        const char* xp = smem + rbuf * ABUF + (xlane[s >> 1] ^ ((2 * (s & 1) + e) << 5));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            Mma32<T>::run(wf[cur][k >> 2], xf[cur][k & 3], acc[k >> 2][k & 3]);
            if (do_read && !(ABL & 2)) {
                if (k < 2) { if (!(ABL & 8)) wf[cur ^ 1][k] = *(const u32x4*)(wp + k * 2048); }
                else if (k < 6) xf[cur ^ 1][k - 2] = *(const u32x4*)(xp + (k - 2) * 4096);
            }
            if (k >= 2 && k - 2 < G && !(ABL & 1)) ld(k - 2);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto no_ld = [](int) {};

    // ---- prologue: band of period 0, W tiles of tap-steps 0 .. NW-2
#pragma unroll
    for (int j = 0; j < 5; ++j) issue_a(0, 0, j);
#pragma unroll
    for (int s = 0; s < NW - 1; ++s) { issue_w(0, s, 0); issue_w(0, s, 1); slot_w = slot_w + 1; }
    wb_nxt = 0; ab_nxt = 0;
    advance(wb_nxt, ab_nxt);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NW - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    {
        const char* wp = smem + W0 + wlane;
        const char* xp = smem + xlane[0];
        wf[0][0] = *(const u32x4*)wp; wf[0][1] = *(const u32x4*)(wp + 2048);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) xf[0][mi] = *(const u32x4*)(xp + mi * 4096);
    }

    if (ABL & 8) { wf[1][0] = wf[0][0]; wf[1][1] = wf[0][1]; }
    int buf = 0;
    auto period = [&](auto LAST_) {
        constexpr bool LAST = decltype(LAST_)::value != 0;
        auto step = [&](auto S_) {
            constexpr int s = decltype(S_)::value;
            // sub-step 0: read (s, 1) while (s, 0) multiplies
            substep(0, true, buf, s, 1, 0, no_ld);
            if (LAST && s == 5) {
                substep(1, false, buf, 0, 0, 0, no_ld);
                return;
            }
            if (!(ABL & 4)) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allowed(s, LAST, NW)) : "memory");
                __builtin_amdgcn_s_barrier();
            }
            // refill behind the barrier: W tile NW-1 tap-steps ahead, the next period's band in steps 0..2
            constexpr int s2 = (s + NW - 1) % 6, pa = (s + NW - 1) / 6;
            constexpr int G = issued(s, LAST, NW);
            const int wbase = pa ? wb_nxt : wb_cur;
            auto ld = [&](int g) {
                if (g < 2) { if (!(ABL & 8)) issue_w(wbase, s2, g); }
                else if (s < 2) issue_a(ab_nxt, buf ^ 1, 2 * s + g - 2);
                else issue_a(ab_nxt, buf ^ 1, 4);
            };
            slot_r = slot_r == NW - 1 ? 0 : slot_r + 1;
            // sub-step 1: read (s + 1, 0) while (s, 1) multiplies
            if (s < 5) substep(1, true, buf, s + 1, 0, G, ld);
            else substep(1, true, buf ^ 1, 0, 0, G, ld);
            if (G > 0) slot_w = slot_w == NW - 1 ? 0 : slot_w + 1;
        };
        step(IC<0>{}); step(IC<1>{}); step(IC<2>{}); step(IC<3>{}); step(IC<4>{}); step(IC<5>{});
    };
    if (!(ABL & 16)) {
    for (int m = 0; m < P - 1; ++m) {
        period(IC<0>{});
        wb_cur = wb_nxt; ab_cur = ab_nxt;
        advance(wb_nxt, ab_nxt);
        buf ^= 1;
    }
    period(IC<1>{});
    }

    // ---- epilogue over frame pixels: only interior pixels are stored.  acc[ni][mi][r]: pixel = q0 + 128 wm + 32 mi + l31,
    // cout = n0 + 64 wn + 32 ni + 8 (r >> 2) + 4 h + (r & 3).  Group pairs (0,1) and (2,3) are exchanged between the lane halves
    // (v_permlane32_swap) so that every lane stores 16 bytes = eight consecutive couts.
    const int epi = a.epi;
    const int fpix = a.x_hp * a.x_wp, nimg = a.M / a.HoWo;
    const int cw = n0 + wn * 64;
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
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {                            // group pair (2 jp, 2 jp + 1)
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
                // lower half keeps its group 2jp and receives the upper half's; upper half receives the lower's 2jp+1
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
        if (a.x_wp >= 32) {                                             // one row wrap at most per 32-pixel advance
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

template <typename T, int NW, int ABL = 0>
static int launch_conv_pipe(const ConvArgs& a, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        constexpr int smem = pipe::W0 + NW * pipe::WSLOT;
        static_assert(smem <= 160 * 1024, "LDS budget");
        static bool attr_set = false;
        if (!attr_set) {
            DBX_HIP(hipFuncSetAttribute((const void*)conv3x3_pipe_kernel<T, NW, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr_set = true;
        }
        hipLaunchKernelGGL((conv3x3_pipe_kernel<T, NW, ABL>), dim3(a.nblocks), dim3(512), smem, s, a);
        DBX_LAUNCH_CHECK();
    }
    return DBX_OK;
}
