// v7: 3x3 / pad-1 convolution with register-streamed weights ("ws"): four fat waves (one per SIMD, the whole 512-register
// file each), persistent workgroups.  Included by conv_igemm.hip (needs ConvArgs, gate_packed16).
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
// weight loads run D steps ahead of their MFMAs.  Round 4: 3 steps for the 1x1 GEMMs with 256-pixel tiles (their ring of four register
// sets has room for it): a K = 16 step of 16 MFMAs is ~512 cycles, an L2 hit ~500-800 ns -- two steps ahead was short of it
#ifndef DBX_WS_D1
#define DBX_WS_D1 3
#endif
constexpr int dist(int KS, int NFX = 8) { return (KS == 1 && NFX == 8) ? DBX_WS_D1 : 2; }
constexpr int DMAX = 3;
constexpr int WL = 2;                           // weight loads per wave and step (64 couts = two 32-row fragments)
// K=16 steps per period.  3x3: period = (ky, 64-channel chunk), step = kx * 4 + 16-channel chunk, ONE band of 128-byte rows read at
// row shifts 0/1/2.  1x1: period = 128 channels, step = sub-band * 4 + 16-channel chunk, TWO sub-bands of 128-byte rows.
constexpr int nstep(int KS) { return KS == 3 ? 12 : 8; }
constexpr int subs(int KS) { return KS == 3 ? 1 : 2; }
// band pieces (8 rows x 128 B) per sub-band for WM wave rows of 256 pixels, per period, and per-wave LDS-DMA slots (4 waves)
constexpr int sub_pieces(int WM, int KS, int NFX = 8) { return (32 * NFX * WM + (KS == 3 ? 2 : 0) + 7) / 8; }   // NFX: fragments of 32 q' per wave row (8; 4: two workgroups per CU)
constexpr int pieces(int WM, int KS, int NFX = 8) { return subs(KS) * sub_pieces(WM, KS, NFX); }
constexpr int slots(int WM, int KS, int NFX = 8) { return (pieces(WM, KS, NFX) + 3) / 4; }
// LDS-DMA loads a wave issues at step i of a period (behind the step's weight loads): the next period's band -- only in
// steps 0 .. NS-D-2: the seam wait at step NS-1 (for the weights issued at step NS-1-D) must cover them all
constexpr int g(int i, int WM, int KS, int NFX = 8) {
    const int D = dist(KS, NFX);
    const int NS = nstep(KS), NA = NS - D - 1;                          // 9 (3x3) / 5 or 4 (1x1) issue steps
    i = ((i % NS) + NS) % NS;
    if (i >= NA) return 0;
    const int s = slots(WM, KS, NFX), lo = s / NA, rem = s % NA;            // 3x3: 9 = 1 per step (WM 1), 17 = 2,..,2,1 (WM 2); 1x1: 16 = 4,3,3,3,3
    return lo + (i < rem ? 1 : 0);
}
constexpr int gsum(int i, int WM, int KS, int NFX = 8) { int n = 0; for (int k = 0; k < i; ++k) n += g(k, WM, KS, NFX); return n; }   // slots before step i
// VMEM operations issued after the weight loads of step j (which go out at step j - D): may still be in flight at its wait.
// (A tile's last period issues a few extra LDS-DMA pieces in its last D steps and older stores may be pending at a tile
// start: both only make a wait longer.)
constexpr int allowed(int j, int WM, int KS, int NFX = 8) {
    const int D = dist(KS, NFX);
    int n = 0;
    for (int i = j - D; i <= j - 1; ++i) n += g(i, WM, KS, NFX) + (i > j - D ? WL : 0);
    return n;
}
constexpr int wld(int WM) { return 2 / WM; }                            // LDS-DMA pieces per wave for one weight step: 8 KB (BN 256) / 4 KB (BN 128)
constexpr int gmax(int WM, int KS, int NFX = 8) { int m = 0; for (int i = 0; i < nstep(KS); ++i) m = g(i, WM, KS, NFX) > m ? g(i, WM, KS, NFX) : m; return m; }
}  // namespace ws

struct WsArgs {
    int mt;              // M tiles
    int base, extra;     // tile t covers base + (t < extra) units of WM fragments (32 q' each), starting at unit t base + min(t, extra)
    int items;           // mt * ntile_n work items
    int hwp;             // H * Wp: q' per image
    int qtot;            // N * H * Wp
    int xpad;            // frame width of x (1; 0 for an unframed 1x1 input): halo columns fx < xpad, fx >= Wp - xpad are dropped
    int dbg;             // ablation bits (lab builds: DBX_WS_DBG = 1 no periods, 2 no stores, 4 no band DMA, 8 no weight loads); 0 in the product library
};
// Lab builds (tools/band_lab.hip) read the bits from the environment.  The product library never does: the 3x3 instantiations see
// the constant 0 (their tests fold away: 225 -> 208 us per conv4 layer), the 1x1 instantiations keep testing the field, which the
// host always sets to 0 -- with the tests folded away the compiler schedules the heads' epilogue 25 % slower (same-box A/B:
// heads forward 901 -> 1165 us, its split data gradient 665 -> 770 us), so the dead tests stay as a code-generation anchor.
#ifdef DBX_LAB
#define WS_DBG(t) ((t).dbg)
#else
#define WS_DBG(t) (KS == 1 ? (t).dbg : 0)
#endif

// EPIK: 0 = the epilogue reads its kind from the arguments at run time; 1 = fixed to BIAS + hash dropout on a single destination
// (the heads' forward GEMM): the flag tests fold away and with them the ReLU / gate / second-destination code.
// EPIK 2 = EPIK 1 + the heads' SECOND 1x1 convs (DenseBox.py:158-162: Conv1x1(768 -> 512) -> Dropout -> Conv1x1(512 -> k), k <= 8) on
// the tile while it is in registers: the 16-byte store chunk of a lane (eight consecutive hidden channels of one pixel, rounded and
// dropped -- exactly what lands in the hidden map) IS the B operand of a v_mfma_f32_32x32x16 whose A operand is the [k rows][16
// channels] slice of that head's second weight (fragment order, a.w2f); four MFMAs per pixel fragment leave a wave's partial sums over
// its 64 channels in four accumulator registers per lane, the four waves are summed through the dead band buffer in a fixed order,
// and the workgroup stores a.part[cout tile][pixel][8] (fp32).  dbx_heads_forward_fused adds a head's two tiles and the bias:
// the 944 MB hidden map is not read back by a second GEMM (conv_igemm_dma<256,64>: 219 us of the step at batch 64).
template <typename T, int WM, int KS, int EPIK = 0, int NFX = 8>
__global__ __launch_bounds__(256, NFX == 8 ? 1 : 2) void conv3x3_ws_kernel(const ConvArgs a, const WsArgs t) {
    using namespace ws;
    constexpr int ES = sizeof(T);
    static_assert(ES == 2, "16-bit types");
    static_assert(KS == 3 || (KS == 1 && WM == 1), "3x3, or 1x1 with 256-cout tiles");
    constexpr int WN = 4 / WM;                                          // waves along the couts
    constexpr int BN = 64 * WN;
    constexpr int NSTEP = nstep(KS), SUBS = subs(KS), PS = sub_pieces(WM, KS, NFX);
    constexpr int AP = pieces(WM, KS, NFX), SL = slots(WM, KS, NFX), ABUF = AP * 1024, SUBB = PS * 1024;
    constexpr int WSTEP = BN * 32;                                      // packed weight bytes per step of one BN-cout tile
    constexpr int D = dist(KS, NFX);
    constexpr int WLZ = 2 * ABUF;                                       // LDS landing zone of a tile's first D weight steps
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;
    const int pix_bytes = a.x_ld * ES;
    const int KC = a.cpt / 8;                                           // 64-channel chunks
    const int P = KS == 3 ? 3 * KC : KC / 2;                            // periods
    const int wp = a.x_wp;

    // ---- persistent schedule: workgroup g runs on XCD g % 8.  One cout tile: item = round * G + (xcd-contiguous index), so the 32
    // workgroups of an XCD work on 32 neighbouring pixel tiles at any time.  Two or four cout tiles (conv4: 512 couts): an XCD
    // keeps to ONE of them for the whole launch -- its workgroups stream the same 2.4 MB of weights, which then stay in its
    // 4 MB L2 (both halves together thrash it, and L2 misses are capped at ~12 B/clk/CU against 16 B/clk/CU of demand);
    // the pixel bands are then fetched by two XCDs instead of one (5 B/clk/CU).
    const int G = gridDim.x;
    const bool by_xcd = (G & 7) == 0 && (a.ntile_n == 2 || a.ntile_n == 4);
    int item, istride;                                                  // linear (tile_m, tile_n) index: tm * ntile_n + tn
    if (by_xcd) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        item = ((xcd / a.ntile_n) * (G >> 3) + j) * a.ntile_n + xcd % a.ntile_n;
        istride = G;                                                    // G / ntile_n pixel tiles per round
    } else {
        item = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
        istride = G;
    }
    if (item >= t.items) return;

    // ---- per-tile state (all uniform)
    struct Tile { int q0, nf, n0, sw; const char* wbase; const char* abase; };
    auto tile_of = [&](int it) {
        Tile r;
        const int tm = it / a.ntile_n, tn = it - tm * a.ntile_n;
        const int u0 = tm * t.base + (tm < t.extra ? tm : t.extra);
        r.nf = t.base + (tm < t.extra ? 1 : 0);
        r.q0 = u0 * (32 * WM);
        r.n0 = tn * BN;
        const int img = r.q0 / t.hwp;
        // first band row that belongs to the next image: pixel index of its first q' in this tile (+ 1 for 3x3, see header)
        r.sw = (img + 1) * t.hwp - r.q0 + (KS == 3 ? 1 : 0);
        r.wbase = a.w + (size_t)tn * P * NSTEP * WSTEP;
        // band row 0 is frame position pos(q0) (3x3, ky = 0: - 1 - Wp), pos(q') = q' + (2 img + 1) Wp xpad
        r.abase = a.x + ((long long)r.q0 + (long long)(2 * img + 1) * wp * t.xpad - (KS == 3 ? 1 + wp : 0)) * pix_bytes;
        return r;
    };

    // ---- weight stream: scalar base walks the packed image step by step, one lane offset for the whole kernel
    const unsigned wvoff = wn * 2048 + lane * 16;
    const char* wptr;
    constexpr int RING = KS == 3 ? 3 : 4;                               // weight register sets: divides NSTEP (12 / 8), > D
    static_assert(NSTEP % RING == 0 && RING > D, "weight ring");
    u32x4 wr[RING][2];
    // s_nop 4: the base may have just been restored from a spill by v_readlane (VALU-written SGPR -> VMEM address needs five
    // wait states; the compiler pads its own instructions, not the inside of an asm statement)
    auto wload = [&](int set) {
        const char* wp = wptr;
        if constexpr (NFX != 8) {
            // (two workgroups per CU: with the 256-register budget the compiler kept this uniform pointer in a VGPR pair and printed it
            //  as the asm statement's "s" operand -- make it an SGPR pair by construction)
            const unsigned long long u = (unsigned long long)wptr;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
            wp = (const char*)(((unsigned long long)hi << 32) | lo);
        }
        asm volatile("s_nop 4\n\t"
                     "global_load_dwordx4 %0, %2, %3\n\t"
                     "global_load_dwordx4 %1, %2, %3 offset:1024"
                     : "=&v"(wr[set][0]), "=&v"(wr[set][1])
                     : "v"(wvoff), "s"(wp)
                     : "memory");
        wptr += WSTEP;
    };

    // ---- band pieces by LDS-DMA: slot i of this wave is piece wave + 4 i; slots past the last piece repeat the wave's last real
    // piece (same bytes to the same place) so that every wave issues the same number of loads (uniform counted waits).
    // Address = uniform base of the piece (scalar arithmetic) + ONE lane offset register; the halo-row skip is a per-lane select
    // against a uniform threshold.  The chunk swizzle goes on the source: chunk ^ ((row >> 1) & 7), row = 8 (wave + 4 i) + lr8.
    const int lr8 = lane >> 3, lc8 = lane & 7;
    const unsigned a_lane = lr8 * pix_bytes + ((lc8 ^ ((4 * wave + (lr8 >> 1)) & 7)) << 4);
    const unsigned skip_bytes = 2 * wp * t.xpad * pix_bytes;
    auto issue_a = [&](const char* abase, int sw, int ab, int buf, int i) {
        int ii = i;
        if (wave + 4 * i >= AP) ii = i - 1;                             // uniform per wave
        const int piece = wave + 4 * ii;
        const int sub = SUBS == 1 ? 0 : piece / PS;                     // 1x1: second 64-channel chunk of the period
        const int row0 = 8 * (piece - sub * PS);                        // uniform
        const char* sbase = abase + row0 * pix_bytes + ab + sub * 128;
        const unsigned voff = a_lane + ((lr8 >= sw - row0) ? skip_bytes : 0u);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + voff),
                                         (__attribute__((address_space(3))) void*)(smem + buf * ABUF + (wave + 4 * ii) * 1024), 16, 0, 0);
    };
    // first D weight steps of a tile by LDS-DMA into the landing zone: piece p of step d = bytes [1024 p, +1024) of that step
    auto issue_w0 = [&](const char* wbase, int d) {
#pragma unroll
        for (int i = 0; i < wld(WM); ++i) {
            const int p = wave * wld(WM) + i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + d * WSTEP + p * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(smem + WLZ + d * WSTEP + p * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read addresses: 128-byte rows, chunk 2 c + h of row l31 + kx (+ 32 mi), swizzle (row >> 1) & 7
    int xlane[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xlane[kx] = (l31 + kx) * 128 + ((h ^ (((l31 + kx) >> 1) & 7)) << 4);   // + wave row: 32 NF wm rows
    // step j of a period reads chunk pair j % 4 at row shift j / 4 (3x3) or of sub-band j / 4 (1x1)
    auto xoff = [&](int j, int xb0) { return KS == 3 ? (xb0 ^ ((j & 3) << 5)) : (j >> 2) * SUBB + (xb0 ^ ((j & 3) << 5)); };

    u32x4 xf[2][NFX];
    int buf = 0;
    Tile cur = tile_of(item);

    // ---- prologue of the first tile: its first band and weight steps (the later tiles' are issued by their predecessors)
#pragma unroll
    for (int i = 0; i < SL; ++i) issue_a(cur.abase, cur.sw, 0, 0, i);
#pragma unroll
    for (int d = 0; d < D; ++d) issue_w0(cur.wbase, d);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    for (;;) {
        // ---- tile start: band 0 and the first D weight steps sit in the LDS (every wave waited for its own pieces: prologue, or
        // the vmcnt(0) in front of the previous epilogue -- LDS-DMA involves no registers, so it may fly across compiler-
        // scheduled code, unlike an asm load whose destination the compiler believes to be ready and may spill or move).
        // The epilogue's stores may still be in flight: they only make the first counted waits of the tile wait longer.
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        wptr = cur.wbase + D * WSTEP;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            wr[d][0] = *(const u32x4*)(smem + WLZ + d * WSTEP + wvoff);
            wr[d][1] = *(const u32x4*)(smem + WLZ + d * WSTEP + wvoff + 1024);
        }
        const int nxt_item = item + istride;
        const bool more = nxt_item < t.items;
        Tile nxt = cur;                                                 // last tile: a harmless re-fetch of its own start

        // One tile, specialised on its fragment count.  The accumulators live and die inside: the two instantiations assign them
        // to different registers, and nothing but scalars and the operand fragments crosses the merge behind the branch.
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
                const char* xp = smem + buf * ABUF + xrow + xoff(0, xlane[0]);
#pragma unroll
                for (int mi = 0; mi < NF; ++mi) xf[0][mi] = *(const u32x4*)(xp + mi * 4096);
            }
            int p_ky = 0, p_kc = 0;
            // ONE period body for every period of the tile (a FIRST / LAST specialisation triples the code and the register
            // pressure: the compiler then spills lane invariants and drains the pipeline around every reload).  What differs is
            // scalar: whose band is prefetched, two skipped waits at the tile start, extra LDS-DMA at the tile end.
            for (int m = 0; m < ((WS_DBG(t) & 1) ? 0 : P); ++m) {
                const bool first = m == 0, last = m == P - 1;
                if (last && more) nxt = tile_of(nxt_item);
                int ab_nxt = 0;
                if (!last) {
                    if (KS == 3) {
                        if (++p_ky == 3) { p_ky = 0; ++p_kc; }                  // ky fastest: the bands of one chunk share their rows
                        ab_nxt = p_ky * wp * pix_bytes + p_kc * 128;
                    } else {
                        ab_nxt = (m + 1) * 256;
                    }
                }
                const char* asrc = last ? nxt.abase : cur.abase;
                const int asw = last ? nxt.sw : cur.sw;
                auto step = [&](auto J_) {
                    constexpr int j = decltype(J_)::value;
                    constexpr int wsx = j % RING, wnx = (j + D) % RING, xs = j & 1;
                    // the weights of steps 0 .. D-1 of a tile came through the LDS: nothing to wait for
                    if (j >= D || !first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allowed(j, WM, KS, NFX)) : "memory");
                    if (j == NSTEP - 1) {
                        // period seam: the next band landed (its loads are older than the weights just waited for) and every
                        // wave is done with this one (its last reads were issued a step ago).  In a tile's last period this
                        // publishes nothing yet (the next tile's band is waited for in front of the epilogue): harmless.
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const int rbuf = j == NSTEP - 1 ? buf ^ 1 : buf;
                    constexpr int jn = (j + 1) % NSTEP;
                    int xb = xlane[KS == 3 ? jn >> 2 : 0];
                    asm volatile("" : "+v"(xb));                        // recompute per step: twelve hoisted address registers spill
                    const char* xp = smem + rbuf * ABUF + xrow + xoff(jn, xb);
                    constexpr int GA = g(j, WM, KS, NFX), G0 = gsum(j, WM, KS, NFX);
#pragma unroll
                    for (int k = 0; k < NM; ++k) {
                        Mma32<T>::run(wr[wsx][k / NF], xf[xs][k % NF], acc[k / NF][k % NF]);
                        // (not in a tile's very last step: its band is the next tile's, still landing -- the tile start reads it)
                        if (k < NF && !(last && j == NSTEP - 1)) xf[xs ^ 1][k] = *(const u32x4*)(xp + k * 4096);
                        // (slots of the step's non-MFMA work between its MFMAs; short steps -- NFX 4: 6 or 8 MFMAs -- bunch them at the end)
                        constexpr int KA0 = NM >= 14 ? NF + 3 : NM - 2, KA1 = NM >= 14 ? NF + 5 : NM - 1, KA2 = NM >= 14 ? NF + 6 : NM - 1;
                        if (k == NF + 1 && !(WS_DBG(t) & 8)) wload(wnx);    // (a tile's last D steps run past its stream: drained, unused)
                        if (k == KA0 && GA > 0 && !(WS_DBG(t) & 4)) issue_a(asrc, asw, ab_nxt, buf ^ 1, G0);
                        if (k == KA1 && GA > 1) issue_a(asrc, asw, ab_nxt, buf ^ 1, G0 + 1);
                        if (k == KA2 && GA > 2) issue_a(asrc, asw, ab_nxt, buf ^ 1, G0 + 2);
                        if (k == NM - 1 && GA > 3) issue_a(asrc, asw, ab_nxt, buf ^ 1, G0 + 3);
                        if (k == KA0 && j >= NSTEP - D && last) issue_w0(nxt.wbase, j - (NSTEP - D));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                step(pipe::IC<0>{}); step(pipe::IC<1>{}); step(pipe::IC<2>{}); step(pipe::IC<3>{}); step(pipe::IC<4>{}); step(pipe::IC<5>{});
                step(pipe::IC<6>{}); step(pipe::IC<7>{});
                if constexpr (KS == 3) { step(pipe::IC<8>{}); step(pipe::IC<9>{}); step(pipe::IC<10>{}); step(pipe::IC<11>{}); }
                buf ^= 1;
            }

            // ---- epilogue.  Everything asynchronous (next tile's band and first weight steps, the over-run weight loads) lands
            // first: the compiler owns the schedule and the register allocation from here to the next tile start.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // acc[ni][mi][r]: pixel q' = q0 + 32 NF wm + 32 mi + l31, cout = cw + 32 ni + 8 (r >> 2) + 4 h + (r & 3).  Group pairs
            // (0,1) and (2,3) are exchanged between the lane halves (v_permlane32_swap): every lane stores 16 bytes = eight
            // consecutive couts; the four stores of a pixel fragment complete one 128-byte line per pixel.
            // split destination (1x1 only: the data gradient of the fusion concat): cout tiles at or past split_c go to y2 / gate2
            const bool second = EPIK == 0 && a.split_c > 0 && cur.n0 >= a.split_c;
            constexpr bool FIX = EPIK >= 1;                            // fixed epilogue: bias + hash dropout
            const int epi = FIX ? (DBX_EPI_BIAS | DBX_EPI_DROPHASH) : (second ? a.epi2 : a.epi);
            char* const ybase = second ? a.y2 : a.y;
            const char* const gbase = second ? a.gate2 : a.gate;
            const int y_hp = second ? a.y2_hp : a.y_hp, y_wp = second ? a.y2_wp : a.y_wp, y_ld = second ? a.y2_ld : a.y_ld, y_pad = second ? a.y2_pad : a.y_pad;
            const int g_hp = second ? a.g2_hp : a.g_hp, g_wp = second ? a.g2_wp : a.g_wp, g_ld = second ? a.g2_ld : a.g_ld, g_pad = second ? a.g2_pad : a.g_pad;
            const int cw = cur.n0 + wn * 64;                            // first cout of this wave (bias / dropout index)
            const int cy = cw - (second ? a.split_c : 0);               // ... within its destination
            f32x4 bias[2][4];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bias[ni][j] = (epi & DBX_EPI_BIAS) ? *(const f32x4*)(a.bias + cw + ni * 32 + 8 * j + 4 * h) * (FIX ? 2.f : 1.f) : (f32x4){0.f, 0.f, 0.f, 0.f};
            int qq = cur.q0 + wm * NF * 32 + l31;
            int n = qq / t.hwp;
            const int rem = qq - n * t.hwp;
            int oy = rem / wp, fx = rem - oy * wp;
            const int H = t.hwp / wp;
            const int Wo = wp - 2 * t.xpad;
            // ReLU gate (data gradients): the four 16-byte gate chunks of a pixel fragment are fetched PD fragments ahead of their use
            // -- with one wave per SIMD a load issued and consumed inside one fragment's code exposes its whole latency, eight
            // times per tile (the gated launches ran 10 % behind the un-gated ones); a second pixel walker runs ahead of the stores
            constexpr int PD = 4;
            u32x4 gring[PD][2][2];
            int g_qq = qq, g_n = n, g_oy = oy, g_fx = fx;
            auto gate_fetch = [&](u32x4 (&dst)[2][2]) {
                const bool okg = g_qq < t.qtot && g_fx >= t.xpad && g_fx < wp - t.xpad;
                if (okg) {
                    const T* gpix = (const T*)gbase + (size_t)((g_n * g_hp + g_oy + g_pad) * g_wp + (g_fx - t.xpad + g_pad)) * (size_t)g_ld + cy + 8 * h;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int jp = 0; jp < 2; ++jp) dst[ni][jp] = *(const u32x4*)(gpix + ni * 32 + 16 * jp);
                }
                g_qq += 32;
                if (wp >= 32) {
                    g_fx += 32;
                    if (g_fx >= wp) { g_fx -= wp; if (++g_oy == H) { g_oy = 0; ++g_n; } }
                } else {
                    g_n = g_qq / t.hwp;
                    const int rr = g_qq - g_n * t.hwp;
                    g_oy = rr / wp; g_fx = rr - g_oy * wp;
                }
            };
            if (EPIK == 0 && (epi & DBX_EPI_GATE)) {
#pragma unroll
                for (int d = 0; d < PD && d < NF; ++d) gate_fetch(gring[d]);
            }
            // EPIK 2: this wave's four [32 rows][16 channels] slices of the second weight (block = hidden channel / 16; the image
            // is dbx_pack_weight mode 4 with 256 rows: eight 1-KiB row blocks per channel block, the first holds rows 0..31)
            u32x4 w2f[2][2];
            char* const red = smem + (buf ^ 1) * ABUF;                  // the band buffer the tile has finished with (seam barrier passed)
            unsigned st_pix[2] = {0u, 0u};
            bool st_ok[2] = {false, false};
            if constexpr (EPIK == 2) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp)
                        w2f[ni][jp] = *(const u32x4*)(a.w2f + (size_t)((cw + ni * 32 + jp * 16) >> 4) * 8192 + lane * 16);
            }
#pragma unroll
            for (int mi = 0; mi < NF; ++mi) {
                const bool ok = qq < t.qtot && fx >= t.xpad && fx < wp - t.xpad;
                const int ox = fx - t.xpad;
                T* ypix = (T*)ybase + (size_t)((n * y_hp + oy + y_pad) * y_wp + (ox + y_pad)) * (size_t)y_ld + cy + 8 * h;
                u32x4 (&gt)[2][2] = gring[mi % PD];
                const unsigned mpix = (unsigned)((n * H + oy) * Wo + ox);   // output pixel index (dropout counter)
                f32x16 acc2;
                if constexpr (EPIK == 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
                    if (mi == wn) { st_pix[0] = mpix; st_ok[0] = ok; }      // (wave-uniform tests: the fragments this wave reduces below)
                    if (mi == wn + 4) { st_pix[1] = mpix; st_ok[1] = ok; }
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    unsigned h32 = 0;                                    // EPIK 1: one hash for this wave's 32 couts of the pixel
                    // (shifted by this lane half's 4 h once: the per-element fields are then immediate bit positions 8 j + i -- with the
                    // shift amounts 8 j + 4 h kept as four lane invariants the compiler spilled them, and every scratch reload waits
                    // vmcnt(0), i.e. for all the output stores in flight)
                    // The select goes through an SGPR-pair lane mask (inline asm): a lane-dependent shift amount 4 h is one more VGPR invariant, and it
                    // was the one the allocator spilled next.
                    if constexpr (FIX) {
                        const unsigned hraw = dbx_drop_hash32(a.drop_seed, mpix, (unsigned)(cw + ni * 32) >> 5);
                        const unsigned long long upper = 0xffffffff00000000ull;        // lanes 32..63 (h = 1)
                        asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(h32) : "v"(hraw), "v"(hraw >> 4), "s"(upper));
                    }
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        u32x2 pk[2];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = 2 * jp + jj;
                            float v[4];
                            if constexpr (FIX) {
                                // 2 (acc + bias) as two v_pk_fma_f32 (bias pre-doubled); a dropped element is cleared by ANDing
                                // with the sign-extended one-bit field of the hash (v_bfe_i32 + v_and_b32)
                                typedef float f32x2 __attribute__((ext_vector_type(2)));
                                const f32x2 two = {2.f, 2.f};
                                const f32x2 lo = (f32x2){acc[ni][mi][4 * j], acc[ni][mi][4 * j + 1]} * two + (f32x2){bias[ni][j][0], bias[ni][j][1]};
                                const f32x2 hi = (f32x2){acc[ni][mi][4 * j + 2], acc[ni][mi][4 * j + 3]} * two + (f32x2){bias[ni][j][2], bias[ni][j][3]};
                                v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
                                // keep bit of channel 8 j + 4 h + i of the 32-block = bit 8 j + i of the pre-shifted hash
                                // (inline asm: the compiler rewrites the builtin into v_cmp + v_cndmask, twice the work)
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    int msk;
                                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(msk) : "v"(h32), "n"(8 * j + i));
                                    v[i] = __builtin_bit_cast(float, __builtin_bit_cast(int, v[i]) & msk);
                                }
                                T p1[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
                                pk[jj] = *(const u32x2*)p1;
                                continue;
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = acc[ni][mi][4 * j + i] + bias[ni][j][i];
                            if (epi & DBX_EPI_RELU) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                            }
                            if (epi & DBX_EPI_DROPHASH) {               // nn.Dropout(0.5): keep bit of (pixel, cout), kept values x 2
                                const unsigned kb = dbx_drop_bits4(a.drop_seed, mpix, (unsigned)(cw + ni * 32 + 8 * j + 4 * h) >> 2);
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[i] = ((kb >> i) & 1u) ? v[i] * 2.f : 0.f;
                            }
                            T p[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
                            pk[jj] = *(const u32x2*)p;
                        }
                        // lower half keeps its group 2jp and receives the upper half's; upper half receives the lower's 2jp+1
                        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                        u32x4 o = (u32x4){r0[0], r1[0], r0[1], r1[1]};
                        if constexpr (EPIK == 2) Mma32<T>::run(w2f[ni][jp], o, acc2);
                        if (ok && !(WS_DBG(t) & 2)) {
                            if (epi & DBX_EPI_GATE) o = gate_packed16(o, gt[ni][jp]);
                            *(u32x4*)(ypix + ni * 32 + 16 * jp) = o;
                        }
                    }
                }
                if (EPIK == 0 && (epi & DBX_EPI_GATE) && mi + PD < NF) gate_fetch(gring[mi % PD]);
                if constexpr (EPIK == 2)         // rows 0..3 (lower half) / 4..7 (upper half) of pixel l31: this wave's partial over its 64 channels
                    *(f32x4*)(red + ((wn * NFX + mi) * 64 + lane) * 16) = (f32x4){acc2[0], acc2[1], acc2[2], acc2[3]};
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
            if constexpr (EPIK == 2) {
                // the four waves' partials, summed in wave order; wave w finishes fragments w and w + 4 (its own lanes' pixels)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int mi = wn + 4 * k;
                    if (mi < NF) {
                        f32x4 s = *(const f32x4*)(red + ((0 * NFX + mi) * 64 + lane) * 16);
#pragma unroll
                        for (int w = 1; w < 4; ++w) s += *(const f32x4*)(red + ((w * NFX + mi) * 64 + lane) * 16);
                        if (st_ok[k]) *(f32x4*)(a.part + ((size_t)(cur.n0 >> 8) * a.M + st_pix[k]) * 8 + 4 * h) = s;
                    }
                }
            }
        };
        if (cur.nf == NFX) body(pipe::IC<NFX>{});
        else body(pipe::IC<NFX - 1>{});
        if (!more) break;
        item = nxt_item;
        cur = nxt;
    }
}

// tile schedule: units of WM fragments; tiles of 7..8 units, their number rounded up to fill whole rounds of CUs
static inline WsArgs ws_schedule(long long qtot, int hwp, int wm, int ntile_n, int ncu, int xpad, int nfx = 8) {
    WsArgs t;
    const long long units = (qtot + 32 * wm - 1) / (32 * wm);
    long long mt = (units + nfx - 1) / nfx;
    const long long wgs = mt * ntile_n;
    if (wgs > ncu) {
        const long long up = (wgs + ncu - 1) / ncu * ncu / ntile_n;     // tiles that fill the last round
        if (up > mt && units / up >= nfx - 1) mt = up;                  // (the kernel has (nfx-1)- and nfx-unit tiles)
    }
    t.mt = (int)mt; t.base = (int)(units / mt); t.extra = (int)(units % mt);
    t.items = (int)(mt * ntile_n); t.hwp = hwp; t.qtot = (int)qtot; t.xpad = xpad;
    t.dbg = 0;
#ifdef DBX_LAB
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("DBX_WS_DBG"); dbg = e ? atoi(e) : 0; } t.dbg = dbg; }
#endif
    return t;
}

template <typename T, int WM, int KS, int EPIK = 0, int NFX = 8>
static int launch_conv_ws(const ConvArgs& a, int n, int h, int xpad, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        constexpr int smem = 2 * ws::pieces(WM, KS, NFX) * 1024 + ws::dist(KS, NFX) * (256 / WM) * 32;
        constexpr int WGS = NFX == 8 ? 1 : 2;                            // workgroups per CU
        static_assert(smem <= 160 * 1024, "LDS budget");
        static DbxDevOnce attr_once; int attr_dev = 0;
        if (attr_once.pending(&attr_dev)) {
            DBX_HIP(hipFuncSetAttribute((const void*)conv3x3_ws_kernel<T, WM, KS, EPIK, NFX>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            attr_once.mark(attr_dev);
        }
        static int ncu = 0;
        if (!ncu) {
            int dev = 0;
            DBX_HIP(hipGetDevice(&dev));
            DBX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        }
        const WsArgs t = ws_schedule((long long)n * h * a.x_wp, h * a.x_wp, WM, a.ntile_n, ncu * WGS, xpad, NFX);
        const int grid = t.items < ncu * WGS ? t.items : ncu * WGS;
        hipLaunchKernelGGL((conv3x3_ws_kernel<T, WM, KS, EPIK, NFX>), dim3(grid), dim3(256), smem, s, a, t);
        DBX_LAUNCH_CHECK();
    }
    return DBX_OK;
}
