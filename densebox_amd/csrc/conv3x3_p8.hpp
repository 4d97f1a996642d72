// Round 5: implicit-GEMM convolution (3x3 / pad 1 on congruent frames, or 1x1) on the 8-phase MFMA core (mma8p.hpp).
// Included by conv_igemm.hip (needs ConvArgs, gate_packed16).  Forward and -- with flipped / transposed packed weights -- data gradient
// of the wide backbone layers (couts a multiple of 256, input channels a multiple of 64, 16-bit types), plain packed weights
// [cout][tap][cin] (dbx_pack_weight modes 0 / 1: no fragment-order image).
//
//   GEMM view: m = output pixel (COMPACT index p = (n H + y) W + x: no halo positions are computed), n = cout, K = (64-channel chunk, tap)
//   with the taps fastest (the nine taps of a chunk re-read the same pixel rows: L1 / L2 hits).  A K tile of the "A" matrix is 256
//   pixels x 128 bytes at frame offset tap(ky, kx) -- each lane's LDS-DMA source is its pixel's frame address (one 32-bit register per
//   piece, computed once per output tile: two divisions), the tap is a scalar offset: no im2col, no bounds checks (zero frame).
//   The "B" matrix is the packed weight as it lies in memory.
//
//   Persistent workgroups (one per CU) with BALANCED tile heights: a tile is 8 or 7 units of 32 pixels (p8::ktiles<.., MI1 = 4 / 3>), their
//   number rounded up to whole rounds of CUs (ws_schedule) -- 450 tiles of 256 pixels on 256 CUs would be two rounds at 88 %.  The load
//   stream never stops at a tile seam: the last two K tiles of a tile stage the first seven half-tiles of the NEXT one, the epilogue
//   (bias, ReLU | ReLU-gate, 16-byte NHWC stores; gate chunks of eight fragments requested ahead of the first store) runs between two
//   phases while those loads fly, and stores are younger than the loads the next counted wait covers.  The two wave groups meet at the
//   seam (p8::FL_TSYNC) so that their epilogues run side by side on each SIMD instead of one after the other, and part again behind it.
#pragma once
#include "mma8p.hpp"

// v if bit B of h is set, else +0 (v_bfe_i32 sign-extends the one-bit field into an AND mask; inline asm: the compiler rewrites the builtin
// form into v_cmp + v_cndmask)
template <int B> __device__ __forceinline__ float p8_keep(float v, unsigned h) {
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(h), "n"(B));
    return __builtin_bit_cast(float, __builtin_bit_cast(int, v) & m);
}

// ---- 2x2 max-pool fused into an epilogue (EPIK 3; DenseBox.py:191, :204: MaxPool2d(2, 2) behind conv2_2 / conv3_4).  The kernels enumerate
// their pixels WINDOW-MAJOR for it (p = 4 w + j: w = compact index of the pooled pixel, j = 2 dy + dx -- a lane's LDS-DMA source address is
// arbitrary, so the order costs nothing), which puts the four pixels of a window into the four lanes of a quad with the same eight
// channels (one 16-byte chunk each, rounded, >= +0 after the ReLU: the 16-bit patterns order like the numbers).  A 4 x 4 transpose over
// the quad (two cndmask / DPP / cndmask rounds) leaves lane j with the window's four values of channels 2 j, 2 j + 1; the window logic of
// conv3x3_c64_kernel's pooled epilogue runs ONCE per lane on packed unsigned halves (first maximum in (0,0), (0,1), (1,0), (1,1) order,
// as ATen; nibble = position | (max > 0) << 2: bitwise what dbx_maxpool2x2_idx takes from the stored map), and the quad stores 16 bytes
// of the pooled map and 4 bytes of nibbles.  ~45 VALU instructions per chunk.
__device__ __forceinline__ void p8_pool_chunk(const u32x4& chunk, int lane, bool ok, char* ppix, unsigned char* ipix) {
    const int j = lane & 3;
    unsigned o[4] = {chunk.x & 0x7fff7fffu, chunk.y & 0x7fff7fffu, chunk.z & 0x7fff7fffu, chunk.w & 0x7fff7fffu};     // (-0 -> +0: keeps the unsigned order)
#pragma unroll
    for (int e = 0; e < 4; e += 2) {                                    // lanes j ^ 1 exchange dwords (e, e + 1)
        const unsigned send = (j & 1) ? o[e] : o[e + 1];
        const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xF, 0xF, true);
        if (j & 1) o[e] = recv; else o[e + 1] = recv;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {                                       // lanes j ^ 2 exchange dwords (e, e + 2)
        const unsigned send = (j & 2) ? o[e] : o[e + 2];
        const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0x4E, 0xF, 0xF, true);
        if (j & 2) o[e] = recv; else o[e + 2] = recv;
    }
    auto pk_max = [](unsigned x, unsigned y) { unsigned d; asm("v_pk_max_u16 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
    auto pk_min = [](unsigned x, unsigned y) { unsigned d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
    auto pk_sub = [](unsigned x, unsigned y) { unsigned d; asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
    auto pk_mad = [](unsigned x, unsigned y, unsigned z) { unsigned d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z)); return d; };
    const unsigned one = 0x00010001u, two = 0x00020002u, four = 0x00040004u;
    const unsigned A = o[0], B = o[1], Cc = o[2], D = o[3];             // window positions (0,0), (0,1), (1,0), (1,1) of channels 2 j (low half), 2 j + 1
    const unsigned t0 = pk_max(A, B), t1 = pk_max(Cc, D), m = pk_max(t0, t1);
    const unsigned h0 = pk_min(t0 ^ A, one);                            // 1: the second column is strictly larger (row 0)
    const unsigned h1 = pk_min(t1 ^ Cc, one);                           //    ... (row 1)
    const unsigned r = pk_min(m ^ t0, one);                             // 1: row 1 is strictly larger
    const unsigned pos = pk_min(m, one);
    const unsigned b0 = pk_mad(r, pk_sub(h1, h0), h0);                  // r ? h1 : h0  (mod 2^16)
    const unsigned nib = pk_mad(pos, four, pk_mad(r, two, b0));
    if (ok) {
        *(unsigned*)(ppix + 4 * j) = m;                                  // channels 2 j, 2 j + 1 of the pooled pixel
        if (ipix) ipix[j] = (unsigned char)((nib & 0xfu) | ((nib >> 12) & 0xf0u));
    }
}

struct P8Args {
    int mt;              // pixel tiles
    int base, extra;     // tile t covers base + (t < extra) units of 32 pixels, starting at unit t base + min(t, extra)
    int items;           // mt * ntile_n
    int nkt;             // K tiles: taps * cin / 64 (even)
    int HW, W;           // output pixels per image / per row (EPIK 3: POOLED pixels per image / per row: the enumeration is window-major)
    float inv_HW, inv_W;
    unsigned long long* tstamp;  // lab builds (-DP8_TILE_STAMPS): per-tile stamp table, else null
    unsigned long long* stamp;   // lab builds (-DDBX_P8_STAMP): {first workgroup in, last workgroup out} of this launch in s_memrealtime ticks (10 ns), else null
};

#ifdef DBX_P8_STAMP
// Launch-boundary measurement WITHOUT a profiler (tools/gpu_p8_stamps.py): every launch of the 8-phase kernels stamps the 100-MHz real-time
// counter when its first workgroup starts and when its last one ends; dbx_lab_p8_stamps copies the table out.
static unsigned long long* g_p8_stamps = nullptr;
static int g_p8_stamp_n = 0;
constexpr int P8_STAMP_MAX = 4096;
static unsigned long long* p8_next_stamp() {
    if (!g_p8_stamps) {
        if (hipMalloc(&g_p8_stamps, sizeof(unsigned long long) * 2 * P8_STAMP_MAX) != hipSuccess) return nullptr;
        std::vector<unsigned long long> init(2 * P8_STAMP_MAX);
        for (int i = 0; i < P8_STAMP_MAX; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
        (void)hipMemcpy(g_p8_stamps, init.data(), sizeof(unsigned long long) * 2 * P8_STAMP_MAX, hipMemcpyHostToDevice);
    }
    if (g_p8_stamp_n >= P8_STAMP_MAX) return nullptr;
    return g_p8_stamps + 2 * (g_p8_stamp_n++);
}
extern "C" int dbx_lab_p8_stamps(unsigned long long* out, int max_n) {
    const int n = g_p8_stamp_n < max_n ? g_p8_stamp_n : max_n;
    if (n > 0 && hipMemcpy(out, g_p8_stamps, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}
// -DP8_TILE_STAMPS (with -DDBX_P8_STAMP): per-tile stamps of the heads forward (EPIK 1): workgroup 0 .. 255's wave 0 writes {tile top, K loop done,
// chunks done, reduction done} of each of its tiles; dbx_lab_p8_tile_stamps copies the table out (tools/gpu_p8_tile_stamps.py)
#ifdef P8_TILE_STAMPS
static unsigned long long* g_p8_tile_stamps = nullptr;
constexpr int P8_TILE_STAMP_MAX = 16384;
static unsigned long long* p8_tile_stamp_table() {
    if (!g_p8_tile_stamps) {
        if (hipMalloc(&g_p8_tile_stamps, sizeof(unsigned long long) * 4 * P8_TILE_STAMP_MAX) != hipSuccess) return nullptr;
        (void)hipMemset(g_p8_tile_stamps, 0, sizeof(unsigned long long) * 4 * P8_TILE_STAMP_MAX);
    }
    return g_p8_tile_stamps;
}
extern "C" int dbx_lab_p8_tile_stamps(unsigned long long* out, int max_items) {
    const int n = max_items < P8_TILE_STAMP_MAX ? max_items : P8_TILE_STAMP_MAX;
    if (!g_p8_tile_stamps || hipMemcpy(out, g_p8_tile_stamps, sizeof(unsigned long long) * 4 * n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}
#define P8_TSTAMP(t, item, k) do { if (EPIK == 1 && (t).tstamp && threadIdx.x == 0 && (item) < P8_TILE_STAMP_MAX) (t).tstamp[4 * (item) + (k)] = wall_clock64(); } while (0)
#else
#define P8_TSTAMP(t, item, k) do { } while (0)
#endif
#define P8_STAMP_IN(t) do { if ((t).stamp && threadIdx.x == 0) atomicMin((t).stamp, wall_clock64()); } while (0)
#define P8_STAMP_OUT(t) do { if ((t).stamp && threadIdx.x == 0) atomicMax((t).stamp + 1, wall_clock64()); } while (0)
#else
#define P8_STAMP_IN(t) do { } while (0)
#define P8_STAMP_OUT(t) do { } while (0)
#define P8_TSTAMP(t, item, k) do { } while (0)
#endif

// EPIK 0: bias and / or ReLU by the arguments' flags; EPIK 2: the ReLU gate of the data gradients, no bias (its own instantiation: the gate
// chunks of a whole tile are in flight together and want the registers the bias would hold).  EPIK 1 (1x1 only): the heads' forward -- bias + hash dropout
// (p = 0.5: kept values doubled, the keep bits of dbx_drop_hash32(seed, pixel, channel / 32): the masks the backward kernels regenerate) and,
// with a.w2f set, the heads' SECOND 1x1 convs (DenseBox.py:158-162: Conv1x1(768 -> 512) -> Dropout -> Conv1x1(512 -> k), k <= 8) on the tile
// while it is in registers: the 16-byte store chunk of a lane (eight consecutive hidden channels of one pixel, rounded and dropped --
// exactly what lands in the hidden map) IS the B operand of a v_mfma_f32_16x16x32 (K slots 8 g .. 8 g + 7 of column = pixel) whose A
// operand is the [k rows][those channels] slice of the head's second weight: a.w2f = plain [64 rows][all hidden channels] image, head i's
// rows 0 .. k_i - 1 in its 512 columns, zero elsewhere, so a lane's A fragment is ONE 16-byte load.  Two MFMAs per pixel fragment
// leave a wave's partial sums over its 64 channels in four registers; the four waves of a group (they share the pixels and split the
// channels) are summed in wave order through 16 KB of LDS behind the stream's buffers and the workgroup stores
// a.part[cout tile][pixel][8] (fp32) -- heads2_finish_kernel adds a head's two tiles and the bias (as for the ws kernel's EPIK 2).
template <typename T, int KS, int FLAGS = p8::FL_STAGGER | p8::FL_TSYNC, int EPIK = 0>
__global__ __launch_bounds__(512, 1) void conv3x3_p8_kernel(const ConvArgs a, const P8Args t) {
    constexpr int ES = sizeof(T);
    static_assert(ES == 2, "16-bit types");
    constexpr int NTAPS = KS * KS;
    static_assert(EPIK != 1 || KS == 1, "the heads epilogue belongs to the 1x1 instantiation");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    P8_STAMP_IN(t);
    const p8::Lanes<16> L = p8::lanes<16>(smem);
    const int lane = L.lane, wave = L.wave;
    const int pix_bytes = a.x_ld * ES;
    const int nkt = t.nkt;
    const int cin_bytes = a.cpt * 16;

    // ---- persistent schedule (as conv3x3_ws_kernel): workgroup g runs on XCD g % 8; with two or four cout tiles an XCD keeps to ONE
    // of them for the whole launch (its workgroups stream the same weights through its L2 in step), otherwise an XCD works on
    // neighbouring pixel tiles
    const int G = gridDim.x;
    const bool by_xcd = (G & 7) == 0 && (a.ntile_n == 2 || a.ntile_n == 4);
    int item;
    if (by_xcd) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        item = ((xcd / a.ntile_n) * (G >> 3) + j) * a.ntile_n + xcd % a.ntile_n;
    } else {
        item = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    }
    if (item >= t.items) { P8_STAMP_OUT(t); return; }

    struct Tile { int p0, nf, n0; };
    auto tile_of = [&](int it) {
        Tile r;
        const int tm = it / a.ntile_n, tn = it - tm * a.ntile_n;
        const int u0 = tm * t.base + (tm < t.extra ? tm : t.extra);
        r.nf = t.base + (tm < t.extra ? 1 : 0);
        r.p0 = u0 * 32;
        r.n0 = tn * 256;
        return r;
    };
    // pixel index -> (image, row, column): float estimate + one correction step (exact for p < 2^24)
    auto split0 = [&](int p, int& n, int& oy, int& ox) {
        n = (int)(((float)p + 0.5f) * t.inv_HW);
        int r = p - __mul24(n, t.HW);                              // (24-bit multiplies run at full rate; n, HW, W < 2^24: the host checks M)
        if (r < 0) { --n; r += t.HW; }
        if (r >= t.HW) { ++n; r -= t.HW; }
        oy = (int)(((float)r + 0.5f) * t.inv_W);
        ox = r - __mul24(oy, t.W);
        if (ox < 0) { --oy; ox += t.W; }
        if (ox >= t.W) { ++oy; ox -= t.W; }
    };
    // EPIK 3 (fused pooling): window-major enumeration p = 4 w + (2 dy + dx), w = compact index of the pooled pixel (t.HW / t.W are the pooled dims)
    auto split = [&](int p, int& n, int& oy, int& ox) {
        if constexpr (EPIK == 3) { split0(p >> 2, n, oy, ox); oy = 2 * oy + ((p >> 1) & 1); ox = 2 * ox + (p & 1); }
        else split0(p, n, oy, ox);
    };
    // the pixel 16 further on (the next fragment row of a wave): at most one row wrap when W >= 16, else divide again
    auto advance16 = [&](int p, int& n, int& oy, int& ox) {
        if (t.W >= 16) {
            ox += 16;
            if (ox >= t.W) { ox -= t.W; if (++oy * t.W >= t.HW) { oy = 0; ++n; } }
        } else split(p, n, oy, ox);
    };
    // tile row (half mh, row r of the half's 128 LDS rows) -> pixel index offset from p0; 7-unit tiles use rows 0..47 of each wave's 64 of half 1
    auto row_index = [&](int nf, int mh, int r) { return mh == 0 ? r : (nf == 8 ? 128 + r : 128 + (r >> 6) * 48 + ((r & 63) < 48 ? (r & 63) : 0)); };
    // ---- LDS-DMA source offsets of a tile's pixel rows: this lane loads LDS row 64 j + 8 wave + (lane >> 3) of each half-tile
    const unsigned chunk16 = p8::src_chunk(wave, lane) << 4;
    auto a_offsets = [&](const Tile& tl, unsigned (&vo)[2][2]) {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int p = tl.p0 + row_index(tl.nf, mh, j * 64 + wave * 8 + (lane >> 3));
                p = p < a.M ? p : a.M - 1;                          // rows past the end: any valid pixel (their results are dropped)
                int n, oy, ox;
                split(p, n, oy, ox);
                vo[mh][j] = (unsigned)((n * a.x_hp + oy + a.x_org) * a.x_wp + ox + a.x_org) * (unsigned)pix_bytes + chunk16;
            }
    };
    const unsigned voB = (unsigned)((wave * 8 + (lane >> 3)) * a.ktot_bytes) + chunk16;

    Tile cur = tile_of(item), nxt = cur;
    unsigned vcur[2][2], vnxt[2][2];
    a_offsets(cur, vcur);
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int j = 0; j < 2; ++j) vnxt[mh][j] = vcur[mh][j];

    // K tile kt of the stream: kt >= nkt is K tile kt - nkt of the NEXT output tile
    auto k_split = [&](int kt, bool& nx, int& chunk, int& tap) {
        nx = kt >= nkt;
        const int k = nx ? kt - nkt : kt;
        if constexpr (NTAPS == 9) { chunk = (k * 7282) >> 16; tap = k - chunk * 9; }       // k / 9 for k < 7000
        else { chunk = k; tap = 0; }
    };
    auto stA = [&](int kt, int mh, unsigned dst) {
        bool nx; int chunk, tap;
        k_split(kt, nx, chunk, tap);
#if defined(P8_ABL) && (P8_ABL & 8)
        // lab, timing only (results are garbage): the A half-tiles of taps 1..8 are not staged -- an UPPER bound on what a halo-tile A operand
        // (one staging per 64-channel chunk for all nine taps) could take out of the K loop.  The counted waits stay safe: fewer loads in flight.
        if (NTAPS == 9 && tap != 0) return;
#endif
        int toff = 0;
        if constexpr (NTAPS == 9) { const int ky = (tap * 11) >> 5, kx = tap - 3 * ky; toff = (ky * a.x_wp + kx) * pix_bytes; }
        const char* b = a.x + toff + chunk * 128;
        p8::glds(b, nx ? vnxt[mh][0] : vcur[mh][0], dst);
        p8::glds(b, nx ? vnxt[mh][1] : vcur[mh][1], dst + 8192);
    };
    auto stB = [&](int kt, int nh, unsigned dst) {
        bool nx; int chunk, tap;
        k_split(kt, nx, chunk, tap);
        const char* b = a.w + (size_t)((nx ? nxt.n0 : cur.n0) + nh * 128) * a.ktot_bytes + tap * cin_bytes + chunk * 128;
        p8::glds(b, voB, dst);
        p8::glds(b + (size_t)64 * a.ktot_bytes, voB, dst + 8192);
    };

    // ---- bias of this lane's couts: n = n0 + 128 nh + 32 wc + 16 ni + 4 (lane >> 4) + 0..3; reloaded when the cout tile changes
    const int epi = a.epi;
    const int g4 = lane >> 4;
    f32x4 bias[2][2];
    int bias_n0 = -1;

    if constexpr (FLAGS & p8::FL_P4) { p8::prologue4(L, stA, stB); p8::start4<FLAGS>(L); }
    else { p8::prologue<16>(L, stA, stB); p8::start<FLAGS>(L); }
    p8::Acc<16> acc;
    for (;;) {
        const int nxt_item = item + G;
        const bool more = nxt_item < t.items;
        if (more) { nxt = tile_of(nxt_item); a_offsets(nxt, vnxt); }          // last tile: the stream re-fetches its own start (never read)
        P8_TSTAMP(t, item, 0);
        p8::zero<16>(acc);
        if (EPIK != 2 && cur.n0 != bias_n0) {
            bias_n0 = cur.n0;
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    bias[nh][ni] = (EPIK == 1 || (epi & DBX_EPI_BIAS)) ? *(const f32x4*)(a.bias + cur.n0 + nh * 128 + L.wc * 32 + ni * 16 + 4 * g4) * (EPIK == 1 ? 2.f : 1.f)
                                                                       : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        auto body = [&](auto MI1_) {
            constexpr int MI1 = decltype(MI1_)::value;
            p8::tile_begin<FLAGS>(L);
            if constexpr (FLAGS & p8::FL_P4) p8::ktiles4<T, FLAGS, MI1>(acc, L, nkt, stA, stB);
#ifdef P8_KT_STAMPS
            else p8::ktiles<T, 16, FLAGS, MI1>(acc, L, nkt, stA, stB, [&](int kt) {
                if (EPIK == 1 && t.tstamp && threadIdx.x == 0 && item < 1024 && kt < 16) t.tstamp[4 * P8_TILE_STAMP_MAX / 2 + item * 8 + (kt >> 1)] = wall_clock64(); });
#else
            else p8::ktiles<T, 16, FLAGS, MI1>(acc, L, nkt, stA, stB);
#endif
            p8::tile_end<FLAGS>(L);
            P8_TSTAMP(t, item, 1);
            if constexpr (EPIK == 1) {
#if defined(P8_ABL) && (P8_ABL & 4)
                if (a.drop_seed != 0x7fffffffu) return;                                                   // lab: no epilogue at all (timing only)
#endif
                // ---- heads forward: 2 (acc + bias) behind the keep bits, rounded, stored; the second convs on the stored chunks
                const int pend = cur.p0 + cur.nf * 32 < a.M ? cur.p0 + cur.nf * 32 : a.M;
                T* const ybase = (T*)a.y + cur.n0;
                const int l15 = lane & 15;
                const int cl = L.wc * 32 + (g4 & 1) * 16 + (g4 >> 1) * 8;            // this lane's eight channels inside the 128-channel half
                u32x4 w2f[2];
                if (a.w2f) {
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
                        w2f[nh] = *(const u32x4*)(a.w2f + ((size_t)l15 * (size_t)(a.ntile_n * 256) + cur.n0 + nh * 128 + cl) * ES);
                }
                char* const red = smem + p8::LDS_BYTES + L.wr * 16384;
                const f32x4 two4 = {2.f, 2.f, 2.f, 2.f};
#pragma unroll
                for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        if (mh == 1 && mi >= MI1) continue;
                        const int p = cur.p0 + row_index(cur.nf, mh, L.wr * 64 + mi * 16 + l15);
                        const bool ok = p < pend;
                        const int pp = ok ? p : cur.p0;
                        int n, oy, ox;                                      // (divided out per row: carrying the coordinates across the rows spilled 10 registers here)
                        split(pp, n, oy, ox);
                        const unsigned yo = (__umul24(__umul24(n, a.y_hp) + oy + a.y_pad, a.y_wp) + ox + a.y_pad) * (unsigned)a.y_ld;      // (< 4 G elements: host check)
                        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int nh = 0; nh < 2; ++nh) {
                            // one hash covers the wave's 32 channels of this pixel; lane row g holds bits 4 g .. of each 16-channel fragment
#if defined(P8_ABL) && (P8_ABL & 2)
                            const unsigned hs = 0xffffffffu;                                                  // lab: no hash (timing only)
#else
                            const unsigned hs = dbx_drop_hash32(a.drop_seed, (unsigned)pp, (unsigned)(cur.n0 + nh * 128 + L.wc * 32) >> 5) >> (4 * g4);
#endif
                            f32x4 v[2];
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni) {
                                v[ni] = __builtin_elementwise_fma(acc.v[mh][nh][mi][ni], two4, bias[nh][ni]);      // (one v_pk_fma_f32 per pair: the compiler's own choice is x + x, then + bias)
                            }
                            // a dropped element is cleared by ANDing with the sign-extended one-bit field of the hash (p8_keep<bit>: the bit
                            // position is an immediate of v_bfe_i32)
                            v[0].x = p8_keep<0>(v[0].x, hs); v[0].y = p8_keep<1>(v[0].y, hs); v[0].z = p8_keep<2>(v[0].z, hs); v[0].w = p8_keep<3>(v[0].w, hs);
                            v[1].x = p8_keep<16>(v[1].x, hs); v[1].y = p8_keep<17>(v[1].y, hs); v[1].z = p8_keep<18>(v[1].z, hs); v[1].w = p8_keep<19>(v[1].w, hs);
                            const u32x4 o = pair_exchange<T>(v[0], v[1]);
                            if (a.w2f) p8::Mma16<T>::run(w2f[nh], o, acc2);
#if defined(P8_ABL) && (P8_ABL & 1)
                            if (ok && o.x == 0x12345678u) *(u32x4*)(ybase + yo + nh * 128 + cl) = o;        // lab: no stores (timing only)
#elif defined(P8_HID_NT)
                            if (ok) __builtin_nontemporal_store(o, (u32x4*)(ybase + yo + nh * 128 + cl));
#else
                            if (ok) *(u32x4*)(ybase + yo + nh * 128 + cl) = o;
#endif
                        }
                        // rows 4 g .. 4 g + 3 of the second convs' outputs for pixel l15: only g < 2 (k <= 8) carries anything
                        if (a.w2f && g4 < 2) *(f32x4*)(red + ((L.wc * 8 + mh * 4 + mi) * 32 + (lane & 31)) * 16) = acc2;
                    }
                }
                P8_TSTAMP(t, item, 2);
                if (a.w2f) {
                    // the group's four waves, summed in wave order; wave wc finishes fragments 2 wc and 2 wc + 1.  The barrier is the
                    // whole workgroup's (every wave runs the same barrier sequence one phase apart: the stagger is unchanged)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    p8::barrier();
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int f = 2 * L.wc + k, mh = f >> 2, mi = f & 3;
                        if (mh == 1 && mi >= MI1) continue;
                        if (lane < 32) {
                            f32x4 sum = *(const f32x4*)(red + ((0 * 8 + f) * 32 + lane) * 16);
#pragma unroll
                            for (int w = 1; w < 4; ++w) sum += *(const f32x4*)(red + ((w * 8 + f) * 32 + lane) * 16);
                            const int p = cur.p0 + row_index(cur.nf, mh, L.wr * 64 + mi * 16 + l15);
                            if (p < pend) *(f32x4*)(a.part + ((size_t)(cur.n0 >> 8) * a.M + p) * 8 + 4 * (lane >> 4)) = sum;
                        }
                    }
                }
                P8_TSTAMP(t, item, 3);
                return;
            }
            if constexpr (EPIK == 3) {
                // ---- bias + ReLU, the full map unless the caller wants the pooled one only, and the 2x2 max-pool of the tile (p8_pool_chunk)
                const int pend = cur.p0 + cur.nf * 32 < a.M ? cur.p0 + cur.nf * 32 : a.M;
                T* const ybase = (T*)a.y + cur.n0;
                const int nc = L.wc * 32 + pair_cout_off(g4, 0);
                const bool full = !(a.epi2 & EPI2_POOL_ONLY);
#pragma unroll
                for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        if (mh == 1 && mi >= MI1) continue;
                        const int p = cur.p0 + row_index(cur.nf, mh, L.wr * 64 + mi * 16 + (lane & 15));
                        const bool ok = p < pend;
                        const int pp = ok ? p : cur.p0;
                        int n, py, px;
                        split0(pp >> 2, n, py, px);
                        const int oy = 2 * py + ((pp >> 1) & 1), ox = 2 * px + (pp & 1);
                        // (element offsets in 32 bits: both maps are < 4 G elements, checked by the host; 24-bit multiplies run at full rate)
                        const unsigned yo = (__umul24(__umul24(n, a.y_hp) + oy + a.y_pad, a.y_wp) + ox + a.y_pad) * (unsigned)a.y_ld;
                        const unsigned po = (__umul24(__umul24(n, a.y2_hp) + py + a.y2_pad, a.y2_wp) + px + a.y2_pad) * (unsigned)a.y2_ld;
#pragma unroll
                        for (int nh = 0; nh < 2; ++nh) {
                            f32x4 v0 = acc.v[mh][nh][mi][0] + bias[nh][0], v1 = acc.v[mh][nh][mi][1] + bias[nh][1];
                            v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
                            v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
                            const u32x4 o = pair_exchange<T>(v0, v1);
                            const int ch = cur.n0 + nh * 128 + nc;           // first of this lane's eight couts
                            if (ok && full) *(u32x4*)(ybase + yo + nh * 128 + nc) = o;
                            p8_pool_chunk(o, lane, ok, (char*)((T*)a.y2 + po + ch),
                                          a.pool_idx ? a.pool_idx + ((unsigned)(pp >> 2) * (unsigned)(a.cout_valid >> 1) + (unsigned)(ch >> 1)) : nullptr);
                        }
                    }
                return;
            }
            // ---- epilogue (compiler-scheduled; no LDS).  Row m of half mh -> output pixel; chunk = eight consecutive couts of it
            const int pend = cur.p0 + cur.nf * 32 < a.M ? cur.p0 + cur.nf * 32 : a.M;
            T* const ybase = (T*)a.y + cur.n0;
            const T* const gbase = (const T*)a.gate + cur.n0;
            const int nc = L.wc * 32 + pair_cout_off(g4, 0);
            unsigned yo[2][4];                                          // element offsets (the tensors are < 4 GB: checked by the host)
            bool ok[2][4];
            u32x4 gt[2][4][2];
            // addresses of the tile's eight pixel rows, and ALL its gate chunks requested before the first store goes out: a load behind a
            // store is issued behind it, and the epilogue has nothing else to hide a memory latency under (the operand fragments' 64
            // registers are free here)
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                int n, oy, ox;                                              // divided out once per half, then stepped 16 pixels per fragment row
                size_t go0 = 0;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    if (mh == 1 && mi >= MI1) continue;
                    const int p = cur.p0 + row_index(cur.nf, mh, L.wr * 64 + mi * 16 + (lane & 15));
                    ok[mh][mi] = p < pend;
                    if (mi == 0) split(p < a.M ? p : a.M - 1, n, oy, ox); else advance16(p, n, oy, ox);
                    yo[mh][mi] = (__umul24(__umul24(n, a.y_hp) + oy + a.y_pad, a.y_wp) + ox + a.y_pad) * (unsigned)a.y_ld;
                    if constexpr (EPIK == 2) {
                        size_t go = (size_t)(__umul24(__umul24(n, a.g_hp) + oy + a.g_pad, a.g_wp) + ox + a.g_pad) * (size_t)a.g_ld;
                        if (mi == 0) go0 = go;
                        if (!ok[mh][mi]) go = go0;                          // rows past the tile's end: any address that exists
#pragma unroll
                        for (int nh = 0; nh < 2; ++nh) gt[mh][mi][nh] = *(const u32x4*)(gbase + go + nh * 128 + nc);
                    }
                }
            }
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    if (mh == 1 && mi >= MI1) continue;
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh) {
                        f32x4 v0 = acc.v[mh][nh][mi][0], v1 = acc.v[mh][nh][mi][1];
                        if constexpr (EPIK == 0) { v0 += bias[nh][0]; v1 += bias[nh][1]; }
                        if (EPIK == 0 && (epi & DBX_EPI_RELU)) {
                            v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
                            v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
                        }
                        u32x4 o = pair_exchange<T>(v0, v1);                  // all lanes: eight consecutive couts at pair_cout_off
                        if constexpr (EPIK == 2) o = gate_packed16(o, gt[mh][mi][nh]);
                        if (ok[mh][mi]) *(u32x4*)(ybase + (size_t)yo[mh][mi] + nh * 128 + nc) = o;
                    }
                }
        };
        if (cur.nf == 8) body(pipe::IC<4>{});
        else body(pipe::IC<3>{});
        if (!more) break;
        item = nxt_item;
        cur = nxt;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int j = 0; j < 2; ++j) vcur[mh][j] = vnxt[mh][j];
    }
    p8::finish<FLAGS>(L);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P8_STAMP_OUT(t);
}

// tile schedule: units of 32 pixels, tiles of 7 or 8 units, their number rounded up to fill whole rounds of CUs
static inline bool p8_schedule(long long M, int ntile_n, int ncu, P8Args& t) {
    const long long units = (M + 31) / 32;
    long long mt = (units + 7) / 8;
    const long long wgs = mt * ntile_n;
    if (wgs > ncu) {
        const long long up = (wgs + ncu - 1) / ncu * ncu / ntile_n;     // tiles that fill the last round
        if (up > mt && units / up >= 7) mt = up;
    }
    t.mt = (int)mt; t.base = (int)(units / mt); t.extra = (int)(units % mt);
    t.items = (int)(mt * ntile_n);
    return t.base == 8 ? t.extra == 0 : t.base == 7;                    // the kernel has 7- and 8-unit tiles
}

// Default phase program: the 3x3 layers run TWO phases per K tile (clusters of 32 MFMAs, p8::ktiles4: half the barriers -- conv4_2 forward
// 187 -> 182 us = 1496 TFLOP/s, the gated data gradients -4 %, step -0.7 % over three alternating pairs), the 1x1 GEMMs keep four phases of 16
// (with two phases the heads forward measured 848 -> 864 us)
template <int KS> constexpr int p8_default_flags() { return p8::FL_STAGGER | p8::FL_TSYNC | (KS == 3 ? p8::FL_P4 : 0); }
template <typename T, int KS, int EPIK = 0, int FLAGS = p8_default_flags<KS>()>
static int launch_conv_p8(const ConvArgs& a, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
#ifdef DBX_P8_AB
        // same-box A/B (tools/build_variant.sh -DDBX_P8_AB): DBX_P8_P4=0 / 1 forces four phases of 16 MFMAs / two phases of 32 per K tile
        if constexpr (FLAGS == p8_default_flags<KS>()) {
            static int p4 = -2;
            if (p4 == -2) { const char* e = getenv("DBX_P8_P4"); p4 = e ? atoi(e) : -1; }
            if (p4 == 0 && (FLAGS & p8::FL_P4)) return launch_conv_p8<T, KS, EPIK, p8::FL_STAGGER | p8::FL_TSYNC>(a, s);
            if (p4 == 1 && !(FLAGS & p8::FL_P4)) return launch_conv_p8<T, KS, EPIK, p8::FL_STAGGER | p8::FL_TSYNC | p8::FL_P4>(a, s);
        }
#endif
        constexpr int LDS = p8::LDS_BYTES + (EPIK == 1 ? 32768 : 0);        // + the two groups' 16-KB reduction areas of the fused second convs
        static DbxDevOnce attr_once; int attr_dev = 0;
        if (attr_once.pending(&attr_dev)) {
            DBX_HIP(hipFuncSetAttribute((const void*)conv3x3_p8_kernel<T, KS, FLAGS, EPIK>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            attr_once.mark(attr_dev);
        }
        static int ncu = 0;
        if (!ncu) {
            int dev = 0;
            DBX_HIP(hipGetDevice(&dev));
            DBX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        }
        P8Args t;
        DBX_REQUIRE(p8_schedule(a.M, a.ntile_n, ncu, t), "conv p8: no 7/8-unit tile schedule for %d pixels", a.M);
        t.nkt = KS * KS * (a.cpt / 8);
        t.HW = a.HoWo; t.W = a.Wo;
        if (EPIK == 3) { t.HW = a.HoWo / 4; t.W = a.Wo / 2; }          // window-major enumeration: the pooled map's dims
        t.inv_HW = 1.0f / (float)t.HW; t.inv_W = 1.0f / (float)t.W;
        t.tstamp = nullptr;
#ifdef P8_TILE_STAMPS
        if (EPIK == 1) t.tstamp = p8_tile_stamp_table();
#endif
        t.stamp = nullptr;
#ifdef DBX_P8_STAMP
        t.stamp = p8_next_stamp();
#endif
        int grid = t.items < ncu ? t.items : ncu;
#ifdef P8_TILE_STAMPS
        if (const char* e = getenv("DBX_P8_MAXWG")) { const int g = atoi(e); if (g > 0 && g < grid) grid = g; }      // lab: fewer CUs at work (is a stall per-CU or chip-wide?)
#endif
        hipLaunchKernelGGL((conv3x3_p8_kernel<T, KS, FLAGS, EPIK>), dim3(grid), dim3(512), LDS, s, a, t);
        DBX_LAUNCH_CHECK();
    }
    return DBX_OK;
}

// ------------------------------------------------------------------------------------------------ 128-cout layers: 512-pixel x 128-cout tiles
// The same kernel on the p8w core (mma8p.hpp): a tile is 16, 15 or 14 units of 32 pixels (m halves 2 and 3 of a wave with four or three
// fragments), one cout tile of 128; everything else (compact pixels, K order, continuous stream, seam-synchronised epilogues) as above.
// conv2_2 (128 -> 128 at 120 x 120) forward and data gradient, conv3_1's data gradient (256 -> 128 at 60 x 60).
template <typename T, int KS, int EPIK = 0>
__global__ __launch_bounds__(512, 1) void conv3x3_p8w_kernel(const ConvArgs a, const P8Args t) {
    constexpr int ES = sizeof(T);
    static_assert(ES == 2 && (EPIK == 0 || EPIK == 2 || EPIK == 3), "16-bit types; bias / ReLU, gate, or bias + ReLU + pooling epilogue");
    constexpr int NTAPS = KS * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    P8_STAMP_IN(t);
    const p8w::Lanes L = p8w::lanes(smem);
    const int lane = L.lane, wave = L.wave;
    const int pix_bytes = a.x_ld * ES;
    const int nkt = t.nkt;
    const int cin_bytes = a.cpt * 16;
    const int G = gridDim.x;
    int item = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;       // an XCD works on neighbouring pixel tiles
    if (item >= t.items) return;

    struct Tile { int p0, nf, n0; };
    auto tile_of = [&](int it) {
        Tile r;
        const int tm = it / a.ntile_n, tn = it - tm * a.ntile_n;
        const int u0 = tm * t.base + (tm < t.extra ? tm : t.extra);
        r.nf = t.base + (tm < t.extra ? 1 : 0);
        r.p0 = u0 * 32;
        r.n0 = tn * 128;
        return r;
    };
    auto split0 = [&](int p, int& n, int& oy, int& ox) {
        n = (int)(((float)p + 0.5f) * t.inv_HW);
        int r = p - __mul24(n, t.HW);                              // (24-bit multiplies run at full rate; n, HW, W < 2^24: the host checks M)
        if (r < 0) { --n; r += t.HW; }
        if (r >= t.HW) { ++n; r -= t.HW; }
        oy = (int)(((float)r + 0.5f) * t.inv_W);
        ox = r - __mul24(oy, t.W);
        if (ox < 0) { --oy; ox += t.W; }
        if (ox >= t.W) { ++oy; ox -= t.W; }
    };
    auto split = [&](int p, int& n, int& oy, int& ox) {              // EPIK 3: window-major enumeration (see p8_pool_chunk)
        if constexpr (EPIK == 3) { split0(p >> 2, n, oy, ox); oy = 2 * oy + ((p >> 1) & 1); ox = 2 * ox + (p & 1); }
        else split0(p, n, oy, ox);
    };
    // tile row (half mh, LDS row r of its 128) -> pixel offset from p0: a half with three fragments per wave uses rows 0..47 of each wave's 64
    auto row_index = [&](int nf, int mh, int r) {
        const int h2 = nf >= 15 ? 128 : 96, h3 = nf == 16 ? 128 : 96;              // pixel rows of halves 2 and 3
        if (mh < 2) return mh * 128 + r;
        const int rows = mh == 2 ? h2 : h3, off = mh == 2 ? 256 : 256 + h2;
        const int w = r & 63;
        return off + (r >> 6) * (rows >> 1) + (w < (rows >> 1) ? w : 0);
    };
    const unsigned chunk16 = p8::src_chunk(wave, lane) << 4;
    auto a_offsets = [&](const Tile& tl, unsigned (&vo)[4][2]) {
#pragma unroll
        for (int mh = 0; mh < 4; ++mh)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int p = tl.p0 + row_index(tl.nf, mh, j * 64 + wave * 8 + (lane >> 3));
                p = p < a.M ? p : a.M - 1;
                int n, oy, ox;
                split(p, n, oy, ox);
                vo[mh][j] = (unsigned)((n * a.x_hp + oy + a.x_org) * a.x_wp + ox + a.x_org) * (unsigned)pix_bytes + chunk16;
            }
    };
    const unsigned voB = (unsigned)((wave * 8 + (lane >> 3)) * a.ktot_bytes) + chunk16;
    Tile cur = tile_of(item), nxt = cur;
    unsigned vcur[4][2], vnxt[4][2];
    a_offsets(cur, vcur);
#pragma unroll
    for (int mh = 0; mh < 4; ++mh)
#pragma unroll
        for (int j = 0; j < 2; ++j) vnxt[mh][j] = vcur[mh][j];
    auto k_split = [&](int kt, bool& nx, int& chunk, int& tap) {
        nx = kt >= nkt;
        const int k = nx ? kt - nkt : kt;
        if constexpr (NTAPS == 9) { chunk = (k * 7282) >> 16; tap = k - chunk * 9; }
        else { chunk = k; tap = 0; }
    };
    auto stA = [&](int kt, int mh, unsigned dst) {
        bool nx; int chunk, tap;
        k_split(kt, nx, chunk, tap);
#if defined(P8_ABL) && (P8_ABL & 8)
        // lab, timing only (results are garbage): the A half-tiles of taps 1..8 are not staged -- an UPPER bound on what a halo-tile A operand
        // (one staging per 64-channel chunk for all nine taps) could take out of the K loop.  The counted waits stay safe: fewer loads in flight.
        if (NTAPS == 9 && tap != 0) return;
#endif
        int toff = 0;
        if constexpr (NTAPS == 9) { const int ky = (tap * 11) >> 5, kx = tap - 3 * ky; toff = (ky * a.x_wp + kx) * pix_bytes; }
        const char* b = a.x + toff + chunk * 128;
        p8::glds(b, nx ? vnxt[mh][0] : vcur[mh][0], dst);
        p8::glds(b, nx ? vnxt[mh][1] : vcur[mh][1], dst + 8192);
    };
    auto stB = [&](int kt, unsigned dst) {
        bool nx; int chunk, tap;
        k_split(kt, nx, chunk, tap);
        const char* b = a.w + (size_t)(nx ? nxt.n0 : cur.n0) * a.ktot_bytes + tap * cin_bytes + chunk * 128;
        p8::glds(b, voB, dst);
        p8::glds(b + (size_t)64 * a.ktot_bytes, voB, dst + 8192);
    };
    const int epi = a.epi;
    const int g4 = lane >> 4;
    f32x4 bias[2];
    int bias_n0 = -1;

    p8w::prologue(L, stA, stB);
    p8w::start();
    p8w::Acc acc;
    for (;;) {
        const int nxt_item = item + G;
        const bool more = nxt_item < t.items;
        if (more) { nxt = tile_of(nxt_item); a_offsets(nxt, vnxt); }
        p8w::zero(acc);
        if (EPIK != 2 && cur.n0 != bias_n0) {
            bias_n0 = cur.n0;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                bias[ni] = (epi & DBX_EPI_BIAS) ? *(const f32x4*)(a.bias + cur.n0 + L.wc * 32 + ni * 16 + 4 * g4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        auto body = [&](auto MI2_, auto MI3_) {
            constexpr int MI2 = decltype(MI2_)::value, MI3 = decltype(MI3_)::value;
            p8w::tile_begin(L);
            p8w::ktiles<T, MI2, MI3>(acc, L, nkt, stA, stB);
            p8w::tile_end(L);
            const int pend = cur.p0 + cur.nf * 32 < a.M ? cur.p0 + cur.nf * 32 : a.M;
            T* const ybase = (T*)a.y + cur.n0;
            const T* const gbase = (const T*)a.gate + cur.n0;
            const int nc = L.wc * 32 + pair_cout_off(g4, 0);
            if constexpr (EPIK == 3) {
                // ---- bias + ReLU, the full map unless the caller wants the pooled one only, and the 2x2 max-pool of the tile (p8_pool_chunk)
                const bool full = !(a.epi2 & EPI2_POOL_ONLY);
                const int ch = cur.n0 + nc;
#pragma unroll
                for (int mh = 0; mh < 4; ++mh)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        if ((mh == 2 && mi >= MI2) || (mh == 3 && mi >= MI3)) continue;
                        const int p = cur.p0 + row_index(cur.nf, mh, L.wr * 64 + mi * 16 + (lane & 15));
                        const bool ok = p < pend;
                        const int pp = ok ? p : cur.p0;
                        int n, py, px;
                        split0(pp >> 2, n, py, px);
                        const int oy = 2 * py + ((pp >> 1) & 1), ox = 2 * px + (pp & 1);
                        // (element offsets in 32 bits: both maps are < 4 G elements, checked by the host; 24-bit multiplies run at full rate)
                        const unsigned yo = (__umul24(__umul24(n, a.y_hp) + oy + a.y_pad, a.y_wp) + ox + a.y_pad) * (unsigned)a.y_ld;
                        const unsigned po = (__umul24(__umul24(n, a.y2_hp) + py + a.y2_pad, a.y2_wp) + px + a.y2_pad) * (unsigned)a.y2_ld;
                        f32x4 v0 = acc.v[mh][mi][0] + bias[0], v1 = acc.v[mh][mi][1] + bias[1];
                        v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
                        v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
                        const u32x4 o = pair_exchange<T>(v0, v1);
                        if (ok && full) *(u32x4*)(ybase + yo + nc) = o;
                        p8_pool_chunk(o, lane, ok, (char*)((T*)a.y2 + po + ch),
                                      a.pool_idx ? a.pool_idx + ((unsigned)(pp >> 2) * (unsigned)(a.cout_valid >> 1) + (unsigned)(ch >> 1)) : nullptr);
                    }
                return;
            }
            // two batches of eight pixel rows: addresses + the batch's gate chunks first, then its stores
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                unsigned yo[2][4];
                bool ok[2][4];
                u32x4 gt[2][4];
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        const int mh = 2 * hb + m2;
                        if ((mh == 2 && mi >= MI2) || (mh == 3 && mi >= MI3)) continue;
                        const int p = cur.p0 + row_index(cur.nf, mh, L.wr * 64 + mi * 16 + (lane & 15));
                        ok[m2][mi] = p < pend;
                        int n, oy, ox;
                        split(ok[m2][mi] ? p : cur.p0, n, oy, ox);
                        yo[m2][mi] = (__umul24(__umul24(n, a.y_hp) + oy + a.y_pad, a.y_wp) + ox + a.y_pad) * (unsigned)a.y_ld;
                        if constexpr (EPIK == 2) {
                            const size_t go = (size_t)(__umul24(__umul24(n, a.g_hp) + oy + a.g_pad, a.g_wp) + ox + a.g_pad) * (size_t)a.g_ld;
                            gt[m2][mi] = *(const u32x4*)(gbase + go + nc);
                        }
                    }
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        const int mh = 2 * hb + m2;
                        if ((mh == 2 && mi >= MI2) || (mh == 3 && mi >= MI3)) continue;
                        f32x4 v0 = acc.v[mh][mi][0], v1 = acc.v[mh][mi][1];
                        if constexpr (EPIK == 0) { v0 += bias[0]; v1 += bias[1]; }
                        if (EPIK == 0 && (epi & DBX_EPI_RELU)) {
                            v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
                            v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
                        }
                        u32x4 o = pair_exchange<T>(v0, v1);
                        if constexpr (EPIK == 2) o = gate_packed16(o, gt[m2][mi]);
                        if (ok[m2][mi]) *(u32x4*)(ybase + (size_t)yo[m2][mi] + nc) = o;
                    }
            }
        };
        if (cur.nf == 16) body(pipe::IC<4>{}, pipe::IC<4>{});
        else if (cur.nf == 15) body(pipe::IC<4>{}, pipe::IC<3>{});
        else body(pipe::IC<3>{}, pipe::IC<3>{});
        if (!more) break;
        item = nxt_item;
        cur = nxt;
#pragma unroll
        for (int mh = 0; mh < 4; ++mh)
#pragma unroll
            for (int j = 0; j < 2; ++j) vcur[mh][j] = vnxt[mh][j];
    }
    p8w::finish();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P8_STAMP_OUT(t);
}

// tiles of 14 .. 16 units of 32 pixels, their number rounded up to whole rounds of CUs
static inline bool p8w_schedule(long long M, int ntile_n, int ncu, P8Args& t) {
    const long long units = (M + 31) / 32;
    long long mt = (units + 15) / 16;
    const long long wgs = mt * ntile_n;
    if (wgs > ncu) {
        const long long up = (wgs + ncu - 1) / ncu * ncu / ntile_n;
        if (up > mt && units / up >= 14) mt = up;
    }
    t.mt = (int)mt; t.base = (int)(units / mt); t.extra = (int)(units % mt);
    t.items = (int)(mt * ntile_n);
    return t.base == 16 ? t.extra == 0 : (t.base == 14 || t.base == 15);
}

template <typename T, int KS, int EPIK = 0>
static int launch_conv_p8w(const ConvArgs& a, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        static DbxDevOnce attr_once; int attr_dev = 0;
        if (attr_once.pending(&attr_dev)) {
            DBX_HIP(hipFuncSetAttribute((const void*)conv3x3_p8w_kernel<T, KS, EPIK>, hipFuncAttributeMaxDynamicSharedMemorySize, p8w::LDS_BYTES));
            attr_once.mark(attr_dev);
        }
        static int ncu = 0;
        if (!ncu) {
            int dev = 0;
            DBX_HIP(hipGetDevice(&dev));
            DBX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        }
        P8Args t;
        DBX_REQUIRE(p8w_schedule(a.M, a.ntile_n, ncu, t), "conv p8w: no 14..16-unit tile schedule for %d pixels", a.M);
        t.nkt = KS * KS * (a.cpt / 8);
        t.HW = a.HoWo; t.W = a.Wo;
        if (EPIK == 3) { t.HW = a.HoWo / 4; t.W = a.Wo / 2; }          // window-major enumeration: the pooled map's dims
        t.inv_HW = 1.0f / (float)t.HW; t.inv_W = 1.0f / (float)t.W;
        t.tstamp = nullptr;
        t.stamp = nullptr;
#ifdef DBX_P8_STAMP
        t.stamp = p8_next_stamp();
#endif
        const int grid = t.items < ncu ? t.items : ncu;
        hipLaunchKernelGGL((conv3x3_p8w_kernel<T, KS, EPIK>), dim3(grid), dim3(512), p8w::LDS_BYTES, s, a, t);
        DBX_LAUNCH_CHECK();
    }
    return DBX_OK;
}
