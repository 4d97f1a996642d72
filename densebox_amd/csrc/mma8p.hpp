// 256 x 256 x 64 "8-phase" MFMA core (round 5): the K loop the 3x3 stack's implicit-GEMM kernels and tools/probe_gemm_8phase.hip share.
//
// Why (profiles/r04_band_stage_stamps.txt): the 8-wave kernels of rounds 1-4 have ONE barrier per 96-MFMA stage; the two waves of a SIMD
// run the same program in step, the older one wins every arbitration, and the younger finishes alone at the single-wave issue rate
// behind it -- 28-32 % of every stage.  Here the two waves of a SIMD never compete: the workgroup's 8 waves are two GROUPS of four (one
// wave of each group per SIMD) that run the same phase program one barrier apart, so while one group issues a pure cluster of MFMAs the
// other issues its LDS reads and LDS-DMA for the next phase (cdna_hip_programming.md "256^2 8-phase template", T3+T4+T5):
//
//      phase p of a group:   ds_read this phase's operand sub-tile ; 2 x global_load_lds (one half-tile of a later K tile)
//                            [p = 4: s_waitcnt vmcnt(6)]  [p = 1: s_waitcnt lgkmcnt(8)]
//                            s_barrier                                   <- the other group's cluster ends here
//                            s_waitcnt lgkmcnt(0) ; s_setprio 1 ; 16 MFMAs (one C quadrant x K = 64) ; s_setprio 0
//                            s_barrier                                   <- the other group's cluster starts here
//
// Geometry: workgroup tile 256 (m) x 256 (n), K tile 64 (128 bytes), 8 waves = 2 (m) x 4 (n).  The tile's m rows and n rows are split in
// HALVES of 128 and a wave owns 64 m rows of EACH m half and 32 n rows of EACH n half (wave (wr, wc): m = 128 mh + 64 wr + ..,
// n = 128 nh + 32 wc + ..), so C quadrant (mh, nh) of the wave needs exactly half-tiles A[mh] and B[nh]: four phases per K tile,
//      p = 1: reads B[0] (4 x ds_read_b128, first) + A[0] (8),  MFMAs (0, 0)        p = 3: reads A[1] (8),  MFMAs (1, 1)
//      p = 2: reads B[1] (4),                                    MFMAs (0, 1)        p = 4: reads nothing,   MFMAs (1, 0)
// LDS: 2 K tiles x {B[0], A[0], B[1], A[1]} x 16 KB = 128 KB.  Half-tiles are staged in the order B0, A0, B1, A1, one per phase, so that the
// wait of phase 4 (at most 6 LDS-DMA = 3 half-tiles still in flight) retires the whole NEXT K tile while three half-tiles of the one after
// it stay in flight across every barrier: vmcnt is never 0 inside the loop.  A half-tile is re-staged one phase after its last read when an
// lgkmcnt in front of the reading phase's first barrier retired those reads (B[0]: read first in phase 1, lgkmcnt(8)), otherwise two phases
// after (A[0]: read p1, staged p3; B[1]: read p2, staged p4; A[1]: read p3, staged p1 of the next tile); a staged tile is read one phase
// after the wait that retires it (both groups have waited and passed a barrier by then).
//
// The LDS image of a half-tile is [128 rows][128 B], written lane-linear by the DMA (a wave instruction = 8 rows); the bank swizzle
// chunk ^ ((row >> 1) & 7) goes on the per-lane SOURCE address and on the read address (conflict-free for the real ds_read_b128 lane groups,
// MI355X_MICROARCH.md LDS table; the same map as the band / ws kernels).
//
// MFMA roles: the matrix "B" (n rows: weights / couts, K-contiguous) is the MFMA's A operand and the matrix "A" (m rows: pixels) its B
// operand, so a lane ends up with 4 (16x16x32) or 4 x 4 (32x32x16) consecutive n of one m: NHWC stores.
#pragma once

namespace p8 {

constexpr int LDS_BYTES = 131072;
constexpr int FL_PRIO = 1;        // s_setprio 1 around the MFMA cluster
constexpr int FL_STAGGER = 2;     // the two wave groups run one barrier apart
constexpr int FL_TSYNC = 8;       // persistent kernels: the wave groups meet at every tile seam (their epilogues run side by side) and part again behind it
constexpr int FL_P4 = 64;         // two phases per K tile (clusters of 32 MFMAs): ktiles4 / prologue4 / start4
constexpr int FL_NODMA = 16;      // probe ablation (wrong results): no LDS-DMA inside the K loop
constexpr int FL_NOREAD = 32;     // probe ablation (wrong results): no fragment reads inside the K loop
constexpr int FL_SAFE = 4;        // debugging: every phase drains vmcnt / lgkmcnt in front of its first barrier (separates layout bugs from ordering bugs)

template <int MF> struct Acc;
// v[mh][nh][mi][ni]: n = 128 nh + 32 wc + 16 ni + 4 (lane >> 4) + j,  m = 128 mh + 64 wr + 16 mi + (lane & 15)
template <> struct Acc<16> { f32x4 v[2][2][4][2]; };
// v[mh][nh][mi]: n = 128 nh + 32 wc + 8 (r >> 2) + 4 (lane >> 5) + (r & 3),  m = 128 mh + 64 wr + 32 mi + (lane & 31)
template <> struct Acc<32> { f32x16 v[2][2][2]; };

template <int MF> __device__ __forceinline__ void zero(Acc<MF>& a) {
    if constexpr (MF == 16) {
#pragma unroll
        for (int i = 0; i < 32; ++i) (&a.v[0][0][0][0])[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) (&a.v[0][0][0])[i][r] = 0.f;
    }
}

template <typename T> struct Mma16;
template <> struct Mma16<_Float16> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma16<__bf16> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

// LDS-DMA of one 1-KiB piece: scalar 64-bit base + one 32-bit lane offset; destination = M0 (wave-uniform) + 16 lane.  Invisible to the
// compiler's waitcnt pass: every wait on these is written by hand.
__device__ __forceinline__ void glds(const char* sbase, unsigned voff, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_read(u32x4& d, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void barrier() { asm volatile("s_barrier" ::: "memory"); }

// Per-lane constants of a wave.
template <int MF> struct Lanes {
    static constexpr int NK = MF == 16 ? 2 : 4;        // read-address registers per operand (the K sub-step is an XOR on the chunk, not an add)
    unsigned ra[NK], rb[NK];                           // ds_read base addresses (A region at 0, B region at 64 KB)
    unsigned dst;                                      // LDS-DMA destination of this wave's piece 0 of half-tile 0 of the A region
    int wave, wr, wc, lane;
};
// f(row) of the piece rows this lane loads: row = 64 j + 8 wave + (lane >> 3)  ->  ((wave & 1) * 4 + (lane >> 4)) & 7
__device__ __forceinline__ unsigned src_chunk(int wave, int lane) { return (unsigned)((lane & 7) ^ (((wave & 1) * 4 + ((lane >> 3) >> 1)) & 7)); }

template <int MF> __device__ __forceinline__ Lanes<MF> lanes(const char* smem) {
    Lanes<MF> L;
    const int tid = threadIdx.x;
    L.lane = tid & 63;
    L.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    L.wr = L.wave >> 2; L.wc = L.wave & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    L.dst = lds0 + L.wave * 1024;
    if constexpr (MF == 16) {
        const int l15 = L.lane & 15, g = L.lane >> 4, s = (l15 >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            L.ra[kk] = lds0 + (L.wr * 64 + l15) * 128 + (((kk * 4 + g) ^ s) << 4);
            L.rb[kk] = lds0 + 65536 + (L.wc * 32 + l15) * 128 + (((kk * 4 + g) ^ s) << 4);
        }
    } else {
        const int l31 = L.lane & 31, h = L.lane >> 5, s = (l31 >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            L.ra[ks] = lds0 + (L.wr * 64 + l31) * 128 + (((ks * 2 + h) ^ s) << 4);
            L.rb[ks] = lds0 + 65536 + (L.wc * 32 + l31) * 128 + (((ks * 2 + h) ^ s) << 4);
        }
    }
    return L;
}

// The phase program.  A "stage" functor issues the wave's two LDS-DMA pieces of one half-tile: stA(kt, mh, dst) / stB(kt, nh, dst) with dst =
// this wave's LDS destination of piece 0 (piece 1 goes to dst + 8192); kt may run past the tile's nkt by up to 2: the caller decides
// what the stream carries there (the next tile of a persistent workgroup, or a clamped re-load nobody reads) -- it must issue
// exactly two LDS-DMA per call in any case, the counted waits rely on it.
//
//   prologue(L, stA, stB)         K tile 0 and three half-tiles of K tile 1 go out (first tile of a workgroup only)
//   start<FLAGS>(L)               K tile 0 has landed for everybody; the wave groups go one barrier apart
//   ktiles<..>(acc, L, nkt, ..)   the 4 nkt phases of one output tile (nkt even); may be called tile after tile -- the seam between
//                                 two tiles is an ordinary K-tile transition of the stream (an epilogue in between touches no LDS)
//   finish<FLAGS>(L)              the groups meet again, nothing is in flight
template <int MF, typename SA, typename SB>
__device__ __forceinline__ void prologue(const Lanes<MF>& L, SA stA, SB stB) {
    stB(0, 0, L.dst + 65536); stA(0, 0, L.dst); stB(0, 1, L.dst + 65536 + 16384); stA(0, 1, L.dst + 16384);
    stB(1, 0, L.dst + 65536 + 32768); stA(1, 0, L.dst + 32768); stB(1, 1, L.dst + 65536 + 49152);
}
template <int FLAGS, int MF>
__device__ __forceinline__ void start(const Lanes<MF>& L) {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    barrier();
    if ((FLAGS & FL_STAGGER) && !(FLAGS & FL_TSYNC) && L.wr == 1) barrier();
}
// FL_TSYNC: around every tile's ktiles()
template <int FLAGS, int MF> __device__ __forceinline__ void tile_begin(const Lanes<MF>& L) {
    if ((FLAGS & FL_STAGGER) && (FLAGS & FL_TSYNC) && L.wr == 1) barrier();
}
template <int FLAGS, int MF> __device__ __forceinline__ void tile_end(const Lanes<MF>& L) {
    if ((FLAGS & FL_STAGGER) && (FLAGS & FL_TSYNC) && L.wr == 0) barrier();
}
template <int FLAGS, int MF>
__device__ __forceinline__ void finish(const Lanes<MF>& L) {
    if ((FLAGS & FL_STAGGER) && !(FLAGS & FL_TSYNC) && L.wr == 0) barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // over-run loads of the stream: nothing lands in the LDS behind this point
    __builtin_amdgcn_sched_barrier(0);
}

// MI1: 16-row fragments per wave in the SECOND m half (16x16x32 only): 4 = 256-row tile, 3 = 224 rows (LDS rows 48..63 of each wave's
// 64 of half-tile A[1] are loaded and ignored): persistent workgroups balance their tile heights with it.
struct NoKtCb { __device__ __forceinline__ void operator()(int) const {} };        // lab: a callback in front of every pair of K tiles (stamps)
template <typename T, int MF, int FLAGS, int MI1 = (MF == 16 ? 4 : 2), typename SA, typename SB, typename CB = NoKtCb>
__device__ __forceinline__ void ktiles(Acc<MF>& acc, const Lanes<MF>& L, int nkt, SA stA, SB stB, CB cb = CB{}) {
    constexpr bool PRIO = FLAGS & FL_PRIO, SAFE = FLAGS & FL_SAFE;
    static_assert(MF == 16 ? (MI1 >= 2 && MI1 <= 4) : MI1 == 2, "tile height");
    u32x4 af[8], bf[2][4];
    auto readA = [&](auto D_, auto H_) {
        constexpr int d = decltype(D_)::value, mh = decltype(H_)::value, base = (d * 2 + mh) * 16384;
        constexpr int NI = mh == 0 ? 4 : MI1;
        if constexpr (MF == 16) {
            lds_read<base + 0 * 2048>(af[0], L.ra[0]); lds_read<base + 1 * 2048>(af[1], L.ra[0]);
            if constexpr (NI > 2) lds_read<base + 2 * 2048>(af[2], L.ra[0]);
            if constexpr (NI > 3) lds_read<base + 3 * 2048>(af[3], L.ra[0]);
            lds_read<base + 0 * 2048>(af[4], L.ra[1]); lds_read<base + 1 * 2048>(af[5], L.ra[1]);
            if constexpr (NI > 2) lds_read<base + 2 * 2048>(af[6], L.ra[1]);
            if constexpr (NI > 3) lds_read<base + 3 * 2048>(af[7], L.ra[1]);
        } else {
            lds_read<base>(af[0], L.ra[0]); lds_read<base + 4096>(af[1], L.ra[0]);
            lds_read<base>(af[2], L.ra[1]); lds_read<base + 4096>(af[3], L.ra[1]);
            lds_read<base>(af[4], L.ra[2]); lds_read<base + 4096>(af[5], L.ra[2]);
            lds_read<base>(af[6], L.ra[3]); lds_read<base + 4096>(af[7], L.ra[3]);
        }
    };
    auto readB = [&](auto D_, auto H_) {
        constexpr int d = decltype(D_)::value, nh = decltype(H_)::value, base = (d * 2 + nh) * 16384;
        if constexpr (MF == 16) {
            lds_read<base>(bf[nh][0], L.rb[0]); lds_read<base + 2048>(bf[nh][1], L.rb[0]);
            lds_read<base>(bf[nh][2], L.rb[1]); lds_read<base + 2048>(bf[nh][3], L.rb[1]);
        } else {
            lds_read<base>(bf[nh][0], L.rb[0]); lds_read<base>(bf[nh][1], L.rb[1]);
            lds_read<base>(bf[nh][2], L.rb[2]); lds_read<base>(bf[nh][3], L.rb[3]);
        }
    };
    // the MFMAs of quadrant (mh, nh); af / bf index = K sub-step major
    auto cluster = [&](auto MH_, auto NH_) {
        constexpr int mh = decltype(MH_)::value, nh = decltype(NH_)::value;
        if constexpr (MF == 16) {
            constexpr int NI = mh == 0 ? 4 : MI1;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int mi = 0; mi < NI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) Mma16<T>::run(bf[nh][kk * 2 + ni], af[kk * 4 + mi], acc.v[mh][nh][mi][ni]);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) Mma32<T>::run(bf[nh][ks], af[ks * 2 + mi], acc.v[mh][nh][mi]);
        }
    };
    auto phase = [&](auto P_, auto D_, int kt) {
        constexpr int P = decltype(P_)::value, d = decltype(D_)::value;
        using pipe::IC;
        constexpr bool RD = !(FLAGS & FL_NOREAD), DMA = !(FLAGS & FL_NODMA);
        if constexpr (P == 1 && RD) { readB(IC<d>{}, IC<0>{}); __builtin_amdgcn_sched_barrier(0); readA(IC<d>{}, IC<0>{}); }
        if constexpr (P == 2 && RD) readB(IC<d>{}, IC<1>{});
        if constexpr (P == 3 && RD) readA(IC<d>{}, IC<1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P == 1 && DMA) stA(kt + 1, 1, L.dst + ((d ^ 1) * 2 + 1) * 16384);
        if constexpr (P == 2 && DMA) stB(kt + 2, 0, L.dst + 65536 + (d * 2 + 0) * 16384);
        if constexpr (P == 3 && DMA) stA(kt + 2, 0, L.dst + (d * 2 + 0) * 16384);
        if constexpr (P == 4 && DMA) stB(kt + 2, 1, L.dst + 65536 + (d * 2 + 1) * 16384);
        if constexpr (SAFE) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if constexpr (P == 4 && DMA) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (P == 1 && RD) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
        if constexpr (P == 1) cluster(IC<0>{}, IC<0>{});
        if constexpr (P == 2) cluster(IC<0>{}, IC<1>{});
        if constexpr (P == 3) cluster(IC<1>{}, IC<1>{});
        if constexpr (P == 4) cluster(IC<1>{}, IC<0>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        barrier();
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        using pipe::IC;
        cb(kt);
        phase(IC<1>{}, IC<0>{}, kt); phase(IC<2>{}, IC<0>{}, kt); phase(IC<3>{}, IC<0>{}, kt); phase(IC<4>{}, IC<0>{}, kt);
        phase(IC<1>{}, IC<1>{}, kt + 1); phase(IC<2>{}, IC<1>{}, kt + 1); phase(IC<3>{}, IC<1>{}, kt + 1); phase(IC<4>{}, IC<1>{}, kt + 1);
    }
}

// Probe variant (tools/probe_gemm_8phase.hip): TWO phases per K tile, 32 MFMAs per cluster -- half the barriers.  Phase X reads B[0], B[1], A[0]
// (16 reads) and runs quadrants (0,0), (0,1); phase Y reads A[1] and runs (1,1), (1,0).  Every phase retires its reads in front of its
// first barrier, so a half-tile may be re-staged one phase after its last read: Y of K tile kt stages B[0], B[1] of kt + 2, X of kt + 1
// stages A[0], A[1] of kt + 2; the wait in Y (at most the 4 LDS-DMA it has just issued in flight) retires the whole next K tile.
// Prologue: K tile 0 and B[0], B[1] of K tile 1 (then vmcnt(4)).
template <typename T, int FLAGS, int MI1 = 4, typename SA, typename SB>
__device__ __forceinline__ void ktiles4(Acc<16>& acc, const Lanes<16>& L, int nkt, SA stA, SB stB) {
    static_assert(MI1 >= 2 && MI1 <= 4, "tile height");
    u32x4 af[8], bf[2][4];
    auto rdA = [&](auto D_, auto H_) {
        constexpr int mh = decltype(H_)::value, base = (decltype(D_)::value * 2 + mh) * 16384, NI = mh == 0 ? 4 : MI1;
        lds_read<base + 0 * 2048>(af[0], L.ra[0]); lds_read<base + 1 * 2048>(af[1], L.ra[0]);
        if constexpr (NI > 2) lds_read<base + 2 * 2048>(af[2], L.ra[0]);
        if constexpr (NI > 3) lds_read<base + 3 * 2048>(af[3], L.ra[0]);
        lds_read<base + 0 * 2048>(af[4], L.ra[1]); lds_read<base + 1 * 2048>(af[5], L.ra[1]);
        if constexpr (NI > 2) lds_read<base + 2 * 2048>(af[6], L.ra[1]);
        if constexpr (NI > 3) lds_read<base + 3 * 2048>(af[7], L.ra[1]);
    };
    auto rdB = [&](auto D_, auto H_) {
        constexpr int nh = decltype(H_)::value, base = (decltype(D_)::value * 2 + nh) * 16384;
        lds_read<base>(bf[nh][0], L.rb[0]); lds_read<base + 2048>(bf[nh][1], L.rb[0]);
        lds_read<base>(bf[nh][2], L.rb[1]); lds_read<base + 2048>(bf[nh][3], L.rb[1]);
    };
    auto cluster = [&](auto MH_, auto NH_) {
        constexpr int mh = decltype(MH_)::value, nh = decltype(NH_)::value, NI = mh == 0 ? 4 : MI1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < NI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) Mma16<T>::run(bf[nh][kk * 2 + ni], af[kk * 4 + mi], acc.v[mh][nh][mi][ni]);
    };
    auto phase = [&](auto P_, auto D_, int kt) {
        constexpr int P = decltype(P_)::value, d = decltype(D_)::value;
        using pipe::IC;
        if constexpr (P == 0) { rdB(IC<d>{}, IC<0>{}); rdB(IC<d>{}, IC<1>{}); rdA(IC<d>{}, IC<0>{}); }
        else rdA(IC<d>{}, IC<1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P == 0) { stA(kt + 1, 0, L.dst + ((d ^ 1) * 2 + 0) * 16384); stA(kt + 1, 1, L.dst + ((d ^ 1) * 2 + 1) * 16384); }
        else { stB(kt + 2, 0, L.dst + 65536 + (d * 2 + 0) * 16384); stB(kt + 2, 1, L.dst + 65536 + (d * 2 + 1) * 16384); }
        if constexpr (P == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P == 0) { cluster(IC<0>{}, IC<0>{}); cluster(IC<0>{}, IC<1>{}); }
        else { cluster(IC<1>{}, IC<1>{}); cluster(IC<1>{}, IC<0>{}); }
        __builtin_amdgcn_sched_barrier(0);
        barrier();
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        using pipe::IC;
        phase(IC<0>{}, IC<0>{}, kt); phase(IC<1>{}, IC<0>{}, kt);
        phase(IC<0>{}, IC<1>{}, kt + 1); phase(IC<1>{}, IC<1>{}, kt + 1);
    }
}
template <typename SA, typename SB>
__device__ __forceinline__ void prologue4(const Lanes<16>& L, SA stA, SB stB) {
    stB(0, 0, L.dst + 65536); stB(0, 1, L.dst + 65536 + 16384); stA(0, 0, L.dst); stA(0, 1, L.dst + 16384);
    stB(1, 0, L.dst + 65536 + 32768); stB(1, 1, L.dst + 65536 + 49152);
}
template <int FLAGS>
__device__ __forceinline__ void start4(const Lanes<16>& L) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    barrier();
    if ((FLAGS & FL_STAGGER) && !(FLAGS & FL_TSYNC) && L.wr == 1) barrier();
}
}  // namespace p8

namespace p8 {
// Epilogue helper: hands every (m row, 8 consecutive n) 16-byte chunk of the wave's accumulators to `f(mh, nh, m_in_half, n_in_half, chunk)`
// with m_in_half = 64 wr + .., n_in_half = 32 wc + .. (rounded to T; fragment pairs exchanged across lane rows / halves so that a
// lane owns eight consecutive n).  `fin(mh, nh, n_in_half, v)` may modify the four fp32 values n_in_half .. +3 first (bias, ReLU).
template <typename T, int MF, int MI1 = (MF == 16 ? 4 : 2), typename FIN, typename F>
__device__ __forceinline__ void for_chunks(Acc<MF>& acc, const Lanes<MF>& L, FIN fin, F f) {
    if constexpr (MF == 16) {
        const int l15 = L.lane & 15, g = L.lane >> 4;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    if (mh == 1 && mi >= MI1) continue;
                    f32x4 v0 = acc.v[mh][nh][mi][0], v1 = acc.v[mh][nh][mi][1];
                    fin(mh, nh, L.wc * 32 + 4 * g, v0);
                    fin(mh, nh, L.wc * 32 + 16 + 4 * g, v1);
                    T pa[4] = {from_f32<T>(v0.x), from_f32<T>(v0.y), from_f32<T>(v0.z), from_f32<T>(v0.w)};
                    T pb[4] = {from_f32<T>(v1.x), from_f32<T>(v1.y), from_f32<T>(v1.z), from_f32<T>(v1.w)};
                    const u32x2 A = *(const u32x2*)pa, B = *(const u32x2*)pb;
                    // rows g = 0, 2 end up with fragment 0, rows 1, 3 with fragment 1, n (g >> 1) * 8 .. +7 of it
                    const auto r0 = __builtin_amdgcn_permlane16_swap(A.x, B.x, false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(A.y, B.y, false, false);
                    f(mh, nh, L.wr * 64 + mi * 16 + l15, L.wc * 32 + (g & 1) * 16 + (g >> 1) * 8, (u32x4){r0[0], r1[0], r0[1], r1[1]});
                }
    } else {
        const int l31 = L.lane & 31, h = L.lane >> 5;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        u32x2 pk[2];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = 2 * jp + jj;
                            f32x4 v = {acc.v[mh][nh][mi][4 * j], acc.v[mh][nh][mi][4 * j + 1], acc.v[mh][nh][mi][4 * j + 2], acc.v[mh][nh][mi][4 * j + 3]};
                            fin(mh, nh, L.wc * 32 + 8 * j + 4 * h, v);
                            T p[4] = {from_f32<T>(v.x), from_f32<T>(v.y), from_f32<T>(v.z), from_f32<T>(v.w)};
                            pk[jj] = *(const u32x2*)p;
                        }
                        // lower half keeps its group 2 jp and receives the upper half's; the upper half ends up with both halves' group 2 jp + 1
                        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                        f(mh, nh, L.wr * 64 + mi * 32 + l31, L.wc * 32 + 16 * jp + 8 * h, (u32x4){r0[0], r1[0], r0[1], r1[1]});
                    }
    }
}
}  // namespace p8

// ------------------------------------------------------------------------------------------------ 512 (m) x 128 (n) variant
// The same phase program for layers with 128 couts: FOUR m halves of 128 rows and ONE n half, K tile 64.  A wave owns 64 rows of each m
// half and 32 n (wave (wr, wc): m = 128 mh + 64 wr + .., n = 32 wc + ..): phase p of a K tile reads A[p - 1] (8 x ds_read_b128; phase 1
// also B, 4 reads, first) and runs the 16 MFMAs of C quadrant p - 1 against the B fragments phase 1 left in registers.
// LDS: 2 K tiles x (A[0..3] + B) x 16 KB = 160 KB (all of it).  Five half-tiles per K tile go out in the order B, A0, A1, A2, A3 over four
// phases -- each one re-staged as early as its last read allows: of K tile kt (parity d) phase 1 stages A2 of kt + 1, phase 2 A3 of kt + 1 AND
// B of kt + 2, phase 3 A0 of kt + 2, phase 4 A1 of kt + 2 -- so that phase 4's vmcnt(6) again leaves exactly the three youngest half-tiles
// in flight and retires the whole next K tile.
namespace p8w {
using p8::barrier; using p8::glds; using p8::lds_read; using p8::Mma16;
constexpr int LDS_BYTES = 163840;
constexpr int A_OFF(int d, int mh) { return (d * 4 + mh) * 16384; }      // A region [0, 128 KB)
constexpr int B_OFF(int d) { return 131072 + d * 16384; }                // B region [128 KB, 160 KB)

// v[mh][mi][ni]: n = 32 wc + 16 ni + 4 (lane >> 4) + j,  m = 128 mh + 64 wr + 16 mi + (lane & 15)
struct Acc { f32x4 v[4][4][2]; };
__device__ __forceinline__ void zero(Acc& a) {
#pragma unroll
    for (int i = 0; i < 32; ++i) (&a.v[0][0][0])[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
struct Lanes {
    unsigned ra[2][2];       // [d][kk]: A read bases (the second K tile's half-tiles lie beyond the 64-KB reach of a ds_read offset)
    unsigned rb[2];          // [kk]
    unsigned dst;            // LDS-DMA destination of this wave's piece 0 at LDS offset 0
    int wave, wr, wc, lane;
};
__device__ __forceinline__ Lanes lanes(const char* smem) {
    Lanes L;
    const int tid = threadIdx.x;
    L.lane = tid & 63;
    L.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    L.wr = L.wave >> 2; L.wc = L.wave & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    L.dst = lds0 + L.wave * 1024;
    const int l15 = L.lane & 15, g = L.lane >> 4, s = (l15 >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const unsigned a0 = lds0 + (L.wr * 64 + l15) * 128 + (((kk * 4 + g) ^ s) << 4);
        L.ra[0][kk] = a0; L.ra[1][kk] = a0 + 65536;
        L.rb[kk] = lds0 + 131072 + (L.wc * 32 + l15) * 128 + (((kk * 4 + g) ^ s) << 4);
    }
    return L;
}
// stA(kt, mh, dst) / stB(kt, dst): as p8 (two LDS-DMA per call); kt may run past nkt by up to 2
template <typename SA, typename SB>
__device__ __forceinline__ void prologue(const Lanes& L, SA stA, SB stB) {
    stB(0, L.dst + B_OFF(0));
#pragma unroll
    for (int mh = 0; mh < 4; ++mh) stA(0, mh, L.dst + A_OFF(0, mh));
    stB(1, L.dst + B_OFF(1)); stA(1, 0, L.dst + A_OFF(1, 0)); stA(1, 1, L.dst + A_OFF(1, 1));
}
__device__ __forceinline__ void start() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); barrier(); }
__device__ __forceinline__ void tile_begin(const Lanes& L) { if (L.wr == 1) barrier(); }
__device__ __forceinline__ void tile_end(const Lanes& L) { if (L.wr == 0) barrier(); }
__device__ __forceinline__ void finish() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

// MI2 / MI3: 16-row fragments per wave in m halves 2 and 3 (4, or 3: a 480- / 448-row tile; halves 0 and 1 are always full)
template <typename T, int MI2, int MI3, typename SA, typename SB>
__device__ __forceinline__ void ktiles(Acc& acc, const Lanes& L, int nkt, SA stA, SB stB) {
    static_assert(MI2 >= 3 && MI2 <= 4 && MI3 >= 3 && MI3 <= MI2, "tile height");
    u32x4 af[8], bf[4];
    auto readA = [&](auto D_, auto H_) {
        constexpr int d = decltype(D_)::value, mh = decltype(H_)::value, base = A_OFF(d, mh) - d * 65536;
        constexpr int NI = mh == 2 ? MI2 : (mh == 3 ? MI3 : 4);
        lds_read<base + 0 * 2048>(af[0], L.ra[d][0]); lds_read<base + 1 * 2048>(af[1], L.ra[d][0]); lds_read<base + 2 * 2048>(af[2], L.ra[d][0]);
        if constexpr (NI > 3) lds_read<base + 3 * 2048>(af[3], L.ra[d][0]);
        lds_read<base + 0 * 2048>(af[4], L.ra[d][1]); lds_read<base + 1 * 2048>(af[5], L.ra[d][1]); lds_read<base + 2 * 2048>(af[6], L.ra[d][1]);
        if constexpr (NI > 3) lds_read<base + 3 * 2048>(af[7], L.ra[d][1]);
    };
    auto readB = [&](auto D_) {
        constexpr int base = decltype(D_)::value * 16384;
        lds_read<base>(bf[0], L.rb[0]); lds_read<base + 2048>(bf[1], L.rb[0]);
        lds_read<base>(bf[2], L.rb[1]); lds_read<base + 2048>(bf[3], L.rb[1]);
    };
    auto cluster = [&](auto MH_) {
        constexpr int mh = decltype(MH_)::value;
        constexpr int NI = mh == 2 ? MI2 : (mh == 3 ? MI3 : 4);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < NI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) Mma16<T>::run(bf[kk * 2 + ni], af[kk * 4 + mi], acc.v[mh][mi][ni]);
    };
    auto phase = [&](auto P_, auto D_, int kt) {
        constexpr int P = decltype(P_)::value, d = decltype(D_)::value;
        using pipe::IC;
        if constexpr (P == 1) { readB(IC<d>{}); __builtin_amdgcn_sched_barrier(0); }
        readA(IC<d>{}, IC<P - 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P == 1) stA(kt + 1, 2, L.dst + A_OFF(d ^ 1, 2));
        if constexpr (P == 2) { stA(kt + 1, 3, L.dst + A_OFF(d ^ 1, 3)); stB(kt + 2, L.dst + B_OFF(d)); }
        if constexpr (P == 3) stA(kt + 2, 0, L.dst + A_OFF(d, 0));
        if constexpr (P == 4) stA(kt + 2, 1, L.dst + A_OFF(d, 1));
        if constexpr (P == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (P == 1) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        cluster(IC<P - 1>{});
        __builtin_amdgcn_sched_barrier(0);
        barrier();
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        using pipe::IC;
        phase(IC<1>{}, IC<0>{}, kt); phase(IC<2>{}, IC<0>{}, kt); phase(IC<3>{}, IC<0>{}, kt); phase(IC<4>{}, IC<0>{}, kt);
        phase(IC<1>{}, IC<1>{}, kt + 1); phase(IC<2>{}, IC<1>{}, kt + 1); phase(IC<3>{}, IC<1>{}, kt + 1); phase(IC<4>{}, IC<1>{}, kt + 1);
    }
}
}  // namespace p8w
