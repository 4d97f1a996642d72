// Heads backward, data gradient of the first 1x1 convs WITHOUT the hidden gradient in memory (round 4).
// Included by conv_igemm.hip (needs gate_packed16, Mma32, GenHid).
//
//   d_x[p][c] = gate(x[p][c] > 0) * sum_h d_hid[p][h] W1[h][c],     d_hid[p][h] = keep[p][h] * scale * sum_k d_out[p][k] W2[k][h]
//
// d_hid (944 MB at batch 64: 2048 hidden channels on 60 x 60) is the B operand of the GEMM and is GENERATED in registers, 32 hidden
// channels x 32 pixels per v_mfma_f32_32x32x16 (common.hpp: GenHid): the generating MFMA's A rows are W2^T's channels in the order
// pi(i) = 16 (i >> 4) + 8 ((i >> 2) & 1) + 4 ((i >> 3) & 1) + (i & 3), so its result registers r = 8 t + e of lane half lh ARE the K
// slots 8 lh + e of the main MFMA's K = 16 step t in the standard fragment order (dbx_pack_weight mode 5 image of W1^T: the one the
// ws kernel takes) -- no shuffle, no LDS round trip: one hash per lane and block, a 256-entry LDS table turns a keep byte into the
// four AND masks of the packed pairs.  Workgroup = 4 waves, tile = 256 COMPACT pixels (no halo positions) x all 256 output channels;
// wave = 64 pixels (two pixel fragments) x 256 channels: 256 accumulator registers, one wave per SIMD.  W1^T streams through a ring of
// four 16-KiB LDS stages (one 32-channel block each, LDS-DMA, contiguous in the fragment image; the ring runs across tiles: every tile
// reads the same 64 blocks), shared by the four waves; W2^T (32 KiB) sits in LDS for the whole launch.  Per block and wave: 32 main
// MFMAs + 2 generating ones, ~60 VALU instructions, 21 LDS reads, one barrier.  Persistent workgroups.
#pragma once

#ifndef HG_ABL
#define HG_ABL 0                        // lab builds: 1 no generation in the loop, 2 no epilogue, 4 no main MFMAs, 8 no weight DMA
#endif
namespace hgen {
constexpr int HB_BYTES = 16384, NST = 4, W2T_BYTES = 2048 * 16 + 64, LUT_BYTES = 256 * 16;
constexpr int SMEM = NST * HB_BYTES + W2T_BYTES + LUT_BYTES;
}
struct HGenArgs {
    GenHid g;
    const char* w1f;                    // fragment-order image of W1^T: 256 rows (output channels) x 512 nh (hidden channels), mode 5
    char* y; const char* gate;          // channel offset applied
    int y_hp, y_wp, y_ld, y_pad, g_hp, g_wp, g_ld, g_pad;
    int npix, ntiles, nhb;              // N H W; tiles of 256 pixels; 32-channel blocks (16 nh)
};

template <typename T>
__global__ __launch_bounds__(256, 1) void heads1_dgrad_gen_kernel(const HGenArgs a) {
    static_assert(sizeof(T) == 2, "16-bit compute types");
    using namespace hgen;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const w2t = smem + NST * HB_BYTES;                            // [hidden channel][8 k] + one zero row
    char* const lut = w2t + W2T_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nhb = a.nhb;
    // ---- tables: scale * W2^T in the compute dtype, the keep-byte masks
    {
        const float sc = a.g.use_hash ? 2.f : 1.f;
        for (int c = tid; c < 32 * nhb; c += 256) {
            const int hd = c >> 9, cl = c & 511;
            const float* wp = a.g.w2[0];
            int k = a.g.k[0];
#pragma unroll
            for (int hh = 1; hh < 4; ++hh)
                if (hd == hh) { wp = a.g.w2[hh]; k = a.g.k[hh]; }
            u32x4 raw;
            T* e = (T*)&raw;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = j < k ? j : k - 1;
                const float v = wp[(size_t)row * 512 + cl];
                e[j] = from_f32<T>(j < k ? v * sc : 0.f);
            }
            *(u32x4*)(w2t + c * 16) = raw;
        }
        if (tid < 4) *(u32x4*)(w2t + 32 * nhb * 16 + tid * 16) = (u32x4){0u, 0u, 0u, 0u};    // (the zero row the upper lane half reads)
        {
            u32x4 m;
#pragma unroll
            for (int p = 0; p < 4; ++p) m[p] = ((tid >> (2 * p)) & 1 ? 0xffffu : 0u) | ((tid >> (2 * p + 1)) & 1 ? 0xffff0000u : 0u);
            *(u32x4*)(lut + tid * 16) = m;
        }
    }
    // generating MFMA: A row i of this lane = hidden channel pi(i) of the block; lanes >= 32 (K 8..15) read the zero row
    const int pi = 16 * (l31 >> 4) + 8 * ((l31 >> 2) & 1) + 4 * ((l31 >> 3) & 1) + (l31 & 3);
    const unsigned w2_base = lh ? (unsigned)(32 * nhb * 16) : (unsigned)(pi * 16), w2_step = lh ? 0u : 512u;
    const unsigned nodrop = a.g.use_hash ? 0u : 0xffffffffu;
    const unsigned ld2 = (unsigned)a.g.ld * 2u, slot2 = (unsigned)a.g.slot * 2u;

    // ---- LDS-DMA of W1^T blocks: global block index gb (runs across tiles), weight block gb % nhb, stage gb % 4; a wave moves 4 KiB
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto glds = [](const char* src, unsigned voff, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
    };
    const unsigned vlane = (unsigned)lane * 16u;
    int wb_next = 0, st_next = 0;                                       // weight block / stage of the next DMA
    auto issue = [&]() {
        const char* src = a.w1f + (size_t)wb_next * HB_BYTES + wave * 4096;
        const unsigned dst = lds0 + st_next * HB_BYTES + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) glds(src + q * 1024, vlane, dst + q * 1024);
        if (++wb_next == nhb) wb_next = 0;
        st_next = (st_next + 1) & 3;
    };

    const int stride = gridDim.x;
    int tile = blockIdx.x;
    if (tile >= a.ntiles) return;                                       // (uniform)
    // per pixel fragment: compact pixel, its hash product, validity
    int m[2]; unsigned hm[2]; bool ok[2];
    auto set_tile = [&](int t) {
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
            const int mm = t * 256 + wave * 64 + pf * 32 + l31;
            ok[pf] = mm < a.npix;
            m[pf] = ok[pf] ? mm : a.npix - 1;
            hm[pf] = (unsigned)m[pf] * 0x9E3779B1u;
        }
    };
    u32x4 dnx[2];                                                       // d_out slots in flight (asm loads: waited for by the counted vmcnt in front of a barrier)
    auto dfetch = [&](int t, int hd) {
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
            int mm = t * 256 + wave * 64 + pf * 32 + l31;
            mm = mm < a.npix ? mm : a.npix - 1;
            const unsigned voff = __umul24((unsigned)mm, ld2) + (unsigned)hd * slot2;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dnx[pf]) : "v"(voff), "s"(a.g.dout) : "memory");
        }
    };
    u32x4 dfr[2];                                                       // B operands of the generating MFMA for the current head
    auto dswap = [&]() {
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) dfr[pf] = (lh || !ok[pf]) ? (u32x4){0u, 0u, 0u, 0u} : dnx[pf];
    };
    u32x4 bfr[2][2][2];                                                 // [buffer][pixel fragment][K16 half]: generated B operands
    auto gen_w = [&](int hb) { return *(const u32x4*)(w2t + w2_base + (unsigned)hb * w2_step); };   // the generating MFMA's A operand
    // Generation of block hb's operands in three pieces a block's schedule spreads out (below): (1) both generating MFMAs -- inline asm
    // with VGPR destinations and a zero C operand: all 256 AGPRs hold the accumulators, and the builtin's result landed in AGPRs too,
    // the compiler parked 16 accumulator registers in VGPRs around every block (80 register moves per block and wave); (2) the two hashes
    // and the four table reads, independent of the MFMAs' results; (3) behind s_nops covering the MFMA-write -> VALU-read hazard the
    // assembler does not see for asm operands: packing to 16-bit pairs and masking.
    f32x16 dd[2];
    u32x4 mk[2][2];
    auto gen_mma = [&](const u32x4& wa) {
        if constexpr (DType<T>::id == DBX_F16)
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, 0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, 0"
                         : "=&v"(dd[0]), "=&v"(dd[1]) : "v"(wa), "v"(dfr[0]), "v"(dfr[1]));
        else
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, 0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, 0"
                         : "=&v"(dd[0]), "=&v"(dd[1]) : "v"(wa), "v"(dfr[0]), "v"(dfr[1]));
    };
    auto gen_masks = [&](int hb) {
        const unsigned cc = a.g.seed ^ ((unsigned)hb * 0x85EBCA77u);
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
            unsigned x = cc ^ hm[pf];
            x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
            const unsigned bits = (x >> (8 * lh)) | nodrop;             // bit 16 t + e: hidden channel 16 t + 8 lh + e of the block
#pragma unroll
            for (int t = 0; t < 2; ++t) mk[pf][t] = *(const u32x4*)(lut + ((bits >> (16 * t)) & 255u) * 16);
        }
    };
    auto gen_pack = [&](int buf) {
        asm volatile("s_nop 15\n\ts_nop 4" : "+v"(dd[0]), "+v"(dd[1]));
#pragma unroll
        for (int pf = 0; pf < 2; ++pf)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                typedef float f32x2v __attribute__((ext_vector_type(2)));
                typedef T t2v __attribute__((ext_vector_type(2)));
                u32x4 o;
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    o[p] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v){dd[pf][8 * t + 2 * p], dd[pf][8 * t + 2 * p + 1]}, t2v)) & mk[pf][t][p];
                bfr[buf][pf][t] = o;
            }
    };
    auto gen = [&](int buf, int hb, const u32x4& wa) { gen_mma(wa); gen_masks(hb); gen_pack(buf); };

    f32x16 acc[2][8];
#pragma unroll
    for (int pf = 0; pf < 2; ++pf)
#pragma unroll
        for (int cf = 0; cf < 8; ++cf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pf][cf][r] = 0.f;

    // ---- prologue: three weight blocks in flight, the first tile's head-0 slots, block 0 generated
    issue(); issue(); issue();
    set_tile(tile);
    dfetch(tile, 0);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(dnx[0]), "+v"(dnx[1]) :: "memory");
    __syncthreads();                                                    // tables written, blocks 0..2 landed
    dswap();
    gen(0, 0, gen_w(0));

    int st = 0;                                                         // stage of the current block
    for (;;) {
        // one tile: blocks hb = 0 .. nhb - 1; generated operands alternate between the two buffers (nhb is even)
        for (int hb = 0; hb < nhb; hb += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int h = hb + par;
                // block h's weights landed (this wave's pieces: issued three blocks ago, two blocks of four pieces behind them; the barrier
                // covers the other waves'), every wave is past block h - 1's stage
                asm volatile("s_waitcnt vmcnt(8)" : "+v"(dnx[0]), "+v"(dnx[1]) :: "memory");
                __builtin_amdgcn_s_barrier();
                // schedule of a block: the eight weight fragments of the first K = 16 half are requested right behind the barrier and land
                // under the generation of the NEXT block's operands (its table reads, two MFMAs, ~60 VALU instructions); each of them is
                // then used for two MFMAs while the matching fragment of the second half is read
                const char* S = smem + st * HB_BYTES + lane * 16;
                const bool last = h + 1 == nhb;
                const u32x4 wa = gen_w(last ? 0 : h + 1);               // (first: the generating MFMAs wait for this read only)
                u32x4 wf0[8], wf1[8];
#pragma unroll
                for (int cf = 0; cf < 8; ++cf) wf0[cf] = *(const u32x4*)(S + cf * 1024);
                __builtin_amdgcn_sched_barrier(0);
                if (!(HG_ABL & 8)) issue();                             // block h + 3 into the stage block h - 1 left
                // the next block's operands: the next head's (or the next tile's first head's) d_out slots arrive three blocks ahead
                if ((h & 15) == 12) {                                   // (uniform)
                    const bool wrap = h + 4 >= nhb;
                    dfetch(wrap ? tile + stride : tile, wrap ? 0 : (h + 4) >> 4);
                }
                if ((h & 15) == 15) {
                    if (last) set_tile(tile + stride);                  // (past the last tile: clamped pixels, results unused)
                    dswap();
                }
                if (!(HG_ABL & 1)) { gen_mma(wa); gen_masks(last ? 0 : h + 1); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cf = 0; cf < 8; ++cf) {
                    wf1[cf] = *(const u32x4*)(S + (8 + cf) * 1024);
                    if (HG_ABL & 4) continue;
                    Mma32<T>::run(wf0[cf], bfr[par][0][0], acc[0][cf]);
                    Mma32<T>::run(wf0[cf], bfr[par][1][0], acc[1][cf]);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // second half: 16 MFMAs with the next block's packing and masking (32 VALU instructions) between them
                if (!(HG_ABL & 1)) gen_pack(par ^ 1);
#pragma unroll
                for (int cf = 0; cf < 8; ++cf) {
                    if (HG_ABL & 4) continue;
                    Mma32<T>::run(wf1[cf], bfr[par][0][1], acc[0][cf]);
                    Mma32<T>::run(wf1[cf], bfr[par][1][1], acc[1][cf]);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                st = (st + 1) & 3;
            }
        }
        // ---- epilogue of the tile (set_tile has already moved m / ok to the next tile: recompute this tile's pixels).  All 32 gate chunks
        // of the wave's two pixel fragments are requested up front: a load consumed right behind its issue exposes a memory latency 32 times
        // per tile with one wave per SIMD (the first version: a third of the tile's time)
        __builtin_amdgcn_sched_barrier(0);
        if (!(HG_ABL & 2)) {
            T* ypix[2];
            bool okp[2];
            u32x4 gt[2][8][2];
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                const int mm = tile * 256 + wave * 64 + pf * 32 + l31;
                okp[pf] = mm < a.npix;
                const int mc = okp[pf] ? mm : 0;
                const int hw = a.g.H * a.g.W;
                const int n = mc / hw, rem = mc - n * hw, oy = rem / a.g.W, ox = rem - oy * a.g.W;
                ypix[pf] = (T*)a.y + (size_t)((n * a.y_hp + oy + a.y_pad) * a.y_wp + ox + a.y_pad) * (size_t)a.y_ld + 8 * lh;
                const T* gpix = (const T*)a.gate + (size_t)((n * a.g_hp + oy + a.g_pad) * a.g_wp + ox + a.g_pad) * (size_t)a.g_ld + 8 * lh;
#pragma unroll
                for (int cf = 0; cf < 8; ++cf)
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) gt[pf][cf][jp] = *(const u32x4*)(gpix + cf * 32 + 16 * jp);
            }
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
#pragma unroll
                for (int cf = 0; cf < 8; ++cf) {
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        u32x2 pk[2];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = 2 * jp + jj;
                            T p4[4] = {from_f32<T>(acc[pf][cf][4 * j]), from_f32<T>(acc[pf][cf][4 * j + 1]), from_f32<T>(acc[pf][cf][4 * j + 2]),
                                       from_f32<T>(acc[pf][cf][4 * j + 3])};
                            pk[jj] = *(const u32x2*)p4;
                        }
                        // lower half keeps its group 2 jp and receives the upper half's; upper half receives the lower's 2 jp + 1: 8 consecutive channels per lane
                        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                        const u32x4 o = gate_packed16((u32x4){r0[0], r1[0], r0[1], r1[1]}, gt[pf][cf][jp]);
                        if (okp[pf]) *(u32x4*)(ypix[pf] + cf * 32 + 16 * jp) = o;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[pf][cf][r] = 0.f;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        tile += stride;
        if (tile >= a.ntiles) break;                                    // (uniform)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the over-run weight blocks land before the workgroup ends
}

template <typename T>
static int launch_heads1_dgrad_gen(const HGenArgs& a, hipStream_t s) {
    static DbxDevOnce attr_once; int attr_dev = 0;
    if (attr_once.pending(&attr_dev)) {
        DBX_HIP(hipFuncSetAttribute((const void*)heads1_dgrad_gen_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, hgen::SMEM));
        attr_once.mark(attr_dev);
    }
    int ncu = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    const int grid = a.ntiles < ncu ? a.ntiles : ncu;
    hipLaunchKernelGGL(heads1_dgrad_gen_kernel<T>, dim3(grid), dim3(256), hgen::SMEM, s, a);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
