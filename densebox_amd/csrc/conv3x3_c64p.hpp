// v8: 64 input channels, 3x3 / pad 1, 64 couts per workgroup -- the halo-tile kernel of conv3x3_c64_kernel with the tile loop
// software-pipelined ("c64p").  Included by conv_igemm.hip (needs ConvArgs, C64Geo, Mma, pair_exchange, dma_swz, EPI2_POOL*).
//
// Why: conv1_2 (64 -> 64 at 240 x 240) is 14 400 tiles of 8 x 32 pixels per 64 patches, 56 per CU, and a tile is only 144 MFMAs
// per wave (4 600 MFMA cycles per SIMD).  In conv3x3_c64_kernel every tile is a closed episode -- vmcnt(0), barrier, issue the next
// halo, first fragment reads, 18 K steps, epilogue arithmetic, stores -- and all eight waves of the one workgroup a CU holds go through
// its phases together: nothing overlaps the ~8 000 cycles per tile that are not MFMA issue (tools/band_lab ablations, round 4: with
// EVERY fragment read removed the kernel still ran at 0.33 of the MFMA peak; the tile, not the LDS, was the limit).  Here
//   * the barrier of tile t sits in front of its LAST K step: by then every wave has issued (and waited for) its last fragment reads
//     of the tile, and the halo of tile t+1 -- issued a whole tile earlier -- has landed: the barrier never waits for data;
//   * behind the barrier the halo of tile t+2 is issued into the buffer tile t has just left, and the first fragments of tile t+1
//     are read UNDER the last eight MFMAs of tile t: a tile never starts from an empty pipe;
//   * a tile's accumulators (32 registers) are handed to the NEXT tile's K loop, which runs its own into the second set and works the
//     previous tile's epilogue (bias, ReLU, rounding, pair exchange, pooling windows + arg-max nibbles, stores) off in four pieces
//     between its first K steps, in the shadow of its MFMAs;
//   * the stores of tile t-1 are ~10 K steps old when tile t reaches its vmcnt(0): it waits for nothing.
// Cout tiles: a workgroup keeps to ONE 64-cout slice (blockIdx.x % ntile_n) whose nine taps (72 KB) stay in its LDS, so the same
// kernel serves conv2_1 (64 -> 128: two slices; the halos are then read by two workgroups, from L2).
#pragma once

template <typename T, bool POOL>
__global__ __launch_bounds__(512) void conv3x3_c64p_kernel(const ConvArgs a, const C64Geo tg) {
    static_assert(sizeof(T) == 2, "16-bit types");
    constexpr int TR = 8, TC = 32, HR = TR + 2, HC = TC + 2, HPX = HR * HC;      // 340 halo pixels of 128 B
    constexpr int WROW = 1152 + 32, W_BYTES = 64 * WROW, IN_BYTES = HPX * 128, IN_STRIDE = IN_BYTES + 512;
    constexpr int PIECES = (HPX + 7) / 8;                                         // 1-KiB pieces (8 pixels): 43
    constexpr int NP = (PIECES + 7) / 8;                                          // per-wave slots: 6
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ws = smem;
    char* In = smem + W_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = a.ntile_n;
    const int tn = blockIdx.x % ntn;                                              // this workgroup's cout slice
    const int n0 = tn * 64;

    // ---- weights of the slice: packed rows [cout][tap][cin] of 1152 B -> LDS rows of 1184 B (register-staged, once per workgroup)
    for (int c = tid; c < 64 * 72; c += 512) {
        const int row = c / 72, ch = c - row * 72;
        *(u32x4*)(Ws + row * WROW + ch * 16) = *(const u32x4*)(a.w + (size_t)(n0 + row) * a.ktot_bytes + ch * 16);
    }

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
    const int tpi = tg.tiles_x * tg.tiles_y;
    struct TileC { int n, y0, x0; };                                              // uniform
    auto coords = [&](int tile) {
        TileC c;
        c.n = tile / tpi;
        const int r = tile - c.n * tpi;
        const int ty = r / tg.tiles_x;
        c.y0 = ty * TR; c.x0 = (r - ty * tg.tiles_x) * TC;
        c.n = __builtin_amdgcn_readfirstlane(c.n); c.y0 = __builtin_amdgcn_readfirstlane(c.y0); c.x0 = __builtin_amdgcn_readfirstlane(c.x0);
        return c;
    };
    auto issue = [&](int tile, int buf) {
        const TileC c = coords(tile);
        char* dst = In + buf * IN_STRIDE;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (pv[i]) {
                int fy = c.y0 + hr_[i]; fy = fy < a.x_hp ? fy : a.x_hp - 1;
                const int p = (wave + 8 * i) * 8 + lp;
                const char* src = a.x + ((size_t)(c.n * a.x_hp + fy) * a.x_wp + (c.x0 + hc_[i])) * pix_bytes + ((lc ^ dma_swz<128>(p)) << 4);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + (wave + 8 * i) * 1024), 16, 0, 0);
            }
        }
    };

    // ---- fragment read offsets (tile-independent).  Pixels: halo pixel p = (row + ky) * 34 + column + fr + kx, chunk kc*4+g.
    // plain: wave w owns tile row w, columns mi * 16 ..; POOL: rows 2 (w >> 1) + mi, columns 16 (w & 1) .. (a 2x2 window = the two
    // fragments of one lane and its lane ^ 1 neighbour)
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

    const int G = gridDim.x / ntn;                                                // workgroups per cout slice
    const int first = blockIdx.x / ntn, stride = G;
    if (first >= tg.ntiles) return;
    const int cb = (lane >> 4) * 4;
    const int epi = a.epi;
    f32x4 bias[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) bias[ni] = (epi & DBX_EPI_BIAS) ? *(const f32x4*)(a.bias + n0 + cb + ni * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};

    u32x4 wf[2][4], xf[2][2];
    auto rd = [&](const char* Xb, int st, u32x4 (&w)[4], u32x4 (&x)[2]) {
        const int t = st >> 1, kc = st & 1;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) w[ni] = *(const u32x4*)(Ws + offW + ni * 16 * WROW + t * 128 + kc * 64);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) x[mi] = *(const u32x4*)(Xb + (offX[t][mi] ^ (kc << 6)));
    };

    // ---- epilogue of ONE finished tile in four (POOL: two) pieces, run between the first K steps of the next tile
    // (`live` false: the first tile has no predecessor -- the arithmetic runs on zeros, nothing is stored)
    auto epilogue_piece = [&](int q, f32x4 (&acc)[4][2], const TileC& c, bool live) {
        if constexpr (!POOL) {
            const int mi = q >> 1, ni = (q & 1) * 2;
            const int oy = c.y0 + wave, ox = c.x0 + mi * 16 + fr;
            const bool ok = live && oy < tg.H && ox < tg.W;
            T* ypix = (T*)a.y + (size_t)((c.n * a.y_hp + oy + a.y_pad) * a.y_wp + (ox + a.y_pad)) * (size_t)a.y_ld + n0;
            f32x4 v0 = acc[ni][mi] + bias[ni], v1 = acc[ni + 1][mi] + bias[ni + 1];
            if (epi & DBX_EPI_RELU) {
                v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
                v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
            }
            u32x4 o = pair_exchange<T>(v0, v1);                                   // all lanes
            if (ok) {
                if (epi & DBX_EPI_GATE) {
                    const T* gpix = (const T*)a.gate + (size_t)((c.n * a.g_hp + oy + a.g_pad) * a.g_wp + (ox + a.g_pad)) * (size_t)a.g_ld + n0;
                    o = gate_packed16(o, *(const u32x4*)(gpix + pair_cout_off(g, ni)));
                }
                *(u32x4*)(ypix + pair_cout_off(g, ni)) = o;
            }
        } else {
            // piece q = cout fragments 2 q, 2 q + 1: both rows of the wave (mi = 0, 1), one 2x2 window per lane pair and channel
            const int oy = c.y0 + 2 * (wave >> 1), ox = c.x0 + (wave & 1) * 16 + fr;          // H, W even: rows oy, oy + 1 together
            const bool ok = live && oy < tg.H && ox < tg.W;
            f32x4 v[2][2];                                                        // [fragment of the pair][row]
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    f32x4 t = acc[2 * q + k][mi] + bias[2 * q + k];
                    if (epi & DBX_EPI_RELU) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
                    v[k][mi] = t;
                }
            if (!(a.epi2 & EPI2_POOL_ONLY)) {
                // the full-resolution map too (training without the arg-max nibbles)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    T* ypix = (T*)a.y + (size_t)((c.n * a.y_hp + oy + mi + a.y_pad) * a.y_wp + (ox + a.y_pad)) * (size_t)a.y_ld;
                    const u32x4 o = pair_exchange<T>(v[0][mi], v[1][mi]);
                    if (ok) *(u32x4*)(ypix + pair_cout_off(g, 2 * q)) = o;
                }
            }
            T* ppix = (T*)a.y2 + (size_t)((c.n * a.y2_hp + (oy >> 1) + a.y2_pad) * a.y2_wp + ((ox >> 1) + a.y2_pad)) * (size_t)a.y2_ld;
            if (a.pool_idx && (epi & DBX_EPI_RELU)) {
                // pooled map + arg-max nibbles from the values ROUNDED to T, on packed unsigned 16-bit halves (post-ReLU values are
                // >= +0: their bit patterns order like the numbers) -- the window logic of conv3x3_c64_kernel<POOL>, bitwise its results
                auto pk_max = [](unsigned x, unsigned y) { unsigned d; asm("v_pk_max_u16 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
                auto pk_min = [](unsigned x, unsigned y) { unsigned d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
                auto pk_sub = [](unsigned x, unsigned y) { unsigned d; asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; };
                auto pk_mad = [](unsigned x, unsigned y, unsigned z) { unsigned d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z)); return d; };
                const unsigned one = 0x00010001u, two = 0x00020002u, four = 0x00040004u;
                unsigned char* ipix = a.pool_idx + ((size_t)(c.n * (tg.H >> 1) + (oy >> 1)) * (tg.W >> 1) + (ox >> 1)) * 32 + g * 2;
                u32x2 M[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    unsigned int nib[2];
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        unsigned int pk[2];
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi) {
                            const T p2[2] = {from_f32<T>(v[k][mi][2 * qq]), from_f32<T>(v[k][mi][2 * qq + 1])};
                            pk[mi] = *(const unsigned int*)p2 & 0x7fff7fffu;      // (-0 -> +0: keeps the unsigned order)
                        }
                        const unsigned A = pk[0], Cc = pk[1];
                        const unsigned Bn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk[0], 0xB1, 0xF, 0xF, true);
                        const unsigned Dn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk[1], 0xB1, 0xF, 0xF, true);
                        const unsigned t0 = pk_max(A, Bn), t1 = pk_max(Cc, Dn), m = pk_max(t0, t1);
                        const unsigned h0 = pk_min(t0 ^ A, one);        // 1: the neighbour column is strictly larger (row 0)
                        const unsigned h1 = pk_min(t1 ^ Cc, one);       //    ... (row 1)
                        const unsigned r = pk_min(m ^ t0, one);         // 1: row 1 is strictly larger
                        const unsigned pos = pk_min(m, one);
                        const unsigned b0 = pk_mad(r, pk_sub(h1, h0), h0);      // r ? h1 : h0  (mod 2^16)
                        nib[qq] = pk_mad(pos, four, pk_mad(r, two, b0));        // halves: channel 2 qq (low), 2 qq + 1 (high)
                        M[k][qq] = m;
                    }
                    const unsigned t = nib[0] | (nib[1] << 8);
                    const unsigned w16 = (t | (t >> 12)) & 0xffffu;     // x | y << 4 | z << 8 | w << 12
                    if (ok && !(fr & 1)) *(unsigned short*)(ipix + (2 * q + k) * 8) = (unsigned short)w16;
                }
                const auto r0 = __builtin_amdgcn_permlane16_swap(M[0].x, M[1].x, false, false);       // as pair_exchange
                const auto r1 = __builtin_amdgcn_permlane16_swap(M[0].y, M[1].y, false, false);
                const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                if (ok && !(fr & 1)) *(u32x4*)(ppix + pair_cout_off(g, 2 * q)) = o;
            } else {
                if (a.pool_idx) {
                    // no ReLU in the epilogue (values of either sign): the same window logic on the fp32 values before rounding
                    unsigned char* ipix = a.pool_idx + ((size_t)(c.n * (tg.H >> 1) + (oy >> 1)) * (tg.W >> 1) + (ox >> 1)) * 32 + g * 2;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        unsigned int w16 = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float a0 = v[k][0][j], c0 = v[k][1][j];
                            const float b0 = dpp_xor1(a0), d0 = dpp_xor1(c0);
                            int arg = 0;
                            float m = a0;
                            if (b0 > m) { m = b0; arg = 1; }
                            if (c0 > m) { m = c0; arg = 2; }
                            if (d0 > m) { m = d0; arg = 3; }
                            w16 |= (unsigned int)(arg | (m > 0.f ? 4 : 0)) << (4 * j);
                        }
                        if (ok && !(fr & 1)) *(unsigned short*)(ipix + (2 * q + k) * 8) = (unsigned short)w16;
                    }
                }
                // rounding to T is monotonic: max of the f32 values, then rounded == max of the rounded values (dbx_maxpool2x2)
                f32x4 m[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    f32x4 t;
                    t.x = fmaxf(v[k][0].x, v[k][1].x); t.y = fmaxf(v[k][0].y, v[k][1].y);
                    t.z = fmaxf(v[k][0].z, v[k][1].z); t.w = fmaxf(v[k][0].w, v[k][1].w);
                    t.x = fmaxf(t.x, dpp_xor1(t.x)); t.y = fmaxf(t.y, dpp_xor1(t.y));
                    t.z = fmaxf(t.z, dpp_xor1(t.z)); t.w = fmaxf(t.w, dpp_xor1(t.w));
                    m[k] = t;
                }
                const u32x4 o = pair_exchange<T>(m[0], m[1]);
                if (ok && !(fr & 1)) *(u32x4*)(ppix + pair_cout_off(g, 2 * q)) = o;
            }
        }
    };

    // ---- one tile: K steps 0..16 (+ the previous tile's epilogue), barrier, next-next halo, first reads of the next tile, step 17
    f32x4 accs[2][4][2];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) accs[1][ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};       // (the first tile's "previous" accumulators)
    auto run_tile = [&](auto CUR_, int tile, const TileC& prev, bool prev_live) {
        constexpr int cur = decltype(CUR_)::value;
        f32x4 (&acc)[4][2] = accs[cur];
        f32x4 (&pacc)[4][2] = accs[cur ^ 1];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* Xb = In + cur * IN_STRIDE;
        const char* Xn = In + (cur ^ 1) * IN_STRIDE;
        auto mm = [&](int st) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) Mma<T>::run(wf[st & 1][ni], xf[st & 1][mi], acc[ni][mi]);
        };
#pragma unroll
        for (int st = 0; st < 17; ++st) {
            rd(Xb, st + 1, wf[(st + 1) & 1], xf[(st + 1) & 1]);
            mm(st);
            // the interleave of reads and MFMAs is pinned (2 MFMA : 2, 2, 1, 1 reads), as in conv3x3_c64_kernel
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if constexpr (POOL) { if (st == 2 || st == 6) epilogue_piece((st - 2) >> 2, pacc, prev, prev_live); }
            else { if (st >= 2 && st < 10 && (st & 1) == 0) epilogue_piece((st - 2) >> 1, pacc, prev, prev_live); }
        }
        // every wave's fragment reads of this tile are issued; mine have returned.  The next tile's halo (issued a tile ago) has landed
        // and the previous tile's stores are ten K steps old
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (tile + 2 * stride < tg.ntiles) issue(tile + 2 * stride, cur);
        rd(Xn, 0, wf[0], xf[0]);                                                  // (past the last tile: a harmless read of stale bytes)
        mm(17);
    };

    // ---- prologue: first halo (and the weights) visible to everyone, second halo on its way, first fragments in flight
    issue(first, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (first + stride < tg.ntiles) issue(first + stride, 1);
    rd(In, 0, wf[0], xf[0]);

    TileC prev = coords(first);
    bool prev_live = false;
    int tile = first;
    for (;;) {
        run_tile(pipe::IC<0>{}, tile, prev, prev_live);
        prev = coords(tile); prev_live = true;
        tile += stride;
        if (tile >= tg.ntiles) {
#pragma unroll
            for (int q = 0; q < (POOL ? 2 : 4); ++q) epilogue_piece(q, accs[0], prev, true);
            break;
        }
        run_tile(pipe::IC<1>{}, tile, prev, prev_live);
        prev = coords(tile);
        tile += stride;
        if (tile >= tg.ntiles) {
#pragma unroll
            for (int q = 0; q < (POOL ? 2 : 4); ++q) epilogue_piece(q, accs[1], prev, true);
            break;
        }
    }
}

template <typename T, bool POOL>
static int launch_conv_c64p(const ConvArgs& a, int n, int h, int w, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        constexpr int smem = 64 * 1184 + 2 * (340 * 128 + 512);
        static_assert(smem <= 160 * 1024, "LDS budget");
        static DbxDevOnce attr_once; int attr_dev = 0;
        if (attr_once.pending(&attr_dev)) {
            DBX_HIP(hipFuncSetAttribute((const void*)conv3x3_c64p_kernel<T, POOL>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
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
        tg.tiles_x = (w + 31) / 32; tg.tiles_y = (h + 7) / 8; tg.ntiles = n * tg.tiles_x * tg.tiles_y; tg.H = h; tg.W = w;
        // one persistent workgroup per CU, a multiple of the cout slices
        int grid = ncu / a.ntile_n * a.ntile_n;
        if (grid > tg.ntiles * a.ntile_n) grid = tg.ntiles * a.ntile_n;
        hipLaunchKernelGGL((conv3x3_c64p_kernel<T, POOL>), dim3(grid), dim3(512), smem, s, a, tg);
        DBX_LAUNCH_CHECK();
    }
    return DBX_OK;
}
