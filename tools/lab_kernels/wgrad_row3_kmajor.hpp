// Experiment (round 2), not compiled into the library: wgrad_row3 with K-major LDS tiles (register transpose at staging,
// ds_read_b128 fragments).  Correct, but 15 % slower than the transpose-read kernel (conv4_2: 341 vs 296 us): the loop is not
// bound by the LDS read rate but by the read latency exposed after every barrier.  Kept for reference; see DESIGN.md.
// K-major staging (the kernel below): a thread fetches a 4(q) x 4(channel) block with four 8-byte loads, transposes it in
// registers (8 v_perm_b32) and writes four 8-byte q-runs, so LDS holds [channel][64 q] and an MFMA operand fragment (8
// consecutive q of one channel) is ONE full-rate ds_read_b128 instead of two half-rate ds_read_b64_tr_b16: 1024 instead of
// 1850 LDS clocks per 64-row step of a workgroup, below the 1536 MFMA clocks.  16-byte chunks of a channel row are XOR-swizzled
// with (c & 7) ^ ((c >> 3) & 7): conflict-free for the fragment reads (8 consecutive channels -> 8 chunks) and for the
// transposed writes (8 channel quads x 2 row groups -> 16 distinct 8-byte slots).  Band rows 64..67 live in a small side tile.
__device__ __forceinline__ int kmaj(int c, int q) {
    return c * 128 + ((((q >> 3) ^ c ^ (c >> 3)) & 7) << 4) + (q & 7) * 2;
}

template <typename T>
__global__ __launch_bounds__(512) void wgrad_row3_kernel(const WgradArgs a) {
    static_assert(sizeof(T) == 2, "16-bit tiles");
    constexpr int R = 64;                                               // frame rows per K step (one global-latency period)
    constexpr int A_BYTES = 128 * 128, XB = 128 * 128, B_BYTES = XB + 128 * 8;   // [128 ch][64 q] (+ [128 ch][4 q] of band rows 64..67)
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
    const long long q0 = (long long)split * a.rows_per_split;
    long long q1 = q0 + a.rows_per_split; if (q1 > a.Q) q1 = a.Q;
    const int nsteps = q1 > q0 ? (int)((q1 - q0 + R - 1) / R) : 0;
    const bool do_bias = (tile_ci == 0 && ky == 0 && a.bpartial != nullptr);

    // loaders: lane -> channel quad cq (8 bytes of a row) and row group rg (rows 4rg..4rg+3); one load instruction covers two
    // whole 256-byte rows.  Threads 0..31 also fetch band rows 64..67.
    const int cq = (lane & 7) | ((lane >> 4) << 3), rg = 2 * wave + ((lane >> 3) & 1);
    const long long arow = (long long)a.dz_ld * 2, brow = (long long)a.x_ld * 2;
    const char* ap = a.dz + ((q0 + 4 * rg) * a.dz_ld + tile_co * 128) * 2LL + cq * 8;
    const long long xrow0 = q0 + (long long)(ky + a.shift0) * a.wp + a.shift0;
    const char* bp = a.x + ((xrow0 + 4 * rg) * a.x_ld + tile_ci * 128) * 2LL + cq * 8;
    const char* bp2 = a.x + ((xrow0 + 64) * a.x_ld + tile_ci * 128) * 2LL + (tid & 31) * 8;
    const bool has2 = tid < 32;
    int woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) woff[i] = kmaj(4 * cq + i, 4 * rg);
    u32x2 areg[4], breg[4], creg[4];
    auto gload = [&](int s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) areg[j] = *(const u32x2*)(ap + s * 64 * arow + j * arow);
#pragma unroll
        for (int j = 0; j < 4; ++j) breg[j] = *(const u32x2*)(bp + s * 64 * brow + j * brow);
        if (has2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) creg[j] = *(const u32x2*)(bp2 + s * 64 * brow + j * brow);
        }
    };
    // 4 rows x 4 channels (row j = {c0 c1 | c2 c3}) -> 4 channels x 4 rows
    auto tr4 = [&](const u32x2* r, u32x2* o) {
        o[0] = (u32x2){__builtin_amdgcn_perm(r[1].x, r[0].x, 0x05040100u), __builtin_amdgcn_perm(r[3].x, r[2].x, 0x05040100u)};
        o[1] = (u32x2){__builtin_amdgcn_perm(r[1].x, r[0].x, 0x07060302u), __builtin_amdgcn_perm(r[3].x, r[2].x, 0x07060302u)};
        o[2] = (u32x2){__builtin_amdgcn_perm(r[1].y, r[0].y, 0x05040100u), __builtin_amdgcn_perm(r[3].y, r[2].y, 0x05040100u)};
        o[3] = (u32x2){__builtin_amdgcn_perm(r[1].y, r[0].y, 0x07060302u), __builtin_amdgcn_perm(r[3].y, r[2].y, 0x07060302u)};
    };
    auto lstore = [&](int buf) {
        u32x2 o[4];
        tr4(areg, o);
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x2*)(As + buf * A_BYTES + woff[i]) = o[i];
        tr4(breg, o);
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x2*)(Bs + buf * B_BYTES + woff[i]) = o[i];
        if (has2) {
            tr4(creg, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) *(u32x2*)(Bs + buf * B_BYTES + XB + (4 * tid + i) * 8) = o[i];
        }
    };
    f32x4 acc[3][4][2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[k][mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    auto bias_acc = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const T* e = (const T*)&areg[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) bsum[i] += to_f32(e[i]);
        }
    };
    if (nsteps > 0) { gload(0); if (do_bias) bias_acc(); lstore(0); }
    __syncthreads();
    // fragment addresses: lane (n = lane & 15, g = lane >> 4) holds q = 32 kk + 8 g .. + 7 of channel n of its 16-block; kk = 1
    // flips bit 6 of the swizzled offset.  The 4-row tail of a 12-row run is chunk + 1 (the side tile for q = 64..67).
    const int n16 = lane & 15, g = lane >> 4;
    int aoff[4], boff[2], b2off[2][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) aoff[mi] = kmaj(wm * 64 + mi * 16 + n16, 8 * g);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int c = wn * 32 + ni * 16 + n16;
        boff[ni] = kmaj(c, 8 * g);
        b2off[ni][0] = kmaj(c, 8 * g + 8);
        b2off[ni][1] = g == 3 ? XB + c * 8 : kmaj(c, 40 + 8 * g);
    }
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps && !(a.dbg & 1)) gload(s + 1);
        const char* Ab = As + buf * A_BYTES;
        const char* Bb = Bs + buf * B_BYTES;
        // software pipeline over the four (kk, ni) units of 3 x 4 MFMAs: the 12-row run of the next unit and half of the
        // next kk's dz fragments are in flight while the current unit computes; sched_group_barrier pins the interleave.
        auto rdA = [&](int kk, int mi) { return *(const u32x4*)(Ab + (aoff[mi] ^ (kk * 64))); };
        struct Run { u32x4 lo; u32x2 hi; };     // rows r..r+11 of one channel column: pairs (0,1)(2,3)(4,5)(6,7) | (8,9)(10,11)
        auto rdRun = [&](int u) {
            Run r;
            r.lo = *(const u32x4*)(Bb + (boff[u & 1] ^ ((u >> 1) * 64)));
            r.hi = *(const u32x2*)(Bb + b2off[u & 1][u >> 1]);
            return r;
        };
        u32x4 af[2][4];
        Run run[2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[0][mi] = rdA(0, mi);
        run[0] = rdRun(0);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = u >> 1, ni = u & 1;
            const Run c = run[u & 1];
            if (u < 3) run[(u + 1) & 1] = rdRun(u + 1);
            if (u < 2) { af[1][2 * u] = rdA(1, 2 * u); af[1][2 * u + 1] = rdA(1, 2 * u + 1); }
            u32x4 bf[3];
            bf[0] = c.lo;
            bf[1] = (u32x4){__builtin_amdgcn_alignbit(c.lo.y, c.lo.x, 16), __builtin_amdgcn_alignbit(c.lo.z, c.lo.y, 16),
                            __builtin_amdgcn_alignbit(c.lo.w, c.lo.z, 16), __builtin_amdgcn_alignbit(c.hi.x, c.lo.w, 16)};
            bf[2] = (u32x4){c.lo.y, c.lo.z, c.lo.w, c.hi.x};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    if constexpr (DType<T>::id == DBX_F16)
                        acc[kx][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bf[kx]), __builtin_bit_cast(f16x8, af[kk][mi]), acc[kx][mi][ni], 0, 0, 0);
                    else
                        acc[kx][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[kx]), __builtin_bit_cast(bf16x8, af[kk][mi]), acc[kx][mi][ni], 0, 0, 0);
                }
                if (u < 2 && kx == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);         // 2 run reads + 2 dz reads
                else if (u < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                else if (u < 3 && kx < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 2 run reads
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
        if (s + 1 < nsteps && !(a.dbg & 2)) { if (do_bias) bias_acc(); lstore(buf ^ 1); }
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
        float* red = (float*)smem;                       // [16 row groups][128]
#pragma unroll
        for (int i = 0; i < 4; ++i) red[rg * 128 + 4 * cq + i] = bsum[i];
        __syncthreads();
        if (tid < 128) {
            float sum = 0.f;
            for (int r = 0; r < 16; ++r) sum += red[r * 128 + tid];
            a.bpartial[(long long)split * a.co_pad + tile_co * 128 + tid] = sum;
        }
    }
}

