// SGD(momentum, weight decay) multi-tensor update and the inference tail: top-K decode + greedy NMS.
// Index bookkeeping must be bit-exact against NumPy/torch-CPU, so floating-point contraction is OFF in this
// file: `areas[i] + areas[j] - w*h` must round the product before the subtraction like NumPy does.
#pragma clang fp contract(off)
#include "common.hpp"

// ---------------------------------------------------------------------------------------------- SGD (DenseBox.py:2001-2004)
// torch.optim.SGD, dampening 0, no Nesterov: g = grad + wd*p; buf = g (first step) | mu*buf + g; p -= lr*buf
// guard (round 6): dbx_grad_guard leaves step_id in guard[0] when any gradient element of the step is not finite (f16 training keeps its
// activation gradients in 16-bit frames and the reference loss is an un-normalised sum: an overflow there reaches every weight gradient
// behind it as inf / NaN); an update launched with that guard and the same step_id then returns without touching anything and counts
// the skipped step in guard[1].  guard == NULL: the plain update.
__global__ void sgd_kernel(float* const* __restrict__ ptrs, const long long* __restrict__ sizes, float lr, float mu, float wd,
                           int first, int* __restrict__ guard, int step_id) {
    if (guard && guard[0] == step_id) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(guard + 1, 1);
        return;
    }
    const int t = blockIdx.y;
    float* p = ptrs[3 * t];
    const float* g = ptrs[3 * t + 1];
    float* b = ptrs[3 * t + 2];
    const long long n = sizes[t];
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        p[i] = dbx_sgd_update(p[i], g[i], b + i, lr, mu, wd, first);
    }
}
extern "C" int dbx_sgd_step_guarded(float* const* ptrs, const int64_t* sizes, int32_t count, int64_t max_size, float lr, float momentum,
                                    float weight_decay, int32_t first_step, int32_t* guard, int32_t step_id, void* stream) {
    DBX_REQUIRE(ptrs && sizes && count > 0, "sgd: empty parameter list");
    int bx = (int)((max_size + 255) / 256);
    bx = bx < 1 ? 1 : (bx > 512 ? 512 : bx);
    hipLaunchKernelGGL(sgd_kernel, dim3(bx, count), dim3(256), 0, (hipStream_t)stream, ptrs, (const long long*)sizes, lr, momentum,
                       weight_decay, first_step, guard, step_id);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_sgd_step(float* const* ptrs, const int64_t* sizes, int32_t count, int64_t max_size, float lr, float momentum,
                            float weight_decay, int32_t first_step, void* stream) {
    return dbx_sgd_step_guarded(ptrs, sizes, count, max_size, lr, momentum, weight_decay, first_step, nullptr, 0, stream);
}
// one pass over the step's flat gradient buffer (46.7 MB for DenseBoxLMLOC: ~12 us): |x| with the exponent field all ones = inf or NaN
__global__ __launch_bounds__(256) void grad_guard_kernel(const float* __restrict__ g, long long n, int* __restrict__ guard, int step_id) {
    const long long n4 = n >> 2;
    const u32x4* g4 = (const u32x4*)g;
    unsigned bad = 0u;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const u32x4 v = g4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) bad |= ((v[e] & 0x7f800000u) == 0x7f800000u) ? 1u : 0u;
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n & 3))
        bad |= ((__float_as_uint(g[(n4 << 2) + threadIdx.x]) & 0x7f800000u) == 0x7f800000u) ? 1u : 0u;
    if (__any((int)bad) && (threadIdx.x & 63) == 0) atomicMax(guard, step_id);
}
extern "C" int dbx_grad_guard(const float* grads, int64_t n, int32_t* guard, int32_t step_id, void* stream) {
    DBX_REQUIRE(grads && guard && n >= 0 && step_id > 0 && ((size_t)grads % 16) == 0, "grad_guard: 16-byte aligned gradients, a guard word pair, step ids from 1");
    if (n == 0) return DBX_OK;
    long long blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(grad_guard_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, grads, (long long)n, guard, step_id);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

// ---------------------------------------------------------------------------------------------- top-K + decode
#define DET_THREADS 1024

__device__ __forceinline__ void block_argmax_g(const float* vals, int n, float* red_v, int* red_i, float& ov, int& oi) {
    const int tid = threadIdx.x;
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = tid; i < n; i += DET_THREADS) {
        const float v = vals[i];
        if (v > bv || bi == 0x7fffffff) { bv = v; bi = i; }        // lower index wins ties; first element seeds
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float v2 = __shfl_down(bv, off); const int i2 = __shfl_down(bi, off);
        if (i2 != 0x7fffffff && (bi == 0x7fffffff || v2 > bv || (v2 == bv && i2 < bi))) { bv = v2; bi = i2; }
    }
    if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < DET_THREADS / 64; ++w) {
            const float v2 = red_v[w]; const int i2 = red_i[w];
            if (i2 != 0x7fffffff && (bi == 0x7fffffff || v2 > bv || (v2 == bv && i2 < bi))) { bv = v2; bi = i2; }
        }
        red_v[0] = bv; red_i[0] = bi;
    }
    __syncthreads();
    ov = red_v[0]; oi = red_i[0];
    __syncthreads();
}

// the same reduction over one cached (value, index) candidate per thread
__device__ __forceinline__ void block_argmax_cached(float bv, int bi, float* red_v, int* red_i, float& ov, int& oi) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float v2 = __shfl_down(bv, off); const int i2 = __shfl_down(bi, off);
        if (i2 != 0x7fffffff && (bi == 0x7fffffff || v2 > bv || (v2 == bv && i2 < bi))) { bv = v2; bi = i2; }
    }
    if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < DET_THREADS / 64; ++w) {
            const float v2 = red_v[w]; const int i2 = red_i[w];
            if (i2 != 0x7fffffff && (bi == 0x7fffffff || v2 > bv || (v2 == bv && i2 < bi))) { bv = v2; bi = i2; }
        }
        red_v[0] = bv; red_i[0] = bi;
    }
    __syncthreads();
    ov = red_v[0]; oi = red_i[0];
    __syncthreads();
}

// greedy NMS over dets[n][dc] (float64): keep list in `keep` (keep[0] = count).  order = score descending, ties by
// higher row index first (= numpy argsort(stable)[::-1]; DenseBox.py:3415).
#define NMS_LDS_MAX 1024
__device__ void nms_block(const double* dets, int n, int dc, double thresh, int* keep, int* order, unsigned char* supp,
                          unsigned long long* mask = nullptr) {
    const int tid = threadIdx.x, nt = blockDim.x;
    // (NaN scores compare false both ways: the ranks below are then not a permutation -- a diverged network must give a strange
    // order, never an out-of-range row index: every slot starts as a valid row)
    for (int i = tid; i < n; i += nt) order[i] = i;
    __syncthreads();
    for (int i = tid; i < n; i += nt) {
        const double si = dets[(size_t)i * dc + 4];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double sj = dets[(size_t)j * dc + 4];
            rank += (sj > si) || (sj == si && j > i);
        }
        order[rank] = i;
        supp[i] = 0;
    }
    __syncthreads();
    if (n <= NMS_LDS_MAX) {
        // boxes in rank order, areas and suppression flags staged in LDS: a greedy round is one LDS pass + one barrier
        __shared__ double bx1[NMS_LDS_MAX], by1[NMS_LDS_MAX], bx2[NMS_LDS_MAX], by2[NMS_LDS_MAX], bar[NMS_LDS_MAX];
        __shared__ unsigned char sp[NMS_LDS_MAX];
        for (int q = tid; q < n; q += nt) {
            const double* d = dets + (size_t)order[q] * dc;
            bx1[q] = d[0]; by1[q] = d[1]; bx2[q] = d[2]; by2[q] = d[3];
            bar[q] = (d[2] - d[0] + 1) * (d[3] - d[1] + 1);
            sp[q] = 0;
        }
        __syncthreads();
        if (mask != nullptr && n > 64) {
            // Many boxes (K = 1000): the greedy loop above costs one barrier per surviving box (~0.6 ms).  Instead (1) every thread
            // fills one row of the suppression matrix -- bit q of row p (q > p, rank order): box p would suppress box q, with the
            // very same fp64 expression -- into `mask` (n x 16 words, global scratch), all pairs in parallel; (2) ONE wave walks the
            // rows in rank order with the removed-set in registers (lane w holds word w), rows fetched 32 at a time.
            const int nw = (n + 63) >> 6;
            for (int p_ = tid; p_ < n; p_ += nt) {
                const double x1 = bx1[p_], y1 = by1[p_], x2 = bx2[p_], y2 = by2[p_], ai = bar[p_];
                for (int w = 0; w < nw; ++w) {
                    unsigned long long bits = 0ull;
                    if (64 * w + 63 > p_) {
                        for (int b = 0; b < 64; ++b) {
                            const int q = 64 * w + b;
                            if (q <= p_ || q >= n) continue;
                            const double xx1 = fmax(x1, bx1[q]), yy1 = fmax(y1, by1[q]), xx2 = fmin(x2, bx2[q]), yy2 = fmin(y2, by2[q]);
                            const double ww = fmax(0.0, xx2 - xx1 + 1), hh = fmax(0.0, yy2 - yy1 + 1);
                            const double inter = ww * hh;
                            const double ovr = inter / (ai + bar[q] - inter);
                            if (!(ovr <= thresh)) bits |= 1ull << b;             // NaN is dropped, like np.where(ovr <= t)
                        }
                    }
                    mask[(size_t)p_ * 16 + w] = bits;
                }
            }
            __threadfence_block();
            __syncthreads();
            if (tid < 64) {
                const int lane = tid;
                unsigned long long removed = 0ull;                  // lanes >= nw carry nothing
                int cnt = 0;
                for (int p0 = 0; p0 < n; p0 += 32) {
                    unsigned long long rows[32];
#pragma unroll
                    for (int r = 0; r < 32; ++r) rows[r] = (lane < nw && p0 + r < n) ? mask[(size_t)(p0 + r) * 16 + lane] : 0ull;
#pragma unroll
                    for (int r = 0; r < 32; ++r) {
                        const int pos = p0 + r;
                        if (pos >= n) break;
                        const unsigned long long word = __shfl(removed, pos >> 6);
                        if (!((word >> (pos & 63)) & 1ull)) {          // (uniform) box `pos` survives
                            if (lane == 0) keep[1 + cnt] = order[pos];
                            ++cnt;
                            removed |= rows[r];
                        }
                    }
                }
                if (lane == 0) keep[0] = cnt;
            }
            return;
        }
        int cnt = 0;
        for (int pos = 0; pos < n; ++pos) {
            if (sp[pos]) continue;                              // uniform: sp[] only changes between barriers
            if (tid == 0) keep[1 + cnt] = order[pos];
            ++cnt;
            const double x1 = bx1[pos], y1 = by1[pos], x2 = bx2[pos], y2 = by2[pos], ai = bar[pos];
            for (int q = pos + 1 + tid; q < n; q += nt) {
                if (sp[q]) continue;
                const double xx1 = fmax(x1, bx1[q]), yy1 = fmax(y1, by1[q]), xx2 = fmin(x2, bx2[q]), yy2 = fmin(y2, by2[q]);
                const double w = fmax(0.0, xx2 - xx1 + 1), h = fmax(0.0, yy2 - yy1 + 1);
                const double inter = w * h;
                const double ovr = inter / (ai + bar[q] - inter);
                if (!(ovr <= thresh)) sp[q] = 1;                 // NaN is dropped, like np.where(ovr <= t)
            }
            __syncthreads();
        }
        if (tid == 0) keep[0] = cnt;
        return;
    }
    int cnt = 0;
    for (int pos = 0; pos < n; ++pos) {
        const int i = order[pos];
        if (supp[i]) continue;                                  // uniform: supp[] only changes between barriers
        if (tid == 0) keep[1 + cnt] = i;
        ++cnt;
        const double x1 = dets[(size_t)i * dc], y1 = dets[(size_t)i * dc + 1], x2 = dets[(size_t)i * dc + 2], y2 = dets[(size_t)i * dc + 3];
        const double ai = (x2 - x1 + 1) * (y2 - y1 + 1);
        for (int q = pos + 1 + tid; q < n; q += nt) {
            const int j = order[q];
            if (supp[j]) continue;
            const double u1 = dets[(size_t)j * dc], v1 = dets[(size_t)j * dc + 1], u2 = dets[(size_t)j * dc + 2], v2 = dets[(size_t)j * dc + 3];
            const double aj = (u2 - u1 + 1) * (v2 - v1 + 1);
            const double xx1 = fmax(x1, u1), yy1 = fmax(y1, v1), xx2 = fmin(x2, u2), yy2 = fmin(y2, v2);
            const double w = fmax(0.0, xx2 - xx1 + 1), h = fmax(0.0, yy2 - yy1 + 1);
            const double inter = w * h;
            const double ovr = inter / (ai + aj - inter);
            if (!(ovr <= thresh)) supp[j] = 1;                   // NaN is dropped, like np.where(ovr <= t)
        }
        __syncthreads();
    }
    if (tid == 0) keep[0] = cnt;
}

struct DetArgs {
    const float* score; const float* loc; const float* lm_heat; const float* lm_loc;
    int rows, cols, K, dc; double thresh;
    double* dets; long long* topk; int* keep; float* work; int* order; unsigned char* supp; unsigned long long* mask;
};

// order-preserving key of a score for the radix select: larger float <-> larger key, -0 == +0, NaN below every number
__device__ __forceinline__ unsigned det_key(float v) {
    if (v != v) return 0u;
    if (v == 0.f) return 0x80000000u;
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// top-K by radix select + sort from this K on; below it the arg-max rounds over the two-level LDS structure are faster (same-box
// A/B at K = 10 on a 128 x 128 map: whole 512 x 512 detect() 0.486 ms with rounds, 0.506 ms with select + a 16-element sort).
#ifndef DET_SELECT_MIN_K
#define DET_SELECT_MIN_K 49
#endif
__global__ __launch_bounds__(DET_THREADS) void detect_kernel(const DetArgs a) {
    __shared__ float red_v[DET_THREADS / 64];
    __shared__ int red_i[DET_THREADS / 64];
    __shared__ int lm_arg[4];
    const int tid = threadIdx.x, n = a.rows * a.cols;
    constexpr int BK = 64, NB_MAX = 4096;
    __shared__ float bmax[NB_MAX];
    __shared__ int bidx[NB_MAX];
    // landmark arg-max per heat-map channel (parse_DetLM, DenseBox.py:3284-3292): identical for every detection
    if (a.lm_heat && !a.lm_loc) {
        for (int j = 0; j < 4; ++j) {
            float v; int idx;
            block_argmax_g(a.lm_heat + (size_t)j * n, n, red_v, red_i, v, idx);
            if (tid == 0) lm_arg[j] = idx;
        }
        __syncthreads();
    }
    // ---- top-K indices into a.topk, in the reference's order: larger score first, lower index on ties
    const bool select = a.K >= DET_SELECT_MIN_K && a.K <= DET_THREADS;
    const bool tourney = !select && a.K <= 48 && n <= 16 * DET_THREADS;
    if (tourney) {
        // Small K on a map of <= 16384 scores (the 512 x 512 input's 128 x 128 map, K = 10): every lane keeps its 16 scores in registers
        // as (key << 32 | ~index) composites -- larger composite = larger score, lower index on ties, the select path's order -- each
        // wave extracts ITS top K by K rounds of a register maximum + a wave reduction (shuffles only, no workgroup barrier), the 16
        // waves leave their sorted lists in LDS and wave 0 merges the 16 K candidates the same way: ~6 us instead of K block-wide
        // arg-max rounds with three barriers each (30 us at K = 10).
        unsigned long long* wcand = (unsigned long long*)bidx;            // [16][48]
        const int lane = tid & 63, wv = tid >> 6;
        unsigned long long comp[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = tid + e * DET_THREADS;
            comp[e] = i < n ? ((unsigned long long)det_key(a.score[i]) << 32) | (unsigned)(~(unsigned)i) : 0ull;
        }
        auto wave_max = [&](unsigned long long v) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(v, off); v = o > v ? o : v; }
            return v;
        };
        for (int r = 0; r < a.K; ++r) {
            unsigned long long best = comp[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) best = comp[e] > best ? comp[e] : best;
            best = wave_max(best);
#pragma unroll
            for (int e = 0; e < 16; ++e) comp[e] = comp[e] == best ? 0ull : comp[e];
            if (lane == 0) wcand[wv * 48 + r] = best;
        }
        __syncthreads();
        if (wv == 0) {
            const int tot = (DET_THREADS / 64) * a.K;                        // <= 768: 12 per lane
            unsigned long long c2[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int f = lane + 64 * j;
                c2[j] = f < tot ? wcand[(f / a.K) * 48 + f % a.K] : 0ull;
            }
            for (int r = 0; r < a.K; ++r) {
                unsigned long long best = c2[0];
#pragma unroll
                for (int j = 1; j < 12; ++j) best = c2[j] > best ? c2[j] : best;
                best = wave_max(best);
#pragma unroll
                for (int j = 0; j < 12; ++j) c2[j] = c2[j] == best ? 0ull : c2[j];
                if (lane == 0) a.topk[r] = (long long)(~(unsigned)(best & 0xffffffffull));
            }
        }
        __syncthreads();
    } else if (select) {
        // K in (48, 1024] (round 3: K = 1000 at 1080p took 4.7 ms as 1000 arg-max rounds): radix select of the K-th largest key
        // (four 8-bit passes, LDS histogram), compaction of the keys above it plus the lowest-index ties, bitonic sort of <= 1024
        // (key, index) pairs in LDS -- ~0.1 ms, the same ranking bit for bit.
        unsigned* hist = (unsigned*)bmax;                        // [256]
        unsigned long long* cand = (unsigned long long*)bidx;    // [1024] (key << 32) | ~index: descending sort = reference order
        __shared__ unsigned sel_prefix, sel_remaining, sel_ties, cand_n, tie_base[DET_THREADS / 64 + 1];
        if (tid == 0) { sel_prefix = 0; sel_remaining = (unsigned)a.K; }
        __syncthreads();
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = sel_prefix;
            for (int i = tid; i < n; i += DET_THREADS) {
                const unsigned k = det_key(a.score[i]);
                if (pass == 0 || (k >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(k >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned rem = sel_remaining, b = 255;
                for (;; --b) { const unsigned c = hist[b]; if (c >= rem || b == 0) break; rem -= c; }
                sel_prefix = prefix | (b << shift); sel_remaining = rem; sel_ties = hist[b];
            }
            __syncthreads();
        }
        const unsigned T = sel_prefix, need = sel_remaining, ties = sel_ties;     // K-th key; how many of its `ties` copies are taken
        if (tid == 0) cand_n = 0;
        __syncthreads();
        for (int i = tid; i < n; i += DET_THREADS) {
            const unsigned k = det_key(a.score[i]);
            if (k > T || (k == T && ties == need)) {
                const unsigned slot = atomicAdd(&cand_n, 1u);
                cand[slot] = ((unsigned long long)k << 32) | (unsigned)(~(unsigned)i);
            }
        }
        __syncthreads();
        if (ties != need) {
            // more copies of the K-th score than places: the lowest indices win.  Thread t owns the contiguous index range
            // [t * seg, (t + 1) * seg): per-thread tie counts -> exclusive block scan -> the first `need` ties in index order.
            const int seg = (n + DET_THREADS - 1) / DET_THREADS, lo = tid * seg, hi = min(n, lo + seg);
            unsigned mine = 0;
            for (int i = lo; i < hi; ++i) mine += det_key(a.score[i]) == T;
            unsigned incl = mine;                                   // inclusive scan inside the wave
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(incl, off); if ((tid & 63) >= off) incl += o; }
            if ((tid & 63) == 63) tie_base[(tid >> 6) + 1] = incl;
            __syncthreads();
            if (tid == 0) { tie_base[0] = 0; for (int w = 1; w <= DET_THREADS / 64; ++w) tie_base[w] += tie_base[w - 1]; }
            __syncthreads();
            unsigned before = tie_base[tid >> 6] + incl - mine;
            const unsigned base_slot = cand_n;
            for (int i = lo; i < hi && before < need; ++i)
                if (det_key(a.score[i]) == T) { cand[base_slot + before] = ((unsigned long long)T << 32) | (unsigned)(~(unsigned)i); ++before; }
            __syncthreads();
        }
        int P2 = 2;                                               // sort size: the power of two >= K (K = 10: 16 elements, 10 exchange steps)
        while (P2 < a.K) P2 <<= 1;
        for (int i = a.K + tid; i < P2; i += DET_THREADS) cand[i] = 0ull;                // padding sorts last
        __syncthreads();
        // bitonic sort, descending, P2 <= 1024 elements: one element per thread
        for (int size = 2; size <= P2; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                const int partner = tid ^ stride;
                unsigned long long me = 0, ot = 0;
                if (tid < P2) { me = cand[tid]; ot = cand[partner]; }
                __syncthreads();
                const bool desc = (tid & size) == 0;                // this block sorts descending
                const bool keep_max = (tid < partner) == desc;
                if (tid < P2) cand[tid] = keep_max ? (me > ot ? me : ot) : (me < ot ? me : ot);
                __syncthreads();
            }
        }
        if (tid < a.K) a.topk[tid] = (long long)(~(unsigned)(cand[tid] & 0xffffffffull));
        __syncthreads();
    } else {
        // the retire-and-rescan rounds go through a working copy of the scores: in LDS when the map fits (<= 128 x 128, the 512 x 512
        // input: a round's store -> 64 reloads of the winner's bucket is an LDS round trip instead of an L2 one), else in global scratch
        constexpr int WORK_LDS = 16384;
        __shared__ float work_lds[WORK_LDS];
        float* const work = n <= WORK_LDS ? work_lds : a.work;
        for (int i = tid; i < n; i += DET_THREADS) work[i] = a.score[i];
        __syncthreads();
        // K rounds of arg-max over a two-level structure: LDS holds the (max, arg-max) of every bucket of 64 consecutive scores;
        // a round reduces the bucket maxima (LDS only) and one wave re-scans the winner's bucket (64 loads in flight at once),
        // instead of every thread re-reading its share of the whole map from global memory: ~4 us per round instead of 24.
        // Same order as a full scan: larger value first, lower index on ties, NaN never beats a number.
        const int nb = (n + BK - 1) / BK;
        const int lane = tid & 63, wv = tid >> 6;
        const bool two_level = nb <= NB_MAX;
        auto scan_bucket = [&](int b) {                              // one wave: arg-max of bucket b -> LDS
            const int i = b * BK + lane;
            float v = i < n ? work[i] : -INFINITY; int vi = i < n ? i : 0x7fffffff;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float v2 = __shfl_down(v, off); const int i2 = __shfl_down(vi, off);
                if (i2 != 0x7fffffff && (vi == 0x7fffffff || v2 > v || (v2 == v && i2 < vi))) { v = v2; vi = i2; }
            }
            if (lane == 0) { bmax[b] = v; bidx[b] = vi; }
        };
        if (two_level) {
            for (int b = wv; b < nb; b += DET_THREADS / 64) scan_bucket(b);
            __syncthreads();
        }
        float bv = -INFINITY; int bi = 0x7fffffff;
        auto rescan = [&]() {                                        // fallback for huge maps: per-thread cached candidate
            bv = -INFINITY; bi = 0x7fffffff;
            for (int i = tid; i < n; i += DET_THREADS) {
                const float v = work[i];
                if (v > bv || bi == 0x7fffffff) { bv = v; bi = i; }
            }
        };
        if (!two_level) rescan();
        for (int k = 0; k < a.K; ++k) {
            float v; int idx;
            if (two_level) {
                float cv = -INFINITY; int ci = 0x7fffffff;
                for (int b = tid; b < nb; b += DET_THREADS) {
                    const float v2 = bmax[b]; const int i2 = bidx[b];
                    if (i2 != 0x7fffffff && (ci == 0x7fffffff || v2 > cv || (v2 == cv && i2 < ci))) { cv = v2; ci = i2; }
                }
                block_argmax_cached(cv, ci, red_v, red_i, v, idx);
                if (wv == 0) {                                       // wave 0 retires the winner and refreshes its bucket
                    if (lane == 0) work[idx] = -INFINITY;
                    __builtin_amdgcn_wave_barrier();
                    __threadfence_block();
                    scan_bucket(idx / BK);
                }
            } else {
                block_argmax_cached(bv, bi, red_v, red_i, v, idx);
                if ((idx & (DET_THREADS - 1)) == tid) { work[idx] = -INFINITY; rescan(); }
            }
            if (tid == 0) a.topk[k] = idx;
            __syncthreads();
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- decode: one thread per detection (the rows no longer sit as dependent global loads inside the selection rounds)
    for (int k = tid; k < a.K; k += DET_THREADS) {
        const int idx = (int)a.topk[k];
        const float xi = (float)(idx % a.cols), yi = (float)(idx / a.cols);
        double* d = a.dets + (size_t)k * a.dc;
        // fp32 subtraction (python int - fp32 tensor), then float()*4.0 in double (DenseBox.py:3334-3343)
        d[0] = (double)(xi - a.loc[idx]) * 4.0;
        d[1] = (double)(yi - a.loc[(size_t)n + idx]) * 4.0;
        d[2] = (double)(xi - a.loc[(size_t)2 * n + idx]) * 4.0;
        d[3] = (double)(yi - a.loc[(size_t)3 * n + idx]) * 4.0;
        d[4] = (double)a.score[idx];
        if (a.dc == 13) {
            if (a.lm_loc) {
                for (int c = 0; c < 8; ++c)
                    d[5 + c] = (double)(((c & 1) ? yi : xi) - a.lm_loc[(size_t)c * n + idx]) * 4.0;   // :3183-3196
            } else {
                for (int j = 0; j < 4; ++j) {
                    d[5 + 2 * j] = (double)(float)(lm_arg[j] % a.cols) * 4.0;
                    d[6 + 2 * j] = (double)(float)(lm_arg[j] / a.cols) * 4.0;
                }
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    nms_block(a.dets, a.K, a.dc, a.thresh, a.keep, a.order, a.supp, a.mask);
}

extern "C" int64_t dbx_detect_scratch_bytes(int32_t rows, int32_t cols, int32_t K) {
    // scores copy, NMS order, suppression flags, + the 16-word-per-box suppression matrix when the boxes fit the LDS path
    return (int64_t)rows * cols * 4 + (int64_t)K * 4 + ((int64_t)K + 255) / 256 * 256 + (K <= NMS_LDS_MAX ? (int64_t)K * 128 : 0) + 256;
}

extern "C" int dbx_detect(const float* score, const float* loc, const float* lm_heat, const float* lm_loc, int32_t rows,
                          int32_t cols, int32_t K, double nms_thresh, double* dets, int32_t det_cols, int64_t* topk_idx,
                          int32_t* keep, void* scratch, void* stream) {
    DBX_REQUIRE(score && loc && dets && topk_idx && keep && scratch, "detect: null argument");
    DBX_REQUIRE(K > 0 && K <= rows * cols, "detect: K=%d out of range", K);
    DBX_REQUIRE(det_cols == 5 || (det_cols == 13 && (lm_heat || lm_loc)), "detect: det_cols must be 5, or 13 with landmark maps");
    DetArgs a;
    a.score = score; a.loc = loc; a.lm_heat = lm_heat; a.lm_loc = lm_loc;
    a.rows = rows; a.cols = cols; a.K = K; a.dc = det_cols; a.thresh = nms_thresh;
    a.dets = dets; a.topk = (long long*)topk_idx; a.keep = keep;
    char* s = (char*)scratch;
    a.work = (float*)s; s += (size_t)rows * cols * 4;
    a.order = (int*)s; s += (size_t)K * 4;
    a.supp = (unsigned char*)s; s += ((size_t)K + 255) / 256 * 256;
    s = (char*)(((size_t)s + 7) & ~(size_t)7);                          // (inside the 256 spare bytes)
    a.mask = K <= NMS_LDS_MAX ? (unsigned long long*)s : nullptr;
    hipLaunchKernelGGL(detect_kernel, dim3(1), dim3(DET_THREADS), 0, (hipStream_t)stream, a);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

__global__ __launch_bounds__(DET_THREADS) void nms_kernel(const double* dets, int n, int dc, double thresh, int* keep, int* order,
                                                          unsigned char* supp) {
    nms_block(dets, n, dc, thresh, keep, order, supp);
}
extern "C" int dbx_nms(const double* dets, int32_t n, int32_t det_cols, double nms_thresh, int32_t* keep, void* scratch,
                       void* stream) {
    DBX_REQUIRE(dets && keep && scratch && n > 0 && det_cols >= 5, "nms: bad arguments");
    int* order = (int*)scratch;
    unsigned char* supp = (unsigned char*)scratch + (size_t)n * 4;
    hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(DET_THREADS), 0, (hipStream_t)stream, dets, n, det_cols, nms_thresh, keep, order, supp);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

// ---------------------------------------------------------------------------------------------- plate rectification
// perspective_transform (DenseBox.py:3446-3481): homography from the four landmark corners to their axis-aligned bounding
// rectangle (cv2.getPerspectiveTransform) and a warp of the whole image to 1.5x its size (cv2.warpPerspective, default
// INTER_LINEAR, constant border 0).  OpenCV is not part of the reference tree (and not installed here), so both follow its
// PUBLISHED algorithm (imgwarp.cpp): 8x8 system in double solved by LU with partial pivoting; inverse map through the
// 3x3 inverse; source coordinates rounded to 1/32 pixel (INTER_BITS = 5); 8-bit bilinear weights in 15-bit fixed point,
// (sum + 2^14) >> 15.  Parity with OpenCV itself is unpinned; the GPU kernel is bit-exact against oracle/.
extern "C" int dbx_perspective_matrix(const float* src_xy, const float* dst_xy, double* m9) {
    DBX_REQUIRE(src_xy && dst_xy && m9, "perspective_matrix: null argument");
    double A[8][9];
    for (int i = 0; i < 4; ++i) {
        const double sx = src_xy[2 * i], sy = src_xy[2 * i + 1], dx = dst_xy[2 * i], dy = dst_xy[2 * i + 1];
        const double r0[9] = {sx, sy, 1, 0, 0, 0, -sx * dx, -sy * dx, dx};
        const double r1[9] = {0, 0, 0, sx, sy, 1, -sx * dy, -sy * dy, dy};
        for (int j = 0; j < 9; ++j) { A[i][j] = r0[j]; A[i + 4][j] = r1[j]; }
    }
    for (int c = 0; c < 8; ++c) {                      // Gaussian elimination, partial pivoting (DECOMP_LU)
        int piv = c;
        for (int r = c + 1; r < 8; ++r) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        DBX_REQUIRE(fabs(A[piv][c]) > 2.220446049250313e-16, "perspective_matrix: degenerate corner configuration");
        if (piv != c) for (int j = 0; j < 9; ++j) { const double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
        const double d = -1.0 / A[c][c];
        for (int r = c + 1; r < 8; ++r) {
            const double f = A[r][c] * d;
            for (int j = c + 1; j < 9; ++j) A[r][j] += f * A[c][j];
        }
    }
    double x[8];
    for (int r = 7; r >= 0; --r) {
        double acc = A[r][8];
        for (int j = r + 1; j < 8; ++j) acc -= A[r][j] * x[j];
        x[r] = acc / A[r][r];
    }
    for (int i = 0; i < 8; ++i) m9[i] = x[i];
    m9[8] = 1.0;
    return DBX_OK;
}

struct WarpArgs { const unsigned char* src; unsigned char* dst; int sh, sw, c, dh, dw; double im[9]; };
__global__ void warp_perspective_u8_kernel(const WarpArgs a) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.dw || y >= a.dh) return;
    const double X0 = a.im[0] * x + a.im[1] * y + a.im[2], Y0 = a.im[3] * x + a.im[4] * y + a.im[5];
    double W = a.im[6] * x + a.im[7] * y + a.im[8];
    W = W != 0.0 ? 32.0 / W : 0.0;
    const double fX = fmax(-2147483648.0, fmin(2147483647.0, X0 * W)), fY = fmax(-2147483648.0, fmin(2147483647.0, Y0 * W));
    const int X = (int)rint(fX), Y = (int)rint(fY);                          // cvRound: nearest, ties to even
    const int sx = X >> 5, sy = Y >> 5, ax = X & 31, ay = Y & 31;
    const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
    const bool x0 = sx >= 0 && sx < a.sw, x1 = sx + 1 >= 0 && sx + 1 < a.sw, y0 = sy >= 0 && sy < a.sh, y1 = sy + 1 >= 0 && sy + 1 < a.sh;
    for (int ch = 0; ch < a.c; ++ch) {
        const int p00 = (x0 && y0) ? a.src[((size_t)sy * a.sw + sx) * a.c + ch] : 0;
        const int p01 = (x1 && y0) ? a.src[((size_t)sy * a.sw + sx + 1) * a.c + ch] : 0;
        const int p10 = (x0 && y1) ? a.src[((size_t)(sy + 1) * a.sw + sx) * a.c + ch] : 0;
        const int p11 = (x1 && y1) ? a.src[((size_t)(sy + 1) * a.sw + sx + 1) * a.c + ch] : 0;
        const int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
        a.dst[((size_t)y * a.dw + x) * a.c + ch] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}
extern "C" int dbx_warp_perspective_u8(const uint8_t* src, int32_t sh, int32_t sw, int32_t c, const double* m9, uint8_t* dst,
                                       int32_t dh, int32_t dw, void* stream) {
    DBX_REQUIRE(src && dst && m9 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && c >= 1 && c <= 4, "warp_perspective: bad arguments");
    // inverse of the 3x3 map (dst -> src), cofactor form in double
    const double* m = m9;
    const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    DBX_REQUIRE(det != 0.0, "warp_perspective: singular matrix");
    const double d = 1.0 / det;
    WarpArgs a;
    a.src = src; a.dst = dst; a.sh = sh; a.sw = sw; a.c = c; a.dh = dh; a.dw = dw;
    a.im[0] = (m[4] * m[8] - m[5] * m[7]) * d; a.im[1] = (m[2] * m[7] - m[1] * m[8]) * d; a.im[2] = (m[1] * m[5] - m[2] * m[4]) * d;
    a.im[3] = (m[5] * m[6] - m[3] * m[8]) * d; a.im[4] = (m[0] * m[8] - m[2] * m[6]) * d; a.im[5] = (m[2] * m[3] - m[0] * m[5]) * d;
    a.im[6] = (m[3] * m[7] - m[4] * m[6]) * d; a.im[7] = (m[1] * m[6] - m[0] * m[7]) * d; a.im[8] = (m[0] * m[4] - m[1] * m[3]) * d;
    hipLaunchKernelGGL(warp_perspective_u8_kernel, dim3((dw + 255) / 256, dh), dim3(256), 0, (hipStream_t)stream, a);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
