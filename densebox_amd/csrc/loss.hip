// Dense per-pixel DenseBox loss on the 60x60 output grid, one workgroup per patch.
//
// Fuses what the reference does with ~10 host<->device hops per step:
//   label maps (DenseBox.py:1556-1914) -> element-wise L2 (:2056-2066) -> hard-negative top-k per sample
//   (:2070-2099) -> mask fill + gray zones (:1368-1553, :2105-2156) -> weighted sums (:2164-2180) -> dL/d(out).
// All index bookkeeping is integer / float32 / float64 arithmetic in exactly the reference's promotion order
// (see oracle/densebox_oracle.py::_rect), so masks and index lists are bit-exact.  HBM-bound and tiny
// (18 maps x 3600 x 4 B per patch): the point of the kernel is removing the syncs, not bandwidth.
#include "common.hpp"

#define HWD 60
#define NPIX 3600
#define LOSS_THREADS 1024

// ---------------------------------------------------------------------------------------------- integer helpers
__device__ __forceinline__ int norm_idx(int v) {            // python slice index normalisation on a dim of 60
    if (v < 0) { v += HWD; if (v < 0) v = 0; }
    else if (v > HWD) v = HWD;
    return v;
}
struct Span { int a, b; };                                  // [a, b)
__device__ __forceinline__ Span py_slice(int start, int stop) {
    Span s; s.a = norm_idx(start); s.b = norm_idx(stop); if (s.b < s.a) s.b = s.a; return s;
}
// centre rectangle along one axis (DenseBox.py:1572-1581 border=0, :1486-1497 border=2)
__device__ __forceinline__ void rect_axis(float c0, float c2, double border, int& org, int& end) {
    const double centre = (double)(c0 + c2) * 0.5;          // float32 add, then python float
    const float w = c2 - c0;
    const float rw = 0.3f * w;                              // ratio * bbox_w is a float32 product (NEP-50)
    const float half = rw * 0.5f;
    if (border == 0.0) {
        org = (int)(centre - (double)half + 0.5);
        end = (int)((double)org + (double)rw + 0.5);
    } else {
        org = (int)(centre - (double)half - border + 0.5);
        end = (int)((double)org + (double)rw + border * 2.0 + 0.5);
    }
}
struct Box { Span py, px, gzy, gzx, coy, cox; int valid; };
__device__ __forceinline__ Box make_box(const float* bb, int valid) {
    Box b; b.valid = valid;
    int ox, ex, oy, ey;
    rect_axis(bb[0], bb[2], 0.0, ox, ex); rect_axis(bb[1], bb[3], 0.0, oy, ey);
    b.px = py_slice(ox, ex + 1); b.py = py_slice(oy, ey + 1);          // positive core (init_score_map)
    rect_axis(bb[0], bb[2], 2.0, ox, ex); rect_axis(bb[1], bb[3], 2.0, oy, ey);
    b.gzx = py_slice(ox, ex); b.gzy = py_slice(oy, ey);                // zeroed block   (mask_gray_zone_cls)
    b.cox = py_slice(ox + 2, ex - 2 + 1); b.coy = py_slice(oy + 2, ey - 2 + 1);   // re-set core
    return b;
}
__device__ __forceinline__ bool in_span(const Span& s, int v) { return v >= s.a && v < s.b; }
__device__ __forceinline__ int lm_coord(float v, int clamp) {          // int(coord + 0.5), optionally clamped to 59
    int x = (int)(v + 0.5f);
    if (clamp) x = x < HWD ? x : HWD - 1;
    return x;
}

// block-wide arg-max of (value, lowest index on ties) over smem[0..NPIX); result broadcast via red_i[0]
__device__ __forceinline__ int block_argmax(const float* vals, float* red_v, int* red_i) {
    const int tid = threadIdx.x;
    float bv = -2.f; int bi = 0x7fffffff;
    for (int i = tid; i < NPIX; i += LOSS_THREADS) {
        const float v = vals[i];
        if (v > bv) { bv = v; bi = i; }                    // ascending i: ties keep the lower index
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(bv, off); const int oi = __shfl_down(bi, off);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < LOSS_THREADS / 64; ++w)
            if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bi)) { bv = red_v[w]; bi = red_i[w]; }
        red_i[0] = bi;
    }
    __syncthreads();
    const int r = red_i[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int w = 0; w < LOSS_THREADS / 64; ++w) t += red[w];
    __syncthreads();
    return t;   // valid on thread 0
}

struct LossArgs { dbx_loss_desc d; dbx_loss_io io; double* partial; unsigned char* maskbuf; };
#define LOSS_APPLY_THREADS 256
#define LOSS_APPLY_BLOCKS ((NPIX + LOSS_APPLY_THREADS - 1) / LOSS_APPLY_THREADS)      // 15 workgroups per patch in the second kernel

__global__ __launch_bounds__(LOSS_THREADS) void loss_kernel(const LossArgs a) {
    __shared__ float negl[NPIX];
    __shared__ unsigned char mask[NPIX];
    __shared__ unsigned char lmask[4][NPIX];
    __shared__ float red_v[LOSS_THREADS / 64];
    __shared__ int red_i[LOSS_THREADS / 64];
    __shared__ double red_d[LOSS_THREADS / 64];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int kind = a.d.kind, N = a.d.n, K = a.d.half_neg;
    const dbx_loss_io& io = a.io;
    const float* bb = io.bbox + 4 * n;
    const int valid = (a.d.use_labels && io.labels) ? (io.labels[n] != 0.f) : 1;
    const Box box = make_box(bb, valid);

    // ---- score loss, positives, negative loss into LDS
    const float* score = io.score + (size_t)n * NPIX;
    int npos = 0;
    for (int i = tid; i < NPIX; i += LOSS_THREADS) {
        const int y = i / HWD, x = i - y * HWD;
        const float gt = (valid && in_span(box.py, y) && in_span(box.px, x)) ? 1.f : 0.f;
        const float d = score[i] - gt;
        negl[i] = d * d * (1.f - gt);
        mask[i] = gt != 0.f;
        npos += gt != 0.f;
    }
    __syncthreads();
    if (io.pos_count) {
        const double t = block_sum((double)npos, red_d);
        if (tid == 0) io.pos_count[n] = (int)t;
    }
    // ---- hard negatives: top-K of the negative loss (descending; DenseBox.py:2083), then the random draws
    if (K > 0 && K <= 48) {
        // register tournament (as detect_kernel's): every lane holds its four losses as (bits << 32 | ~index) composites -- the losses
        // are >= +0, so their bit patterns order like the numbers; larger composite = larger loss, LOWER index on ties, the order of
        // the arg-max rounds below -- each wave extracts its top K with register maxima + shuffles, wave 0 merges the 16 lists:
        // no workgroup barrier per selected pixel (24 -> 5 us at K = 11).  A NaN loss ranks with the zeros.
        __shared__ unsigned long long wcand[(LOSS_THREADS / 64) * 48];
        const int lane = tid & 63, wv = tid >> 6;
        unsigned long long comp[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = tid + e * LOSS_THREADS;
            const float v = i < NPIX ? negl[i] : 0.f;
            const unsigned key = v != v ? 0u : __float_as_uint(v);
            comp[e] = i < NPIX ? ((unsigned long long)key << 32) | (unsigned)(~(unsigned)i) : 0ull;
        }
        auto wave_max = [&](unsigned long long v) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(v, off); v = o > v ? o : v; }
            return v;
        };
        for (int r = 0; r < K; ++r) {
            unsigned long long best = comp[0];
#pragma unroll
            for (int e = 1; e < 4; ++e) best = comp[e] > best ? comp[e] : best;
            best = wave_max(best);
#pragma unroll
            for (int e = 0; e < 4; ++e) comp[e] = comp[e] == best ? 0ull : comp[e];
            if (lane == 0) wcand[wv * 48 + r] = best;
        }
        __syncthreads();
        if (wv == 0) {
            const int tot = (LOSS_THREADS / 64) * K;                         // <= 768: 12 per lane
            unsigned long long c2[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int f = lane + 64 * j;
                c2[j] = f < tot ? wcand[(f / K) * 48 + f % K] : 0ull;
            }
            for (int r = 0; r < K; ++r) {
                unsigned long long best = c2[0];
#pragma unroll
                for (int j = 1; j < 12; ++j) best = c2[j] > best ? c2[j] : best;
                best = wave_max(best);
#pragma unroll
                for (int j = 0; j < 12; ++j) c2[j] = c2[j] == best ? 0ull : c2[j];
                if (lane == 0) {
                    const int idx = (int)(~(unsigned)(best & 0xffffffffull));
                    mask[idx] = 1;
                    if (io.neg_idx) io.neg_idx[(size_t)n * 2 * K + r] = idx;
                }
            }
        }
        __syncthreads();
    } else {
        for (int k = 0; k < K; ++k) {
            const int idx = block_argmax(negl, red_v, red_i);
            if (tid == 0) {
                negl[idx] = -1.f;
                mask[idx] = 1;
                if (io.neg_idx) io.neg_idx[(size_t)n * 2 * K + k] = idx;
            }
            __syncthreads();
        }
    }
    for (int k = tid; k < K; k += LOSS_THREADS) {
        const long long r = io.rand_neg[(size_t)n * K + k];
        if (io.neg_idx) io.neg_idx[(size_t)n * 2 * K + K + k] = r;
        if (r >= 0 && r < NPIX) mask[(int)r] = 1;                      // ids outside [0,3600) skipped (:1390)
    }
    __syncthreads();
    // ---- gray zone after selection (:2108): zero the block, re-set the core
    if (valid) {
        for (int i = tid; i < NPIX; i += LOSS_THREADS) {
            const int y = i / HWD, x = i - y * HWD;
            if (in_span(box.gzy, y) && in_span(box.gzx, x)) mask[i] = 0;
        }
        __syncthreads();
        for (int i = tid; i < NPIX; i += LOSS_THREADS) {
            const int y = i / HWD, x = i - y * HWD;
            if (in_span(box.coy, y) && in_span(box.cox, x)) mask[i] = 1;
        }
        __syncthreads();
    }

    // ---- landmark heat-map masks (kinds 1, 2): per channel top-1 hard negative + 1 random + 5x5 gray zone
    int lmx[4] = {0, 0, 0, 0}, lmy[4] = {0, 0, 0, 0};
    if (kind != 0) {
        const float* vt = io.vertices + 8 * n;
        const int clamp = a.d.use_labels;
        for (int j = 0; j < 4; ++j) { lmx[j] = lm_coord(vt[2 * j], clamp); lmy[j] = lm_coord(vt[2 * j + 1], clamp); }
        for (int j = 0; j < 4; ++j) {
            const float* lm = io.lm + ((size_t)n * 4 + j) * NPIX;
            const int pidx = lmy[j] * HWD + lmx[j];
            for (int i = tid; i < NPIX; i += LOSS_THREADS) {
                const float gt = (valid && i == pidx) ? 1.f : 0.f;
                const float d = lm[i] - gt;
                negl[i] = d * d * (1.f - gt);
                lmask[j][i] = gt != 0.f;
            }
            __syncthreads();
            const int hard = block_argmax(negl, red_v, red_i);
            if (tid == 0) {
                const long long r = io.lm_rand_neg[((size_t)j * N + n)];
                lmask[j][hard] = 1;
                if (r >= 0 && r < NPIX) lmask[j][(int)r] = 1;
                if (io.lm_neg_idx) { io.lm_neg_idx[((size_t)j * N + n) * 2] = hard; io.lm_neg_idx[((size_t)j * N + n) * 2 + 1] = r; }
            }
            __syncthreads();
            if (valid) {                                                // mask_gray_zone_lm (:1453-1462)
                const Span sy = py_slice(lmy[j] - 2, lmy[j] + 3), sx = py_slice(lmx[j] - 2, lmx[j] + 3);
                for (int i = tid; i < 25; i += LOSS_THREADS) {
                    const int y = sy.a + i / 5, x = sx.a + i % 5;
                    if (y < sy.b && x < sx.b) lmask[j][y * HWD + x] = 0;
                }
                __syncthreads();
                if (tid == 0) lmask[j][pidx] = 1;
                __syncthreads();
            }
        }
    }

    // ---- masks to the scratch buffer: the weighted sums and the gradients are a second, full-grid kernel (the 18 maps of a patch are
    // 260 KB of loads and stores: 36 of this kernel's 79 us when one workgroup per patch did them)
    unsigned char* mb = a.maskbuf + (size_t)n * 5 * NPIX;
    for (int i = tid; i < NPIX; i += LOSS_THREADS) {
        mb[i] = mask[i];
        if (kind != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) mb[(1 + j) * NPIX + i] = lmask[j][i];
        }
    }
}

// second kernel: one lane per output pixel, 15 workgroups per patch; the same expressions in the same order per pixel, the patch's sums
// as 15 partials added in a fixed order by loss_finish_kernel
__global__ __launch_bounds__(LOSS_APPLY_THREADS) void loss_apply_kernel(const LossArgs a) {
    __shared__ double red_d[LOSS_APPLY_THREADS / 64];
    const int n = blockIdx.y, tid = threadIdx.x, i = blockIdx.x * LOSS_APPLY_THREADS + tid;
    const int kind = a.d.kind;
    const dbx_loss_io& io = a.io;
    const float* bb = io.bbox + 4 * n;
    const int valid = (a.d.use_labels && io.labels) ? (io.labels[n] != 0.f) : 1;
    const Box box = make_box(bb, valid);
    const float* score = io.score + (size_t)n * NPIX;
    const unsigned char* mb = a.maskbuf + (size_t)n * 5 * NPIX;
    int lmx[4] = {0, 0, 0, 0}, lmy[4] = {0, 0, 0, 0};
    if (kind != 0) {
        const float* vt = io.vertices + 8 * n;
        const int clamp = a.d.use_labels;
        for (int j = 0; j < 4; ++j) { lmx[j] = lm_coord(vt[2 * j], clamp); lmy[j] = lm_coord(vt[2 * j + 1], clamp); }
    }
    const float l_loc = a.d.lambda_loc, l_det = (kind == 0) ? 1.f : a.d.lambda_det, l_lm = a.d.lambda_lm;
    double s_cls = 0, s_loc = 0, s_lm = 0, s_lmloc = 0, s_rf = 0;
    if (i < NPIX) {
        const int y = i / HWD, x = i - y * HWD;
        const float m = (float)mb[i];
        const float gt = (valid && in_span(box.py, y) && in_span(box.px, x)) ? 1.f : 0.f;
        if (io.mask_cls) io.mask_cls[(size_t)n * NPIX + i] = m;
        {
            const float d = score[i] - gt;
            s_cls += (double)(m * (d * d));
            if (io.d_score) io.d_score[(size_t)n * NPIX + i] = 2.f * l_det * m * d;
        }
        const float mg = m * gt;
        const float fx = (float)x, fy = (float)y;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float tgt = valid ? ((c & 1) ? fy : fx) - bb[c] : 0.f;             // init_loc_map / init_loc
            const size_t o = ((size_t)n * 4 + c) * NPIX + i;
            const float d = io.loc[o] - tgt;
            s_loc += (double)(mg * (d * d));
            if (io.d_loc) io.d_loc[o] = 2.f * l_det * l_loc * mg * d;
        }
        if (kind != 0) {
            const float d = io.rf[(size_t)n * NPIX + i] - gt;
            s_rf += (double)(m * (d * d));
            if (io.d_rf) io.d_rf[(size_t)n * NPIX + i] = 2.f * m * d;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ml = (float)mb[(1 + j) * NPIX + i];
                const float hg = (valid && i == lmy[j] * HWD + lmx[j]) ? 1.f : 0.f;
                const size_t o = ((size_t)n * 4 + j) * NPIX + i;
                const float dl = io.lm[o] - hg;
                s_lm += (double)(ml * (dl * dl));
                if (io.d_lm) io.d_lm[o] = 2.f * l_lm * ml * dl;
                if (io.mask_lm) io.mask_lm[o] = ml;
            }
            if (kind == 2) {
                const float* vt = io.vertices + 8 * n;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float tgt = valid ? ((c & 1) ? fy : fx) - vt[c] : 0.f;     // init_lm_locmap(_pn)
                    const size_t o = ((size_t)n * 8 + c) * NPIX + i;
                    const float d = io.lmloc[o] - tgt;
                    s_lmloc += (double)(mg * (d * d));
                    if (io.d_lmloc) io.d_lmloc[o] = 2.f * mg * d;
                }
            }
        }
    }
    // full = l_det*(cls + l_loc*loc) + l_lm*lm + lmloc + rf      (DenseBox.py:2166-2180, :2711-2723, :2917)
    double mine = (double)l_det * (s_cls + (double)l_loc * s_loc) + (double)l_lm * s_lm + s_lmloc + s_rf;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((tid & 63) == 0) red_d[tid >> 6] = mine;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < LOSS_APPLY_THREADS / 64; ++w) t += red_d[w];
        a.partial[(size_t)n * LOSS_APPLY_BLOCKS + blockIdx.x] = t;
    }
}

// n patches x `per` partials: every thread adds the partials of its patches in order, thread 0 then adds the patch sums in patch order
// (fixed order: deterministic; one thread walking all 960 partials took 49 us of dependent global loads)
__global__ __launch_bounds__(256) void loss_finish_kernel(const double* partial, int n, int per, float* loss) {
    __shared__ double psum[256];
    double run = 0.0;                                         // thread 0 only: running total over groups of 256 patches
    for (int base = 0; base < n; base += 256) {
        const int p = base + threadIdx.x;
        double t = 0.0;
        if (p < n) for (int i = 0; i < per; ++i) t += partial[(size_t)p * per + i];
        psum[threadIdx.x] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int cnt = n - base < 256 ? n - base : 256;
            for (int i = 0; i < cnt; ++i) run += psum[i];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)run;
}

extern "C" int64_t dbx_loss_scratch_bytes(int32_t n) {          // per-workgroup partial sums (double) + the five mask planes of every patch
    return (int64_t)n * LOSS_APPLY_BLOCKS * 8 + (int64_t)n * 5 * NPIX + 64;
}
extern "C" int dbx_loss_forward_backward(const dbx_loss_desc* d, const dbx_loss_io* io, void* scratch, void* stream) {
    DBX_REQUIRE(d && io && scratch, "loss: null argument");
    DBX_REQUIRE(d->kind >= 0 && d->kind <= 2 && d->n > 0, "loss: bad kind/n");
    DBX_REQUIRE(d->half_neg >= 0 && d->half_neg <= NPIX, "loss: half_neg out of range");
    DBX_REQUIRE(io->bbox && io->score && io->loc && io->loss && (d->half_neg == 0 || io->rand_neg), "loss: missing tensors");
    if (d->kind != 0) DBX_REQUIRE(io->vertices && io->lm && io->rf && io->lm_rand_neg, "loss: landmark tensors missing");
    if (d->kind == 2) DBX_REQUIRE(io->lmloc != nullptr, "loss: lm_loc missing");
    LossArgs a; a.d = *d; a.io = *io; a.partial = (double*)scratch;
    a.maskbuf = (unsigned char*)scratch + (size_t)d->n * LOSS_APPLY_BLOCKS * sizeof(double);
    hipLaunchKernelGGL(loss_kernel, dim3(d->n), dim3(LOSS_THREADS), 0, (hipStream_t)stream, a);
    DBX_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_apply_kernel, dim3(LOSS_APPLY_BLOCKS, d->n), dim3(LOSS_APPLY_THREADS), 0, (hipStream_t)stream, a);
    DBX_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)scratch, d->n, LOSS_APPLY_BLOCKS, io->loss);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

// ---------------------------------------------------------------------------------------------- stand-alone label / mask ops
__global__ void count_positives_kernel(const float* bbox, const float* labels, int n, int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int valid = labels ? labels[i] != 0.f : 1;
    const Box b = make_box(bbox + 4 * i, valid);
    out[i] = valid ? (b.py.b - b.py.a) * (b.px.b - b.px.a) : 0;
}
extern "C" int dbx_count_positives(const float* bbox, const float* labels, int32_t n, int32_t* out, void* stream) {
    hipLaunchKernelGGL(count_positives_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, bbox, labels, n, out);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

__global__ void init_score_map_kernel(const float* bbox, const float* labels, float* out) {
    const int n = blockIdx.x;
    const int valid = labels ? labels[n] != 0.f : 1;
    const Box b = make_box(bbox + 4 * n, valid);
    for (int i = threadIdx.x; i < NPIX; i += blockDim.x) {
        const int y = i / HWD, x = i - y * HWD;
        out[(size_t)n * NPIX + i] = (valid && in_span(b.py, y) && in_span(b.px, x)) ? 1.f : 0.f;
    }
}
extern "C" int dbx_init_score_map(const float* bbox, const float* labels, int32_t n, float* out, void* stream) {
    hipLaunchKernelGGL(init_score_map_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, bbox, labels, out);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

__global__ void init_offset_map_kernel(const float* coords, const float* labels, int c, float* out) {
    const int n = blockIdx.x;
    const int valid = labels ? labels[n] != 0.f : 1;
    for (int i = threadIdx.x; i < c * NPIX; i += blockDim.x) {
        const int ch = i / NPIX, p = i - ch * NPIX;
        const int y = p / HWD, x = p - y * HWD;
        out[(size_t)n * c * NPIX + i] = valid ? ((ch & 1) ? (float)y : (float)x) - coords[n * c + ch] : 0.f;
    }
}
extern "C" int dbx_init_offset_map(const float* coords, const float* labels, int32_t n, int32_t c, float* out, void* stream) {
    hipLaunchKernelGGL(init_offset_map_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, coords, labels, c, out);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

__global__ void init_lm_heatmap_kernel(const float* vert, const float* labels, int clamp, float* out, int* err) {
    const int n = blockIdx.x;
    const int valid = labels ? labels[n] != 0.f : 1;
    for (int i = threadIdx.x; i < 4 * NPIX; i += blockDim.x) out[(size_t)n * 4 * NPIX + i] = 0.f;
    __syncthreads();
    if (threadIdx.x < 4 && valid) {
        const int j = threadIdx.x;
        int x = lm_coord(vert[8 * n + 2 * j], clamp), y = lm_coord(vert[8 * n + 2 * j + 1], clamp);
        if (x >= HWD || y >= HWD) { if (err) atomicAdd(err, 1); x = min(x, HWD - 1); y = min(y, HWD - 1); }
        out[((size_t)n * 4 + j) * NPIX + y * HWD + x] = 1.f;
    }
}
extern "C" int dbx_init_lm_heatmap(const float* vertices, const float* labels, int32_t n, int32_t clamp, float* out, void* stream) {
    hipLaunchKernelGGL(init_lm_heatmap_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, vertices, labels, clamp, out, (int*)nullptr);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

// mask[N,1,60,60] fp32, in place (DenseBox.py:1368-1402)
__global__ void mask_by_sel_kernel(float* mask, int n, const long long* pos, long long npos, const long long* neg, int nneg) {
    const long long total = npos + (long long)n * nneg;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        if (i < npos) {
            const long long* p = pos + 4 * i;                      // [n, c(=0), y, x]
            mask[((size_t)p[0] + p[1]) * NPIX + p[2] * HWD + p[3]] = 1.f;
        } else {
            const long long j = i - npos;
            const long long id = neg[j];
            if (id >= 0 && id < NPIX) mask[(size_t)(j / nneg) * NPIX + id] = 1.f;
        }
    }
}
extern "C" int dbx_mask_by_sel(float* mask, int32_t n, const int64_t* pos_idx, int64_t n_pos, const int64_t* neg_idx,
                               int32_t n_neg, void* stream) {
    const long long total = n_pos + (long long)n * n_neg;
    if (total == 0) return DBX_OK;
    hipLaunchKernelGGL(mask_by_sel_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mask, n,
                       (const long long*)pos_idx, (long long)n_pos, (const long long*)neg_idx, n_neg);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

__global__ void mask_gray_zone_cls_kernel(float* mask, const float* bbox, const float* labels) {
    const int n = blockIdx.x;
    const int valid = labels ? labels[n] != 0.f : 1;
    if (!valid) return;
    const Box b = make_box(bbox + 4 * n, valid);
    for (int i = threadIdx.x; i < NPIX; i += blockDim.x) {
        const int y = i / HWD, x = i - y * HWD;
        float v = mask[(size_t)n * NPIX + i];
        if (in_span(b.gzy, y) && in_span(b.gzx, x)) v = 0.f;
        if (in_span(b.coy, y) && in_span(b.cox, x)) v = 1.f;
        mask[(size_t)n * NPIX + i] = v;
    }
}
extern "C" int dbx_mask_gray_zone_cls(float* mask, const float* bbox, const float* labels, int32_t n, void* stream) {
    hipLaunchKernelGGL(mask_gray_zone_cls_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, mask, bbox, labels);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

// mask [N,1,60,60] (one landmark channel view made contiguous by the caller), pos rows [n,0,y,x]; positives of one
// channel never share a sample, so the per-positive 5x5 blocks are independent (DenseBox.py:1453-1462)
__global__ void mask_gray_zone_lm_kernel(float* mask, const long long* pos, long long npos) {
    const long long p = blockIdx.x;
    if (p >= npos) return;
    const long long n = pos[4 * p];
    const int y = (int)pos[4 * p + 2], x = (int)pos[4 * p + 3];
    const Span sy = py_slice(y - 2, y + 3), sx = py_slice(x - 2, x + 3);
    const int i = threadIdx.x;
    if (i < 25) {
        const int yy = sy.a + i / 5, xx = sx.a + i % 5;
        if (yy < sy.b && xx < sx.b) mask[(size_t)n * NPIX + yy * HWD + xx] = 0.f;
    }
    __syncthreads();
    if (i == 0) mask[(size_t)n * NPIX + y * HWD + x] = 1.f;
}
extern "C" int dbx_mask_gray_zone_lm(float* mask, int32_t n, const int64_t* pos_idx, int64_t n_pos, void* stream) {
    if (n_pos == 0) return DBX_OK;
    hipLaunchKernelGGL(mask_gray_zone_lm_kernel, dim3((unsigned)n_pos), dim3(64), 0, (hipStream_t)stream, mask,
                       (const long long*)pos_idx, (long long)n_pos);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
