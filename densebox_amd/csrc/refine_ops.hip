// Refine branch of DenseBoxLM / DenseBoxLMLOC (DenseBox.py:464-471, :729-736) by its linear structure, forward AND backward:
//
//     x5 = MaxPool2d(2,2)(cat(landmarks, score))        [n][5][ph][pw]
//     refine = conv6_3(Upsample(conv6_2(conv6_1(x5))))   3x3 (5 -> 64), 5x5 (64 -> 64), bilinear align_corners, 1x1 (64 -> 1)
//
// Nothing after the pooling is non-linear and the 1x1 conv commutes with the up-sampling (bilinear weights sum to 1), so
// refine = up(s), s = W * x5 + b with ONE un-padded 7x7 kernel W[c][u][v] = sum_m sum_{a+i=u, b+j=v} V[m][a][b] w1[m][c][i][j],
// V[m][a][b] = sum_n w3[n] w2[n][m][a][b]  (dbx_fold_refine, aux_ops.hip).  The backward pass needs the intermediate 64-channel
// maps for nothing either.  With g = up^T(dL/d refine) (one channel on the small grid) and the 245 + 1 numbers
//
//     G1[c][u][v] = sum_{n,p} g[n](p) x5[n][c](p + (u,v)),      Sg = sum_{n,p} g[n](p)            (= dL/dW, dL/db of the folded conv)
//
// every parameter gradient of the three convs is a small contraction of G1, Sg and the weights:
//
//     G2[m][a][b] = sum_{c,i,j} w1[m][c][i][j] G1[c][a+i][b+j] + b1[m] Sg          (= sum_p g(p) rf_1[m](p + (a,b)))
//     dw3[n]      = sum_{m,a,b} w2[n][m][a][b] G2[m][a][b] + b2[n] Sg               db3 = Sg
//     dw2[n][m][a][b] = w3[n] G2[m][a][b]                                             db2[n] = w3[n] Sg
//     dw1[m][c][i][j] = sum_{a,b} V[m][a][b] G1[c][a+i][b+j]                          db1[m] = (sum_{a,b} V[m][a][b]) Sg
//
// and the gradient of the branch's input is the transposed folded conv of g pushed through the pooling's arg-max:
//     d x5[n][c](q) = sum_{u,v} W[c][u][v] g[n](q - (u,v)).
// Five fp32 kernels on tiny maps replace three MFMA convs, their three weight gradients + reductions, three data gradients, two
// up-sampling passes over 64 channels and the layout kernels around them (0.31 ms of a 10.2-ms training step -> 0.04 ms); every sum
// runs in a fixed order (bitwise repeatable), and there is no 16-bit rounding inside the branch.
#include "common.hpp"

namespace {
constexpr int RCI = 5, RTAPS = 49, RG1 = RCI * RTAPS + 1;              // G1 (245) + Sg

// ATen's align_corners=True source coordinate (same arithmetic as aux_ops.hip::bilin_coef)
__device__ __forceinline__ void rf_coef(int d, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
    // no FMA contraction here: the compiler unrolls a grid-stride loop by two into v_pk_* instructions and folds scale * d - i0 into one
    // (packed) fma there but not in the scalar remainder iteration -- l1 then differs by an ulp of src and a value depends on where in the
    // batch its plane sits (HIP's __fmul_rn is a plain product: it does not stop the contraction, the pragma does)
#pragma clang fp contract(off)
    const float src = scale * (float)d;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}
inline float rf_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

// g[n][sy][sx] = sum over the output pixels (y, x) whose bilinear footprint holds (sy, sx) of their weight x d[n][y][x]: a gather with a
// fixed summation order (y ascending, x ascending) instead of a scatter with atomics.
__global__ void refine_up_t_kernel(const float* __restrict__ d, int n, int ho, int wo, float* __restrict__ g, int hs, int ws, float sy, float sx) {
    const int64_t total = (int64_t)n * hs * ws;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(i % ws), py = (int)((i / ws) % hs);
        const int64_t im = i / ((int64_t)ws * hs);
        // candidate output rows / columns: those whose i0 is py - 1 or py (a superset, filtered exactly below)
        int ylo = sy > 0.f ? (int)((float)(py - 1) / sy) - 1 : 0, yhi = sy > 0.f ? (int)((float)(py + 1) / sy) + 2 : ho - 1;
        int xlo = sx > 0.f ? (int)((float)(px - 1) / sx) - 1 : 0, xhi = sx > 0.f ? (int)((float)(px + 1) / sx) + 2 : wo - 1;
        ylo = ylo < 0 ? 0 : ylo; xlo = xlo < 0 ? 0 : xlo; yhi = yhi > ho - 1 ? ho - 1 : yhi; xhi = xhi > wo - 1 ? wo - 1 : xhi;
        const float* dp = d + im * ho * wo;
        float acc = 0.f;
        for (int y = ylo; y <= yhi; ++y) {
            int y0, y1; float ly0, ly1;
            rf_coef(y, sy, hs, y0, y1, ly0, ly1);
            const float wy = (y0 == py ? ly0 : 0.f) + (y1 == py ? ly1 : 0.f);
            if (wy == 0.f) continue;
            for (int x = xlo; x <= xhi; ++x) {
                int x0, x1; float lx0, lx1;
                rf_coef(x, sx, ws, x0, x1, lx0, lx1);
                const float wx = (x0 == px ? lx0 : 0.f) + (x1 == px ? lx1 : 0.f);
                if (wx != 0.f) acc = fmaf(wy * wx, dp[y * wo + x], acc);
            }
        }
        g[i] = acc;
    }
}

// RSPLIT workgroups per image, each with a band of rows of g and the x5 rows under it in LDS; thread t < 245 owns G1 entry t, thread 245 Sg
constexpr int RSPLIT = 4;
__global__ __launch_bounds__(256) void refine_g1_kernel(const float* __restrict__ lm, const float* __restrict__ sc, int h, int w,
                                                        const float* __restrict__ g, float* __restrict__ partial) {
    extern __shared__ float sm[];
    const int ph = h / 2, pw = w / 2, oh = ph - 6, ow = pw - 6;
    const int n = blockIdx.x / RSPLIT, part = blockIdx.x % RSPLIT, tid = threadIdx.x;
    const int rows = (oh + RSPLIT - 1) / RSPLIT, y0 = part * rows < oh ? part * rows : oh, y1 = y0 + rows < oh ? y0 + rows : oh;   // rows [y0, y1) of g
    const int xr = y1 > y0 ? y1 - y0 + 6 : 0;                                                                  // x5 rows y0 .. y1 + 5
    float* x5 = sm;                                  // [5][xr][pw]
    float* gs = sm + RCI * (rows + 6) * pw;          // [rows][ow]
    for (int e = tid; e < RCI * xr * pw; e += 256) {
        const int c = e / (xr * pw), r = e % (xr * pw), py = y0 + r / pw, px = r % pw;
        const float* p = (c < 4 ? lm + ((size_t)n * 4 + c) * h * w : sc + (size_t)n * h * w) + (size_t)(2 * py) * w + 2 * px;
        x5[e] = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[w], p[w + 1]));
    }
    for (int e = tid; e < (y1 - y0) * ow; e += 256) gs[e] = g[((size_t)n * oh + y0) * ow + e];
    __syncthreads();
    if (tid < RG1) {
        float acc = 0.f;
        if (tid < RCI * RTAPS) {
            const int c = tid / RTAPS, u = (tid % RTAPS) / 7, v = tid % 7;
            const float* xc = x5 + (c * xr + u) * pw + v;
            for (int y = 0; y < y1 - y0; ++y)
                for (int x = 0; x < ow; ++x) acc = fmaf(gs[y * ow + x], xc[y * pw + x], acc);
        } else {
            for (int e = 0; e < (y1 - y0) * ow; ++e) acc += gs[e];
        }
        partial[(size_t)blockIdx.x * RG1 + tid] = acc;
    }
}

struct RefineW { const float *w1, *b1, *w2, *b2, *w3, *b3, *V; float *dw1, *db1, *dw2, *db2, *dw3, *db3; };
// every workgroup sums the per-image partials in image order (G1, Sg), builds G2 in LDS from the LDS copy of w1 and then writes its share of
// the outputs: blocks 0 .. cm-1: row n = block of dw2 (+ dw3[n], db2[n]); block cm: dw1, db1, db3 (V comes from dbx_fold_refine)
__global__ __launch_bounds__(256) void refine_wgrad_kernel(const float* __restrict__ partial, int nimg, int cm, RefineW a) {
    __shared__ float G1[RG1], G2[64 * 25], V[64 * 25], w1s[64 * RCI * 9], red[256];
    const int tid = threadIdx.x;
    for (int e = tid; e < cm * RCI * 9; e += 256) w1s[e] = a.w1[e];
    if (tid < RG1) {
        float acc = 0.f;
#pragma unroll 16
        for (int n = 0; n < nimg; ++n) acc += partial[(size_t)n * RG1 + tid];          // (image, band) order: fixed
        G1[tid] = acc;
    }
    const int blk = blockIdx.x;
    if (blk == cm)
        for (int e = tid; e < cm * 25; e += 256) V[e] = a.V[e];
    __syncthreads();
    const float Sg = G1[RG1 - 1];
    for (int e = tid; e < cm * 25; e += 256) {
        const int m = e / 25, aa = (e % 25) / 5, bb = e % 5;
        float acc = a.b1[m] * Sg;
        for (int c = 0; c < RCI; ++c)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) acc = fmaf(w1s[(m * RCI + c) * 9 + i * 3 + j], G1[(c * 7 + aa + i) * 7 + bb + j], acc);
        G2[e] = acc;
    }
    __syncthreads();
    if (blk < cm) {
        const int n = blk;
        const float w3n = a.w3[n];
        float part = 0.f;
        for (int e = tid; e < cm * 25; e += 256) {
            a.dw2[(size_t)n * cm * 25 + e] = w3n * G2[e];
            part = fmaf(a.w2[(size_t)n * cm * 25 + e], G2[e], part);
        }
        red[tid] = part;
        __syncthreads();
        if (tid == 0) {
            float acc = a.b2[n] * Sg;
            for (int t = 0; t < 256; ++t) acc += red[t];
            a.dw3[n] = acc;
            a.db2[n] = w3n * Sg;
        }
    } else {
        for (int e = tid; e < cm * RCI * 9; e += 256) {
            const int m = e / (RCI * 9), c = (e / 9) % RCI, i = (e % 9) / 3, j = e % 3;
            float acc = 0.f;
            for (int aa = 0; aa < 5; ++aa)
                for (int bb = 0; bb < 5; ++bb) acc = fmaf(V[m * 25 + aa * 5 + bb], G1[(c * 7 + aa + i) * 7 + bb + j], acc);
            a.dw1[e] = acc;
        }
        for (int m = tid; m < cm; m += 256) {
            float sv = 0.f;
            for (int ab = 0; ab < 25; ++ab) sv += V[m * 25 + ab];
            a.db1[m] = sv * Sg;
        }
        if (tid == 0) a.db3[0] = Sg;
    }
}

// d x5 = transposed folded conv of g, routed through the pooling arg-max (first maximum in (0,0),(0,1),(1,0),(1,1) order, ATen) and added
// to the incoming gradients of the landmark / score heads: out_lm = g_lm + d(cat)[0:4], out_sc = g_sc + d(cat)[4].  One lane per 2x2
// window and channel; windows past the pooled extent (odd h / w) pass the incoming gradient through.
__global__ void refine_dgrad_kernel(const float* __restrict__ lm, const float* __restrict__ sc, int n, int h, int w, const float* __restrict__ g,
                                    const float* __restrict__ wf, const float* __restrict__ g_lm, const float* __restrict__ g_sc,
                                    float* __restrict__ out_lm, float* __restrict__ out_sc) {
    __shared__ float wsm[RCI * RTAPS];
    for (int e = threadIdx.x; e < RCI * RTAPS; e += blockDim.x) wsm[e] = wf[e];
    __syncthreads();
    const int ph = h / 2, pw = w / 2, oh = ph - 6, ow = pw - 6, wh = (h + 1) / 2, ww = (w + 1) / 2;
    const int64_t total = (int64_t)n * RCI * wh * ww;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int wx = (int)(i % ww), wy = (int)((i / ww) % wh), c = (int)((i / ((int64_t)ww * wh)) % RCI);
        const int im = (int)(i / ((int64_t)ww * wh * RCI));
        const size_t plane = c < 4 ? ((size_t)im * 4 + c) * h * w : (size_t)im * h * w;
        const float* xin = (c < 4 ? lm : sc) + plane;
        const float* gin = c < 4 ? (g_lm ? g_lm + plane : nullptr) : (g_sc ? g_sc + plane : nullptr);
        float* dst = (c < 4 ? out_lm : out_sc) + plane;
        float dv = 0.f;
        int arg = -1;
        if (wy < ph && wx < pw) {
            const float* gp = g + (size_t)im * oh * ow;
            for (int u = 0; u < 7; ++u) {
                const int y = wy - u;
                if (y < 0 || y >= oh) continue;
                for (int v = 0; v < 7; ++v) {
                    const int x = wx - v;
                    if (x >= 0 && x < ow) dv = fmaf(wsm[(c * 7 + u) * 7 + v], gp[y * ow + x], dv);
                }
            }
            const float* p = xin + (size_t)(2 * wy) * w + 2 * wx;
            float m = p[0]; arg = 0;
            if (p[1] > m) { m = p[1]; arg = 1; }
            if (p[w] > m) { m = p[w]; arg = 2; }
            if (p[w + 1] > m) { m = p[w + 1]; arg = 3; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int y = 2 * wy + (k >> 1), x = 2 * wx + (k & 1);
            if (y >= h || x >= w) continue;
            const float in = gin ? gin[(size_t)y * w + x] : 0.f;
            dst[(size_t)y * w + x] = in + (k == arg ? dv : 0.f);
        }
    }
}
}  // namespace

extern "C" int64_t dbx_refine_backward_scratch_bytes(int32_t n, int32_t h, int32_t w) {
    const int64_t oh = h / 2 - 6, ow = w / 2 - 6;
    return ((int64_t)n * oh * ow + (int64_t)n * RSPLIT * RG1 + 64) * 4;
}

// d_refine [n][1][h][w], landmark / score = the heads' fp32 NCHW outputs of the forward pass, w_fold / v_fold = dbx_fold_refine's 7x7 kernel
// and its V[cm][5][5] = sum_n w3[n] w2[n][m].
// g_landmark / g_score: incoming gradients of those heads (may be null = zero); out_landmark [n][4][h][w] / out_score [n][1][h][w]
// receive incoming + the branch's contribution.  dw* / db*: fp32 parameter gradients in the parameters' own layouts (overwritten).
extern "C" int dbx_refine_backward(const float* d_refine, const float* landmark, const float* score, int32_t n, int32_t h, int32_t w,
                                   const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                                   int32_t cm, const float* w_fold, const float* v_fold, const float* g_landmark, const float* g_score, float* out_landmark,
                                   float* out_score, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, void* scratch,
                                   void* stream) {
    DBX_REQUIRE(d_refine && landmark && score && w1 && b1 && w2 && b2 && w3 && b3 && w_fold && v_fold && out_landmark && out_score && dw1 && db1 && dw2 &&
                    db2 && dw3 && db3 && scratch, "refine_backward: null argument");
    DBX_REQUIRE(n >= 1 && h / 2 >= 7 && w / 2 >= 7 && cm >= 1 && cm <= 64, "refine_backward: H/2, W/2 >= 7, mid channels <= 64");
    const int ph = h / 2, pw = w / 2, oh = ph - 6, ow = pw - 6;
    const int rows = (oh + RSPLIT - 1) / RSPLIT;
    const size_t lds = ((size_t)RCI * (rows + 6) * pw + (size_t)rows * ow) * 4;
    DBX_REQUIRE(lds <= 150 * 1024, "refine_backward: map too large for the per-image LDS tile (%d x %d)", h, w);
    hipStream_t s = (hipStream_t)stream;
    float* g = (float*)scratch;
    float* partial = g + (size_t)n * oh * ow;
    const int64_t tot_g = (int64_t)n * oh * ow;
    hipLaunchKernelGGL(refine_up_t_kernel, dim3((unsigned)((tot_g + 255) / 256)), dim3(256), 0, s, d_refine, n, h, w, g, oh, ow, rf_scale(oh, h), rf_scale(ow, w));
    DBX_LAUNCH_CHECK();
    static DbxDevOnce attr_once; int attr_dev = 0;
    if (attr_once.pending(&attr_dev)) {
        DBX_HIP(hipFuncSetAttribute((const void*)refine_g1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_once.mark(attr_dev);
    }
    hipLaunchKernelGGL(refine_g1_kernel, dim3(n * RSPLIT), dim3(256), lds, s, landmark, score, h, w, g, partial);
    DBX_LAUNCH_CHECK();
    RefineW a = {w1, b1, w2, b2, w3, b3, v_fold, dw1, db1, dw2, db2, dw3, db3};
    hipLaunchKernelGGL(refine_wgrad_kernel, dim3(cm + 1), dim3(256), 0, s, partial, n * RSPLIT, cm, a);
    DBX_LAUNCH_CHECK();
    const int64_t tot_d = (int64_t)n * RCI * ((h + 1) / 2) * ((w + 1) / 2);
    hipLaunchKernelGGL(refine_dgrad_kernel, dim3((unsigned)((tot_d + 255) / 256)), dim3(256), 0, s, landmark, score, n, h, w, g, w_fold, g_landmark, g_score,
                       out_landmark, out_score);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
