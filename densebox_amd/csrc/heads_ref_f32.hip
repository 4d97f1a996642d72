// fp32 instantiations of the two GENERATING consumers of the heads' hidden gradient (dbx_heads1_wgrad_gen / dbx_heads1_dgrad_gen).
//
// The 16-bit training step never holds d_hid = keep * scale * (d_out W2) in memory: its two 60x60 consumers generate it inside their MFMA
// loops (conv_wgrad.hip: wgrad_wide2_kernel<T, true>; heads_gen.hpp).  The exact-fp32 path exists to be compared element-wise with the
// gradients captured from the reference's own training loop (tests/test_hip_backward.py, 2e-4); with these kernels it runs the SAME call
// structure as the 16-bit step (DBX_F32_LIN=1: heads backward by linearity of the up-sampling + generated hidden gradient), so the algebra
// and the plumbing the bench times -- which slice of dW1 comes from which grid, the dropout scale, the hash seed, the W2 pointers, the
// per-head k -- are pinned to the reference at fp32 tolerance.  They are plain one-thread-per-output kernels: correctness vehicles for the
// parity suite, not tuned (the fp32 path is never benchmarked).  Reference: autograd through nn.Conv2d(768, 512, 1) -> nn.Dropout ->
// nn.Conv2d(512, k, 1), DenseBox.py:158-162, :717-726.
#include "common.hpp"

namespace {

struct RefGen {
    const float* dout;           // compact [N * H * W][ld] fp32, one slot of `slot` channels per head
    const float* w2[4];          // fp32 [k][512] per head
    int k[4];
    int ld, slot, nh, npix;
    unsigned seed; int use_hash;
};

// d_hid[m][ch] (ch = global hidden channel, 0 .. 512 nh): the keep bit is bit ch % 32 of dbx_drop_hash32(seed, m, ch / 32) -- the forward's
__device__ __forceinline__ float ref_dhid(const RefGen& g, int m, int ch) {
    const int hd = ch >> 9, cl = ch & 511;
    const float* wp = g.w2[0];
    int k = g.k[0];
#pragma unroll
    for (int hh = 1; hh < 4; ++hh)
        if (hd == hh) { wp = g.w2[hh]; k = g.k[hh]; }
    if (g.use_hash && !((dbx_drop_hash32(g.seed, (unsigned)m, (unsigned)ch >> 5) >> (ch & 31)) & 1u)) return 0.f;
    const float* d = g.dout + (size_t)m * g.ld + hd * g.slot;
    float s = 0.f;
    for (int j = 0; j < k; ++j) s += d[j] * wp[(size_t)j * 512 + cl];
    return g.use_hash ? 2.f * s : s;
}

// dW1[ch][ci_off + c] = sum_m d_hid[m][ch] x[m][c],  db1[ch] = sum_m d_hid[m][ch]; one workgroup per hidden channel, pixels in ascending order
__global__ __launch_bounds__(256) void heads1_wgrad_gen_f32_kernel(RefGen g, FrameGeo x, int H, int W, int ci, float* __restrict__ dw, int ci_total,
                                                                   int ci_off, float* __restrict__ db) {
    const int ch = blockIdx.x;
    __shared__ float s_dh[256];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                                // channels tid, tid + 256, ... (ci <= 1024)
    float bacc = 0.f;
    for (int m0 = 0; m0 < g.npix; m0 += 256) {
        const int mm = m0 + (int)threadIdx.x;
        __syncthreads();
        s_dh[threadIdx.x] = mm < g.npix ? ref_dhid(g, mm, ch) : 0.f;
        __syncthreads();
        const int lim = g.npix - m0 < 256 ? g.npix - m0 : 256;
        for (int i = 0; i < lim; ++i) {
            const int m = m0 + i;
            const float dh = s_dh[i];
            const int n = m / (H * W), r = m - n * H * W, y = r / W, xx = r - y * W;
            const float* xp = (const float*)x.base + geo_pix(x, n, y, xx);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = (int)threadIdx.x + 256 * q;
                if (c < ci) acc[q] += dh * xp[c];
            }
            bacc += dh;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = (int)threadIdx.x + 256 * q;
        if (c < ci) dw[(size_t)ch * ci_total + ci_off + c] = acc[q];
    }
    if (db && threadIdx.x == 0) db[ch] = bacc;
}

// y[m][c] = (gate[m][c] > 0) * sum_ch d_hid[m][ch] Wt[c][ch]; one workgroup per pixel, Wt = plain data-gradient image (rows = the 1x1 conv's
// input channels, K = hidden channels, row stride `wld`)
__global__ __launch_bounds__(256) void heads1_dgrad_gen_f32_kernel(RefGen g, const float* __restrict__ wt, int wld, int nhid, FrameGeo y, FrameGeo gate,
                                                                   int H, int W) {
    extern __shared__ float s_hid[];                                    // [nhid]
    const int m = blockIdx.x;
    for (int ch = threadIdx.x; ch < nhid; ch += 256) s_hid[ch] = ref_dhid(g, m, ch);
    __syncthreads();
    const int n = m / (H * W), r = m - n * H * W, yy = r / W, xx = r - yy * W;
    float* yp = (float*)y.base + geo_pix(y, n, yy, xx);
    const float* gp = (const float*)gate.base + geo_pix(gate, n, yy, xx);
    for (int c = threadIdx.x; c < y.c; c += 256) {
        const float* wr = wt + (size_t)c * wld;
        float s = 0.f;
        for (int ch = 0; ch < nhid; ++ch) s += s_hid[ch] * wr[ch];
        yp[c] = gp[c] > 0.f ? s : 0.f;
    }
}

int fill(RefGen& g, const dbx_view* d_out, const float* const* w2, const int32_t* k, int nh, int use_hash, unsigned seed) {
    DBX_REQUIRE(nh >= 1 && nh <= 4 && d_out->pad == 0 && d_out->c % nh == 0, "heads gen (fp32): d_out is the compact map with one slot per head");
    g.dout = (const float*)d_out->ptr + d_out->c_off;
    for (int i = 0; i < 4; ++i) {
        g.w2[i] = i < nh ? w2[i] : nullptr; g.k[i] = i < nh ? k[i] : 0;
        if (i < nh) DBX_REQUIRE(w2[i] && k[i] >= 1 && k[i] <= d_out->c / nh, "heads gen (fp32): k in 1..slot");
    }
    g.ld = d_out->ld; g.slot = d_out->c / nh; g.nh = nh; g.npix = d_out->n * d_out->h * d_out->w; g.seed = seed; g.use_hash = use_hash ? 1 : 0;
    return DBX_OK;
}

}  // namespace

int dbx_internal_heads1_wgrad_gen_f32(const dbx_view* d_out, const dbx_view* x, const float* const* w2, const int32_t* k, int nh, int use_hash,
                                      unsigned seed, int ci, float* dw, int ci_total, int ci_off, float* db, hipStream_t s) {
    RefGen g;
    if (int rc = fill(g, d_out, w2, k, nh, use_hash, seed)) return rc;
    DBX_REQUIRE(d_out->n == x->n && d_out->h == x->h && d_out->w == x->w && ci >= 1 && ci <= x->c && ci <= 1024 && ci_off >= 0 && ci_off + ci <= ci_total,
                "heads1_wgrad_gen (fp32): d_out and x are maps of the same pixels, the column slice lies inside dw");
    hipLaunchKernelGGL(heads1_wgrad_gen_f32_kernel, dim3(512 * nh), dim3(256), 0, s, g, make_geo<float>(x), x->h, x->w, ci, dw, ci_total, ci_off, db);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

int dbx_internal_heads1_dgrad_gen_f32(const dbx_view* d_out, const float* const* w2, const int32_t* k, int nh, int use_hash, unsigned seed,
                                      const void* w1t_plain, const dbx_view* y, const dbx_view* gate, hipStream_t s) {
    RefGen g;
    if (int rc = fill(g, d_out, w2, k, nh, use_hash, seed)) return rc;
    DBX_REQUIRE(d_out->n == y->n && d_out->h == y->h && d_out->w == y->w && gate->n == y->n && gate->h == y->h && gate->w == y->w && gate->c >= y->c,
                "heads1_dgrad_gen (fp32): d_out (compact), y and gate are maps of the same pixels");
    const int nhid = 512 * nh;
    hipLaunchKernelGGL(heads1_dgrad_gen_f32_kernel, dim3(g.npix), dim3(256), nhid * sizeof(float), s, g, (const float*)w1t_plain, nhid, nhid,
                       make_geo<float>(y), make_geo<float>(gate), y->h, y->w);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
