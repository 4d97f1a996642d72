// HBM-bound helper kernels around the convolutions: layout conversion, 2x2 max-pool, bilinear
// resampling (align_corners=True) and their backward passes.  All operate on framed NHWC views and
// move 16 bytes per lane (8 f16/bf16 or 4 f32 channels).
#include "common.hpp"
#include <mutex>

// ---------------------------------------------------------------------------------------------- error plumbing
static thread_local char g_err[512] = "";
void dbx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* dbx_last_error(void) { return g_err; }
extern "C" int dbx_version(void) { return DBX_ABI_VERSION; }
extern "C" int dbx_device_arch(int device) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) { dbx_set_error("hipGetDeviceProperties failed"); return DBX_ERR_HIP; }
    int arch = 0;
    const char* s = p.gcnArchName;   // "gfx950:sramecc+:xnack-"
    if (s[0] == 'g' && s[1] == 'f' && s[2] == 'x') arch = (int)strtol(s + 3, nullptr, 16) == 0x950 ? 950 : (int)strtol(s + 3, nullptr, 10);
    return arch;
}

template <typename T> struct Vec { static constexpr int N = 16 / sizeof(T); };

template <typename T> __device__ __forceinline__ void load_vec(const T* p, float (&v)[16 / sizeof(T)]) {
    const u32x4 raw = *(const u32x4*)p;
    const T* e = (const T*)&raw;
#pragma unroll
    for (int i = 0; i < 16 / (int)sizeof(T); ++i) v[i] = to_f32(e[i]);
}
template <typename T> __device__ __forceinline__ void store_vec(T* p, const float (&v)[16 / sizeof(T)]) {
    u32x4 raw;
    T* e = (T*)&raw;
#pragma unroll
    for (int i = 0; i < 16 / (int)sizeof(T); ++i) e[i] = from_f32<T>(v[i]);
    *(u32x4*)p = raw;
}

static inline int grid_for(int64_t work, int block = 256) {
    int64_t b = (work + block - 1) / block;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

#define VIEW_VEC_CHECK(T, v, name)                                                                             \
    DBX_REQUIRE(((size_t)(v)->ptr % 16) == 0 && ((v)->ld * sizeof(T)) % 16 == 0 && ((v)->c_off * sizeof(T)) % 16 == 0 && \
                    ((v)->c * sizeof(T)) % 16 == 0, name ": view must be 16-byte aligned/strided")

// ---------------------------------------------------------------------------------------------- NCHW fp32 -> framed
template <typename T>
__global__ void nchw_to_framed_kernel(const float* __restrict__ x, int csrc, FrameGeo y) {
    constexpr int V = Vec<T>::N;
    const int cg = y.c / V;
    const int64_t total = (int64_t)y.n * y.h * y.w * cg;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // lanes run along x fastest (coalesced reads of each source plane); channel group slowest within a pixel row
        const int px = (int)(i % y.w);
        const int g = (int)((i / y.w) % cg);
        const int py = (int)((i / ((int64_t)y.w * cg)) % y.h);
        const int n = (int)(i / ((int64_t)y.w * cg * y.h));
        float v[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int c = g * V + j;
            v[j] = c < csrc ? x[(((int64_t)n * csrc + c) * y.h + py) * y.w + px] : 0.f;
        }
        store_vec<T>((T*)y.base + geo_pix(y, n, py, px) + g * V, v);
    }
}
template <typename T> static int nchw_to_framed_t(const float* x, int csrc, const dbx_view* y, hipStream_t s) {
    VIEW_VEC_CHECK(T, y, "nchw_to_framed");
    FrameGeo g = make_geo<T>(y);
    const int64_t total = (int64_t)y->n * y->h * y->w * (y->c / Vec<T>::N);
    hipLaunchKernelGGL(nchw_to_framed_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, x, csrc, g);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_nchw_to_framed(int32_t dtype, const float* x_nchw, int32_t c_src, const dbx_view* y, void* stream) {
    DBX_DISPATCH_DTYPE(dtype, nchw_to_framed_t, x_nchw, c_src, y, (hipStream_t)stream);
}

// The heads' gradients in one launch: up to four fp32 NCHW tensors (k[i] planes each) into consecutive `slot`-channel ranges of ONE
// framed view (dL/d(head outputs): one slot per head, channels k[i] .. slot-1 of a slot zero) -- four launches of ~9 us each otherwise.
struct SlotSrc { const float* x[4]; int k[4]; };
template <typename T>
__global__ void nchw_to_framed_slots_kernel(SlotSrc src, int nslots, int slot, FrameGeo y) {
    constexpr int V = Vec<T>::N;
    const int gps = slot / V, cg = nslots * gps;                       // channel groups per slot, in all
    const int64_t total = (int64_t)y.n * y.h * y.w * cg;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(i % y.w);
        const int g = (int)((i / y.w) % cg);
        const int py = (int)((i / ((int64_t)y.w * cg)) % y.h);
        const int n = (int)(i / ((int64_t)y.w * cg * y.h));
        const int si = g / gps, c0 = (g % gps) * V;
        const float* x = si == 0 ? src.x[0] : si == 1 ? src.x[1] : si == 2 ? src.x[2] : src.x[3];
        const int k = si == 0 ? src.k[0] : si == 1 ? src.k[1] : si == 2 ? src.k[2] : src.k[3];
        float v[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int c = c0 + j;
            v[j] = (x && c < k) ? x[(((int64_t)n * k + c) * y.h + py) * y.w + px] : 0.f;
        }
        store_vec<T>((T*)y.base + geo_pix(y, n, py, px) + si * slot + c0, v);
    }
}
template <typename T> static int nchw_to_framed_slots_t(const float* const* xs, const int32_t* ks, int nslots, int slot, const dbx_view* y, hipStream_t s) {
    VIEW_VEC_CHECK(T, y, "nchw_to_framed_slots");
    DBX_REQUIRE(nslots >= 1 && nslots <= 4 && slot % Vec<T>::N == 0 && y->c == nslots * slot, "nchw_to_framed_slots: 1..4 slots of a multiple of the vector width, view of nslots * slot channels");
    SlotSrc src;
    for (int i = 0; i < 4; ++i) {
        src.x[i] = i < nslots ? xs[i] : nullptr; src.k[i] = i < nslots ? ks[i] : 0;
        if (i < nslots) DBX_REQUIRE(ks[i] >= 0 && ks[i] <= slot, "nchw_to_framed_slots: k must fit its slot");
    }
    FrameGeo g = make_geo<T>(y);
    const int64_t total = (int64_t)y->n * y->h * y->w * (y->c / Vec<T>::N);
    hipLaunchKernelGGL(nchw_to_framed_slots_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, src, nslots, slot, g);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_nchw_to_framed_slots(int32_t dtype, const float* const* x_nchw, const int32_t* k, int32_t nslots, int32_t slot, const dbx_view* y,
                                        void* stream) {
    if (!x_nchw || !k || !y) { dbx_set_error("nchw_to_framed_slots: null argument"); return DBX_ERR_ARG; }
    DBX_DISPATCH_DTYPE(dtype, nchw_to_framed_slots_t, x_nchw, k, nslots, slot, y, (hipStream_t)stream);
}

template <typename T>
__global__ void framed_to_nchw_kernel(FrameGeo x, float* __restrict__ y) {
    const int64_t total = (int64_t)x.n * x.c * x.h * x.w;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(i % x.w);
        const int py = (int)((i / x.w) % x.h);
        const int c = (int)((i / ((int64_t)x.w * x.h)) % x.c);
        const int n = (int)(i / ((int64_t)x.w * x.h * x.c));
        y[i] = to_f32(((const T*)x.base)[geo_pix(x, n, py, px) + c]);
    }
}
template <typename T> static int framed_to_nchw_t(const dbx_view* x, float* y, hipStream_t s) {
    FrameGeo g = make_geo<T>(x);
    const int64_t total = (int64_t)x->n * x->c * x->h * x->w;
    hipLaunchKernelGGL(framed_to_nchw_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, g, y);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_framed_to_nchw_f32(int32_t dtype, const dbx_view* x, float* y_nchw, void* stream) {
    DBX_DISPATCH_DTYPE(dtype, framed_to_nchw_t, x, y_nchw, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- max-pool 2x2 stride 2 (floor)
// idx (optional): one nibble per pooled element -- bits 0..1 = position of the arg-max in (0,0),(0,1),(1,0),(1,1) order (the FIRST
// element equal to the maximum, as ATen's max_pool2d), bit 2 = (max > 0).  Dense [n][h/2][w/2][c/2] bytes, channel c in byte c/2
// (low nibble = even channel).  The backward pass then needs neither the un-pooled map nor the pooled one (dbx_maxpool2x2_bwd_idx).
template <typename T> struct IdxWord { typedef unsigned int type; };                 // 8 channels per lane: 8 nibbles
template <> struct IdxWord<float> { typedef unsigned short type; };                  // 4 channels per lane
template <typename T>
__global__ void maxpool_kernel(FrameGeo x, FrameGeo y, unsigned char* __restrict__ idx) {
    constexpr int V = Vec<T>::N;
    const int cg = y.c / V;
    const int64_t total = (int64_t)y.n * y.h * y.w * cg;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const int px = (int)((i / cg) % y.w);
        const int py = (int)((i / ((int64_t)cg * y.w)) % y.h);
        const int n = (int)(i / ((int64_t)cg * y.w * y.h));
        const T* p = (const T*)x.base + geo_pix(x, n, 2 * py, 2 * px) + g * V;
        float a[V], b[V], c[V], d[V], o[V];
        load_vec<T>(p, a);
        load_vec<T>(p + x.ld, b);
        load_vec<T>(p + (size_t)x.wp * x.ld, c);
        load_vec<T>(p + (size_t)x.wp * x.ld + x.ld, d);
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = fmaxf(fmaxf(a[j], b[j]), fmaxf(c[j], d[j]));
        store_vec<T>((T*)y.base + geo_pix(y, n, py, px) + g * V, o);
        if (idx) {
            unsigned int word = 0;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                int arg = 0;
                float m = a[j];
                if (b[j] > m) { m = b[j]; arg = 1; }
                if (c[j] > m) { m = c[j]; arg = 2; }
                if (d[j] > m) { m = d[j]; arg = 3; }
                word |= (unsigned int)(arg | (m > 0.f ? 4 : 0)) << (4 * j);
            }
            typedef typename IdxWord<T>::type W;
            *(W*)(idx + ((((size_t)n * y.h + py) * y.w + px) * (size_t)(y.c / 2)) + g * (V / 2)) = (W)word;
        }
    }
}
template <typename T> static int maxpool_t(const dbx_view* x, const dbx_view* y, void* idx, hipStream_t s) {
    VIEW_VEC_CHECK(T, x, "maxpool x"); VIEW_VEC_CHECK(T, y, "maxpool y");
    DBX_REQUIRE(y->h == x->h / 2 && y->w == x->w / 2 && y->c == x->c && y->n == x->n, "maxpool: shape mismatch");
    DBX_REQUIRE(!idx || ((size_t)idx % 4) == 0, "maxpool: idx must be 4-byte aligned");
    const int64_t total = (int64_t)y->n * y->h * y->w * (y->c / Vec<T>::N);
    hipLaunchKernelGGL(maxpool_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, make_geo<T>(x), make_geo<T>(y), (unsigned char*)idx);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_maxpool2x2(int32_t dtype, const dbx_view* x, const dbx_view* y, void* stream) {
    DBX_DISPATCH_DTYPE(dtype, maxpool_t, x, y, nullptr, (hipStream_t)stream);
}
extern "C" int64_t dbx_maxpool_idx_bytes(int32_t n, int32_t h, int32_t w, int32_t c) {
    return (int64_t)n * (h / 2) * (w / 2) * (c / 2);
}
extern "C" int dbx_maxpool2x2_idx(int32_t dtype, const dbx_view* x, const dbx_view* y, void* idx, void* stream) {
    if (!idx) { dbx_set_error("maxpool2x2_idx: null idx"); return DBX_ERR_ARG; }
    DBX_DISPATCH_DTYPE(dtype, maxpool_t, x, y, idx, (hipStream_t)stream);
}

// backward: one lane per (2x2 window, channel group): every byte of x, dy and dx moves exactly once.  The arg-max is
// the FIRST window element equal to the maximum in (0,0),(0,1),(1,0),(1,1) order -- ATen's max_pool2d keeps the earlier
// element on ties.  Windows beyond the pooled extent (odd H/W, floor mode) cover pixels no output saw: gradient 0.
template <typename T>
__global__ void maxpool_bwd_kernel(FrameGeo x, FrameGeo dy, FrameGeo dx, int ph, int pw, int accumulate, int relu_gate) {
    constexpr int V = Vec<T>::N;
    const int cg = x.c / V;
    const int wh = (x.h + 1) >> 1, ww = (x.w + 1) >> 1;               // windows incl. the ragged last row/column
    const int64_t total = (int64_t)x.n * wh * ww * cg;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const int wx = (int)((i / cg) % ww);
        const int wy = (int)((i / ((int64_t)cg * ww)) % wh);
        const int n = (int)(i / ((int64_t)cg * ww * wh));
        const bool covered = wy < ph && wx < pw;
        const size_t p00 = geo_pix(x, n, 2 * wy, 2 * wx) + g * V;
        const size_t d00 = geo_pix(dx, n, 2 * wy, 2 * wx) + g * V;
        const bool has_x1 = 2 * wx + 1 < x.w, has_y1 = 2 * wy + 1 < x.h;
        float o[4][V];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < V; ++j) o[k][j] = 0.f;
        if (covered) {
            float q[4][V], gd[V];
            const T* p = (const T*)x.base + p00;
            load_vec<T>(p, q[0]);
            load_vec<T>(p + x.ld, q[1]);
            load_vec<T>(p + (size_t)x.wp * x.ld, q[2]);
            load_vec<T>(p + (size_t)x.wp * x.ld + x.ld, q[3]);
            load_vec<T>((const T*)dy.base + geo_pix(dy, n, wy, wx) + g * V, gd);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                int arg = 0;
                float m = q[0][j];
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (q[k][j] > m) { m = q[k][j]; arg = k; }
                float v = gd[j];
                if (relu_gate && !(m > 0.f)) v = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k][j] = (k == arg) ? v : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((k & 1) && !has_x1) continue;
            if ((k >> 1) && !has_y1) continue;
            T* dst = (T*)dx.base + d00 + (size_t)(k >> 1) * dx.wp * dx.ld + (size_t)(k & 1) * dx.ld;
            if (accumulate) {
                float old[V];
                load_vec<T>(dst, old);
#pragma unroll
                for (int j = 0; j < V; ++j) o[k][j] += old[j];
            }
            store_vec<T>(dst, o[k]);
        }
    }
}
template <typename T>
static int maxpool_bwd_t(const dbx_view* x, const dbx_view* dy, const dbx_view* dx, int accumulate, int relu_gate, hipStream_t s) {
    VIEW_VEC_CHECK(T, x, "maxpool_bwd x"); VIEW_VEC_CHECK(T, dy, "maxpool_bwd dy"); VIEW_VEC_CHECK(T, dx, "maxpool_bwd dx");
    DBX_REQUIRE(dy->h == x->h / 2 && dy->w == x->w / 2 && dx->h == x->h && dx->w == x->w && dx->c == x->c && dy->c == x->c,
                "maxpool_bwd: shape mismatch");
    const int64_t total = (int64_t)x->n * ((x->h + 1) / 2) * ((x->w + 1) / 2) * (x->c / Vec<T>::N);
    hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, make_geo<T>(x), make_geo<T>(dy),
                       make_geo<T>(dx), dy->h, dy->w, accumulate, relu_gate);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_maxpool2x2_bwd(int32_t dtype, const dbx_view* x, const dbx_view* dy, const dbx_view* dx,
                                  int32_t accumulate, int32_t relu_gate, void* stream) {
    DBX_DISPATCH_DTYPE(dtype, maxpool_bwd_t, x, dy, dx, accumulate, relu_gate, (hipStream_t)stream);
}

// backward from the arg-max nibbles the forward wrote (dbx_maxpool2x2_idx / dbx_conv_forward_pool_idx): reads dy and half a byte
// per pooled element instead of the four un-pooled values -- pool1 at batch 64: 0.62 GB instead of 1.18 GB moved.  Same results
// as dbx_maxpool2x2_bwd on the map the nibbles were taken from (relu_gate uses bit 2).
template <typename T>
__global__ void maxpool_bwd_idx_kernel(const unsigned char* __restrict__ idx, FrameGeo dy, FrameGeo dx, int accumulate, int relu_gate) {
    constexpr int V = Vec<T>::N;
    typedef typename IdxWord<T>::type W;
    const int cg = dx.c / V;
    const int ph = dy.h, pw = dy.w;
    const int wh = (dx.h + 1) >> 1, ww = (dx.w + 1) >> 1;             // windows incl. the ragged last row/column
    const int64_t total = (int64_t)dx.n * wh * ww * cg;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const int wx = (int)((i / cg) % ww);
        const int wy = (int)((i / ((int64_t)cg * ww)) % wh);
        const int n = (int)(i / ((int64_t)cg * ww * wh));
        const bool covered = wy < ph && wx < pw;
        const size_t d00 = geo_pix(dx, n, 2 * wy, 2 * wx) + g * V;
        const bool has_x1 = 2 * wx + 1 < dx.w, has_y1 = 2 * wy + 1 < dx.h;
        float o[4][V];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < V; ++j) o[k][j] = 0.f;
        if (covered) {
            float gd[V];
            load_vec<T>((const T*)dy.base + geo_pix(dy, n, wy, wx) + g * V, gd);
            const unsigned int word = *(const W*)(idx + ((((size_t)n * ph + wy) * pw + wx) * (size_t)(dx.c / 2)) + g * (V / 2));
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const unsigned int nib = (word >> (4 * j)) & 15u;
                const int arg = (int)(nib & 3u);
                const float v = (relu_gate && !(nib & 4u)) ? 0.f : gd[j];
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k][j] = (k == arg) ? v : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((k & 1) && !has_x1) continue;
            if ((k >> 1) && !has_y1) continue;
            T* dst = (T*)dx.base + d00 + (size_t)(k >> 1) * dx.wp * dx.ld + (size_t)(k & 1) * dx.ld;
            if (accumulate) {
                float old[V];
                load_vec<T>(dst, old);
#pragma unroll
                for (int j = 0; j < V; ++j) o[k][j] += old[j];
            }
            store_vec<T>(dst, o[k]);
        }
    }
}
template <typename T>
static int maxpool_bwd_idx_t(const void* idx, const dbx_view* dy, const dbx_view* dx, int accumulate, int relu_gate, hipStream_t s) {
    VIEW_VEC_CHECK(T, dy, "maxpool_bwd_idx dy"); VIEW_VEC_CHECK(T, dx, "maxpool_bwd_idx dx");
    DBX_REQUIRE(dy->h == dx->h / 2 && dy->w == dx->w / 2 && dy->c == dx->c && dy->n == dx->n, "maxpool_bwd_idx: shape mismatch");
    DBX_REQUIRE(((size_t)idx % 4) == 0, "maxpool_bwd_idx: idx must be 4-byte aligned");
    const int64_t total = (int64_t)dx->n * ((dx->h + 1) / 2) * ((dx->w + 1) / 2) * (dx->c / Vec<T>::N);
    hipLaunchKernelGGL(maxpool_bwd_idx_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, (const unsigned char*)idx, make_geo<T>(dy),
                       make_geo<T>(dx), accumulate, relu_gate);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_maxpool2x2_bwd_idx(int32_t dtype, const void* idx, const dbx_view* dy, const dbx_view* dx,
                                      int32_t accumulate, int32_t relu_gate, void* stream) {
    if (!idx || !dy || !dx) { dbx_set_error("maxpool2x2_bwd_idx: null argument"); return DBX_ERR_ARG; }
    DBX_DISPATCH_DTYPE(dtype, maxpool_bwd_idx_t, idx, dy, dx, accumulate, relu_gate, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- bilinear, align_corners=True
// ATen: scale = (in-1)/(out-1) in float; src = scale*dst; i0 = (int)src; i1 = i0 + (i0 < in-1); l1 = src - i0.
__device__ __forceinline__ void bilin_coef(int d, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
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
static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

template <typename T>
__global__ void upsample_kernel(FrameGeo x, FrameGeo y, float sy, float sx) {
    constexpr int V = Vec<T>::N;
    const int cg = y.c / V;
    const int64_t total = (int64_t)y.n * y.h * y.w * cg;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const int px = (int)((i / cg) % y.w);
        const int py = (int)((i / ((int64_t)cg * y.w)) % y.h);
        const int n = (int)(i / ((int64_t)cg * y.w * y.h));
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        bilin_coef(py, sy, x.h, y0, y1, ly0, ly1);
        bilin_coef(px, sx, x.w, x0, x1, lx0, lx1);
        float a[V], b[V], c[V], d[V], o[V];
        const T* base = (const T*)x.base + g * V;
        load_vec<T>(base + geo_pix(x, n, y0, x0), a);
        load_vec<T>(base + geo_pix(x, n, y0, x1), b);
        load_vec<T>(base + geo_pix(x, n, y1, x0), c);
        load_vec<T>(base + geo_pix(x, n, y1, x1), d);
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = ly0 * (lx0 * a[j] + lx1 * b[j]) + ly1 * (lx0 * c[j] + lx1 * d[j]);
        store_vec<T>((T*)y.base + geo_pix(y, n, py, px) + g * V, o);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void upsample_rows_kernel(FrameGeo x, FrameGeo y, float sy, float sx) {
    constexpr int V = Vec<T>::N;
    const int cg = y.c / V;
    const int n = blockIdx.x / y.h, py = blockIdx.x - n * y.h;
    int y0, y1;
    float ly0, ly1;
    bilin_coef(py, sy, x.h, y0, y1, ly0, ly1);
    const T* r0 = (const T*)x.base + geo_pix(x, n, y0, 0);
    const T* r1 = (const T*)x.base + geo_pix(x, n, y1, 0);
    T* out = (T*)y.base + geo_pix(y, n, py, 0);
    const int xs = (int)(geo_pix(x, n, y0, 1) - geo_pix(x, n, y0, 0)), ys = (int)(geo_pix(y, n, py, 1) - geo_pix(y, n, py, 0));
    const int total = y.w * cg;
    // U items per thread with all their loads issued before the first store (the stores may alias the source rows as far as the compiler
    // knows: one item per iteration exposed a load latency per item -- 89 us at batch 64 for 236 MB of output)
    constexpr int U = 4;
    for (int e0 = threadIdx.x; e0 < total; e0 += 256 * U) {
        float a[U][V], b[U][V], c[U][V], d[U][V], lx0[U], lx1[U];
        int px[U], g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + 256 * u < total ? e0 + 256 * u : total - 1;       // (clamped: unconditional loads)
            px[u] = e / cg; g[u] = e - px[u] * cg;
            int x0, x1;
            bilin_coef(px[u], sx, x.w, x0, x1, lx0[u], lx1[u]);
            load_vec<T>(r0 + (size_t)x0 * xs + g[u] * V, a[u]);
            load_vec<T>(r0 + (size_t)x1 * xs + g[u] * V, b[u]);
            load_vec<T>(r1 + (size_t)x0 * xs + g[u] * V, c[u]);
            load_vec<T>(r1 + (size_t)x1 * xs + g[u] * V, d[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float o[V];
#pragma unroll
            for (int j = 0; j < V; ++j) o[j] = ly0 * (lx0[u] * a[u][j] + lx1[u] * b[u][j]) + ly1 * (lx0[u] * c[u][j] + lx1[u] * d[u][j]);
            if (e0 + 256 * u < total) store_vec<T>(out + (size_t)px[u] * ys + g[u] * V, o);
        }
    }
}
template <typename T> static int upsample_t(const dbx_view* x, const dbx_view* y, hipStream_t s) {
    VIEW_VEC_CHECK(T, x, "upsample x"); VIEW_VEC_CHECK(T, y, "upsample y");
    DBX_REQUIRE(x->c == y->c && x->n == y->n, "upsample: channel/batch mismatch");
    // one workgroup per output row: the row's two source rows and weights are uniform, a thread walks (pixel, 16-byte channel
    // group) pairs with 32-bit index math (the flat-index kernel spent its time in 64-bit divisions: 2.3 TB/s -> write-bound)
    hipLaunchKernelGGL(upsample_rows_kernel<T>, dim3(y->n * y->h), dim3(256), 0, s, make_geo<T>(x), make_geo<T>(y),
                       ac_scale(x->h, y->h), ac_scale(x->w, y->w));
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_upsample_bilinear(int32_t dtype, const dbx_view* x, const dbx_view* y, void* stream) {
    DBX_DISPATCH_DTYPE(dtype, upsample_t, x, y, (hipStream_t)stream);
}

// backward in gather form (deterministic, no atomics): each source pixel sums the destination pixels that
// interpolate from it, recomputing the forward coefficients.
template <typename T>
__global__ void upsample_bwd_kernel(FrameGeo dy, FrameGeo dx, FrameGeo gate, int has_gate, float sy, float sx) {
    constexpr int V = Vec<T>::N;
    const int cg = dx.c / V;
    const int64_t total = (int64_t)dx.n * dx.h * dx.w * cg;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const int ix = (int)((i / cg) % dx.w);
        const int iy = (int)((i / ((int64_t)cg * dx.w)) % dx.h);
        const int n = (int)(i / ((int64_t)cg * dx.w * dx.h));
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        // destination rows whose source coordinate lies in (iy-1, iy+1)
        int oy_lo = sy > 0.f ? (int)floorf((float)(iy - 1) / sy) : 0;
        int oy_hi = sy > 0.f ? (int)ceilf((float)(iy + 1) / sy) : dy.h - 1;
        int ox_lo = sx > 0.f ? (int)floorf((float)(ix - 1) / sx) : 0;
        int ox_hi = sx > 0.f ? (int)ceilf((float)(ix + 1) / sx) : dy.w - 1;
        oy_lo = max(oy_lo, 0); ox_lo = max(ox_lo, 0);
        oy_hi = min(oy_hi, dy.h - 1); ox_hi = min(ox_hi, dy.w - 1);
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1; float ly0, ly1;
            bilin_coef(oy, sy, dx.h, y0, y1, ly0, ly1);
            const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1; float lx0, lx1;
                bilin_coef(ox, sx, dx.w, x0, x1, lx0, lx1);
                const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
                if (wx == 0.f) continue;
                float d[V];
                load_vec<T>((const T*)dy.base + geo_pix(dy, n, oy, ox) + g * V, d);
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] += wy * wx * d[j];
            }
        }
        if (has_gate) {
            float gt[V];
            load_vec<T>((const T*)gate.base + geo_pix(gate, n, iy, ix) + g * V, gt);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] = gt[j] > 0.f ? acc[j] : 0.f;
        }
        store_vec<T>((T*)dx.base + geo_pix(dx, n, iy, ix) + g * V, acc);
    }
}
// One workgroup per source row: the destination rows that interpolate from it and their weights are found once (same
// bilin_coef arithmetic as the forward), and so is, per source column, the window of <= 8 destination columns with its weights
// (LDS table).  A thread then only loads and accumulates, in the same (oy, ox) ascending order as the flat kernel above.
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_rows_kernel(FrameGeo dy, FrameGeo dx, FrameGeo gate, int has_gate, float sy, float sx) {
    constexpr int V = Vec<T>::N, WIN = 8, MAXR = 16;
    extern __shared__ __attribute__((aligned(16))) char up_smem[];
    int* s_ox0 = (int*)up_smem;                         // [dx.w]
    float* s_wx = (float*)(s_ox0 + dx.w);               // [dx.w][WIN]
    __shared__ int s_oy[MAXR];
    __shared__ float s_wy[MAXR];
    __shared__ int s_ny;
    const int cg = dx.c / V;
    const int n = blockIdx.x / dx.h, iy = blockIdx.x - n * dx.h;
    if (threadIdx.x == 0) {
        int lo = sy > 0.f ? (int)floorf((float)(iy - 1) / sy) : 0, hi = sy > 0.f ? (int)ceilf((float)(iy + 1) / sy) : dy.h - 1;
        lo = max(lo, 0); hi = min(hi, dy.h - 1);
        int cnt = 0;
        for (int oy = lo; oy <= hi && cnt < MAXR; ++oy) {
            int y0, y1; float ly0, ly1;
            bilin_coef(oy, sy, dx.h, y0, y1, ly0, ly1);
            const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
            if (wy != 0.f) { s_oy[cnt] = oy; s_wy[cnt] = wy; ++cnt; }
        }
        s_ny = cnt;
    }
    for (int ix = threadIdx.x; ix < dx.w; ix += 256) {
        int lo = sx > 0.f ? (int)floorf((float)(ix - 1) / sx) : 0, hi = sx > 0.f ? (int)ceilf((float)(ix + 1) / sx) : dy.w - 1;
        lo = max(lo, 0); hi = min(hi, dy.w - 1);
        int first = -1;
        float w[WIN];
#pragma unroll
        for (int k = 0; k < WIN; ++k) w[k] = 0.f;
        for (int ox = lo; ox <= hi; ++ox) {
            int x0, x1; float lx0, lx1;
            bilin_coef(ox, sx, dx.w, x0, x1, lx0, lx1);
            const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
            if (wx != 0.f) {
                if (first < 0) first = ox;
                if (ox - first < WIN) w[ox - first] = wx;
            }
        }
        s_ox0[ix] = first < 0 ? 0 : first;
#pragma unroll
        for (int k = 0; k < WIN; ++k) s_wx[ix * WIN + k] = w[k];
    }
    __syncthreads();
    const int ny = s_ny, total = dx.w * cg;
    const int dys = (int)(geo_pix(dy, n, 0, 1) - geo_pix(dy, n, 0, 0));
    for (int e = threadIdx.x; e < total; e += 256) {
        const int ix = e / cg, g = e - ix * cg;
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        const int ox0 = s_ox0[ix];
        for (int r = 0; r < ny; ++r) {
            const float wy = s_wy[r];
            const T* row = (const T*)dy.base + geo_pix(dy, n, s_oy[r], 0) + g * V;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const float wx = s_wx[ix * WIN + k];
                if (wx == 0.f) continue;
                float d[V];
                load_vec<T>(row + (size_t)(ox0 + k) * dys, d);
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] += wy * wx * d[j];
            }
        }
        if (has_gate) {
            float gt[V];
            load_vec<T>((const T*)gate.base + geo_pix(gate, n, iy, ix) + g * V, gt);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] = gt[j] > 0.f ? acc[j] : 0.f;
        }
        store_vec<T>((T*)dx.base + geo_pix(dx, n, iy, ix) + g * V, acc);
    }
}
// Wide maps (round 3: the 2048-channel hidden gradient, 944 MB): one workgroup per PAIR of source rows, one thread per 16-byte
// channel group.  A thread walks the destination columns once: per column it loads the <= 8 destination rows that interpolate
// from either source row (each element exactly once per workgroup, 4 KiB contiguous per load instruction across the workgroup),
// reduces them over y with the two rows' weights, and adds the result to two rolling accumulators per source row (the columns
// x0 and x0 + 1 it interpolates from); an accumulator is stored when the walk leaves its column.  Same coefficients as the
// forward (bilin_coef), fp32 accumulation, y before x.
template <typename T>
__global__ __launch_bounds__(256, 4) void upsample_bwd_walk_kernel(FrameGeo dy, FrameGeo dx, FrameGeo gate, int has_gate, float sy, float sx, int pairs, int nseg) {
    constexpr int V = Vec<T>::N, MAXR = 8;
    extern __shared__ __attribute__((aligned(16))) char up_smem[];
    int* s_x0 = (int*)up_smem;                          // [dy.w] first source column of destination column ox
    float* s_l0 = (float*)(s_x0 + dy.w);                // [dy.w] weight of source column x0 (both weights when x1 == x0)
    float* s_l1 = s_l0 + dy.w;                          // [dy.w] weight of source column x0 + 1 (0 when x1 == x0)
    __shared__ int s_oy[MAXR];
    __shared__ float s_w[2][MAXR];
    __shared__ int s_ny, s_oxr[2];
    // workgroup b runs on XCD b % 8: give every XCD a contiguous range of row pairs (neighbouring pairs share destination rows).
    // A row pair is cut into nseg segments of source columns, one workgroup each (more waves in flight: a thread's walk is a serial
    // chain of ~60 load / accumulate rounds); a segment walks the destination columns that touch its source columns and drops what
    // falls outside them.
    const int nwg = gridDim.x;
    int b = blockIdx.x;
    { const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, j = b >> 3; b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j; }
    const int seg = b % nseg; b /= nseg;
    const int c_lo = (int)((long long)dx.w * seg / nseg), c_hi = (int)((long long)dx.w * (seg + 1) / nseg);
    const int n = b / pairs, iy0 = 2 * (b - n * pairs);
    const bool two = iy0 + 1 < dx.h;
    if (threadIdx.x == 0) {
        const int iy1 = two ? iy0 + 1 : iy0;
        int lo = sy > 0.f ? (int)floorf((float)(iy0 - 1) / sy) : 0, hi = sy > 0.f ? (int)ceilf((float)(iy1 + 1) / sy) : dy.h - 1;
        lo = max(lo, 0); hi = min(hi, dy.h - 1);
        int cnt = 0;
        for (int oy = lo; oy <= hi && cnt < MAXR; ++oy) {
            int y0, y1; float ly0, ly1;
            bilin_coef(oy, sy, dx.h, y0, y1, ly0, ly1);
            const float wa = (y0 == iy0 ? ly0 : 0.f) + (y1 == iy0 ? ly1 : 0.f);
            const float wb = two ? (y0 == iy0 + 1 ? ly0 : 0.f) + (y1 == iy0 + 1 ? ly1 : 0.f) : 0.f;
            if (wa != 0.f || wb != 0.f) { s_oy[cnt] = oy; s_w[0][cnt] = wa; s_w[1][cnt] = wb; ++cnt; }
        }
        s_ny = cnt;
    }
    for (int ox = threadIdx.x; ox < dy.w; ox += 256) {
        int x0, x1; float lx0, lx1;
        bilin_coef(ox, sx, dx.w, x0, x1, lx0, lx1);
        s_x0[ox] = x0; s_l0[ox] = x1 == x0 ? lx0 + lx1 : lx0; s_l1[ox] = x1 == x0 ? 0.f : lx1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                             // destination columns with x0 in [c_lo - 1, c_hi - 1]
        int lo = dy.w, hi = -1;
        for (int ox = 0; ox < dy.w; ++ox) { const int x0 = s_x0[ox]; if (x0 >= c_lo - 1 && x0 <= c_hi - 1) { if (ox < lo) lo = ox; hi = ox; } }
        s_oxr[0] = lo; s_oxr[1] = hi;
    }
    __syncthreads();
    const int ox_lo = __builtin_amdgcn_readfirstlane(s_oxr[0]), ox_hi = __builtin_amdgcn_readfirstlane(s_oxr[1]);
    const int ny = __builtin_amdgcn_readfirstlane(s_ny), cg = dx.c / V;
    const int dys = (int)(geo_pix(dy, n, 0, 1) - geo_pix(dy, n, 0, 0)), dxs = (int)(geo_pix(dx, n, 0, 1) - geo_pix(dx, n, 0, 0));
    const int gts = (int)(geo_pix(gate, n, 0, 1) - geo_pix(gate, n, 0, 0));
    // uniform element offsets of the destination rows (rows past ny repeat row 0 with weight 0: fixed trip count, no divergence)
    // (read from LDS, i.e. into VGPRs: readfirstlane moves them to scalar registers -- 32 VGPRs, one wave per SIMD more)
    unsigned rowoff[MAXR];
    float wa[MAXR], wb[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        rowoff[r] = (unsigned)__builtin_amdgcn_readfirstlane((int)geo_pix(dy, n, s_oy[r < ny ? r : 0], 0));
        wa[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, r < ny ? s_w[0][r] : 0.f)));
        wb[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, r < ny ? s_w[1][r] : 0.f)));
    }
    const size_t o0 = geo_pix(dx, n, iy0, 0), o1 = geo_pix(dx, n, two ? iy0 + 1 : iy0, 0);
    const size_t g0 = geo_pix(gate, n, iy0, 0), g1 = geo_pix(gate, n, two ? iy0 + 1 : iy0, 0);
    for (int g = threadIdx.x; g < cg; g += 256) {
        const T* in = (const T*)dy.base + g * V;                        // + uniform offsets: one lane offset register
        T* out = (T*)dx.base + g * V;
        const T* gt = (const T*)gate.base + g * V;
        float c0[V], c1[V], n0[V], n1[V];
#pragma unroll
        for (int j = 0; j < V; ++j) c0[j] = c1[j] = n0[j] = n1[j] = 0.f;
        auto flush = [&](int col, float (&a0)[V], float (&a1)[V]) {
            if (col < c_lo || col >= c_hi) return;                      // (uniform) another segment's column
            if (has_gate) {
                float ga[V], gb[V];
                load_vec<T>(gt + g0 + (size_t)col * gts, ga); load_vec<T>(gt + g1 + (size_t)col * gts, gb);
#pragma unroll
                for (int j = 0; j < V; ++j) { a0[j] = ga[j] > 0.f ? a0[j] : 0.f; a1[j] = gb[j] > 0.f ? a1[j] : 0.f; }
            }
            store_vec<T>(out + o0 + (size_t)col * dxs, a0);
            if (two) store_vec<T>(out + o1 + (size_t)col * dxs, a1);
        };
        if (ox_hi < ox_lo) continue;
        int cur = s_x0[ox_lo];
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const int x0 = s_x0[ox];
            const float l0 = s_l0[ox], l1 = s_l1[ox];
            // all of the column's loads go out first (16 bytes each, raw), then one row at a time is converted and accumulated
            u32x4 raw[MAXR];
#pragma unroll
            for (int r = 0; r < MAXR; ++r)
                if (r < ny) raw[r] = *(const u32x4*)(in + (rowoff[r] + (unsigned)(ox * dys)));
            if (x0 != cur) {                                          // (uniform) the walk left column `cur`
                flush(cur, c0, c1);
#pragma unroll
                for (int j = 0; j < V; ++j) { c0[j] = n0[j]; c1[j] = n1[j]; n0[j] = 0.f; n1[j] = 0.f; }
                cur = x0;
            }
            float v0[V], v1[V];
#pragma unroll
            for (int j = 0; j < V; ++j) v0[j] = v1[j] = 0.f;
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                if (r < ny) {
                    const T* e = (const T*)&raw[r];
#pragma unroll
                    for (int j = 0; j < V; ++j) { const float d = to_f32(e[j]); v0[j] += wa[r] * d; v1[j] += wb[r] * d; }
                }
            }
#pragma unroll
            for (int j = 0; j < V; ++j) { c0[j] += l0 * v0[j]; c1[j] += l0 * v1[j]; n0[j] += l1 * v0[j]; n1[j] += l1 * v1[j]; }
        }
        flush(cur, c0, c1);
        if (cur + 1 < dx.w) flush(cur + 1, n0, n1);
    }
}

template <typename T>
static int upsample_bwd_t(const dbx_view* dy, const dbx_view* dx, const dbx_view* gate, hipStream_t s) {
    VIEW_VEC_CHECK(T, dy, "upsample_bwd dy"); VIEW_VEC_CHECK(T, dx, "upsample_bwd dx");
    DBX_REQUIRE(dx->c == dy->c && dx->n == dy->n, "upsample_bwd: channel/batch mismatch");
    if (gate) { VIEW_VEC_CHECK(T, gate, "upsample_bwd gate"); DBX_REQUIRE(gate->h == dx->h && gate->w == dx->w && gate->c == dx->c, "upsample_bwd: gate shape"); }
    const int64_t total = (int64_t)dx->n * dx->h * dx->w * (dx->c / Vec<T>::N);
    FrameGeo gg = gate ? make_geo<T>(gate) : make_geo<T>(dx);
    const float sy = ac_scale(dx->h, dy->h), sx = ac_scale(dx->w, dy->w);
    // up-sampling by ~2 (every destination column interpolates from x0 and x0 + 1 with x0 non-decreasing, <= 8 destination rows per
    // source-row pair) on maps with >= 64 channel groups: the column walk
    const bool walk_ok = sy > 0.45f && sy < 1.f && sx > 0.45f && sx < 1.f && dx->c / Vec<T>::N >= 64 && dy->w <= 4096;
    // the tabulated kernel covers up-sampling factors up to ~3 (<= 8 destination columns and <= 16 rows per source pixel); its
    // table (36 B per source column) stays inside the 64 KB of dynamic LDS a launch gets without an attribute
    const bool rows_ok = sy > 0.34f && sx > 0.34f && dx->w <= 1536;
    if (walk_ok) {
        const int pairs = (dx->h + 1) / 2;
        // enough workgroups for ~12 waves per SIMD over the channel-group rounds of a thread (at most one segment per 8 source columns)
        const int rounds = (dx->c / Vec<T>::N + 255) / 256;
        int nseg = (int)(12288 / ((int64_t)dx->n * pairs * 4 * rounds) + 1);
        if (nseg > dx->w / 8) nseg = dx->w / 8;
        if (nseg < 1) nseg = 1;
        hipLaunchKernelGGL(upsample_bwd_walk_kernel<T>, dim3(dx->n * pairs * nseg), dim3(256), (size_t)dy->w * 12, s, make_geo<T>(dy), make_geo<T>(dx), gg,
                           gate ? 1 : 0, sy, sx, pairs, nseg);
    } else if (rows_ok)
        hipLaunchKernelGGL(upsample_bwd_rows_kernel<T>, dim3(dx->n * dx->h), dim3(256), (size_t)dx->w * (4 + 8 * 4), s, make_geo<T>(dy), make_geo<T>(dx), gg,
                           gate ? 1 : 0, sy, sx);
    else
        hipLaunchKernelGGL(upsample_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, make_geo<T>(dy), make_geo<T>(dx), gg,
                           gate ? 1 : 0, sy, sx);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_upsample_bilinear_bwd(int32_t dtype, const dbx_view* dy, const dbx_view* dx, const dbx_view* gate, void* stream) {
    DBX_DISPATCH_DTYPE(dtype, upsample_bwd_t, dy, dx, gate, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- channel-slice scatter
// y[n,py,px, c_dst_off + c] = x_nchw[n,c,py,px] for c < c_src; other channels untouched (they stay 0 from
// the workspace memset).  Used to build the refine-branch input cat(landmarks, score) (DenseBox.py:464).
template <typename T>
__global__ void nchw_to_framed_ch_kernel(const float* __restrict__ x, int csrc, FrameGeo y, int c_dst_off) {
    const int64_t total = (int64_t)y.n * csrc * y.h * y.w;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(i % y.w);
        const int py = (int)((i / y.w) % y.h);
        const int c = (int)((i / ((int64_t)y.w * y.h)) % csrc);
        const int n = (int)(i / ((int64_t)y.w * y.h * csrc));
        ((T*)y.base)[geo_pix(y, n, py, px) + c_dst_off + c] = from_f32<T>(x[i]);
    }
}
template <typename T> static int nchw_to_framed_ch_t(const float* x, int csrc, const dbx_view* y, int c_dst_off, hipStream_t s) {
    DBX_REQUIRE(c_dst_off + csrc <= y->c, "nchw_to_framed_ch: slice out of range");
    const int64_t total = (int64_t)y->n * csrc * y->h * y->w;
    hipLaunchKernelGGL(nchw_to_framed_ch_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, x, csrc, make_geo<T>(y), c_dst_off);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_nchw_to_framed_ch(int32_t dtype, const float* x_nchw, int32_t c_src, const dbx_view* y,
                                     int32_t c_dst_off, void* stream) {
    DBX_DISPATCH_DTYPE(dtype, nchw_to_framed_ch_t, x_nchw, c_src, y, c_dst_off, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- channel-slice accumulate
// dst[n,y,x, c_dst_off + j] += src[n,y,x, c_src_off + j], j < n_ch.  Routes the refine branch's input gradient
// (d cat(landmarks, score), DenseBox.py:464) back onto the landmark / score head gradients.
template <typename T>
__global__ void framed_add_ch_kernel(FrameGeo src, int c_src_off, int n_ch, FrameGeo dst, int c_dst_off) {
    const int64_t total = (int64_t)src.n * src.h * src.w * n_ch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % n_ch);
        const int px = (int)((i / n_ch) % src.w);
        const int py = (int)((i / ((int64_t)n_ch * src.w)) % src.h);
        const int n = (int)(i / ((int64_t)n_ch * src.w * src.h));
        T* d = (T*)dst.base + geo_pix(dst, n, py, px) + c_dst_off + j;
        *d = from_f32<T>(to_f32(*d) + to_f32(((const T*)src.base)[geo_pix(src, n, py, px) + c_src_off + j]));
    }
}
template <typename T>
static int framed_add_ch_t(const dbx_view* src, int c_src_off, int n_ch, const dbx_view* dst, int c_dst_off, hipStream_t s) {
    DBX_REQUIRE(src->n == dst->n && src->h == dst->h && src->w == dst->w, "framed_add_ch: shape mismatch");
    DBX_REQUIRE(c_src_off + n_ch <= src->c && c_dst_off + n_ch <= dst->c, "framed_add_ch: slice out of range");
    const int64_t total = (int64_t)src->n * src->h * src->w * n_ch;
    hipLaunchKernelGGL(framed_add_ch_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, make_geo<T>(src), c_src_off, n_ch,
                       make_geo<T>(dst), c_dst_off);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_framed_add_ch(int32_t dtype, const dbx_view* src, int32_t c_src_off, int32_t n_ch, const dbx_view* dst,
                                 int32_t c_dst_off, void* stream) {
    DBX_DISPATCH_DTYPE(dtype, framed_add_ch_t, src, c_src_off, n_ch, dst, c_dst_off, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- heads, 512 -> k backward
// d_hid[m, 512h + c] = 2*mask[m, 512h + c] * sum_{j<k_h} d_out_h[m, j] * W2_h[j][c]   for all heads h in one pass
// (the data gradient of Conv1x1(512->k) behind nn.Dropout, DenseBox.py:158-162; k <= 8).  Rank-k and write-bound
// (1 KiB per pixel per head): a streaming kernel -- one wave per (pixel, head), 16 bytes per lane, the lane's k x 8
// slice of the fp32 master weights in registers -- so that a pixel's 512*nh channels leave as one contiguous burst.
struct Head2Args { const float* w2[4]; int k[4]; int nh; int slot; };

// The lane's k x V slice of its head's fp32 weights.  The head's pointer and k are SELECTED from the kernel arguments (constant
// indices: scalar registers) and the rows fetched with 16-byte loads of a clamped row index -- indexing the argument arrays with
// the per-lane head number made the compiler read them through memory, one dependent pointer load + one scalar load per
// element, each behind s_waitcnt vmcnt(0): 128 serialised round trips (~50 us) at the head of every workgroup.
template <int V>
__device__ __forceinline__ void head2_load_w(const Head2Args& ha, int hd, int c0, bool active, float (&w)[8][V], int& k_out) {
    const float* wp = ha.w2[0];
    int k = ha.k[0];
#pragma unroll
    for (int hh = 1; hh < 4; ++hh)
        if (hd == hh && ha.w2[hh]) { wp = ha.w2[hh]; k = ha.k[hh]; }
    k_out = k;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = j < k ? j : (k > 0 ? k - 1 : 0);
        const bool use = active && j < k;
#pragma unroll
        for (int q = 0; q < V / 4; ++q) {
            const f32x4 v = *(const f32x4*)(wp + (size_t)row * 512 + c0 + 4 * q);
            w[j][4 * q] = use ? v.x : 0.f; w[j][4 * q + 1] = use ? v.y : 0.f; w[j][4 * q + 2] = use ? v.z : 0.f; w[j][4 * q + 3] = use ? v.w : 0.f;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(512) void head2_dgrad_kernel(FrameGeo dout, Head2Args ha, FrameGeo dhid,
                                                          const unsigned char* __restrict__ mask, int mask_ld, int use_hash,
                                                          unsigned drop_seed) {
    constexpr int V = Vec<T>::N;                  // channels per lane
    constexpr int LPH = 512 / V;                  // lanes per (pixel, head): 64 for 16-bit types, 128 for f32
    const int lph_all = LPH * ha.nh;              // lanes per pixel
    const int ppb = blockDim.x / lph_all > 0 ? blockDim.x / lph_all : 1;
    const int sub = threadIdx.x % lph_all;        // position inside the pixel
    const int hd = sub / LPH, c0 = (sub % LPH) * V;
    const bool active = threadIdx.x < ppb * lph_all;
    int k;
    float w[8][V];
    head2_load_w<V>(ha, hd, c0, active, w, k);
    if (!active) return;
    const int nrows = dhid.n * dhid.h;            // one workgroup per image row: no division in the pixel loop
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int n = row / dhid.h, py = row - n * dhid.h;
        for (int px = threadIdx.x / lph_all; px < dhid.w; px += ppb) {
            const int64_t m = (int64_t)row * dhid.w + px;
            const T* g = (const T*)dout.base + geo_pix(dout, n, py, px) + hd * ha.slot;
            float gj[8];
            if constexpr (sizeof(T) == 2) {       // a slot is >= 8 channels = one 16-byte load (padding channels are 0)
                float tmp[V];
                load_vec<T>(g, tmp);
#pragma unroll
                for (int j = 0; j < 8; ++j) gj[j] = tmp[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) gj[j] = j < k ? g[j] : 0.f;
            }
            float o[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += gj[j] * w[j][i];
                o[i] = acc;
            }
            if (mask) {
                const unsigned char* mk = mask + (size_t)m * mask_ld + hd * 512 + c0;
                unsigned char mb[V];
                if constexpr (V == 8) *(u32x2*)mb = *(const u32x2*)mk; else *(unsigned int*)mb = *(const unsigned int*)mk;
#pragma unroll
                for (int i = 0; i < V; ++i) o[i] = mb[i] ? o[i] * 2.f : 0.f;
            } else if (use_hash) {                      // same keep bits as the forward epilogue (DBX_EPI_DROPHASH)
#pragma unroll
                for (int q = 0; q < V / 4; ++q) {
                    const unsigned kb = dbx_drop_bits4(drop_seed, (unsigned)m, (unsigned)(hd * 512 + c0) / 4 + q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[4 * q + i] = (kb >> i & 1u) ? o[4 * q + i] * 2.f : 0.f;
                }
            }
            store_vec<T>((T*)dhid.base + geo_pix(dhid, n, py, px) + hd * 512 + c0, o);
        }
    }
}
template <typename T>
static int head2_dgrad_t(const dbx_view* dout, const float* const* w2, const int32_t* k, int nh, const dbx_view* dhid,
                         const uint8_t* mask, int mask_ld, int use_hash, unsigned drop_seed, hipStream_t s) {
    VIEW_VEC_CHECK(T, dhid, "head2_dgrad d_hid");
    DBX_REQUIRE(nh >= 1 && nh <= 4 && dhid->c == 512 * nh && dout->c % nh == 0 && dout->c / nh >= 8, "head2_dgrad: nh in 1..4, d_hid of 512*nh channels, d_out of nh slots >= 8 channels");
    DBX_REQUIRE(dout->n == dhid->n && dout->h == dhid->h && dout->w == dhid->w, "head2_dgrad: shape mismatch");
    DBX_REQUIRE(((size_t)dout->ptr % 16) == 0 && (dout->ld * sizeof(T)) % 16 == 0 && (dout->c_off * sizeof(T)) % 16 == 0 &&
                    ((dout->c / nh) * sizeof(T)) % 16 == 0, "head2_dgrad: d_out alignment");
    Head2Args ha;
    ha.nh = nh; ha.slot = dout->c / nh;
    for (int i = 0; i < 4; ++i) { ha.w2[i] = i < nh ? w2[i] : nullptr; ha.k[i] = i < nh ? k[i] : 0; if (i < nh) DBX_REQUIRE(k[i] >= 1 && k[i] <= 8 && w2[i] && ((size_t)w2[i] % 16) == 0, "head2_dgrad: k in 1..8, 16-byte aligned weights"); }
    const int lph_all = (512 / Vec<T>::N) * nh;
    const int threads = lph_all <= 256 ? 256 : 512;
    int blocks = dhid->n * dhid->h; blocks = blocks > 8192 ? 8192 : blocks;
    hipLaunchKernelGGL(head2_dgrad_kernel<T>, dim3(blocks), dim3(threads), 0, s, make_geo<T>(dout), ha, make_geo<T>(dhid), mask, mask_ld, use_hash, drop_seed);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_head2_dgrad(int32_t dtype, const dbx_view* d_out, const float* const* w2, const int32_t* k, int32_t nh,
                               const dbx_view* d_hid, const uint8_t* dropmask, int32_t dropmask_ld, int32_t use_hash,
                               uint32_t drop_seed, void* stream) {
    if (!d_out || !w2 || !k || !d_hid) { dbx_set_error("head2_dgrad: null argument"); return DBX_ERR_ARG; }
    DBX_DISPATCH_DTYPE(dtype, head2_dgrad_t, d_out, w2, k, nh, d_hid, dropmask, dropmask_ld, use_hash, (unsigned)drop_seed, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- stage-2 head weight gradient
// dW2[head][j][c] = sum_pixels d_out[pixel][head slot][j] * hid[pixel][512*head + c]  (k <= 8 outputs per head), and
// db2[head][j] = sum_pixels d_out[..][j].  Rank-k updates against the 944 MB hidden map: one streaming pass for all
// heads (the per-head GEMM tiles would stride through it five times with 64x64 tiles that are 88 % padding).  Lanes are
// laid out as in head2_dgrad (one 16-byte chunk of one head's 512 channels per lane, whole pixels per workgroup);
// every lane keeps its 8 x V accumulator block in registers, workgroups own fixed image rows, and the per-workgroup
// partials are summed in a fixed order by head2_wgrad_reduce_kernel (bitwise repeatable).
//
// DG = true (dbx_head2_backward) also produces the data gradient of head2_dgrad_kernel for the pixel it is accumulating:
// d_out and the weights are in registers already, so the 1 GB d_hid write overlaps the 1 GB hid read in ONE pass over the
// pixels instead of a write-only pass followed by a read-only pass (HBM moves ~5 TB/s mixed, ~3.5 TB/s either way alone).
template <typename T, bool DG>
__global__ __launch_bounds__(512) void head2_wgrad_kernel(FrameGeo dout, FrameGeo hid, int nh, int slot, float* __restrict__ partial,
                                                          float* __restrict__ bpartial, Head2Args ha, FrameGeo dhid,
                                                          const unsigned char* __restrict__ mask, int mask_ld, int use_hash,
                                                          unsigned drop_seed) {
    constexpr int V = Vec<T>::N;
    constexpr int LPH = 512 / V;
    const int lph_all = LPH * nh;
    const int ppb = blockDim.x / lph_all > 0 ? blockDim.x / lph_all : 1;
    const int sub = threadIdx.x % lph_all, grp = threadIdx.x / lph_all;
    const int hd = sub / LPH, c0 = (sub % LPH) * V;
    const bool active = threadIdx.x < ppb * lph_all;
    typedef float f32x2 __attribute__((ext_vector_type(2)));        // pairs: the accumulation compiles to v_pk_fma_f32
    f32x2 acc[8][V / 2];
    float bs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bs[j] = 0.f;
#pragma unroll
        for (int i = 0; i < V / 2; ++i) acc[j][i] = (f32x2){0.f, 0.f};
    }
    // a workgroup owns a contiguous range of pixels; each lane group walks it with stride ppb, the next pixel's two
    // 16-byte loads in flight while the current one is accumulated (the loop is latency-bound otherwise)
    const long long npix = (long long)hid.n * hid.h * hid.w;
    const long long per = (npix + gridDim.x - 1) / gridDim.x;
    const long long p0 = (long long)blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
    float w[8][V];
    if constexpr (DG) { int k; head2_load_w<V>(ha, hd, c0, active, w, k); }
    if (active && p0 + grp < p1) {
        long long p = p0 + grp;
        int n = (int)(p / ((long long)hid.h * hid.w));
        int rem = (int)(p - (long long)n * hid.h * hid.w);
        int py = rem / hid.w, px = rem - py * hid.w;
        int nc = n, pyc = py, pxc = px;                                   // coordinates of the pixel being consumed (DG)
        // raw 16-byte chunks of three pixels ahead stay in flight (two loads each); converted when consumed
        constexpr int D = 3;
        u32x4 graw[D], ghi[D], hraw[D];                                   // ghi: channels 4..7 of an f32 slot
        auto fetch = [&](int d) {
            const T* g = (const T*)dout.base + geo_pix(dout, n, py, px) + hd * slot;
            graw[d] = *(const u32x4*)g;
            if constexpr (sizeof(T) == 4) ghi[d] = *(const u32x4*)(g + 4);
            hraw[d] = *(const u32x4*)((const T*)hid.base + geo_pix(hid, n, py, px) + hd * 512 + c0);
        };
        auto advance = [&]() {
            p += ppb; px += ppb;
            while (px >= hid.w) { px -= hid.w; if (++py == hid.h) { py = 0; ++n; } }
        };
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (p < p1) fetch(d);
            advance();
        }
        long long pc = p0 + grp;                                         // pixel being consumed
        while (pc < p1) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (pc < p1) {
                    float gj[8], h[V];
                    if constexpr (sizeof(T) == 2) {
                        const T* ge = (const T*)&graw[d];
#pragma unroll
                        for (int j = 0; j < 8; ++j) gj[j] = to_f32(ge[j]);
                    } else {
                        const float* ge = (const float*)&graw[d];
                        const float* gh = (const float*)&ghi[d];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { gj[j] = ge[j]; gj[j + 4] = gh[j]; }
                    }
                    const T* he = (const T*)&hraw[d];
#pragma unroll
                    for (int i = 0; i < V; ++i) h[i] = to_f32(he[i]);
                    if (p < p1) fetch(d);
                    advance();
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        bs[j] += gj[j];
                        const f32x2 g2 = {gj[j], gj[j]};
#pragma unroll
                        for (int i = 0; i < V / 2; ++i) acc[j][i] = g2 * (f32x2){h[2 * i], h[2 * i + 1]} + acc[j][i];
                    }
                    if constexpr (DG) {                                   // same arithmetic and order as head2_dgrad_kernel
                        float o[V];
#pragma unroll
                        for (int i = 0; i < V; ++i) {
                            float a2 = 0.f;
#pragma unroll
                            for (int j = 0; j < 8; ++j) a2 += gj[j] * w[j][i];
                            o[i] = a2;
                        }
                        if (mask) {
                            const unsigned char* mk = mask + (size_t)pc * mask_ld + hd * 512 + c0;
                            unsigned char mb[V];
                            if constexpr (V == 8) *(u32x2*)mb = *(const u32x2*)mk; else *(unsigned int*)mb = *(const unsigned int*)mk;
#pragma unroll
                            for (int i = 0; i < V; ++i) o[i] = mb[i] ? o[i] * 2.f : 0.f;
                        } else if (use_hash) {
#pragma unroll
                            for (int q = 0; q < V / 4; ++q) {
                                const unsigned kb = dbx_drop_bits4(drop_seed, (unsigned)pc, (unsigned)(hd * 512 + c0) / 4 + q);
#pragma unroll
                                for (int i = 0; i < 4; ++i) o[4 * q + i] = (kb >> i & 1u) ? o[4 * q + i] * 2.f : 0.f;
                            }
                        }
                        store_vec<T>((T*)dhid.base + geo_pix(dhid, nc, pyc, pxc) + hd * 512 + c0, o);
                    }
                }
                if constexpr (DG) { pxc += ppb; while (pxc >= hid.w) { pxc -= hid.w; if (++pyc == hid.h) { pyc = 0; ++nc; } } }
                pc += ppb;
            }
        }
    }
    // combine the ppb pixel groups of this workgroup in a fixed order through LDS, then one partial block per workgroup
    __shared__ float red[512 * 9];
    float* P = partial + (size_t)blockIdx.x * nh * 8 * 512;
    for (int j = 0; j < 8; ++j) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < V; ++i) red[threadIdx.x * V + i] = acc[j][i >> 1][i & 1];
        if (V == 8) {}                                                   // (red holds 512 * V floats at most: V <= 8)
        __syncthreads();
        if (threadIdx.x < lph_all) {
            float o[V];
#pragma unroll
            for (int i = 0; i < V; ++i) o[i] = 0.f;
            for (int q = 0; q < ppb; ++q)
#pragma unroll
                for (int i = 0; i < V; ++i) o[i] += red[(q * lph_all + sub) * V + i];
#pragma unroll
            for (int i = 0; i < V; ++i) P[((size_t)hd * 8 + j) * 512 + c0 + i] = o[i];
        }
    }
    __syncthreads();
    if (active && c0 == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(grp * 4 + hd) * 8 + j] = bs[j];
    }
    __syncthreads();
    if (threadIdx.x < nh * 8) {
        const int hh = threadIdx.x / 8, j = threadIdx.x % 8;
        float o = 0.f;
        for (int q = 0; q < ppb; ++q) o += red[(q * 4 + hh) * 8 + j];
        bpartial[(size_t)blockIdx.x * 32 + hh * 8 + j] = o;
    }
}
struct Head2Out { float* dw[4]; float* db[4]; int k[4];
                  // where head hd's partial blocks are: the launch that produced them wrote [blk][nhl][8][512] floats at poff (+ [blk][32] bias
                  // sums at boff), head hd is its local head hl; nblk[hd] blocks (one launch for all heads: poff 0, nhl = nh, hl = hd)
                  long long poff[4], boff[4]; int nblk[4], nhl[4], hl[4]; };
// 64 consecutive (head, j, c) elements per workgroup; wave w of 16 sums partial blocks w, w+16, ... (four loads in flight),
// the 16 wave sums are added in a fixed order.
__global__ __launch_bounds__(1024) void head2_wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bpartial, int nblk,
                                                                   int nh, Head2Out o) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ne = nh * 8 * 512, e = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < ne) {                                                       // (a workgroup's 64 elements belong to one head)
        const int hd = e / 4096;
        const size_t stride = (size_t)o.nhl[hd] * 4096;
        const float* pp = partial + o.poff[hd] + (size_t)o.hl[hd] * 4096 + (e - hd * 4096);
        const int nb = o.nblk[hd];
        int b = w;
        for (; b + 48 < nb; b += 64) {
            s0 += pp[(size_t)b * stride]; s1 += pp[(size_t)(b + 16) * stride];
            s2 += pp[(size_t)(b + 32) * stride]; s3 += pp[(size_t)(b + 48) * stride];
        }
        for (; b < nb; b += 16) s0 += pp[(size_t)b * stride];
    } else if (e < ne + nh * 8) {
        const int q = e - ne, hd = q / 8;
        const float* bp = bpartial + o.boff[hd] + o.hl[hd] * 8 + (q % 8);
        for (int b = w; b < o.nblk[hd]; b += 16) s0 += bp[(size_t)b * 32];
    }
    red[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0) {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) sum += red[q][lane];
        if (e < ne) {
            const int hd = e / (8 * 512), j = (e / 512) % 8, c = e % 512;
            if (j < o.k[hd]) o.dw[hd][j * 512 + c] = sum;
        } else if (e < ne + nh * 8) {
            const int q = e - ne, hd = q / 8, j = q % 8;
            if (j < o.k[hd] && o.db[hd]) o.db[hd][j] = sum;
        }
    }
}
static int head2_wgrad_blocks(int rows) { return rows < 512 ? rows : 512; }
extern "C" int64_t dbx_head2_wgrad_scratch_bytes(int32_t nh, int32_t rows) {
    return (int64_t)head2_wgrad_blocks(rows) * ((int64_t)nh * 8 * 512 + 32) * 4;
}
template <typename T>
static int head2_wgrad_t(const dbx_view* dout, const dbx_view* hid, const int32_t* k, int nh, float* const* dw, float* const* db,
                         void* scratch, hipStream_t s, const float* const* w2 = nullptr, const dbx_view* dhid = nullptr,
                         const uint8_t* mask = nullptr, int mask_ld = 0, int use_hash = 0, unsigned drop_seed = 0) {
    VIEW_VEC_CHECK(T, hid, "head2_wgrad hid");
    DBX_REQUIRE(nh >= 1 && nh <= 4 && hid->c == 512 * nh && dout->c % nh == 0 && dout->c / nh >= 8, "head2_wgrad: nh in 1..4, hid of 512*nh channels, d_out of nh slots >= 8 channels");
    DBX_REQUIRE(dout->n == hid->n && dout->h == hid->h && dout->w == hid->w, "head2_wgrad: shape mismatch");
    DBX_REQUIRE(((size_t)dout->ptr % 16) == 0 && (dout->ld * sizeof(T)) % 16 == 0 && (dout->c_off * sizeof(T)) % 16 == 0 &&
                    ((dout->c / nh) * sizeof(T)) % 16 == 0, "head2_wgrad: d_out alignment");
    Head2Out o;
    for (int i = 0; i < 4; ++i) {
        o.dw[i] = i < nh ? dw[i] : nullptr; o.db[i] = (i < nh && db) ? db[i] : nullptr; o.k[i] = i < nh ? k[i] : 0;
        if (i < nh) DBX_REQUIRE(k[i] >= 1 && k[i] <= 8 && dw[i], "head2_wgrad: k in 1..8");
    }
    const int lph_all = (512 / Vec<T>::N) * nh;
    const int threads = lph_all <= 256 ? 256 : 512;
    const int blocks = head2_wgrad_blocks(hid->n * hid->h);
    float* partial = (float*)scratch;
    float* bpartial = partial + (size_t)blocks * nh * 8 * 512;
    Head2Args ha;
    ha.nh = nh; ha.slot = dout->c / nh;
    for (int i = 0; i < 4; ++i) { ha.w2[i] = (w2 && i < nh) ? w2[i] : nullptr; ha.k[i] = i < nh ? k[i] : 0; }
    if (dhid) {
        VIEW_VEC_CHECK(T, dhid, "head2_backward d_hid");
        DBX_REQUIRE(w2 && dhid->c == 512 * nh && dout->n == dhid->n && dout->h == dhid->h && dout->w == dhid->w, "head2_backward: d_hid of 512*nh channels on the d_out grid");
        for (int i = 0; i < nh; ++i) DBX_REQUIRE(w2[i] && ((size_t)w2[i] % 16) == 0, "head2_backward: null or unaligned weight");
        hipLaunchKernelGGL((head2_wgrad_kernel<T, true>), dim3(blocks), dim3(threads), 0, s, make_geo<T>(dout), make_geo<T>(hid), nh, dout->c / nh,
                           partial, bpartial, ha, make_geo<T>(dhid), mask, mask_ld, use_hash, drop_seed);
    } else
        hipLaunchKernelGGL((head2_wgrad_kernel<T, false>), dim3(blocks), dim3(threads), 0, s, make_geo<T>(dout), make_geo<T>(hid), nh, dout->c / nh,
                           partial, bpartial, ha, make_geo<T>(hid), (const unsigned char*)nullptr, 0, 0, 0u);
    DBX_LAUNCH_CHECK();
    for (int i = 0; i < 4; ++i) { o.poff[i] = 0; o.boff[i] = 0; o.nblk[i] = blocks; o.nhl[i] = nh; o.hl[i] = i; }
    const int total = nh * 8 * 512 + nh * 8;
    hipLaunchKernelGGL(head2_wgrad_reduce_kernel, dim3((total + 63) / 64), dim3(1024), 0, s, partial, bpartial, blocks, nh, o);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_head2_wgrad(int32_t dtype, const dbx_view* d_out, const dbx_view* hid, const int32_t* k, int32_t nh,
                               float* const* dw, float* const* db, void* scratch, void* stream) {
    if (!d_out || !hid || !k || !dw || !scratch) { dbx_set_error("head2_wgrad: null argument"); return DBX_ERR_ARG; }
    DBX_DISPATCH_DTYPE(dtype, head2_wgrad_t, d_out, hid, k, nh, dw, db, scratch, (hipStream_t)stream);
}

// Stage-2 heads backward in one pass: dbx_head2_wgrad + dbx_head2_dgrad (bitwise the same results as the two calls).
extern "C" int dbx_head2_backward(int32_t dtype, const dbx_view* d_out, const dbx_view* hid, const float* const* w2, const int32_t* k,
                                  int32_t nh, const dbx_view* d_hid, const uint8_t* dropmask, int32_t dropmask_ld, int32_t use_hash,
                                  uint32_t drop_seed, float* const* dw, float* const* db, void* scratch, void* stream) {
    if (!d_out || !hid || !w2 || !k || !d_hid || !dw || !scratch) { dbx_set_error("head2_backward: null argument"); return DBX_ERR_ARG; }
    DBX_DISPATCH_DTYPE(dtype, head2_wgrad_t, d_out, hid, k, nh, dw, db, scratch, (hipStream_t)stream, w2, d_hid, dropmask, dropmask_ld,
                       use_hash, (unsigned)drop_seed);
}

// ---------------------------------------------------------------------------------------------- stage-2 heads backward + up^T
// dbx_head2_backward_up: dbx_head2_backward AND the transposed bilinear up-sampling of the hidden gradient it writes
// (d_g44 = up^T(d_hid): dbx_upsample_bilinear_bwd without a gate) in ONE pass over the hidden map.  The separate up^T pass re-read
// the 944 MB d_hid that had just been written (311 us at batch 64).  Here a workgroup owns (64-channel slice, half of the image width)
// and streams rows top to bottom, image after image -- 8 lanes x 16 bytes per pixel, 128-byte segments of a pixel's channels -- so
// the hidden gradient of a row is in registers when the rows of conv4_4's grid that interpolate from it are accumulated: every lane
// keeps the y-reduced values of the two source rows its pixel column feeds (va, vb); when the walk leaves a source row the row goes
// to an LDS ring and NB rows at a time are reduced over x and stored.  The two halves of a row overlap by the few destination
// columns both sides' source columns interpolate from (their d_hid is computed twice, stored and accumulated by the owner only), so
// the halves are independent workgroups: 256 threads, two per CU -- one's barriers and x reductions run under the other's streaming
// (3 % faster than one 512-thread workgroup per CU walking whole rows; what the kernel's speed hangs on is in the comments at the row
// loop: unconditional prefetch loads, scalar row bases, a branch-free body).  Coefficients, order of accumulation (y before x, ascending) and rounding
// points (d_hid rounded to T first) are those of upsample_bwd_walk_kernel; d_hid is bitwise head2_dgrad_kernel's.  The
// weight-gradient partials are one block per workgroup (fixed order), combined by head2_wgrad_reduce_kernel as before.
#ifndef H2U_D
#define H2U_D 2
#endif
#ifndef H2U_NB
#define H2U_NB 4
#endif
#ifndef H2U_WGS
#define H2U_WGS 512
#endif
struct H2UHalf { int px_lo, px_n, own_lo, own_hi, ix_lo, ix_n; };      // pixel columns walked / owned [own_lo, own_hi), source columns reduced
struct H2UPlan { int halves; H2UHalf h[2]; };
// KJ: the d_out channels multiplied out (the launch's heads have k <= KJ: one launch per run of heads with the same rounded-up k --
// one loop body per kernel, unlike the per-k copies inside one kernel that overflowed the instruction cache); hd0 / nhl: the launch's
// first head and head count (its hidden channels are hid's [512 hd0, 512 (hd0 + nhl)), its partial blocks [blk][nhl][8][512]).
template <typename T, int KJ>
__global__ __launch_bounds__(256, 2) void head2_backward_up_kernel(FrameGeo dout, FrameGeo hid, FrameGeo dhid, FrameGeo dg, int hd0, int nhl, int slot,
                                                                   Head2Args ha, float* __restrict__ partial, float* __restrict__ bpartial,
                                                                   const unsigned char* __restrict__ mask, int mask_ld, int use_hash,
                                                                   unsigned drop_seed, float sy, float sx, H2UPlan plan, int nostore) {
    constexpr int V = 8, CS = 64, LPP = CS / V, D = H2U_D, PW = 32, NB = H2U_NB, NT = 6;
    static_assert(sizeof(T) == 2, "16-bit element types");
    __shared__ int s_x0[64], s_lo[32], s_hi[32];
    __shared__ float s_l0[64], s_l1[64];
    extern __shared__ __attribute__((aligned(16))) char h2u_smem[];
    float* const s_v = (float*)h2u_smem;                                // ring of 2 * NB row buffers [PW][CS] fp32
    const int nsl = nhl * (512 / CS), H = hid.h, W = hid.w;
    int b = blockIdx.x;
    const int sl = b % nsl; b /= nsl;
    const int half = b % plan.halves, grp = b / plan.halves, G = gridDim.x / (nsl * plan.halves);   // images grp, grp + G, ...
    const H2UHalf hf = half ? plan.h[1] : plan.h[0];
    const int hd = hd0 + sl / (512 / CS), cbase = (sl % (512 / CS)) * CS;
    const int pl = threadIdx.x / LPP, ch = threadIdx.x % LPP, c0 = cbase + ch * V;
    const int px = hf.px_lo + pl;
    const bool active = pl < hf.px_n, owner = active && px >= hf.own_lo && px < hf.own_hi;
    if (threadIdx.x < W) {
        int x0, x1; float lx0, lx1;
        bilin_coef((int)threadIdx.x, sx, dg.w, x0, x1, lx0, lx1);
        s_x0[threadIdx.x] = x0; s_l0[threadIdx.x] = x1 == x0 ? lx0 + lx1 : lx0; s_l1[threadIdx.x] = x1 == x0 ? 0.f : lx1;
    }
    __syncthreads();
    if (threadIdx.x < hf.ix_n) {                 // destination columns that interpolate from source column ix or ix - 1 (x0 is non-decreasing)
        const int ix = hf.ix_lo + threadIdx.x;
        int lo = W, hi = -1;
        for (int ox = 0; ox < W; ++ox) { const int x0 = s_x0[ox]; if (x0 == ix || x0 == ix - 1) { if (ox < lo) lo = ox; hi = ox; } }
        s_lo[threadIdx.x] = lo; s_hi[threadIdx.x] = hi;
    }
    __syncthreads();
    // the x reduction of a finished source row: thread -> (source column, channel quad), its <= NT destination columns (fewer than
    // 2 / sx + 1) as row-buffer positions and their weights in registers
    const int tl = threadIdx.x / (CS / 4), tq = threadIdx.x % (CS / 4), tix = hf.ix_lo + tl;
    const bool xtask = tl < hf.ix_n;
    int tlo = 0, tcnt = 0;
    float tcf[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) tcf[t] = 0.f;
    if (xtask) {
        const int lo = s_lo[tl];
        tcnt = s_hi[tl] - lo + 1; tlo = lo - hf.px_lo;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ox = lo + t < W ? lo + t : W - 1;
            tcf[t] = s_x0[ox] == tix ? s_l0[ox] : s_l1[ox];
        }
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc[KJ][V / 2];
    float bs[KJ], w[8][V];
    f32x2 va[V / 2], vb[V / 2], w2[KJ][V / 2];
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
        bs[j] = 0.f;
#pragma unroll
        for (int i = 0; i < V / 2; ++i) acc[j][i] = (f32x2){0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < V / 2; ++i) va[i] = vb[i] = (f32x2){0.f, 0.f};
    { int k; head2_load_w<V>(ha, hd, c0, active, w, k); }
    if (nostore) {                                                      // the consumers generate d_hid with W2 in the compute dtype (GenHid): the same values here
#pragma unroll
        for (int j = 0; j < KJ; ++j)
#pragma unroll
            for (int i = 0; i < V; ++i) w[j][i] = to_f32(from_f32<T>(w[j][i]));
    }
    // round 6: the weights as pairs (the products run as v_pk_fma_f32: half the issue slots of the eight v_fma_mix per d_out channel) with the
    // dropout scale folded in (x 2 is exact: sum_j g_j (2 w_j) has the bits of 2 sum_j g_j w_j), so a kept element needs no multiply
    {
        const float dscale = (mask || use_hash) ? 2.f : 1.f;
#pragma unroll
        for (int j = 0; j < KJ; ++j)
#pragma unroll
            for (int i = 0; i < V / 2; ++i) w2[j][i] = (f32x2){w[j][2 * i] * dscale, w[j][2 * i + 1] * dscale};
    }
    // addresses = uniform row base (scalar registers) + a per-lane 32-bit element offset fixed for the whole walk (geo_pix per pixel is a
    // 64-bit vector multiply: quarter-rate instructions)
    auto row_of = [](const FrameGeo& g, int img, int y) { return ((size_t)(img * g.hp + y + g.pad) * g.wp) * (size_t)g.ld; };
    const int pxl = active ? px : hf.px_lo + hf.px_n - 1;               // idle lanes load the last pixel column (no predicated loads, below)
    const unsigned lo_dout = (unsigned)((pxl + dout.pad) * dout.ld + hd * slot), lo_hid = (unsigned)((pxl + hid.pad) * hid.ld + hd * 512 + c0);
    const unsigned lo_dhid = (unsigned)((px + dhid.pad) * dhid.ld + hd * 512 + c0);
    const unsigned lo_dg = (unsigned)((tix + dg.pad) * dg.ld + hd * 512 + cbase + tq * 4);
    const unsigned lo_mask = (unsigned)(pxl * mask_ld + hd * 512 + c0);
    const int R = ((hid.n - grp + G - 1) / G) * H;                      // rows this workgroup walks
    int fn = grp, foy = 0, fr = 0;                                      // the next row to fetch (the pipeline runs across images; it stops at the last row)
    u32x4 graw[D], hraw[D];
    auto fetch = [&](int d) {
        graw[d] = *(const u32x4*)((const T*)dout.base + row_of(dout, fn, foy) + lo_dout);
        hraw[d] = *(const u32x4*)((const T*)hid.base + row_of(hid, fn, foy) + lo_hid);
        if (fr + 1 < R) { ++fr; if (++foy == H) { foy = 0; fn += G; } }
    };
    int n = grp, wr = 0, pend = 0, pend_iy = 0;                         // image; ring write index, rows waiting for their x reduction (consecutive)
    auto flush = [&]() {                                                // called by every thread of the workgroup
        __syncthreads();
        if (xtask) {
            for (int p = 0; p < pend; ++p) {
                const float* buf = s_v + ((wr - pend + p) % (2 * NB)) * (PW * CS);
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < NT; ++t) {                          // the taps' loads are independent: one LDS latency per row
                    const int o = tlo + t < PW ? tlo + t : PW - 1;
                    const f32x4 a = *(const f32x4*)&buf[o * CS + tq * 4];
                    if (t < tcnt) { c.x = fmaf(tcf[t], a.x, c.x); c.y = fmaf(tcf[t], a.y, c.y); c.z = fmaf(tcf[t], a.z, c.z); c.w = fmaf(tcf[t], a.w, c.w); }
                }
                u32x2 raw;
                T* e = (T*)&raw;
                e[0] = from_f32<T>(c.x); e[1] = from_f32<T>(c.y); e[2] = from_f32<T>(c.z); e[3] = from_f32<T>(c.w);
                *(u32x2*)((T*)dg.base + row_of(dg, n, pend_iy + p) + lo_dg) = raw;
            }
        }
        pend = 0;
    };
    auto finalize = [&](int iy, const f32x2 (&v)[V / 2]) {      // source row iy of image n is complete
        float* buf = s_v + (wr % (2 * NB)) * (PW * CS);
        if (active) {
            *(f32x4*)&buf[pl * CS + ch * V] = (f32x4){v[0].x, v[0].y, v[1].x, v[1].y};
            *(f32x4*)&buf[pl * CS + ch * V + 4] = (f32x4){v[2].x, v[2].y, v[3].x, v[3].y};
        }
        if (pend == 0) pend_iy = iy;
        ++wr; ++pend;
        if (pend == NB) flush();
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d);
    // The row body has no branches: idle and halo lanes compute on the clamped loads and only the d_hid store is predicated (their sums are
    // dropped after the walk) -- a per-lane branch around the body costs a copy of every live accumulator at its end -- and all eight d_out
    // channels are multiplied out although the heads have k = 1, 4, 4, 8 (the rest is zero padding): skipping them behind scalar
    // branches measured 5-10 % SLOWER (620-660 us against 593), specialised copies of the walk per k 50 % slower (four loop bodies in
    // the instruction cache of a CU pair).  ~250 VALU instructions per row and wave, 2/3 of the issue slots (rocprofv3 SQ_INSTS_VALU).
    int cur = 0, oy = 0;
    for (int r0 = 0; r0 < R; r0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int r = r0 + d;
            // The loads are unconditional (rows past the end re-read the last row, idle lanes the last pixel column): a load under a
            // condition makes its destination a phi, the compiler copies it at the end of the block and WAITS for the load there -- the
            // row just requested instead of the one needed D rows later
            float gj[KJ], h[V];
            {
                const T* ge = (const T*)&graw[d];
                const T* he = (const T*)&hraw[d];
#pragma unroll
                for (int j = 0; j < KJ; ++j) gj[j] = to_f32(ge[j]);
#pragma unroll
                for (int i = 0; i < V; ++i) h[i] = to_f32(he[i]);
            }
            fetch(d);
            if (r < R) {                                                // (uniform)
                int y0, y1; float ly0, ly1;
                bilin_coef(oy, sy, dg.h, y0, y1, ly0, ly1);
                while (y0 > cur) {                                      // (uniform) the walk left source row `cur`
                    finalize(cur, va);
#pragma unroll
                    for (int i = 0; i < V / 2; ++i) { va[i] = vb[i]; vb[i] = (f32x2){0.f, 0.f}; }
                    ++cur;
                }
                const float wa = (y0 == cur ? ly0 : 0.f) + (y1 == cur ? ly1 : 0.f);
                const float wb = (y0 == cur + 1 ? ly0 : 0.f) + (y1 == cur + 1 ? ly1 : 0.f);
                {
                    f32x2 o2[V / 2];                                  // same arithmetic and order as head2_dgrad_kernel (its terms past k are + 0; x 2 folded into w2)
#pragma unroll
                    for (int i = 0; i < V / 2; ++i) o2[i] = (f32x2){0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < KJ; ++j) {
                        bs[j] += gj[j];
                        const f32x2 g2 = {gj[j], gj[j]};
#pragma unroll
                        for (int i = 0; i < V / 2; ++i) acc[j][i] = g2 * (f32x2){h[2 * i], h[2 * i + 1]} + acc[j][i];
#pragma unroll
                        for (int i = 0; i < V / 2; ++i) o2[i] = g2 * w2[j][i] + o2[i];
                    }
                    float o[V];
#pragma unroll
                    for (int i = 0; i < V / 2; ++i) { o[2 * i] = o2[i].x; o[2 * i + 1] = o2[i].y; }
                    const long long pc0 = ((long long)n * H + oy) * W;          // (uniform) index of the row's first pixel
                    if (mask) {
                        unsigned char mb[V];
                        *(u32x2*)mb = *(const u32x2*)(mask + (size_t)pc0 * mask_ld + lo_mask);
#pragma unroll
                        for (int i = 0; i < V; ++i) o[i] = mb[i] ? o[i] : 0.f;
                    } else if (use_hash) {
                        // one hash covers the chunk's eight channels (they lie in one 32-channel block); a dropped element is cleared by ANDing
                        // with its keep bit, sign-extended
                        const unsigned kb8 = dbx_drop_hash32(drop_seed, (unsigned)pc0 + (unsigned)px, (unsigned)(hd * 512 + c0) >> 5) >> ((unsigned)c0 & 31u);
#pragma unroll
                        for (int i = 0; i < V; ++i)
                            o[i] = __builtin_bit_cast(float, __builtin_bit_cast(int, o[i]) & ((int)(kb8 << (31 - i)) >> 31));
                    }
                    u32x4 raw;
                    T* oe = (T*)&raw;
#pragma unroll
                    for (int i = 0; i < V; ++i) oe[i] = from_f32<T>(o[i]);
                    if (owner && !nostore) *(u32x4*)((T*)dhid.base + row_of(dhid, n, oy) + lo_dhid) = raw;
                    const f32x2 wa2 = {wa, wa}, wb2 = {wb, wb};
#pragma unroll
                    for (int i = 0; i < V / 2; ++i) {
                        const f32x2 dv = {to_f32(oe[2 * i]), to_f32(oe[2 * i + 1])};
                        va[i] = wa2 * dv + va[i]; vb[i] = wb2 * dv + vb[i];
                    }
                }
                if (++oy == H) {                                        // (uniform) the image is done: its last source rows, next image
                    finalize(cur, va);
                    if (cur + 1 < dg.h) finalize(cur + 1, vb);
                    if (pend) flush();
#pragma unroll
                    for (int i = 0; i < V / 2; ++i) va[i] = vb[i] = (f32x2){0.f, 0.f};
                    cur = 0; oy = 0; n += G;
                }
            }
        }
    }
    if (!owner) {                                                       // idle lanes and the other half's columns
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            bs[j] = 0.f;
#pragma unroll
            for (int i = 0; i < V / 2; ++i) acc[j][i] = (f32x2){0.f, 0.f};
        }
    }
    // weight-gradient partial of this workgroup: the eight pixel columns of a wave by shuffles, the four waves through LDS, fixed order
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float a = acc[j][i >> 1][i & 1];
            a += __shfl_xor(a, 8); a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
            acc[j][i >> 1][i & 1] = a;
        }
        float bsum = bs[j];
        bsum += __shfl_xor(bsum, 8); bsum += __shfl_xor(bsum, 16); bsum += __shfl_xor(bsum, 32);
        bs[j] = bsum;
    }
    __syncthreads();
    float* red = s_v;                                                   // [wave][j][64 channels] + [wave][8] bias sums
    if (lane < LPP) {
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            *(f32x4*)&red[(wv * 8 + j) * CS + lane * V] = (f32x4){acc[j][0][0], acc[j][0][1], acc[j][1][0], acc[j][1][1]};
            *(f32x4*)&red[(wv * 8 + j) * CS + lane * V + 4] = (f32x4){acc[j][2][0], acc[j][2][1], acc[j][3][0], acc[j][3][1]};
        }
#pragma unroll
        for (int j = KJ; j < 8; ++j) {                                  // (rows past the launch's k: zeros, the reduction reads all eight)
            *(f32x4*)&red[(wv * 8 + j) * CS + lane * V] = (f32x4){0.f, 0.f, 0.f, 0.f};
            *(f32x4*)&red[(wv * 8 + j) * CS + lane * V + 4] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[4 * 8 * CS + wv * 8 + j] = j < KJ ? bs[j < KJ ? j : 0] : 0.f;
        }
    }
    __syncthreads();
    {
        const size_t blk = (size_t)grp * plan.halves + half;
        const int c = threadIdx.x % CS;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = threadIdx.x / CS + 4 * jj;
            float o = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) o += red[(q * 8 + j) * CS + c];
            partial[(blk * nhl * 8 + (hd - hd0) * 8 + j) * 512 + cbase + c] = o;
        }
        if (cbase == 0 && threadIdx.x < 8) {
            float bsum = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) bsum += red[4 * 8 * CS + q * 8 + threadIdx.x];
            bpartial[blk * 32 + (hd - hd0) * 8 + threadIdx.x] = bsum;
        }
    }
}
// The split of a row between two workgroups (host side; the same float arithmetic as bilin_coef: one product, truncation, clamp).
static bool h2u_plan(int W, int ws, float sx, H2UPlan& p) {
    if (W > 64 || ws > 32 || W < 1 || ws < 1) return false;
    int x0[64];
    for (int ox = 0; ox < W; ++ox) { int v = (int)(sx * (float)ox); x0[ox] = v > ws - 1 ? ws - 1 : v; }
    if (W <= 32 && ws <= 16) { p.halves = 1; p.h[0] = p.h[1] = H2UHalf{0, W, 0, W, 0, ws}; return true; }
    const int S = ws / 2;                                               // source columns [0, S) and [S, ws)
    if (S < 1 || S > 16 || ws - S > 16) return false;
    int hiL = -1, loR = W;
    for (int ox = 0; ox < W; ++ox) { if (x0[ox] <= S - 1) hiL = ox; if (x0[ox] >= S - 1 && ox < loR) loR = ox; }
    if (hiL + 1 > 32 || W - loR > 32 || loR > hiL + 1) return false;
    const int P = (loR + hiL + 1) / 2;                                  // ownership boundary inside the overlap
    p.halves = 2;
    p.h[0] = H2UHalf{0, hiL + 1, 0, P, 0, S};
    p.h[1] = H2UHalf{loR, W - loR, P, W, S, ws - S};
    return true;
}
template <typename T>
static int head2_backward_up_t(const dbx_view* dout, const dbx_view* hid, const float* const* w2, const int32_t* k, int nh, const dbx_view* dhid,
                               const uint8_t* mask, int mask_ld, int use_hash, unsigned drop_seed, float* const* dw, float* const* db,
                               void* scratch, const dbx_view* dg, hipStream_t s) {
    VIEW_VEC_CHECK(T, dg, "head2_backward_up d_g44");
    DBX_REQUIRE(dg->n == hid->n && dg->c == hid->c, "head2_backward_up: d_g44 has the hidden map's batch and channels");
    const float sy = ac_scale(dg->h, hid->h), sx = ac_scale(dg->w, hid->w);
    H2UPlan plan;
    bool fused = sizeof(T) == 2 && sy > 0.f && sy < 1.f && sx > 0.45f && sx < 1.f && hid->c % 64 == 0 && hid->h >= 2 && h2u_plan(hid->w, dg->w, sx, plan);
    if (const char* e = getenv("DBX_HEAD2_UP")) fused = fused && atoi(e) != 0;
    const bool nostore = dhid->ptr == nullptr;                          // the hidden gradient is not wanted in memory (its consumers generate it)
    DBX_REQUIRE(!nostore || (fused && !mask), "head2_backward_up without d_hid: needs the one-pass kernel (dbx_head2_backward_up_fused) and hash / no dropout");
    if (!fused) {                                                       // the two passes
        const int rc = head2_wgrad_t<T>(dout, hid, k, nh, dw, db, scratch, s, w2, dhid, mask, mask_ld, use_hash, drop_seed);
        return rc != DBX_OK ? rc : upsample_bwd_t<T>(dhid, dg, nullptr, s);
    }
    if constexpr (sizeof(T) == 2) {
        VIEW_VEC_CHECK(T, hid, "head2_backward_up hid");
        if (!nostore) VIEW_VEC_CHECK(T, dhid, "head2_backward_up d_hid");
        DBX_REQUIRE(nh >= 1 && nh <= 4 && hid->c == 512 * nh && dout->c % nh == 0 && dout->c / nh >= 8, "head2_backward_up: nh in 1..4, hid of 512*nh channels, d_out of nh slots >= 8 channels");
        DBX_REQUIRE(dout->n == hid->n && dout->h == hid->h && dout->w == hid->w && dhid->n == hid->n && dhid->h == hid->h && dhid->w == hid->w && dhid->c == hid->c,
                    "head2_backward_up: shape mismatch");
        DBX_REQUIRE(((size_t)dout->ptr % 16) == 0 && (dout->ld * sizeof(T)) % 16 == 0 && (dout->c_off * sizeof(T)) % 16 == 0 &&
                        ((dout->c / nh) * sizeof(T)) % 16 == 0, "head2_backward_up: d_out alignment");
        Head2Out o;
        Head2Args ha;
        ha.nh = nh; ha.slot = dout->c / nh;
        for (int i = 0; i < 4; ++i) {
            o.dw[i] = i < nh ? dw[i] : nullptr; o.db[i] = (i < nh && db) ? db[i] : nullptr; o.k[i] = i < nh ? k[i] : 0;
            ha.w2[i] = i < nh ? w2[i] : nullptr; ha.k[i] = i < nh ? k[i] : 0;
            if (i < nh) DBX_REQUIRE(k[i] >= 1 && k[i] <= 8 && dw[i] && w2[i] && ((size_t)w2[i] % 16) == 0, "head2_backward_up: k in 1..8, 16-byte aligned weights");
        }
        // One launch per run of consecutive heads with the same k rounded up to 1 / 2 / 4 / 8 (round 4: the kernel is VALU-bound --
        // ~250 instructions per row and wave with all eight d_out channels multiplied out -- and the heads have k = 1, 4, 4, 8: the
        // k = 1 launch runs 36 % fewer, the k = 4 launch 21 % fewer; DBX_HEAD2_UP_SPLIT=0: one launch with eight channels).
        // Two workgroups per CU (248 VGPRs x 256 threads): `groups` workgroups per (channel slice, half), each walking every groups-th image.
        constexpr size_t smem = (size_t)2 * H2U_NB * 32 * 64 * 4;       // the ring (>= the 4 x 8 x 64 + 32 floats of the final reduction)
        static_assert(smem >= (4 * 8 * 64 + 32) * 4 && 2 * smem + 4096 <= 160 * 1024, "LDS budget of two workgroups per CU");
        static DbxDevOnce attr_once; int attr_dev = 0;
        if (attr_once.pending(&attr_dev)) {
            DBX_HIP(hipFuncSetAttribute((const void*)head2_backward_up_kernel<T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            DBX_HIP(hipFuncSetAttribute((const void*)head2_backward_up_kernel<T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            DBX_HIP(hipFuncSetAttribute((const void*)head2_backward_up_kernel<T, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            DBX_HIP(hipFuncSetAttribute((const void*)head2_backward_up_kernel<T, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            attr_once.mark(attr_dev);
        }
        static int split = -1;
        if (split < 0) { const char* e = getenv("DBX_HEAD2_UP_SPLIT"); split = e ? atoi(e) : 1; }
        auto kup = [&](int kk) { return !split ? 8 : kk <= 1 ? 1 : kk <= 2 ? 2 : kk <= 4 ? 4 : 8; };
        float* const pbase = (float*)scratch;
        long long pused = 0;                                            // floats of the scratch handed out
        int blocks_max = 0;
        for (int h0 = 0; h0 < nh;) {
            int h1 = h0 + 1;
            while (h1 < nh && kup(k[h1]) == kup(k[h0])) ++h1;
            const int nhl = h1 - h0, nsl = nhl * 8, kj = kup(k[h0]);
            int groups = H2U_WGS / (nsl * plan.halves);
            groups = groups < 1 ? 1 : groups > hid->n ? hid->n : groups;
            const int blocks = groups * plan.halves;                    // <= 2 n <= n h: all launches together stay inside dbx_head2_wgrad_scratch_bytes
            float* partial = pbase + pused;
            float* bpartial = partial + (size_t)blocks * nhl * 8 * 512;
            for (int i = h0; i < h1; ++i) { o.poff[i] = pused; o.boff[i] = pused + (long long)blocks * nhl * 8 * 512; o.nblk[i] = blocks; o.nhl[i] = nhl; o.hl[i] = i - h0; }
            pused += (long long)blocks * (nhl * 8 * 512 + 32);
            blocks_max = blocks > blocks_max ? blocks : blocks_max;
#define H2U_LAUNCH(KJ)                                                                                                              \
            hipLaunchKernelGGL((head2_backward_up_kernel<T, KJ>), dim3(blocks * nsl), dim3(256), smem, s, make_geo<T>(dout), make_geo<T>(hid),    \
                               make_geo<T>(nostore ? hid : dhid), make_geo<T>(dg), h0, nhl, dout->c / nh, ha, partial, bpartial, mask, mask_ld,    \
                               use_hash, drop_seed, sy, sx, plan, nostore ? 1 : 0)
            if (kj == 1) H2U_LAUNCH(1); else if (kj == 2) H2U_LAUNCH(2); else if (kj == 4) H2U_LAUNCH(4); else H2U_LAUNCH(8);
#undef H2U_LAUNCH
            DBX_LAUNCH_CHECK();
            h0 = h1;
        }
        for (int i = nh; i < 4; ++i) { o.poff[i] = 0; o.boff[i] = 0; o.nblk[i] = 0; o.nhl[i] = 1; o.hl[i] = 0; }
        float* partial = pbase; float* bpartial = pbase;                // (the reduction takes each head's offsets from o)
        const int blocks = blocks_max;
        const int total = nh * 8 * 512 + nh * 8;
        hipLaunchKernelGGL(head2_wgrad_reduce_kernel, dim3((total + 63) / 64), dim3(1024), 0, s, partial, bpartial, blocks, nh, o);
        DBX_LAUNCH_CHECK();
    }
    return DBX_OK;
}
// 1 when dbx_head2_backward_up runs as ONE pass for these maps (the form that can leave d_hid out: d_hid->ptr == NULL)
template <typename T> static int head2_up_fused_t(const dbx_view* hid, const dbx_view* dg) {
    const float sy = ac_scale(dg->h, hid->h), sx = ac_scale(dg->w, hid->w);
    H2UPlan plan;
    bool fused = sizeof(T) == 2 && sy > 0.f && sy < 1.f && sx > 0.45f && sx < 1.f && hid->c % 64 == 0 && hid->h >= 2 && h2u_plan(hid->w, dg->w, sx, plan);
    if (const char* e = getenv("DBX_HEAD2_UP")) fused = fused && atoi(e) != 0;
    return fused ? 1 : 0;
}
extern "C" int dbx_head2_backward_up_fused(int32_t dtype, const dbx_view* hid, const dbx_view* d_g44) {
    if (!hid || !d_g44) return 0;
    switch (dtype) { case DBX_F16: return head2_up_fused_t<_Float16>(hid, d_g44); case DBX_BF16: return head2_up_fused_t<__bf16>(hid, d_g44); default: return 0; }
}
extern "C" int dbx_head2_backward_up(int32_t dtype, const dbx_view* d_out, const dbx_view* hid, const float* const* w2, const int32_t* k,
                                     int32_t nh, const dbx_view* d_hid, const uint8_t* dropmask, int32_t dropmask_ld, int32_t use_hash,
                                     uint32_t drop_seed, float* const* dw, float* const* db, void* scratch, const dbx_view* d_g44, void* stream) {
    if (!d_out || !hid || !w2 || !k || !d_hid || !dw || !scratch || !d_g44) { dbx_set_error("head2_backward_up: null argument"); return DBX_ERR_ARG; }
    DBX_DISPATCH_DTYPE(dtype, head2_backward_up_t, d_out, hid, w2, k, nh, d_hid, dropmask, dropmask_ld, use_hash, (unsigned)drop_seed, dw, db,
                       scratch, d_g44, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- multi-tensor weight packing
// One launch re-packs every parameter after an optimizer step (fp32 OIHW -> compute-dtype GEMM layouts, both the forward
// and the transposed/flipped dgrad copy) and refreshes the padded fp32 bias vectors.  Same element mapping as
// dbx_pack_weight (conv_igemm.hip); the table lives in device memory and is built once by the host.
struct PackJob { const float* src; void* dst; int co, ci, taps, mode; long long ktot; int cin_pad, row_off, k_off, rows_lim; };   // rows_lim: rows of dst (0: not checked)
template <typename T>
__global__ void pack_multi_kernel(const PackJob* __restrict__ jobs) {
    const PackJob j = jobs[blockIdx.y];
    const int total = j.co * j.ci * j.taps, taps = j.taps, ci = j.ci;           // (a layer has < 2^31 weights: 32-bit index math)
    T* wp = (T*)j.dst;
    const bool fwd = j.mode == 0 || j.mode == 4;
    const int rl = j.mode >= 4 ? (int)j.ktot : j.rows_lim;
    if (sizeof(T) == 2 && j.mode != 2 && (fwd ? ci : j.co) % 8 == 0 && (j.k_off & 7) == 0 && (j.cin_pad & 7) == 0 && taps <= 25) {
        // 16-bit weights whose packed K index runs over a multiple of 8 source channels.  The source is [o][c][tap] (tap fastest),
        // the packed images want eight consecutive K elements (input channels of one (row, tap) in the forward layouts, output
        // channels in the transposed ones) as one 16-byte chunk: a workgroup stages a tile of 8 (o) x 32 (c) x taps fp32 weights in LDS
        // with coalesced loads (each o row of the tile is 32 * taps contiguous floats), then every thread assembles whole chunks from
        // LDS and stores them coalesced -- no scattered 2-byte stores (the element-wise loop below: 130 us per step for all
        // parameters) and no strided 4-byte gathers from global memory.
        __shared__ float tile[8][32 * 25];
        const int ot = (j.co + 7) / 8, ctn = (ci + 31) / 32, ntiles = ot * ctn, row_f = 32 * taps;
        for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
            const int o0 = (tl / ctn) * 8, c0 = (tl % ctn) * 32;
            const int cw = min(32, ci - c0), ow = min(8, j.co - o0);                 // valid extent of this tile
            __syncthreads();
            for (int e = threadIdx.x; e < 8 * row_f; e += blockDim.x) {
                const int r = e / row_f, f = e - r * row_f;
                tile[r][f] = (r < ow && f < cw * taps) ? j.src[((o0 + r) * ci + c0) * taps + f] : 0.f;
            }
            __syncthreads();
            // forward layouts: chunk = (o, tap, 8 c);  transposed: chunk = (c, tap, 8 o)
            const int nch = fwd ? 8 * taps * 4 : 32 * taps;
            for (int q = threadIdx.x; q < nch; q += blockDim.x) {
                int row, col, tap;
                float v[8];
                if (fwd) {
                    const int k8 = q & 3, t = (q >> 2) % taps, r = (q >> 2) / taps;
                    if (r >= ow || 8 * k8 >= cw) continue;
                    row = j.row_off + o0 + r; col = j.k_off + c0 + 8 * k8; tap = t;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = tile[r][(8 * k8 + e) * taps + t];
                } else {
                    const int c = q % 32, t = q / 32;
                    if (c >= cw) continue;
                    row = j.row_off + c0 + c; col = j.k_off + o0; tap = taps - 1 - t;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = tile[e][c * taps + t];
                }
                if (row < 0 || col < 0 || col + 8 > j.cin_pad || (rl > 0 && row >= rl)) continue;
                u32x4 raw;
                T* e8 = (T*)&raw;
#pragma unroll
                for (int e = 0; e < 8; ++e) e8[e] = from_f32<T>(v[e]);
                const long long di = j.mode >= 4 ? (long long)dbx_frag_index(row, tap, col, j.cin_pad, (int)j.ktot, taps)
                                                 : (long long)row * j.ktot + (long long)tap * j.cin_pad + col;
                *(u32x4*)(wp + di) = raw;
            }
        }
        return;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const float v = j.src[i];
        if (j.mode == 2) { ((float*)j.dst)[j.row_off + i] = v; continue; }        // bias: plain copy into the padded vector
        const int q = i / taps, t = i - q * taps;
        const int o = q / ci, c = q - o * ci;
        {   // an element whose destination lies outside the packed matrix is skipped: negative offsets cut a channel range out of a wider tensor
            const int row = j.row_off + (fwd ? o : c), col = j.k_off + (fwd ? c : o);
            if (row < 0 || col < 0 || col >= j.cin_pad || (rl > 0 && row >= rl)) continue;
        }
        if (j.mode == 0) wp[(long long)(j.row_off + o) * j.ktot + (long long)t * j.cin_pad + j.k_off + c] = from_f32<T>(v);
        else if (j.mode == 1) wp[(long long)(j.row_off + c) * j.ktot + (long long)(taps - 1 - t) * j.cin_pad + j.k_off + o] = from_f32<T>(v);
        else if (j.mode == 4) wp[dbx_frag_index(j.row_off + o, t, j.k_off + c, j.cin_pad, (int)j.ktot, taps)] = from_f32<T>(v);
        else wp[dbx_frag_index(j.row_off + c, taps - 1 - t, j.k_off + o, j.cin_pad, (int)j.ktot, taps)] = from_f32<T>(v);
    }
}
template <typename T> static int pack_multi_t(const void* jobs, int count, long long max_elems, hipStream_t s) {
    int bx = (int)((max_elems + 255) / 256);
    bx = bx < 1 ? 1 : (bx > 256 ? 256 : bx);
    hipLaunchKernelGGL(pack_multi_kernel<T>, dim3(bx, count), dim3(256), 0, s, (const PackJob*)jobs);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_pack_multi(int32_t dtype, const void* jobs, int32_t count, int64_t max_elems, void* stream) {
    DBX_REQUIRE(jobs && count > 0, "pack_multi: empty job table");
    DBX_DISPATCH_DTYPE(dtype, pack_multi_t, jobs, count, (long long)max_elems, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- SGD update + re-packing in one pass
// After an optimizer step every parameter is read again to refresh its packed copies (pack_multi_kernel above: 53 us per step at batch 64
// next to sgd_kernel's 43).  Here ONE job per parameter applies the update (dbx_sgd_update: the bits of sgd_kernel) while the 8 x 32 x
// taps tile is staged in LDS and emits the chunks of ALL its packed images (forward, transposed, fragment-order, channel slices: up to
// four destinations) from that tile: the parameter is read once, the gradient and the momentum buffer once, and no second launch follows.
// pidx < 0: no gradient this step (the job only re-packs).  ndst == 0: a parameter nobody packs (plain update).
struct PackDst { void* dst; long long ktot; int mode, cin_pad, row_off, k_off, rows_lim, pad_; };     // as PackJob's destination fields
struct SgdPackJob { float* p; int pidx, co, ci, taps, ndst, tiled; PackDst d[4]; };
template <typename T>
__device__ __forceinline__ void pack_scatter_elem(const PackDst& d, int o, int c, int t, int i, int taps, float v) {
    if (d.mode == 2) { ((float*)d.dst)[d.row_off + i] = v; return; }
    const bool fwd = d.mode == 0 || d.mode == 4;
    const int rl = d.mode >= 4 ? (int)d.ktot : d.rows_lim;
    const int row = d.row_off + (fwd ? o : c), col = d.k_off + (fwd ? c : o), tap = fwd ? t : taps - 1 - t;
    if (row < 0 || col < 0 || col >= d.cin_pad || (rl > 0 && row >= rl)) return;
    T* wp = (T*)d.dst;
    if (d.mode >= 4) wp[dbx_frag_index(row, tap, col, d.cin_pad, (int)d.ktot, taps)] = from_f32<T>(v);
    else wp[(long long)row * d.ktot + (long long)tap * d.cin_pad + col] = from_f32<T>(v);
}
template <typename T>
__global__ __launch_bounds__(256) void sgd_pack_kernel(const SgdPackJob* __restrict__ jobs, float* const* __restrict__ ptrs, float lr, float mu,
                                                       float wd, int first, int* __restrict__ guard, int step_id) {
    if (guard && guard[0] == step_id) {                                 // (dbx_grad_guard found a non-finite gradient: nothing changes, the packed images stay valid)
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(guard + 1, 1);
        return;
    }
    const SgdPackJob& j = jobs[blockIdx.y];
    float* p = j.p;
    const int pidx = j.pidx, taps = j.taps, ci = j.ci, co = j.co, ndst = j.ndst;
    const float* g = pidx >= 0 ? ptrs[3 * pidx + 1] : nullptr;
    float* b = pidx >= 0 ? ptrs[3 * pidx + 2] : nullptr;
    const int total = co * ci * taps;
    if (j.tiled && sizeof(T) == 2) {
        // tile = 8 (o) x CW (c) x taps fp32 weights in LDS; CW = 32 for filters, 256 for 1x1 layers (their 32-channel tiles would hold 256
        // numbers).  Staging keeps a batch of loads in flight per thread (the update's stores to p / buf would otherwise fence every
        // following load: the pointers may alias as far as the compiler knows) and walks (row, offset) without a division.
        __shared__ float tile[8][32 * 25];
        const int CW = taps == 1 ? 256 : 32, nk8 = CW / 8;
        const int ot = (co + 7) / 8, ctn = (ci + CW - 1) / CW, ntiles = ot * ctn, row_f = CW * taps;
        constexpr int U = 3;
        for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
            const int o0 = (tl / ctn) * 8, c0 = (tl % ctn) * CW;
            const int cw = min(CW, ci - c0), ow = min(8, co - o0);
            __syncthreads();
            int r = 0, f = threadIdx.x;
            while (f >= row_f) { f -= row_f; ++r; }
            for (int e = threadIdx.x; e < 8 * row_f; e += U * 256) {
                float pv[U], gv[U], bv[U];
                int ii[U], rr[U], ff[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    rr[u] = r; ff[u] = f;
                    const bool ok = r < ow && f < cw * taps;               // (r >= 8: past the tile -- ow <= 8)
                    ii[u] = ok ? ((o0 + r) * ci + c0) * taps + f : -1;
                    pv[u] = ok ? p[ii[u]] : 0.f;
                    if (g && ok) { gv[u] = g[ii[u]]; bv[u] = first ? 0.f : b[ii[u]]; }
                    f += 256;
                    while (f >= row_f) { f -= row_f; ++r; }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (g && ii[u] >= 0) {
                        float bb = bv[u];
                        pv[u] = dbx_sgd_update(pv[u], gv[u], &bb, lr, mu, wd, first);
                        b[ii[u]] = bb; p[ii[u]] = pv[u];
                    }
                    if (rr[u] < 8) tile[rr[u]][ff[u]] = pv[u];
                }
            }
            __syncthreads();
            for (int di = 0; di < ndst; ++di) {
                const PackDst& d = j.d[di];
                T* wp = (T*)d.dst;
                const bool fwd = d.mode == 0 || d.mode == 4;
                const int rl = d.mode >= 4 ? (int)d.ktot : d.rows_lim;
                const int nch = fwd ? 8 * taps * nk8 : CW * taps;        // forward layouts: chunk = (o, tap, 8 c);  transposed: (c, tap, 8 o)
                for (int q = threadIdx.x; q < nch; q += blockDim.x) {
                    int row, col, tap;
                    float v[8];
                    if (fwd) {
                        const int k8 = q % nk8, q2 = q / nk8, t = q2 % taps, r2 = q2 / taps;
                        if (r2 >= ow || 8 * k8 >= cw) continue;
                        row = d.row_off + o0 + r2; col = d.k_off + c0 + 8 * k8; tap = t;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = tile[r2][(8 * k8 + e) * taps + t];
                    } else {
                        const int c = q % CW, t = q / CW;
                        if (c >= cw) continue;
                        row = d.row_off + c0 + c; col = d.k_off + o0; tap = taps - 1 - t;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = tile[e][c * taps + t];
                    }
                    if (row < 0 || col < 0 || col + 8 > d.cin_pad || (rl > 0 && row >= rl)) continue;
                    u32x4 raw;
                    T* e8 = (T*)&raw;
#pragma unroll
                    for (int e = 0; e < 8; ++e) e8[e] = from_f32<T>(v[e]);
                    const long long off = d.mode >= 4 ? (long long)dbx_frag_index(row, tap, col, d.cin_pad, (int)d.ktot, taps)
                                                      : (long long)row * d.ktot + (long long)tap * d.cin_pad + col;
                    *(u32x4*)(wp + off) = raw;
                }
            }
        }
        return;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        float v = p[i];
        if (g) { v = dbx_sgd_update(v, g[i], b + i, lr, mu, wd, first); p[i] = v; }
        if (ndst == 0) continue;
        const int q = i / taps, t = i - q * taps;
        const int o = q / ci, c = q - o * ci;
        for (int di = 0; di < ndst; ++di) pack_scatter_elem<T>(j.d[di], o, c, t, i, taps, v);
    }
}
template <typename T> static int sgd_pack_t(const void* jobs, int count, long long max_elems, float* const* ptrs, float lr, float mu, float wd,
                                            int first, int* guard, int step_id, hipStream_t s) {
    int bx = (int)((max_elems + 255) / 256);
    bx = bx < 1 ? 1 : (bx > 256 ? 256 : bx);
    hipLaunchKernelGGL(sgd_pack_kernel<T>, dim3(bx, count), dim3(256), 0, s, (const SgdPackJob*)jobs, ptrs, lr, mu, wd, first, guard, step_id);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_sgd_pack_step_guarded(int32_t dtype, const void* jobs, int32_t count, int64_t max_elems, float* const* ptrs, float lr,
                                         float momentum, float weight_decay, int32_t first_step, int32_t* guard, int32_t step_id, void* stream) {
    DBX_REQUIRE(jobs && count > 0, "sgd_pack: empty job table");
    static_assert(sizeof(PackDst) == 40 && sizeof(SgdPackJob) == 192, "job record layout (include/densebox_hip.h)");
    DBX_DISPATCH_DTYPE(dtype, sgd_pack_t, jobs, count, (long long)max_elems, ptrs, lr, momentum, weight_decay, first_step, guard, step_id,
                       (hipStream_t)stream);
}
extern "C" int dbx_sgd_pack_step(int32_t dtype, const void* jobs, int32_t count, int64_t max_elems, float* const* ptrs, float lr, float momentum,
                                 float weight_decay, int32_t first_step, void* stream) {
    return dbx_sgd_pack_step_guarded(dtype, jobs, count, max_elems, ptrs, lr, momentum, weight_decay, first_step, nullptr, 0, stream);
}

// ---------------------------------------------------------------------------------------------- eval-mode head folding
// The heads are Conv1x1(768->512) -> Dropout -> Conv1x1(512->k) with NO non-linearity (DenseBox.py:158-162); in eval
// mode Dropout is the identity, so the pair is one linear map:  W = W2 W1  [k x 768],  b = W2 b1 + b2.
// fp32 in, fp32 out; run once per weight version, not per image.
__global__ void fold_heads_kernel(const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w1,
                                  const float* __restrict__ b1, int k, float* __restrict__ w, float* __restrict__ b) {
    const int j = blockIdx.x;                       // output channel of this head
    for (int c = threadIdx.x; c < 768; c += blockDim.x) {
        float acc = 0.f;
        for (int h = 0; h < 512; ++h) acc = fmaf(w2[j * 512 + h], w1[h * 768 + c], acc);
        w[j * 768 + c] = acc;
    }
    if (threadIdx.x == 0) {
        float acc = b2[j];
        for (int h = 0; h < 512; ++h) acc = fmaf(w2[j * 512 + h], b1[h], acc);
        b[j] = acc;
    }
}
extern "C" int dbx_fold_heads(const float* w2, const float* b2, const float* w1, const float* b1, int32_t k, float* w_out,
                              float* b_out, void* stream) {
    DBX_REQUIRE(w2 && b2 && w1 && b1 && w_out && b_out && k >= 1, "fold_heads: bad arguments");
    hipLaunchKernelGGL(fold_heads_kernel, dim3(k), dim3(256), 0, (hipStream_t)stream, w2, b2, w1, b1, k, w_out, b_out);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

// Refine branch in eval mode (DenseBox.py:464-471): pool4 -> conv6_1 (3x3) -> conv6_2 (5x5) -> bilinear up -> conv6_3 (1x1) has no
// non-linearity after the pooling, the 1x1 conv commutes with the up-sampling (bilinear weights sum to 1, so constants pass through),
// and two un-padded cross-correlations compose into one: the branch is ONE un-padded 7x7 conv from `ci` channels to 1, then the
// up-sampling of that single map.   W[c][u][v] = sum_m sum_{a+i=u, b+j=v} (sum_n w3[n] w2[n][m][a][b]) w1[m][c][i][j],
// b = b3 + sum_n w3[n] b2[n] + sum_m (sum_{a,b} V[m][a][b]) b1[m].  fp32 in, fp32 out; once per weight version.
// V[m][a][b] = sum_n w3[n] w2[n][m][a][b]: 32 outputs per workgroup, eight lanes per output (eight mid channels each), summed by a fixed
// shuffle tree -- a single workgroup walking the 410 KB of conv6_2 weights took 150 us, and training re-folds after every optimizer step
__global__ __launch_bounds__(256) void fold_refine_v_kernel(const float* __restrict__ w2, const float* __restrict__ w3, int cm, float* __restrict__ V) {
    const int e = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
    float acc = 0.f;
    if (e < cm * 25) {
        const int m = e / 25, ab = e % 25;
        for (int n = part; n < cm; n += 8) acc = fmaf(w3[n], w2[((size_t)n * cm + m) * 25 + ab], acc);
    }
    acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
    if (part == 0 && e < cm * 25) V[e] = acc;
}
// one workgroup per input channel c: W[c][u][v] from V and w1 (both staged in LDS), 49 outputs x 5 chunks of mid channels, fixed order
__global__ __launch_bounds__(256) void fold_refine_w_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ b2,
                                                            const float* __restrict__ w3, const float* __restrict__ b3, const float* __restrict__ Vg,
                                                            int ci, int cm, float* __restrict__ w, float* __restrict__ b) {
    __shared__ float V[64 * 25];
    __shared__ float w1s[64 * 9];
    __shared__ float red[5][49];
    __shared__ float bred[256];
    const int tid = threadIdx.x, c = blockIdx.x;
    for (int e = tid; e < cm * 25; e += 256) V[e] = Vg[e];
    for (int e = tid; e < cm * 9; e += 256) w1s[e] = w1[((size_t)(e / 9) * ci + c) * 9 + e % 9];
    __syncthreads();
    if (tid < 245) {
        const int uv = tid % 49, ch = tid / 49, u = uv / 7, v = uv % 7;
        const int per = (cm + 4) / 5, m0 = ch * per, m1 = m0 + per < cm ? m0 + per : cm;
        float acc = 0.f;
        for (int m = m0; m < m1; ++m)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    const int a2 = u - i, b2i = v - j;
                    if (a2 >= 0 && a2 < 5 && b2i >= 0 && b2i < 5) acc = fmaf(V[m * 25 + a2 * 5 + b2i], w1s[m * 9 + i * 3 + j], acc);
                }
        red[ch][uv] = acc;
    }
    __syncthreads();
    if (tid < 49) w[c * 49 + tid] = (((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid]) + red[4][tid];
    if (c == 0) {
        float part = 0.f;
        for (int m = tid; m < cm; m += 256) {
            float sv = 0.f;
            for (int ab = 0; ab < 25; ++ab) sv += V[m * 25 + ab];
            part += sv * b1[m] + w3[m] * b2[m];
        }
        bred[tid] = part;
        __syncthreads();
        if (tid == 0) {
            float acc = b3[0];
            for (int t = 0; t < 256; ++t) acc += bred[t];
            b[0] = acc;
        }
    }
}
extern "C" int dbx_fold_refine(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                               int32_t ci, int32_t cm, float* w_out, float* b_out, float* v_out, void* stream) {
    DBX_REQUIRE(w1 && b1 && w2 && b2 && w3 && b3 && w_out && b_out && v_out && ci >= 1 && cm >= 1 && cm <= 64, "fold_refine: bad arguments (mid channels <= 64)");
    hipLaunchKernelGGL(fold_refine_v_kernel, dim3((cm * 25 + 31) / 32), dim3(256), 0, (hipStream_t)stream, w2, w3, cm, v_out);
    DBX_LAUNCH_CHECK();
    hipLaunchKernelGGL(fold_refine_w_kernel, dim3(ci), dim3(256), 0, (hipStream_t)stream, w1, b1, b2, w3, b3, v_out, ci, cm, w_out, b_out);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

// The folded branch in one fp32 kernel: cat(landmarks, score) -> MaxPool2d(2, 2) -> the 7x7 conv of dbx_fold_refine, straight from the
// heads' fp32 NCHW outputs (no layout conversion, no 16-bit rounding inside the branch).  A workgroup makes 16 x 16 outputs from a
// 22 x 22 x 5 pooled tile it builds in LDS; 245 FMAs per output in a fixed order.
__global__ __launch_bounds__(256) void refine_eval_kernel(const float* __restrict__ lm, const float* __restrict__ sc, int h, int w,
                                                          const float* __restrict__ wf, const float* __restrict__ bf, float* __restrict__ out) {
    constexpr int T = 16, PT = T + 6, CI = 5;
    __shared__ float tile[CI][PT][PT + 1];
    __shared__ float wsm[CI * 49];
    const int ph = h / 2, pw = w / 2, oh = ph - 6, ow = pw - 6;
    const int n = blockIdx.z, oy0 = blockIdx.y * T, ox0 = blockIdx.x * T;
    const int tid = threadIdx.x;
    for (int e = tid; e < CI * 49; e += 256) wsm[e] = wf[e];
    for (int e = tid; e < CI * PT * PT; e += 256) {
        const int c = e / (PT * PT), r = e % (PT * PT), py = oy0 + r / PT, px = ox0 + r % PT;
        float v = 0.f;
        if (py < ph && px < pw) {
            const float* p = (c < 4 ? lm + ((size_t)n * 4 + c) * h * w : sc + (size_t)n * h * w) + (size_t)(2 * py) * w + 2 * px;
            v = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[w], p[w + 1]));
        }
        tile[c][r / PT][r % PT] = v;
    }
    __syncthreads();
    const int ty = tid / T, tx = tid % T, oy = oy0 + ty, ox = ox0 + tx;
    if (oy < oh && ox < ow) {
        float acc = bf[0];
        for (int c = 0; c < CI; ++c)
#pragma unroll
            for (int u = 0; u < 7; ++u)
#pragma unroll
                for (int v = 0; v < 7; ++v) acc = fmaf(wsm[(c * 7 + u) * 7 + v], tile[c][ty + u][tx + v], acc);
        out[((size_t)n * oh + oy) * ow + ox] = acc;
    }
}
extern "C" int dbx_refine_eval(const float* landmark_nchw, const float* score_nchw, int32_t n, int32_t h, int32_t w, const float* w_fold,
                               const float* b_fold, float* out_small, void* stream) {
    DBX_REQUIRE(landmark_nchw && score_nchw && w_fold && b_fold && out_small && n >= 1 && h / 2 >= 7 && w / 2 >= 7, "refine_eval: bad arguments (H/2, W/2 >= 7)");
    const int oh = h / 2 - 6, ow = w / 2 - 6;
    hipLaunchKernelGGL(refine_eval_kernel, dim3((ow + 15) / 16, (oh + 15) / 16, n), dim3(256), 0, (hipStream_t)stream, landmark_nchw, score_nchw, h, w,
                       w_fold, b_fold, out_small);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

// bilinear up-sampling (align_corners=True, ATen's arithmetic: bilin_coef) of fp32 NCHW planes -- the folded refine branch's single map
__global__ void upsample_nchw_f32_kernel(const float* __restrict__ x, int planes, int hi, int wi, float* __restrict__ y, int ho, int wo,
                                         float sy, float sx) {
    const int64_t total = (int64_t)planes * ho * wo;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(i % wo), py = (int)((i / wo) % ho);
        const int64_t pl = i / ((int64_t)wo * ho);
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        bilin_coef(py, sy, hi, y0, y1, ly0, ly1);
        bilin_coef(px, sx, wi, x0, x1, lx0, lx1);
        const float* p = x + pl * hi * wi;
        {   // (no contraction: see bilin_coef)
#pragma clang fp contract(off)
            const float top = lx0 * p[y0 * wi + x0] + lx1 * p[y0 * wi + x1];
            const float bot = lx0 * p[y1 * wi + x0] + lx1 * p[y1 * wi + x1];
            y[i] = ly0 * top + ly1 * bot;
        }
    }
}
extern "C" int dbx_upsample_bilinear_nchw_f32(const float* x, int32_t planes, int32_t hi, int32_t wi, float* y, int32_t ho, int32_t wo,
                                              void* stream) {
    DBX_REQUIRE(x && y && planes >= 1 && hi >= 1 && wi >= 1 && ho >= 1 && wo >= 1, "upsample_nchw: bad arguments");
    const int64_t total = (int64_t)planes * ho * wo;
    hipLaunchKernelGGL(upsample_nchw_f32_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, planes, hi, wi, y, ho, wo,
                       ac_scale(hi, ho), ac_scale(wi, wo));
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}

// ---------------------------------------------------------------------------------------------- uint8 HWC image -> network input
// torchvision's ToTensor + Normalize of the reference datasets (DenseBox.py:766-772, :3613-3619) fused with the layout
// change: y[n,py,px,c] = ((u8 / 255) - mean[c]) / std[c] in fp32 (true divisions, like ATen), rounded to the compute
// dtype, channels 3.. of the framed view zero-filled.  12 bytes in, 16 bytes out per pixel.
template <typename T>
__global__ void u8hwc_to_framed_kernel(const unsigned char* __restrict__ x, FrameGeo y, float m0, float m1, float m2, float s0,
                                       float s1, float s2) {
    constexpr int V = Vec<T>::N;
    const int cg = y.c / V;
    const int64_t total = (int64_t)y.n * y.h * y.w;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(i % y.w);
        const int py = (int)((i / y.w) % y.h);
        const int n = (int)(i / ((int64_t)y.w * y.h));
        const unsigned char* p = x + i * 3;
        float v[3];
        v[0] = ((float)p[0] / 255.0f - m0) / s0;
        v[1] = ((float)p[1] / 255.0f - m1) / s1;
        v[2] = ((float)p[2] / 255.0f - m2) / s2;
        T* dst = (T*)y.base + geo_pix(y, n, py, px);
        for (int g = 0; g < cg; ++g) {
            float o[V];
#pragma unroll
            for (int j = 0; j < V; ++j) { const int c = g * V + j; o[j] = c < 3 ? v[c] : 0.f; }
            store_vec<T>(dst + g * V, o);
        }
    }
}
template <typename T> static int u8hwc_to_framed_t(const uint8_t* x, const dbx_view* y, const float* mean, const float* stdv, hipStream_t s) {
    VIEW_VEC_CHECK(T, y, "u8hwc_to_framed");
    DBX_REQUIRE(y->c >= 3 && mean && stdv, "u8hwc_to_framed: need >= 3 channels and mean/std");
    const int64_t total = (int64_t)y->n * y->h * y->w;
    hipLaunchKernelGGL(u8hwc_to_framed_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, x, make_geo<T>(y), mean[0], mean[1], mean[2],
                       stdv[0], stdv[1], stdv[2]);
    DBX_LAUNCH_CHECK();
    return DBX_OK;
}
extern "C" int dbx_u8hwc_to_framed(int32_t dtype, const uint8_t* x_nhwc, const dbx_view* y, const float* mean3, const float* std3,
                                   void* stream) {
    DBX_DISPATCH_DTYPE(dtype, u8hwc_to_framed_t, x_nhwc, y, mean3, std3, (hipStream_t)stream);
}
