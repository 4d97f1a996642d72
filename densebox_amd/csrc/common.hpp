// Shared device/host helpers for libdensebox_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "../../include/densebox_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

void dbx_set_error(const char* fmt, ...);

#define DBX_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            dbx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return DBX_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

#define DBX_LAUNCH_CHECK()                                                                    \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) {                                                               \
            dbx_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return DBX_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

#define DBX_REQUIRE(cond, ...)                                                                \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            dbx_set_error(__VA_ARGS__);                                                       \
            return DBX_ERR_ARG;                                                               \
        }                                                                                     \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies per DEVICE: one "done" bit per device ordinal, so a process that drives
// several GPUs sets it on each of them (a single process-wide flag made the first launch on a second device fail); thread-safe.
struct DbxDevOnce {
    std::atomic<uint64_t> done{0};
    bool pending(int* dev_out) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        *dev_out = dev;
        return !((done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull);
    }
    void mark(int dev) { done.fetch_or(1ull << (dev & 63), std::memory_order_release); }
};

namespace pipe { template <int N> struct IC { static constexpr int value = N; }; }   // compile-time indices for generic lambdas

static inline int dbx_esize(int dtype) { return dtype == DBX_F32 ? 4 : 2; }

// ---- device-side dtype helpers ---------------------------------------------------------------
template <typename T> struct DType;
template <> struct DType<_Float16> { static constexpr int id = DBX_F16; };
template <> struct DType<__bf16> { static constexpr int id = DBX_BF16; };
template <> struct DType<float> { static constexpr int id = DBX_F32; };

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// frame geometry of a dbx_view flattened for kernels
struct FrameGeo {
    char* base;        // byte pointer to element (0, -pad, -pad, c_off)
    int n, h, w, pad;  // logical dims
    int hp, wp;        // framed dims
    int ld;            // elements per pixel
    int c;             // channels in view
};

template <typename T>
static inline FrameGeo make_geo(const dbx_view* v) {
    FrameGeo g;
    g.base = (char*)v->ptr + (size_t)v->c_off * sizeof(T);
    g.n = v->n; g.h = v->h; g.w = v->w; g.pad = v->pad;
    g.hp = v->h + 2 * v->pad; g.wp = v->w + 2 * v->pad;
    g.ld = v->ld; g.c = v->c;
    return g;
}

// byte offset of logical pixel (n,y,x) in a frame
__device__ __forceinline__ size_t geo_pix(const FrameGeo& g, int n, int y, int x) {
    return ((size_t)(n * g.hp + y + g.pad) * g.wp + (x + g.pad)) * (size_t)g.ld;
}

// Counter-based dropout keep bits (p = 0.5).  One lowbias32 finaliser over (seed, pixel m, 32-channel block c32) yields the keep
// bits of that block's 32 channels (bit j <-> channel 32 c32 + j): stateless, so forward and backward regenerate identical masks,
// and an epilogue that owns 8+ channels of a pixel pays one hash for all of them.
__device__ __forceinline__ unsigned dbx_drop_hash32(unsigned seed, unsigned m, unsigned c32) {
    unsigned x = seed ^ (m * 0x9E3779B1u) ^ (c32 * 0x85EBCA77u);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
// bit j of the result is the keep bit of channel 4*c4 + j of pixel m
__device__ __forceinline__ unsigned dbx_drop_bits4(unsigned seed, unsigned m, unsigned c4) {
    return (dbx_drop_hash32(seed, m, c4 >> 3) >> ((c4 & 7u) * 4u)) & 15u;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T> struct Mma32;
template <> struct Mma32<_Float16> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma32<__bf16> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

// The heads' hidden gradient as its consumers GENERATE it (round 4: dbx_heads_backward_gen_*): d_hid = keep * scale * (d_out W2) with
// d_out of <= 8 channels per head -- a 32-channel x 32-pixel tile of it is ONE v_mfma_f32_32x32x16 (A = scale * W2^T [32 ch x 8 k, the
// upper K half zero], B = d_out^T [8 k x 32 px]) plus one hash per lane: D[i][j] has lane j = pixel, 16 channels i = 8 (r / 4) +
// 4 (lane / 32) + r % 4 -- all inside the 32-channel block one dbx_drop_hash32 covers.  W2 and d_out are rounded to the compute dtype
// (the forward's second convs use the same rounded W2); the products are exact in fp32, the sum of <= 8 of them is the MFMA's.
struct GenHid {
    const char* dout;            // [N * H * W][ld] compute dtype, pad 0; one slot of >= 8 channels per head, channels >= k are zero
    const float* w2[4];          // fp32 [k][512] per head (conv5_2 weights)
    int k[4];
    int ld, slot, nh;            // elements per pixel / per head slot
    int H, W, pad;               // the hidden map; pad: of the frame the consumer walks (d_out itself has none)
    unsigned seed; int use_hash; // use_hash 0: no dropout (scale 1), 1: keep bits of dbx_drop_hash32 (scale 2)
};
// A operand of the generating MFMA for the row `ch` (channel inside head hd) this lane holds; lanes >= 32 (K 8..15) hold zeros
template <typename T>
__device__ __forceinline__ u32x4 genhid_wfrag(const GenHid& gh, int hd, int ch, int lane) {
    const float* wp = gh.w2[0];
    int k = gh.k[0];
#pragma unroll
    for (int hh = 1; hh < 4; ++hh)
        if (hd == hh && gh.w2[hh]) { wp = gh.w2[hh]; k = gh.k[hh]; }
    const float sc = gh.use_hash ? 2.f : 1.f;
    u32x4 raw;
    T* e = (T*)&raw;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = j < k ? j : (k > 0 ? k - 1 : 0);
        const float v = wp[(size_t)row * 512 + ch];
        e[j] = from_f32<T>((lane < 32 && j < k) ? v * sc : 0.f);
    }
    return raw;
}
// pixel of frame position q (inside one image's frame, row-major over wp columns): compact index m = (img * H + y) * W + x, or -1 in the halo
__device__ __forceinline__ int genhid_pixel(const GenHid& gh, int img, int q_in_img, int wp, float inv_wp) {
    const int fy = (int)(((float)q_in_img + 0.5f) * inv_wp);         // exact: (q + 0.5) / wp is >= 0.5 / wp away from an integer
    const int fx = q_in_img - fy * wp;
    const int y = fy - gh.pad, x = fx - gh.pad;
    return ((unsigned)y < (unsigned)gh.H && (unsigned)x < (unsigned)gh.W) ? (img * gh.H + y) * gh.W + x : -1;
}
// B operand: the pixel's d_out slot (16 bytes) in the lower lane half, zeros in the upper one and for halo pixels.  The load is
// unconditional (halo pixels read pixel 0) and the zeroing happens where the value is consumed: a select behind the load would wait for it
__device__ __forceinline__ u32x4 genhid_load_dout(const GenHid& gh, int hd, int m) {
    const int mm = m < 0 ? 0 : m;
    return *(const u32x4*)(gh.dout + ((size_t)mm * gh.ld + hd * gh.slot) * 2);
}
__device__ __forceinline__ u32x4 genhid_bfrag(const u32x4& raw, int m, int lane) {
    return (m < 0 || lane >= 32) ? (u32x4){0u, 0u, 0u, 0u} : raw;
}

// Fragment-order weight image (conv3x3_ws.hpp; dbx_pack_weight modes 4/5).  One 1-KiB block = the A operand of one v_mfma_f32_32x32x16:
// [lane = 32 * ((k % 16) / 8) + row % 32][k % 8]; blocks ordered [row / BN][period][step][(row % BN) / 32] with
//   3x3: period = 3 * (k / 64) + ky,  step = kx * 4 + (k % 64) / 16   (tap = 3 ky + kx; 12 steps per period; the three ky
//        bands of one 64-channel chunk are consecutive periods: the second and third re-read rows the first left in L2)
//   1x1: period = k / 128,           step = (k % 128) / 16           (8 steps per period)
__host__ __device__ inline size_t dbx_frag_index(int row, int tap, int k, int cin_pad, int rows_pad, int taps) {
    const int bn = rows_pad % 256 == 0 ? 256 : 128;
    int period, step, nstep, nper;
    if (taps == 9) {
        const int KC = cin_pad / 64, ky = tap / 3, kx = tap - 3 * ky;
        period = 3 * (k / 64) + ky; step = kx * 4 + (k % 64) / 16; nstep = 12; nper = 3 * KC;
    } else {
        period = k / 128; step = (k % 128) / 16; nstep = 8; nper = cin_pad / 128;
    }
    const int lane = 32 * ((k % 16) / 8) + row % 32;
    const size_t blk = (((size_t)(row / bn) * nper + period) * nstep + step) * (bn / 32) + (row % bn) / 32;
    return (blk * 64 + lane) * 8 + k % 8;
}

// torch.optim.SGD (dampening 0, no Nesterov; DenseBox.py:2001-2004) on one element: g = grad + wd p; buf = g (first step) | mu buf + g;
// p -= lr buf.  No FMA contraction: sgd_kernel and sgd_pack_kernel must produce the same bits (and the ones torch's CPU loop does).
__device__ __forceinline__ float dbx_sgd_update(float pv, float gr, float* buf, float lr, float mu, float wd, int first) {
#pragma clang fp contract(off)
    const float gv = gr + wd * pv;
    const float bv = first ? gv : mu * *buf + gv;
    *buf = bv;
    return pv - lr * bv;
}

#define DBX_DISPATCH_DTYPE(dtype, FN, ...)                          \
    switch (dtype) {                                                \
        case DBX_F16: return FN<_Float16>(__VA_ARGS__);             \
        case DBX_BF16: return FN<__bf16>(__VA_ARGS__);              \
        case DBX_F32: return FN<float>(__VA_ARGS__);                \
        default: dbx_set_error("bad dtype %d", (int)(dtype)); return DBX_ERR_DTYPE; \
    }
