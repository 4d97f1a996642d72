// Kernel lab for the 3x3 convolution kernels (dev tool, not product): times the dispatcher's variants on the backbone
// shapes with random operands and checks every variant against a plain one-thread-per-output reference kernel.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/band_lab.hip -o tools/band_lab
// run:   tools/band_lab [variants, e.g. 0,5,6] [reps]
#include "../densebox_amd/csrc/conv_igemm.hip"
// conv_igemm.hip calls into conv_wgrad.hip for the fused dgrad + weight-gradient entry point; the lab links one translation unit
int dbx_internal_wgrad_reduce(const float*, const float*, int, int, int, int, int, int, float*, float*, int, hipStream_t) { return DBX_ERR_ARG; }
#include <vector>
#include <string>
#include <math.h>
#include <string.h>

static thread_local char g_err[512] = "";
void dbx_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int n, h, w, ci, co; int epi; };

__global__ void fill_kernel(__bf16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 0x9E3779B1u ^ seed; x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
        p[i] = (__bf16)(((int)(x & 0xffff) - 32768) * (scale / 32768.f));
    }
}
// interior of the framed tensor only (the frame stays zero)
__global__ void fill_framed(__bf16* p, int n, int h, int w, int c, unsigned seed, float scale) {
    const size_t tot = (size_t)n * h * w * c;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = i % c; size_t r = i / c; const int x = r % w; r /= w; const int y = r % h; const int im = r / h;
        unsigned v = (unsigned)i * 0x9E3779B1u ^ seed; v ^= v >> 16; v *= 0x7FEB352Du; v ^= v >> 15; v *= 0x846CA68Bu; v ^= v >> 16;
        p[(((size_t)im * (h + 2) + y + 1) * (w + 2) + x + 1) * c + ch] = (__bf16)(((int)(v & 0xffff) - 32768) * (scale / 32768.f));
    }
}
// reference: one thread per (pixel, cout), fp32 accumulate in (tap, ci) order
__global__ void ref_conv(const __bf16* x, const __bf16* wp, const float* bias, const __bf16* gate, float* out, int n, int h, int w, int ci,
                         int co, int epi, int npix, const int* pix) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npix * co) return;
    const int c = t % co, pi = pix[t / co];
    const int px = pi % w, py = (pi / w) % h, im = pi / (w * h);
    float acc = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        const __bf16* xr = x + (((size_t)im * (h + 2) + py + tap / 3) * (w + 2) + px + tap % 3) * ci;
        const __bf16* wr = wp + (size_t)c * 9 * ci + (size_t)tap * ci;
        for (int k = 0; k < ci; ++k) acc += (float)xr[k] * (float)wr[k];
    }
    if (epi & DBX_EPI_BIAS) acc += bias[c];
    if (epi & DBX_EPI_RELU) acc = fmaxf(acc, 0.f);
    if (epi & DBX_EPI_GATE) { if (!((float)gate[(((size_t)im * (h + 2) + py + 1) * (w + 2) + px + 1) * co + c] > 0.f)) acc = 0.f; }
    out[t] = acc;
}
__global__ void gather_out(const __bf16* y, float* out, int h, int w, int co, int npix, const int* pix) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npix * co) return;
    const int c = t % co, pi = pix[t / co];
    const int px = pi % w, py = (pi / w) % h, im = pi / (w * h);
    out[t] = (float)y[(((size_t)im * (h + 2) + py + 1) * (w + 2) + px + 1) * co + c];
}
__global__ void frame_sum(const __bf16* y, int n, int h, int w, int c, float* out) {
    // |sum| over the frame pixels (must stay zero)
    float s = 0.f;
    const int hp = h + 2, wp = w + 2;
    const size_t tot = (size_t)n * hp * wp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const int fx = i % wp, fy = (i / wp) % hp;
        if (fx == 0 || fx == wp - 1 || fy == 0 || fy == hp - 1)
            for (int k = 0; k < c; ++k) s += fabsf((float)y[i * c + k]);
    }
    atomicAdd(out, s);
}

__global__ void retile_w(const __bf16* wp, __bf16* wt, int co, int ci, int bn) {
    const size_t tot = (size_t)co * 9 * ci;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % ci; const int tap = (i / ci) % 9; const int o = i / ((size_t)9 * ci);
        wt[dbx_frag_index(o, tap, c, ci, bn == 256 ? 256 : 128, 9)] = wp[i];
    }
}

int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    std::vector<int> variants = {0, 5};
    if (argc > 1) { variants.clear(); char* tok = strtok(argv[1], ","); while (tok) { variants.push_back(atoi(tok)); tok = strtok(nullptr, ","); } }
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int only = argc > 3 ? atoi(argv[3]) : -1;
    const int RG = DBX_EPI_BIAS | DBX_EPI_RELU;
    const Shape shapes[] = {
        {"conv4_2 512->512 @30 N64", 64, 30, 30, 512, 512, RG},
        {"conv4_1 256->512 @30 N64", 64, 30, 30, 256, 512, RG},
        {"conv3_2 256->256 @60 N64", 64, 60, 60, 256, 256, RG},
        {"conv3_1 128->256 @60 N64", 64, 60, 60, 128, 256, RG},
        {"dgrad4  512->512 @30 gate", 64, 30, 30, 512, 512, DBX_EPI_GATE},
        {"small   128->256 @17x23 N3", 3, 17, 23, 128, 256, RG},
        {"conv2_2 128->128 @120 N64", 64, 120, 120, 128, 128, RG},
        {"conv2_1 64->128 @120 N64", 64, 120, 120, 64, 128, RG},
        {"dgrad3_1 256->128 @60 gate", 64, 60, 60, 256, 128, DBX_EPI_GATE},
        {"odd 192->256 @33x47 N5", 5, 33, 47, 192, 256, RG},
        {"odd 128->128 @41x29 N7 gate", 7, 41, 29, 128, 128, DBX_EPI_GATE | DBX_EPI_BIAS},
        {"conv1_2 64->64 @240 N64", 64, 240, 240, 64, 64, RG},
        {"dgrad1_2 64->64 @240 N64 gate", 64, 240, 240, 64, 64, DBX_EPI_GATE},
    };
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int si = 0; si < (int)(sizeof shapes / sizeof shapes[0]); ++si) {
        if (only >= 0 && si != only) continue;
        const Shape& S = shapes[si];
        const int hp = S.h + 2, wp = S.w + 2;
        const size_t xpix = (size_t)S.n * hp * wp;
        const size_t guard = (size_t)(8 * wp > 576 + wp ? 8 * wp : 576 + wp);
        __bf16 *xb, *yb, *gb, *wpk; float* bias;
        CK(hipMalloc(&xb, (xpix + 2 * guard) * S.ci * 2)); CK(hipMemset(xb, 0, (xpix + 2 * guard) * S.ci * 2));
        CK(hipMalloc(&yb, (xpix + 2 * guard) * S.co * 2)); CK(hipMemset(yb, 0, (xpix + 2 * guard) * S.co * 2));
        CK(hipMalloc(&gb, (xpix + 2 * guard) * S.co * 2)); CK(hipMemset(gb, 0, (xpix + 2 * guard) * S.co * 2));
        __bf16* wtl;
        CK(hipMalloc(&wpk, (size_t)S.co * 9 * S.ci * 2)); CK(hipMalloc(&bias, S.co * 4));
        CK(hipMalloc(&wtl, (size_t)S.co * 9 * S.ci * 2 + 65536)); CK(hipMemset(wtl, 0, (size_t)S.co * 9 * S.ci * 2 + 65536));
        __bf16* x0 = xb + guard * S.ci; __bf16* y0 = yb + guard * S.co; __bf16* g0 = gb + guard * S.co;
        fill_framed<<<1024, 256, 0, st>>>(x0, S.n, S.h, S.w, S.ci, 17u + si, 1.0f);
        fill_framed<<<1024, 256, 0, st>>>(g0, S.n, S.h, S.w, S.co, 91u + si, 1.0f);
        fill_kernel<<<1024, 256, 0, st>>>(wpk, (size_t)S.co * 9 * S.ci, 5u + si, sqrtf(6.0f / (9.f * S.ci)));
        retile_w<<<1024, 256, 0, st>>>(wpk, wtl, S.co, S.ci, S.co % 256 == 0 ? 256 : 128);
        { std::vector<float> b(S.co); for (int i = 0; i < S.co; ++i) b[i] = 0.01f * ((i * 37) % 29 - 14); CK(hipMemcpy(bias, b.data(), S.co * 4, hipMemcpyHostToDevice)); }
        // sample pixels for the reference: corners, edges, tile seams, a stride through the rest
        std::vector<int> pix;
        const int M = S.n * S.h * S.w;
        for (int i = 0; i < M && (int)pix.size() < 4096; i += (M / 1500 > 0 ? M / 1500 : 1)) pix.push_back(i);
        for (int im : {0, S.n - 1}) for (int yy : {0, 1, S.h - 1}) for (int xx = 0; xx < S.w; ++xx) pix.push_back((im * S.h + yy) * S.w + xx);
        int* dpix; CK(hipMalloc(&dpix, pix.size() * 4)); CK(hipMemcpy(dpix, pix.data(), pix.size() * 4, hipMemcpyHostToDevice));
        const int nout = (int)pix.size() * S.co;
        float *ref, *got, *fsum; CK(hipMalloc(&ref, nout * 4)); CK(hipMalloc(&got, nout * 4)); CK(hipMalloc(&fsum, 4));
        ref_conv<<<(nout + 255) / 256, 256, 0, st>>>(x0, wpk, bias, g0, ref, S.n, S.h, S.w, S.ci, S.co, S.epi, (int)pix.size(), dpix);
        std::vector<float> href(nout), hgot(nout);
        CK(hipMemcpyAsync(href.data(), ref, nout * 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));

        dbx_conv_desc d; d.dtype = DBX_BF16; d.kh = d.kw = 3; d.cpad = 1; d.cin_pad = S.ci; d.cout_pad = S.co; d.epilogue = S.epi; d.drop_seed = 0;
        dbx_view xv{x0, S.n, S.h, S.w, 1, S.ci, 0, S.ci}, yv{y0, S.n, S.h, S.w, 1, S.co, 0, S.co}, gv{g0, S.n, S.h, S.w, 1, S.co, 0, S.co};
        const double flop = 2.0 * M * 9.0 * S.ci * S.co;
        printf("%s  (%.1f GFLOP)\n", S.name, flop * 1e-9);
        for (int v : variants) {
            g_conv_variant_override = v;
            CK(hipMemsetAsync(yb, 0, (xpix + 2 * guard) * S.co * 2, st));
            const bool frag = v == 8 || v == 7 || (v >= 70 && v < 80);
            const __bf16* wuse = frag ? wtl : wpk;
            d.epilogue = S.epi | (v == 8 ? DBX_CONV_WFRAG : 0);
#ifdef BAND_TS
            // -DBAND_TS: the band kernel stamps the shader clock around its stage barrier (three per stage and wave): how long a wave waits for
            // its loads, for the barrier, and how long its three taps take -- per half of the workgroup's waves
            unsigned long long* tsb; CK(hipMalloc(&tsb, 64 * 16 * 64 * 3 * 8)); CK(hipMemset(tsb, 0, 64 * 16 * 64 * 3 * 8));
            int rc = conv_forward_t<__bf16>(&d, &xv, wuse, bias, &yv, (S.epi & DBX_EPI_GATE) ? &gv : nullptr, nullptr, 0, st, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, (float*)tsb);
            CK(hipStreamSynchronize(st));
            {
                std::vector<unsigned long long> h(64 * 16 * 64 * 3);
                CK(hipMemcpy(h.data(), tsb, h.size() * 8, hipMemcpyDeviceToHost));
                const int NWV = 8;
                for (int wg : {0, 9, 33}) for (int wv : {0, 5}) {
                    const unsigned long long* t = h.data() + ((size_t)wg * NWV + wv) * 64 * 3;
                    double sw = 0, sb = 0, sc = 0; int n = 0;
                    for (int ks = 0; ks < 63 && t[(ks + 1) * 3]; ++ks) { sw += t[ks * 3 + 1] - t[ks * 3]; sb += t[ks * 3 + 2] - t[ks * 3 + 1]; sc += t[(ks + 1) * 3] - t[ks * 3 + 2]; ++n; }
                    if (n) printf("    wg %2d wave %d, mean of %d stages: load wait %.0f  barrier wait %.0f  three taps %.0f  (sum %.0f clocks; the stage's MFMAs of both waves of a SIMD: 3072)\n", wg, wv, n, sw / n, sb / n, sc / n, (sw + sb + sc) / n);
                }
            }
            CK(hipFree(tsb));
            if (rc) printf("  variant %d: error %s\n", v, g_err);
            continue;                                       // (the stamped kernel writes through a.part: no un-stamped timing runs)
#else
            int rc = conv_forward_t<__bf16>(&d, &xv, wuse, bias, &yv, (S.epi & DBX_EPI_GATE) ? &gv : nullptr, nullptr, 0, st);
#endif
            if (rc) { printf("  variant %d: error %s\n", v, g_err); continue; }
            CK(hipStreamSynchronize(st));
            gather_out<<<(nout + 255) / 256, 256, 0, st>>>(y0, got, S.h, S.w, S.co, (int)pix.size(), dpix);
            CK(hipMemsetAsync(fsum, 0, 4, st));
            frame_sum<<<256, 256, 0, st>>>(y0, S.n, S.h, S.w, S.co, fsum);
            float hf; CK(hipMemcpyAsync(hgot.data(), got, nout * 4, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&hf, fsum, 4, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            double maxerr = 0, maxref = 0; int bad = 0;
            for (int i = 0; i < nout; ++i) {
                const double e = fabs((double)hgot[i] - href[i]); maxerr = e > maxerr ? e : maxerr; maxref = fabs(href[i]) > maxref ? fabs(href[i]) : maxref;
                if (e > 0.02 + 0.01 * fabs(href[i])) {
                    if (bad < 12 && getenv("LAB_DEBUG")) {
                        const int pi = pix[i / S.co], c = i % S.co;
                        printf("    bad: img %d y %d x %d (m %d) cout %d got %.4f ref %.4f\n", pi / (S.h * S.w), (pi / S.w) % S.h, pi % S.w, pi, c, hgot[i], href[i]);
                    }
                    ++bad;
                }
            }
            if (bad && getenv("LAB_DEBUG")) {   // bad-pixel histogram over images / rows
                std::vector<int> bimg(S.n, 0), brow(S.h, 0), bcol(S.w, 0), bc(S.co / 32, 0);
                for (int i = 0; i < nout; ++i) if (fabs((double)hgot[i] - href[i]) > 0.02 + 0.01 * fabs(href[i])) {
                    const int pi = pix[i / S.co]; ++bimg[pi / (S.h * S.w)]; ++brow[(pi / S.w) % S.h]; ++bcol[pi % S.w]; ++bc[(i % S.co) / 32];
                }
                printf("    bad per image:"); for (int v2 : bimg) printf(" %d", v2); printf("\n    bad per row:"); for (int v2 : brow) printf(" %d", v2);
                printf("\n    bad per col:"); for (int v2 : bcol) printf(" %d", v2); printf("\n    bad per cout/32:"); for (int v2 : bc) printf(" %d", v2); printf("\n");
            }
            // timing
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < 3; ++i) conv_forward_t<__bf16>(&d, &xv, wuse, bias, &yv, (S.epi & DBX_EPI_GATE) ? &gv : nullptr, nullptr, 0, st);
            float best = 1e9f, tot = 0;
            for (int r = 0; r < reps; ++r) {
                CK(hipEventRecord(e0, st));
                conv_forward_t<__bf16>(&d, &xv, wuse, bias, &yv, (S.epi & DBX_EPI_GATE) ? &gv : nullptr, nullptr, 0, st);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; tot += ms;
            }
            printf("  variant %d: avg %.1f us  best %.1f us  %.0f TFLOP/s (avg)  %.0f (best)   maxerr %.4f (max|ref| %.2f) bad %d frame %.3g %s\n", v,
                   tot / reps * 1e3, best * 1e3, flop / (tot / reps * 1e-3) * 1e-12, flop / (best * 1e-3) * 1e-12, maxerr, maxref, bad, hf,
                   (bad || hf != 0.f) ? "FAIL" : "ok");
        }
        CK(hipFree(xb)); CK(hipFree(yb)); CK(hipFree(gb)); CK(hipFree(wpk)); CK(hipFree(wtl)); CK(hipFree(bias)); CK(hipFree(dpix)); CK(hipFree(ref)); CK(hipFree(got)); CK(hipFree(fsum));
    }
    return 0;
}
