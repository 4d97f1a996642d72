// Calibration of the 8-phase MFMA core (densebox_amd/csrc/mma8p.hpp) as a plain GEMM: C[M][N] = A[M][K] . B[N][K]^T, f16 / bf16 operands,
// fp32 accumulation, C in the operand type.  M, N multiples of 256, K a multiple of 128.  Driven by tools/gpu_gemm_8phase.py (operands,
// check against torch.mm, timing, profiles/r05_gemm_8phase.txt).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tools/probe_gemm_8phase.hip -o tools/labbin/libprobe_gemm_8phase.so
#include "../densebox_amd/csrc/common.hpp"
#include "../densebox_amd/csrc/mma8p.hpp"

void dbx_set_error(const char*, ...) {}

struct GemmArgs {
    const char* A; const char* B; char* C;
    int M, N, K;
    int tm, tn;          // tiles
    int gm;              // tile rows per group of the workgroup order
};

// variant = MF (16 / 32) and FLAGS of p8
template <typename T, int MF, int FLAGS>
__global__ __launch_bounds__(512, 1) void gemm8p_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // workgroup order: blocks b, b + 8, .. run on one XCD (b % 8) -> give an XCD a contiguous run of the grouped tile order
    // (groups of gm tile rows, column-major inside a group: the 32 workgroups of an XCD share gm A panels and 32 / gm B panels)
    const int nwg = gridDim.x;
    int s = blockIdx.x;
    if ((nwg & 7) == 0) s = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
    const int per_group = a.gm * a.tn;
    const int grp = s / per_group, r = s - grp * per_group;
    const int rows = a.tm - grp * a.gm < a.gm ? a.tm - grp * a.gm : a.gm;       // last group may be short
    const int tm = grp * a.gm + r % rows, tn = r / rows;
    const int m0 = tm * 256, n0 = tn * 256;
    const size_t lda = (size_t)a.K * 2;

    const p8::Lanes<MF> L = p8::lanes<MF>(smem);
    const unsigned vo = (unsigned)((L.wave * 8 + (L.lane >> 3)) * lda) + (p8::src_chunk(L.wave, L.lane) << 4);
    const char* const Ab = a.A + (size_t)m0 * lda;
    const char* const Bb = a.B + (size_t)n0 * lda;
    p8::Acc<MF> acc;
    p8::zero<MF>(acc);
    const int nkt = a.K / 64;
    // tiles past the end of K are clamped to the last one (their loads land in buffers nobody reads any more)
    auto stA = [&](int kt, int mh, unsigned dst) {
        kt = kt < nkt ? kt : nkt - 1;
        const char* b = Ab + (size_t)(mh * 128) * lda + (size_t)kt * 128;
        p8::glds(b, vo, dst); p8::glds(b + 64 * lda, vo, dst + 8192);
    };
    auto stB = [&](int kt, int nh, unsigned dst) {
        kt = kt < nkt ? kt : nkt - 1;
        const char* b = Bb + (size_t)(nh * 128) * lda + (size_t)kt * 128;
        p8::glds(b, vo, dst); p8::glds(b + 64 * lda, vo, dst + 8192);
    };
    if constexpr (FLAGS & 64) {           // the two-phases-per-K-tile variant (32 MFMAs per cluster)
        if constexpr (MF == 16) {
            p8::prologue4(L, stA, stB);
            p8::start4<FLAGS>(L);
            p8::ktiles4<T, FLAGS>(acc, L, nkt, stA, stB);
            p8::finish<FLAGS>(L);
        }
    } else {
        p8::prologue<MF>(L, stA, stB);
        p8::start<FLAGS>(L);
        p8::ktiles<T, MF, FLAGS>(acc, L, nkt, stA, stB);
        p8::finish<FLAGS>(L);
    }
    T* const C = (T*)a.C;
    p8::for_chunks<T, MF>(acc, L, [](int, int, int, f32x4&) {},
        [&](int mh, int nh, int m, int n, const u32x4& o) { *(u32x4*)(C + (size_t)(m0 + mh * 128 + m) * a.N + n0 + nh * 128 + n) = o; });
}

template <typename T, int MF, int FLAGS>
static int launch(const GemmArgs& a, hipStream_t st) {
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute((const void*)gemm8p_kernel<T, MF, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, p8::LDS_BYTES) != hipSuccess) return -2;
        once = true;
    }
    hipLaunchKernelGGL((gemm8p_kernel<T, MF, FLAGS>), dim3(a.tm * a.tn), dim3(512), p8::LDS_BYTES, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// variant: 0 = 16x16x32 prio + stagger (the template), 1 = 32x32x16 prio + stagger, 2 = 16 stagger without prio, 3 = 16 prio without stagger,
//          4 = 16 SAFE (drained waits, no stagger), 5 = 32 stagger without prio, 6 = 32 SAFE, 7 = 16 stagger, two phases per K tile (32-MFMA clusters),
//          8 / 9 / 10 = ablations of variant 2 (wrong results: timing only): no LDS-DMA in the loop / no fragment reads / neither
extern "C" int p8_gemm(int variant, int dtype, const void* A, const void* B, void* C, int M, int N, int K, int gm, void* stream) {
    if (M % 256 || N % 256 || K % 128 || K < 128) return -1;
    GemmArgs a;
    a.A = (const char*)A; a.B = (const char*)B; a.C = (char*)C; a.M = M; a.N = N; a.K = K;
    a.tm = M / 256; a.tn = N / 256; a.gm = gm < 1 ? 1 : (gm > a.tm ? a.tm : gm);
    hipStream_t st = (hipStream_t)stream;
    using namespace p8;
#define V(T)                                                                         \
    switch (variant) {                                                               \
        case 0: return launch<T, 16, FL_PRIO | FL_STAGGER>(a, st);                   \
        case 1: return launch<T, 32, FL_PRIO | FL_STAGGER>(a, st);                   \
        case 2: return launch<T, 16, FL_STAGGER>(a, st);                             \
        case 3: return launch<T, 16, FL_PRIO>(a, st);                                \
        case 4: return launch<T, 16, FL_SAFE>(a, st);                                \
        case 5: return launch<T, 32, FL_STAGGER>(a, st);                             \
        case 6: return launch<T, 32, FL_SAFE>(a, st);                                \
        case 7: return launch<T, 16, FL_STAGGER | 64>(a, st);                        \
        case 8: return launch<T, 16, FL_STAGGER | FL_NODMA>(a, st);                  \
        case 9: return launch<T, 16, FL_STAGGER | FL_NOREAD>(a, st);                 \
        case 10: return launch<T, 16, FL_STAGGER | FL_NODMA | FL_NOREAD>(a, st);     \
        default: return -4;                                                          \
    }
    if (dtype == DBX_F16) { V(_Float16) }
    if (dtype == DBX_BF16) { V(__bf16) }
    return -5;
}
