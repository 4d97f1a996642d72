// L2 -> CU fill-rate probe (gfx950): every workgroup (512 threads, one per CU) streams a private `span`-byte window of
// a buffer over and over with 16-byte-per-lane loads; small spans stay in L1/L2, so the result is the attainable
// L2 -> LDS / L2 -> VGPR rate per CU with all 256 CUs loading at once.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_fill_rate.hip -o /tmp/probe_fill_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void probe(const char* buf, unsigned* out, long long span, long long stride, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const char* base = buf + (long long)blockIdx.x * stride;
    u32x4 acc = {0, 0, 0, 0};
    long long off = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {                    // 4 x 8 KB per iteration
            const char* p = base + off + u * 8192 + tid * 16;
            if (MODE == 0) {
                const u32x4 v = *(const u32x4*)p;
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            } else {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(smem + u * 8192 + wave * 1024), 16, 0, 0);
            }
        }
        off += 32768; if (off >= span) off = 0;
        if (MODE == 1 && (it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    if (MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); acc.x = ((unsigned*)smem)[tid]; }
    out[blockIdx.x * 512 + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}
// tile-like pattern: a wave instruction covers ROWS rows of (1024/ROWS) bytes, rows `pitch` bytes apart (an im2col piece);
// a workgroup's 8 waves x 4 instructions cover 32 x ROWS rows; the window advances along the row (K) by 1024/ROWS bytes per
// iteration and wraps after `kbytes`, like a K loop over a [rows][K] operand that lives in L2.
template <int ROWS>
__global__ __launch_bounds__(512) void probe_tile(const char* buf, unsigned* out, int pitch, int kbytes, long long stride, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RB = 1024 / ROWS, CPR = RB / 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const char* base = buf + (long long)blockIdx.x * stride + (long long)(lane / CPR) * pitch + (lane % CPR) * 16;
    int k = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (long long)((u * 8 + wave) * ROWS) * pitch + k),
                                             (__attribute__((address_space(3))) void*)(smem + u * 8192 + wave * 1024), 16, 0, 0);
        k += RB; if (k >= kbytes) k = 0;
        if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
    out[blockIdx.x * 512 + tid] = ((unsigned*)smem)[tid];
}
template <int ROWS> void run_tile(const char* buf, unsigned* out, int pitch, int kbytes, long long stride) {
    const int iters = 4000, blocks = 256;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe_tile<ROWS><<<blocks, 512, 32768>>>(buf, out, pitch, kbytes, stride, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); probe_tile<ROWS><<<blocks, 512, 32768>>>(buf, out, pitch, kbytes, stride, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double bytes_per_cu = (double)iters * 32768;
    printf("tile rows/instr %2d (%4d B per row) pitch %5d K %5d B, window stride %6lld KB: %6.1f B/clk/CU\n", ROWS, 1024 / ROWS, pitch, kbytes,
           stride >> 10, bytes_per_cu / (ms * 1e-3 * 2.4e9));
}
template <int MODE> void run(const char* name, const char* buf, unsigned* out, long long span, long long stride) {
    const int iters = 4000, blocks = 256;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    probe<MODE><<<blocks, 512, 32768>>>(buf, out, span, stride, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); probe<MODE><<<blocks, 512, 32768>>>(buf, out, span, stride, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double bytes_per_cu = (double)iters * 32768;
    printf("%-12s span %8lld KB stride %8lld KB: %6.1f B/clk/CU  (%.2f TB/s aggregate)\n", name, span >> 10, stride >> 10,
           bytes_per_cu / (ms * 1e-3 * 2.4e9), bytes_per_cu * blocks / (ms * 1e-3) / 1e12);
}
int main() {
    char* buf; unsigned* out;
    const long long total = 1LL << 32;
    (void)hipMalloc(&buf, total); (void)hipMemset(buf, 1, total); (void)hipMalloc(&out, 256 * 512 * 4);
    // private windows: L1-resident (32 KB), L2-resident (256 KB/CU -> 8 MB/XCD?), beyond L2 (4 MB/CU), HBM (16 MB/CU)
    for (long long span : {32LL << 10, 128LL << 10, 1LL << 20, 16LL << 20}) {
        run<0>("vgpr", buf, out, span, span);
        run<1>("lds-dma", buf, out, span, span);
    }
    // shared window: all CUs read the same 4 MB (weights-like)
    run<0>("vgpr shared", buf, out, 4LL << 20, 0);
    run<1>("dma shared", buf, out, 4LL << 20, 0);
    // [512 rows][K] operand tiles resident in L2 (512 rows x pitch per CU, or one shared by all CUs)
    for (int pitch : {1536, 4096}) {
        run_tile<16>(buf, out, pitch, pitch, 0);
        run_tile<8>(buf, out, pitch, pitch, 0);
        run_tile<4>(buf, out, pitch, pitch, 0);
        run_tile<16>(buf, out, pitch, pitch, 512LL * pitch);
        run_tile<8>(buf, out, pitch, pitch, 512LL * pitch);
    }
    return 0;
}
