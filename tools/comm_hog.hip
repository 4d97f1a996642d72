// A stand-in for a communication kernel's channel workgroups: G workgroups of `threads` threads, each spinning for `us` microseconds
// (wall clock, 100 MHz) with a hard iteration bound.  tools/gpu_comm_interference.py launches it on a side stream during the training step.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void hog_kernel(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    int it = 0;
    while (wall_clock64() - t0 < ticks && it < (1 << 24)) { ++it; __builtin_amdgcn_s_sleep(8); }
    if (sink && it == -1) *sink = it;
}
extern "C" int comm_hog(int groups, int threads, int us, void* stream) {
    hipLaunchKernelGGL(hog_kernel, dim3(groups), dim3(threads), 0, (hipStream_t)stream, (long long)us * 100, (int*)nullptr);
    return (int)hipGetLastError();
}
