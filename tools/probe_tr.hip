// One-off hardware probe: semantics of ds_read_b64_tr_b16 on gfx950 (used to design the wgrad kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int mode) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  int l = threadIdx.x;
  int addr_elems = mode == 0 ? l * 4 : ((l & 15) * 64 + (l >> 4) * 4);
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(lds + addr_elems));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  for (int mode = 0; mode < 2; ++mode) {
    k<<<1, 64>>>(d, mode); short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (lane addr = %s)\n", mode, mode == 0 ? "lane*4 elems (contiguous)" : "(l&15)*64 + (l>>4)*4 elems (row-major 16 rows x 64)");
    for (int l = 0; l < 64; ++l) { printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
  }
  return 0;
}
