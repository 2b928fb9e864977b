// Hardware probe: semantics of ds_read_b64_tr_b16 (gfx950) for the weight-gradient (TN) GEMM operand loads.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4;
#define PITCH 40
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  int l = threadIdx.x;
  int off = (l >> 4) * 1024 + ((l & 15) >> 2) * PITCH + (l & 3) * 4;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + off));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) {
      int expect = (l >> 4) * 1024 + j * PITCH + (l & 15);
      printf(" %5d%s", h[l * 4 + j], h[l * 4 + j] == expect ? "" : "!");
      bad += h[l * 4 + j] != expect;
    }
    printf("\n");
  }
  printf("TRPROBE mismatches=%d (expect v[j] = group_base + j*PITCH + (lane&15))\n", bad);
  return 0;
}
