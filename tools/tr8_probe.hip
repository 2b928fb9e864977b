// What does ds_read_b64_tr_b8 return?  (gfx950; the ISA manual is not in this image.)  LDS is filled with distinct 16-bit tags
// at byte granularity (byte i holds i & 0xff, and a second run holds i >> 8), every lane passes the address base + lane * STRIDE,
// and the 8 bytes each lane gets back are printed as source byte offsets.
// build: hipcc --offload-arch=gfx950 -O2 tools/tr8_probe.hip -o tools/tr8_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) int i32x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__global__ void probe(int stride, int hi, uint32_t* out, int which) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = hi ? (unsigned char)(i >> 8) : (unsigned char)(i & 0xff);
  __syncthreads();
  const unsigned char* p = lds + threadIdx.x * stride;
  if (which == 0) {
    i32x2_t v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2_t*)p);
    out[threadIdx.x * 2] = (uint32_t)v[0]; out[threadIdx.x * 2 + 1] = (uint32_t)v[1];
  } else {
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
    out[threadIdx.x * 2] = (uint16_t)v[0] | ((uint32_t)(uint16_t)v[1] << 16); out[threadIdx.x * 2 + 1] = (uint16_t)v[2] | ((uint32_t)(uint16_t)v[3] << 16);
  }
}
int main() {
  uint32_t *d, lo[128], hi[128];
  hipMalloc(&d, 512);
  for (int which = 0; which < 2; ++which)
    for (int stride : {8, 256}) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, stride, 0, d, which); hipMemcpy(lo, d, 512, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, stride, 1, d, which); hipMemcpy(hi, d, 512, hipMemcpyDeviceToHost);
      printf("%s, lane address = base + lane * %d: source byte offset of each returned byte\n", which ? "ds_read_b64_tr_b16" : "ds_read_b64_tr_b8", stride);
      for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 8; ++j) {
          const int b_lo = (lo[l * 2 + j / 4] >> (8 * (j % 4))) & 0xff, b_hi = (hi[l * 2 + j / 4] >> (8 * (j % 4))) & 0xff;
          printf(" %5d", b_hi * 256 + b_lo);
        }
        printf("\n");
      }
    }
  return 0;
}
