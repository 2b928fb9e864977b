// MFMA issue-rate probe (tuning tool): back-to-back v_mfma_f32_32x32x16_bf16 on 8 independent accumulators per wave,
// operands in registers (random bits), 1 or 2 waves per SIMD, with and without a barrier every 8 MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <int NT, int MODE>
__global__ __launch_bounds__(NT) void k(const uint4* __restrict__ in, float* out, int iters) {
  extern __shared__ char smem[];
  bf16x8_t a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8_t, in[threadIdx.x * 6 + i]);
  for (int i = 0; i < 2; ++i) b[i] = __builtin_bit_cast(bf16x8_t, in[threadIdx.x * 6 + 4 + i]);
  f32x16_t acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const char* lp = smem + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
  for (int it = 0; it < iters; ++it) {
    if (MODE & 2) {   // fragment reads from LDS (6 x 16 bytes per 8 MFMAs)
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(lp + i * 1024 + (it & 3) * 32768);
      for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const bf16x8_t*>(lp + (4 + i) * 1024 + (it & 3) * 32768);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE & 4) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[2 * i]) : "v"(a[i]), "v"(b[0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[2 * i + 1]) : "v"(a[i]), "v"(b[1]));
      } else {
        acc[2 * i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[0], acc[2 * i], 0, 0, 0);
        acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[1], acc[2 * i + 1], 0, 0, 0);
      }
    }
    if (MODE & 1) __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int NT, int MODE>
void run(const uint4* in, float* out, const char* what, bool zero) {
  const int iters = 20000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<NT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NT, MODE>), dim3(256), dim3(NT), 131072, 0, in, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NT, MODE>), dim3(256), dim3(NT), 131072, 0, in, out, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = 256.0 * (NT / 64) * iters * 8 * 32768.0;
  printf("%-52s %s: %8.1f us  %.3f PFLOP/s\n", what, zero ? "zero data  " : "random data", ms * 1e3, flop / ms / 1e12);
}
int main() {
  uint4* in; float* out;
  hipMalloc(&in, 512 * 6 * 16); hipMalloc(&out, 256 * 512 * 4);
  for (int z = 0; z < 2; ++z) {
    uint32_t h[512 * 6 * 4]; uint32_t s = 99;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; uint32_t lo = ((s >> 3) & 0x80ffu) | 0x3f00u, hi = ((s >> 17) & 0x80ffu) | 0x3f00u; v = z ? 0u : (lo | (hi << 16)); }
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<256, 0>(in, out, "1 wave/SIMD, registers", z);
    run<512, 0>(in, out, "2 waves/SIMD, registers", z);
    run<512, 1>(in, out, "2 waves/SIMD, registers, barrier per 8 MFMAs", z);
    run<512, 2>(in, out, "2 waves/SIMD, LDS fragments", z);
    run<512, 3>(in, out, "2 waves/SIMD, LDS fragments, barrier per 8 MFMAs", z);
    run<256, 2>(in, out, "1 wave/SIMD, LDS fragments", z);
    run<512, 4>(in, out, "2 waves/SIMD, registers, AGPR accumulators", z);
    run<512, 6>(in, out, "2 waves/SIMD, LDS fragments, AGPR accumulators", z);
    run<512, 7>(in, out, "2 waves/SIMD, LDS frags, AGPR acc, barrier", z);
    run<256, 6>(in, out, "1 wave/SIMD, LDS fragments, AGPR accumulators", z);
  }
  return 0;
}
