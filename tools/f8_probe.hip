// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 operands, unit scales): which bytes of a lane feed which (row, k)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(const uint8_t* A, const uint8_t* B, float* C, int mode) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  // hypothesis `mode`: 0: lane = row (lane % 32), k block = lane / 32, 32 consecutive k per lane
  //                    1: k = 16 * j + 8 * (lane / 32) + (0..7) for j = 0..3 (four 8-byte groups interleaved like the x16 MFMA)
  uint8_t ab[32], bb[32];
  for (int i = 0; i < 32; ++i) {
    int kk = mode == 0 ? (lane / 32) * 32 + i : 16 * (i / 8) + 8 * (lane / 32) + (i % 8);
    ab[i] = A[(lane % 32) * 64 + kk];
    bb[i] = B[(lane % 32) * 64 + kk];
  }
  for (int i = 0; i < 8; ++i) {
    a[i] = ab[4 * i] | (ab[4 * i + 1] << 8) | (ab[4 * i + 2] << 16) | (ab[4 * i + 3] << 24);
    b[i] = bb[4 * i] | (bb[4 * i + 1] << 8) | (bb[4 * i + 2] << 16) | (bb[4 * i + 3] << 24);
  }
  f32x16 acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
    C[row * 32 + col] = acc[r];      // C[m][n] = sum_k A[m][k] B[n][k]
  }
}
static float e4m3(uint8_t v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}
int main() {
  std::vector<uint8_t> A(32 * 64), B(32 * 64);
  for (int i = 0; i < 32 * 64; ++i) { A[i] = (uint8_t)(0x30 + (i * 7) % 24) ^ ((i & 5) == 5 ? 0x80 : 0); B[i] = (uint8_t)(0x2c + (i * 13) % 28) ^ ((i & 3) == 3 ? 0x80 : 0); }
  uint8_t *dA, *dB; float* dC;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 32 * 32 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  std::vector<float> ref(32 * 32, 0.f), C(32 * 32);
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int kk = 0; kk < 64; ++kk) s += e4m3(A[m * 64 + kk]) * e4m3(B[n * 64 + kk]); ref[m * 32 + n] = s; }
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, mode);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    float err = 0, mx = 0;
    for (int i = 0; i < 1024; ++i) { err = fmaxf(err, fabsf(C[i] - ref[i])); mx = fmaxf(mx, fabsf(ref[i])); }
    printf("layout hypothesis %d: max abs err %.4g (max |ref| %.4g) C[0]=%.4f ref[0]=%.4f\n", mode, err, mx, C[0], ref[0]);
  }
  return 0;
}
