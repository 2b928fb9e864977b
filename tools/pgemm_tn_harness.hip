// Correctness + timing harness of pgemm_tn_kernel (tn_pgemm.h) against a plain f32 reference (tuning tool).
//   tools/pgemm_tn_harness [rows np nq]
#include "../include/titanet_amd.h"
#include "../titanet_amd/csrc/tn_pgemm.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s\n", (int)e, #x); return 1; } } while (0)
template <class F> float timeit(F f, int n = 10) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) f(i);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) f(i + 2);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / n;
}
static unsigned short rnd_bf16(uint32_t& s) {
  s = s * 1664525u + 1013904223u;
  const float f = ((int)(s >> 8) - (1 << 23)) / (float)(1 << 23);
  uint32_t u; memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
// out[c][k] = sum_r P[r][c] Q[r][k], one thread per output, rows split over blockIdx.z with atomics
__global__ void ref_tn(const bf16_t* P, int ldp, const bf16_t* Q, int ldq, int rows, float* out, int ldo, int np, int nq) {
  const int k = blockIdx.x * 64 + (threadIdx.x & 63), c = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int r0 = blockIdx.z * 1200, r1 = min(rows, r0 + 1200);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += bf2f(P[(size_t)r * ldp + c]) * bf2f(Q[(size_t)r * ldq + k]);
  atomicAdd(out + (size_t)c * ldo + k, s);
}
int main(int argc, char** argv) {
  const int rows = argc > 3 ? atoi(argv[1]) : 256 * 300, np = argc > 3 ? atoi(argv[2]) : 1024, nq = argc > 3 ? atoi(argv[3]) : 1024;
  printf("TN GEMM rows=%d np=%d nq=%d (%.1f GFLOP)\n", rows, np, nq, 2.0 * rows * np * nq / 1e9);
  std::vector<unsigned short> hp((size_t)rows * np), hq((size_t)rows * nq);
  uint32_t s = 777u;
  for (auto& v : hp) v = rnd_bf16(s);
  for (auto& v : hq) v = rnd_bf16(s);
  bf16_t *P[2], *Q[2]; float *out, *ref;
  for (int i = 0; i < 2; ++i) {
    CK(hipMalloc(&P[i], hp.size() * 2)); CK(hipMalloc(&Q[i], hq.size() * 2));
    CK(hipMemcpy(P[i], hp.data(), hp.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(Q[i], hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&out, (size_t)np * nq * 4)); CK(hipMalloc(&ref, (size_t)np * nq * 4));
  CK(hipMemset(out, 0, (size_t)np * nq * 4)); CK(hipMemset(ref, 0, (size_t)np * nq * 4));
  hipLaunchKernelGGL(ref_tn, dim3(nq / 64, np / 4, (rows + 1199) / 1200), dim3(256), 0, 0, P[0], np, Q[0], nq, rows, ref, nq, np, nq);
  PGemmTnArgs a{P[0], np, np, Q[0], nq, nq, rows, out, nq, 0, 0};
  int rc = launch_pgemm_tn(a, 0);
  if (rc) { printf("launch failed %d\n", rc); return 1; }
  CK(hipDeviceSynchronize());
  std::vector<float> ho((size_t)np * nq), hr((size_t)np * nq);
  CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0; size_t wi = 0;
  for (size_t i = 0; i < ho.size(); ++i) { const double d = fabs((double)ho[i] - hr[i]); if (d > worst) { worst = d; wi = i; } scale = fmax(scale, fabs((double)hr[i])); }
  printf("max abs diff %.4g at (%zu, %zu) [%.5g vs %.5g], max |ref| %.4g -> relative %.3g\n", worst, wi / nq, wi % nq, ho[wi], hr[wi], scale, worst / scale);
  const double flop = 2.0 * rows * np * nq;
  for (int wgs : {256, 192, 128, 96, 64}) {
    float us = timeit([&](int i) { a.P = P[i & 1]; a.Q = Q[i & 1]; launch_pgemm_tn(a, 0, wgs); });
    printf("pgemm_tn_kernel (%3d wgs) : %8.2f us  %.3f PFLOP/s   (P + Q once = %.0f MB -> %.2f TB/s)\n", wgs, us, flop / us / 1e9, (hp.size() + hq.size()) * 2 / 1e6,
           (hp.size() + hq.size()) * 2 / us / 1e6);
  }
  return 0;
}
