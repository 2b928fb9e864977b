// Correctness + timing harness of the fp8 weight-gradient contraction (tn_pgemm.h: pgemm_tn_f8_batched_kernel) against a host
// reference in double (tuning tool).     tools/pgemm_tn_f8_harness [rows H layers]
#include "../include/titanet_amd.h"
#include "../titanet_amd/csrc/tn_pgemm.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s\n", (int)e, #x); return 1; } } while (0)
static float e4m3(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  const float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}
int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 76800, H = argc > 2 ? atoi(argv[2]) : 1024, L = argc > 3 ? atoi(argv[3]) : 4;
  const int U = (H / 256) * (H / 256);
  printf("fp8 TN contraction: %d layers of rows=%d, %d x %d (%.1f GFLOP each)\n", L, rows, H, H, 2.0 * rows * H * H / 1e9);
  std::vector<uint8_t> hp((size_t)rows * H), hq((size_t)rows * H), he(H);
  uint32_t s = 4242u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  for (auto& v : hp) { uint8_t b = (uint8_t)(rnd() & 0x7f); if ((b & 0x7f) == 0x7f) b = 0x7e; v = (uint8_t)(b | ((rnd() & 1) << 7)); }      // (no NaN pattern)
  for (auto& v : hq) { uint8_t b = (uint8_t)(0x20 + rnd() % 0x30); v = (uint8_t)(b | ((rnd() & 1) << 7)); }
  for (auto& v : he) v = (uint8_t)(120 + rnd() % 12);
  std::vector<uint8_t*> P(L), Q(L); std::vector<float*> out(L);
  uint8_t* ce;
  CK(hipMalloc(&ce, H)); CK(hipMemcpy(ce, he.data(), H, hipMemcpyHostToDevice));
  for (int l = 0; l < L; ++l) {
    CK(hipMalloc(&P[l], hp.size())); CK(hipMalloc(&Q[l], hq.size())); CK(hipMalloc(&out[l], (size_t)H * H * 4));
    CK(hipMemcpy(P[l], hp.data(), hp.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(Q[l], hq.data(), hq.size(), hipMemcpyHostToDevice));
    CK(hipMemset(out[l], 0, (size_t)H * H * 4));
  }
  std::vector<PGemmTnF8Desc> hd(L);
  for (int l = 0; l < L; ++l) hd[l] = PGemmTnF8Desc{P[l], Q[l], out[l], ce, H, H, H, H / 256};
  PGemmTnF8Desc* dd; CK(hipMalloc(&dd, sizeof(PGemmTnF8Desc) * L)); CK(hipMemcpy(dd, hd.data(), sizeof(PGemmTnF8Desc) * L, hipMemcpyHostToDevice));
  int rc = launch_pgemm_tn_f8_batched(dd, L, rows, U, nullptr, 0, 0, 256, H);
  if (rc) { printf("launch failed %d\n", rc); return 1; }
  CK(hipDeviceSynchronize());
  // host reference on a sample of outputs (every 37th element of layer 0 and the last layer)
  for (int l : {0, L - 1}) {
    std::vector<float> ho((size_t)H * H);
    CK(hipMemcpy(ho.data(), out[l], ho.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0; size_t n = 0;
    for (size_t idx = 0; idx < ho.size(); idx += 37 * 7 + 1) {
      const int o = (int)(idx / H), c = (int)(idx % H);
      double acc = 0;
      for (int r = 0; r < rows; ++r) acc += (double)e4m3(hp[(size_t)r * H + o]) * (double)e4m3(hq[(size_t)r * H + c]);
      acc *= ldexp(1.0, (int)he[o] - 127);
      worst = fmax(worst, fabs(acc - ho[idx])); scale = fmax(scale, fabs(acc)); ++n;
    }
    printf("layer %d: %zu sampled outputs, max abs diff %.4g, max |ref| %.4g -> relative %.3g\n", l, n, worst, scale, worst / scale);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) launch_pgemm_tn_f8_batched(dd, L, rows, U, nullptr, 0, 0, 256, H);
  CK(hipDeviceSynchronize());
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int it = 0; it < reps; ++it) launch_pgemm_tn_f8_batched(dd, L, rows, U, nullptr, 0, 0, 256, H);
  hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("pgemm_tn_f8_batched: %.1f us per launch of %d layers = %.1f us per layer, %.3f PFLOP/s, %.2f TB/s of operand bytes\n", us, L, us / L,
         2.0 * rows * H * H * L / us / 1e9, 2.0 * rows * H * L / us / 1e6);
  // the bf16 launch on the same shapes (operands twice the bytes)
  {
    std::vector<bf16_t*> Pb(L), Qb(L);
    for (int l = 0; l < L; ++l) { CK(hipMalloc(&Pb[l], (size_t)rows * H * 2)); CK(hipMalloc(&Qb[l], (size_t)rows * H * 2)); CK(hipMemset(Pb[l], 0x3c, (size_t)rows * H * 2)); CK(hipMemset(Qb[l], 0x3c, (size_t)rows * H * 2)); }
    std::vector<PGemmTnDesc> hb(L);
    for (int l = 0; l < L; ++l) hb[l] = PGemmTnDesc{Pb[l], Qb[l], out[l], H, H, H, H / 256};
    PGemmTnDesc* db; CK(hipMalloc(&db, sizeof(PGemmTnDesc) * L)); CK(hipMemcpy(db, hb.data(), sizeof(PGemmTnDesc) * L, hipMemcpyHostToDevice));
    for (int it = 0; it < 2; ++it) launch_pgemm_tn_batched(db, L, rows, U, nullptr, 0, 0, 256, H);
    CK(hipDeviceSynchronize());
    hipEventRecord(e0, 0);
    for (int it = 0; it < reps; ++it) launch_pgemm_tn_batched(db, L, rows, U, nullptr, 0, 0, 256, H);
    hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
    hipEventElapsedTime(&ms, e0, e1);
    printf("pgemm_tn_batched (bf16): %.1f us per layer\n", ms * 1e3 / reps / L);
  }
  return 0;
}
