// Correctness + timing harness of the pipelined GEMMs (tn_pgemm.h) against the generic gemm_nt_kernel (tuning tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/pgemm_harness.hip -o tools/pgemm_harness
//   tools/pgemm_harness [M N K]
#include "../include/titanet_amd.h"
#include "../titanet_amd/csrc/tn_pgemm.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s\n", (int)e, #x); return 1; } } while (0)
template <class F> float timeit(F f, int n = 20) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f(i);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) f(i + 3);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / n;
}
static unsigned short rnd_bf16(uint32_t& s) {
  s = s * 1664525u + 1013904223u;
  const float f = ((int)(s >> 8) - (1 << 23)) / (float)(1 << 23);      // uniform [-1, 1)
  uint32_t u; memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
int main(int argc, char** argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 256 * 300, N = argc > 3 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024;
  const int NSET = 3;
  const int VAR = argc > 4 ? atoi(argv[4]) : 0;
  printf("NT GEMM  M=%d N=%d K=%d  (%.1f GFLOP)\n", M, N, K, 2.0 * M * N * K / 1e9);
  std::vector<unsigned short> ha((size_t)M * K), hw((size_t)N * K);
  uint32_t s = 12345u;
  for (auto& v : ha) v = rnd_bf16(s);
  for (auto& v : hw) v = rnd_bf16(s);
  std::vector<bf16_t*> A(NSET), Y(NSET);
  for (int i = 0; i < NSET; ++i) {
    CK(hipMalloc(&A[i], (size_t)M * K * 2)); CK(hipMalloc(&Y[i], (size_t)(M + 256) * N * 2)); CK(hipMemset(Y[i] + (size_t)M * N, 0x5a, (size_t)256 * N * 2));
    CK(hipMemcpy(A[i], ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  }
  bf16_t *W, *Yref; float *bias, *stats, *stats_ref, *stats_tmp;
  CK(hipMalloc(&stats_tmp, TN_NREP * 2 * 4096 * 4));
  CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&Yref, (size_t)M * N * 2));
  CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&stats, TN_NREP * 2 * N * 4)); CK(hipMalloc(&stats_ref, TN_NREP * 2 * N * 4));
  { std::vector<float> hb(N); for (int i = 0; i < N; ++i) hb[i] = 0.01f * (i % 17) - 0.05f; CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice)); }
  CK(hipMemset(stats, 0, TN_NREP * 2 * N * 4)); CK(hipMemset(stats_ref, 0, TN_NREP * 2 * N * 4));
  GemmShape g{M, N, K, W};
  // ---- reference: the generic kernel
  {
    ProdPlain::Args pa{A[0], K, BnAct{}};
    memset(&pa.act, 0, sizeof(pa.act));
    EpiStoreArgs ea{Yref, N, bias, stats_ref, RowMask{nullptr, 0}, nullptr};
    int rc = launch_gemm<bf16_t, 2, 4, ProdPlain, EpiStore>(g, pa, ea, 0, 0);
    if (rc) { printf("reference launch failed %d\n", rc); return 1; }
  }
  {
    PGemmNtArgs pa{A[0], K};
    PGemmEpiArgs ea{Y[0], N, bias, stats, nullptr};
    int rc = launch_pgemm_nt(g, pa, ea, 0, 256);
    if (rc) { printf("pgemm launch failed %d\n", rc); return 1; }
  }
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 2; ++rep) {
    if (rep == 1) {      // again under load (the timing loop's launches in flight): a race would show here
      CK(hipMemset(stats, 0, TN_NREP * 2 * N * 4));
      PGemmNtArgs pa{A[0], K};
      PGemmEpiArgs ea{Y[0], N, bias, stats, nullptr};
      for (int i = 0; i < 6; ++i) { pa.A = A[i % NSET]; ea.Y = Y[(i + 1) % NSET]; ea.stats = i == 5 ? stats : stats_tmp; if (i == 5) ea.Y = Y[0]; launch_pgemm_nt(g, pa, ea, 0, 256); }
      CK(hipDeviceSynchronize());
      printf("second check (back-to-back launches):\n");
    }
    std::vector<unsigned short> y((size_t)M * N), yr((size_t)M * N);
    CK(hipMemcpy(y.data(), Y[0], y.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(yr.data(), Yref, yr.size() * 2, hipMemcpyDeviceToHost));
    size_t bad = 0, first = (size_t)-1; double maxd = 0;
    for (size_t i = 0; i < y.size(); ++i) {
      uint32_t a = (uint32_t)y[i] << 16, b = (uint32_t)yr[i] << 16; float fa, fb; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
      const double d = fabs((double)fa - fb);
      if (d > maxd) maxd = d;
      if (y[i] != yr[i]) { if (first == (size_t)-1) first = i; ++bad; }
    }
    printf("output: %zu of %zu elements differ bitwise, max abs diff %.4g (first at row %zu col %zu)\n", bad, y.size(), maxd,
           first == (size_t)-1 ? 0 : first / N, first == (size_t)-1 ? 0 : first % N);
    {
      std::vector<unsigned short> can((size_t)256 * N);
      CK(hipMemcpy(can.data(), Y[0] + (size_t)M * N, can.size() * 2, hipMemcpyDeviceToHost));
      size_t hit = 0;
      for (auto c : can) hit += c != 0x5a5a;
      printf("rows beyond M: %zu elements overwritten\n", hit);
    }
    std::vector<float> st(TN_NREP * 2 * N), sr(TN_NREP * 2 * N);
    CK(hipMemcpy(st.data(), stats, st.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(sr.data(), stats_ref, sr.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int which = 0; which < 2; ++which)
      for (int n = 0; n < N; ++n) {
        double a = 0, b = 0;
        for (int r = 0; r < TN_NREP; ++r) { a += st[(r * 2 + which) * N + n]; b += sr[(r * 2 + which) * N + n]; }
        worst = fmax(worst, fabs(a - b) / (fabs(b) + 1.0));
      }
    printf("statistics: worst relative difference of the column sums %.3g\n", worst);
  }
  const double flop = 2.0 * M * N * K;
  {
    ProdPlain::Args pa{A[0], K, BnAct{}};
    memset(&pa.act, 0, sizeof(pa.act));
    EpiStoreArgs ea{Yref, N, bias, stats_ref, RowMask{nullptr, 0}, nullptr};
    float us = timeit([&](int i) { pa.X = A[i % NSET]; ea.Y = Y[i % NSET]; launch_gemm<bf16_t, 2, 4, ProdPlain, EpiStore>(g, pa, ea, 0, 0); });
    printf("gemm_nt_kernel (generic)  : %8.2f us  %.3f PFLOP/s\n", us, flop / us / 1e9);
  }
  if (K == 512 && N % 256 == 0) {
    // the two resident-weight kernels side by side (variant 1: register prefetch, 2: LDS-DMA ring), forward (bias + statistics)
    // and data-gradient (plain) forms; outputs compared bitwise
    for (int epi = 0; epi < 2; ++epi) {
      std::vector<unsigned short> y1((size_t)M * N), y2((size_t)M * N), y3((size_t)M * N), y4((size_t)M * N);
      for (int var = 1; var <= 4; ++var) {
        PGemmNtArgs pa{A[0], K};
        PGemmEpiArgs ea{Y[0], N, epi ? bias : nullptr, epi ? stats_tmp : nullptr, nullptr};
        CK(hipMemset(Y[0], 0, (size_t)M * N * 2));
        int rc = launch_rwgemm_k512(g, pa, ea, 0, 256, var);
        if (rc) { printf("rwgemm variant %d: rc %d\n", var, rc); continue; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy((var == 1 ? y1 : var == 2 ? y2 : var == 3 ? y3 : y4).data(), Y[0], y1.size() * 2, hipMemcpyDeviceToHost));
        float us = timeit([&](int i) { pa.A = A[i % NSET]; ea.Y = Y[i % NSET]; launch_rwgemm_k512(g, pa, ea, 0, 256, var); });
        printf("rwgemm_k512 variant %d %-28s: %8.2f us  %.3f PFLOP/s  %.2f TB/s of its own bytes\n", var, epi ? "(bias + statistics)" : "(plain)", us, flop / us / 1e9,
               ((double)M * K * 2 + (double)M * N * 2) / us / 1e6);
#ifdef RW_STAMPS
        if (var == 2) {
          unsigned long long z[4] = {0, 0, 0, 0}, d[4];
          CK(hipMemcpyToSymbol(HIP_SYMBOL(rw_dbg), z, sizeof(z)));
          launch_rwgemm_k512(g, pa, ea, 0, 256, 2);
          CK(hipDeviceSynchronize());
          CK(hipMemcpyFromSymbol(d, HIP_SYMBOL(rw_dbg), sizeof(d)));
          const double tiles = (double)d[3];      // tile iterations summed over waves
          printf("    stamps per wave and tile (cycles of the 100 MHz counter x ...): wait + barrier %.1f, MFMA loop %.1f, output %.1f  (%.0f wave-tiles)\n",
                 d[0] / tiles, d[1] / tiles, d[2] / tiles, tiles);
        }
#endif
      }
      size_t bad = 0; double maxd = 0, maxv = 0; size_t big = 0;
      for (size_t i = 0; i < y1.size(); ++i) {
        bad += y1[i] != y2[i];
        uint32_t a = (uint32_t)y1[i] << 16, b = (uint32_t)y2[i] << 16; float fa, fb; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
        const double d = fabs((double)fa - fb);
        if (d > maxd) maxd = d;
        if (fabs(fa) > maxv) maxv = fabs(fa);
        big += d > 0.02 * (fabs(fa) + 1.0);
      }
      printf("  variant 2 vs variant 1: %zu of %zu output elements differ, max abs diff %.4g (max |y| %.3g), %zu beyond 2 %%\n", bad, y1.size(), maxd, maxv, big);
      {
        size_t bad3 = 0; double maxd3 = 0;
        for (size_t i = 0; i < y1.size(); ++i) {
          bad3 += y3[i] != y2[i];
          uint32_t a = (uint32_t)y3[i] << 16, b = (uint32_t)y2[i] << 16; float fa, fb; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
          maxd3 = fmax(maxd3, fabs((double)fa - fb));
        }
        printf("  variant 3 (64 columns per wave) vs variant 2: %zu of %zu output elements differ, max abs diff %.4g\n", bad3, y1.size(), maxd3);
        size_t bad4 = 0; double maxd4 = 0;
        for (size_t i = 0; i < y1.size(); ++i) {
          bad4 += y4[i] != y2[i];
          uint32_t a = (uint32_t)y4[i] << 16, b = (uint32_t)y2[i] << 16; float fa, fb; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
          maxd4 = fmax(maxd4, fabs((double)fa - fb));
        }
        printf("  variant 4 (two 256-thread workgroups per CU, 128-column tiles) vs variant 2: %zu of %zu output elements differ, max abs diff %.4g\n", bad4, y1.size(), maxd4);
      }
    }
  }
  for (int wgs : {256, 240}) {
    PGemmNtArgs pa{A[0], K};
    PGemmEpiArgs ea{Y[0], N, bias, stats, nullptr};
    float us = timeit([&](int i) { pa.A = A[i % NSET]; ea.Y = Y[i % NSET]; launch_pgemm_nt(g, pa, ea, 0, wgs); });
    printf("pgemm_nt_kernel (%3d wgs) : %8.2f us  %.3f PFLOP/s\n", wgs, us, flop / us / 1e9);
  }
  for (int wgs : {256, 248, 232}) {
    PGemmNtArgs pa{A[0], K};
    PGemmEpiArgs ea{Y[0], N, bias, stats, nullptr};
    float us = timeit([&](int i) { pa.A = A[i % NSET]; ea.Y = Y[i % NSET]; launch_pgemm_nt_t<0>(g, pa, ea, 0, wgs, false); });
    printf("pgemm_nt_kernel (%3d wgs, uneven rounds) : %8.2f us  %.3f PFLOP/s\n", wgs, us, flop / us / 1e9);
  }
  {
    PGemmNtArgs pa{A[0], K};
    PGemmEpiArgs ea{Y[0], N, bias, stats, nullptr};
    float us;
#define DBGRUN(D, what) us = timeit([&](int i) { pa.A = A[i % NSET]; ea.Y = Y[i % NSET]; launch_pgemm_nt_t<D>(g, pa, ea, 0, 256); }); printf("  dbg %-40s: %8.2f us\n", what, us);
    DBGRUN(1, "no MFMA")
    DBGRUN(2, "no DMA")
    DBGRUN(8, "no stores")
    DBGRUN(9, "no MFMA, no stores")
    DBGRUN(10, "no DMA, no stores")
    DBGRUN(16, "K rotated per workgroup")
    DBGRUN(32, "nt on the A stream")
    DBGRUN(64, "nt on the W stream")
    DBGRUN(48, "K rotated + nt on A")
    DBGRUN(25, "stream only (no MFMA, no stores), K rotated")
    DBGRUN(41, "stream only, nt on A")
  }
  return 0;
}
