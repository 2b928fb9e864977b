// Timing harness for dgrad_dw_v6 (tuning tool): rotating buffer sets (cold).
#include "../titanet_amd/csrc/tn_v2_bwd_kernels.h"
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s\n", (int)e, #x); return 1; } } while (0)
int main(int argc, char** argv) {
  const int M = 256 * 300, C = 256, T = 300, NSET = 8;
  std::vector<bf16_t*> dZ(NSET), Y(NSET), X(NSET), OUT(NSET);
  bf16_t* W; float *stats, *bs, *gamma, *beta, *wdw, *gacc, *bsx;
  std::vector<unsigned short> hx((size_t)M * C);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (unsigned short)((0x3c00 + (i * 7919u) % 0x300) ^ ((i & 1) << 15));
  for (int s = 0; s < NSET; ++s) {
    CK(hipMalloc(&dZ[s], (size_t)M * C * 2)); CK(hipMalloc(&Y[s], (size_t)M * C * 2)); CK(hipMalloc(&X[s], (size_t)M * C * 2)); CK(hipMalloc(&OUT[s], (size_t)M * C * 2));
    CK(hipMemcpy(dZ[s], hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(Y[s], hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(X[s], hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&W, C * C * 2)); CK(hipMemcpy(W, hx.data(), C * C * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&stats, 8 * 2 * C * 4)); CK(hipMalloc(&bs, 8 * 2 * C * 4)); CK(hipMalloc(&gamma, C * 4)); CK(hipMalloc(&beta, C * 4));
  CK(hipMalloc(&wdw, C * 3 * 4)); CK(hipMalloc(&gacc, 8 * 4 * C * 4)); CK(hipMalloc(&bsx, 8 * 2 * C * 4));
  std::vector<float> ones(C * 3, 0.3f); CK(hipMemcpy(gamma, ones.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(beta, ones.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wdw, ones.data(), C * 3 * 4, hipMemcpyHostToDevice));
  { std::vector<float> hs(8 * 2 * C, 0.f); for (int c = 0; c < C; ++c) { hs[c] = 0.1f * M; hs[C + c] = 1.5f * M; } CK(hipMemcpy(stats, hs.data(), hs.size() * 4, hipMemcpyHostToDevice)); }
  CK(hipMemset(bs, 0, 8 * 2 * C * 4)); CK(hipMemset(gacc, 0, 8 * 4 * C * 4)); CK(hipMemset(bsx, 0, 8 * 2 * C * 4));
  uint4* swz; CK(hipMalloc(&swz, C * C * 2));
  { SwzDesc hd{W, swz, C, C}; SwzDesc* dd; CK(hipMalloc(&dd, sizeof(hd))); CK(hipMemcpy(dd, &hd, sizeof(hd), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(swizzle256_kernel<0>, dim3(16, 1), dim3(256), 0, 0, dd); }
  DgradDwArgs a; memset(&a, 0, sizeof(a));
  a.bn.fstats = stats; a.bn.bsums = bs; a.bn.gamma = gamma; a.bn.inv_n = 1.f / M; a.bn.eps = 1e-5f; a.bn.batch = 1.f;
  a.Wswz = swz; a.wdw = wdw; a.gacc = gacc; a.bsumsX = bsx; a.M = M; a.T = T;
  a.actX.mode = 1; a.actX.stats = stats; a.actX.gamma = gamma; a.actX.beta = beta; a.actX.inv_n = 1.f / M; a.actX.eps = 1e-5f; a.actX.relu = 1;
  a.actX.drop_thr = 6554; a.actX.inv_keep = 1.f / 0.9f; a.actX.drop_key = 12345;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 4; ++which) {
    const bool p2 = which & 1;
    auto go = [&](int it) { const int s = it % NSET; a.dZ = dZ[s]; a.Y = Y[s]; a.X = X[s]; a.OUT = OUT[s];
                            return launch_dgrad_dw_v6(a, 256, 0, p2); };
    for (int it = 0; it < 4; ++it) { int rc = go(it); if (rc) { printf("launch rc %d\n", rc); return 1; } }
    CK(hipDeviceSynchronize());
    hipEventRecord(e0, 0);
    for (int it = 0; it < 40; ++it) go(it);
    hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s<7> %s: %.2f us per launch (%.2f TB/s over 4 passes)\n", "dgrad_dw_v6", p2 ? "two barriers per tile" : "three barriers per tile", ms * 1e3f / 40, 4.0 * M * C * 2 / (ms * 1e-3 / 40) / 1e12);
  }
  // round 5: the matrix-pipe share of a fused pointwise weight gradient (WGX probe: +16 MFMAs per wave and tile + their
  // transposing fragment reads, no accumulator flush, no depthwise-output recompute) against the plain kernel, alternating
  {
    const size_t tiles = (size_t)(2 * V6_R * V2_AP + V6_R * V2_C) * sizeof(bf16_t);
    const size_t red = (size_t)8 * 6 * V2_C * sizeof(float);
    const size_t smem = (tiles > red ? tiles : red) + (size_t)10 * V2_C * sizeof(float);
    a.ntiles = (a.M + V6_OUT - 1) / V6_OUT;
    typedef void (*kern_t)(DgradDwArgs);
    kern_t ks_[5] = {dgrad_dw_v6_kernel<7, false, false, false, 0, 0>, dgrad_dw_v6_kernel<7, false, false, false, 0, 4>,
                     dgrad_dw_v6_kernel<7, false, false, false, 0, 3>, dgrad_dw_v6_kernel<7, false, false, false, 0, 6>,
                     dgrad_dw_v6_kernel<7, false, false, false, 1, 0>};
    const char* names[5] = {"compiler-scheduled MFMA phase", "hand-scheduled, 4 fragment reads in flight", "hand-scheduled, 3 in flight",
                            "hand-scheduled, 6 in flight", "+ 16 weight-gradient MFMAs per wave and tile (WGX probe)"};
    for (auto k : ks_) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    for (int rep = 0; rep < 3; ++rep)
      for (int which = 0; which < 5; ++which) {
        auto go = [&](int it) { const int s = it % NSET; a.dZ = dZ[s]; a.Y = Y[s]; a.X = X[s]; a.OUT = OUT[s];
                                hipLaunchKernelGGL(ks_[which], dim3(256), dim3(V2_NT), smem, 0, a); };
        for (int it = 0; it < 4; ++it) go(it);
        CK(hipDeviceSynchronize());
        hipEventRecord(e0, 0);
        for (int it = 0; it < 51; ++it) go(it);
        hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("dgrad_dw_v6<7> %s: %.2f us per launch\n", names[which], ms * 1e3f / 51);
      }
    // same inputs, compiler-scheduled vs hand-scheduled: bitwise (same MFMA order)
    {
      std::vector<unsigned short> o0((size_t)M * C), o1((size_t)M * C);
      a.dZ = dZ[0]; a.Y = Y[0]; a.X = X[0]; a.OUT = OUT[2]; hipLaunchKernelGGL(ks_[0], dim3(256), dim3(V2_NT), smem, 0, a);
      a.OUT = OUT[3]; hipLaunchKernelGGL(ks_[1], dim3(256), dim3(V2_NT), smem, 0, a);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(o0.data(), OUT[2], o0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), OUT[3], o1.size() * 2, hipMemcpyDeviceToHost));
      size_t diff = 0; for (size_t i = 0; i < o0.size(); ++i) diff += o0[i] != o1[i];
      printf("hand-scheduled vs compiler-scheduled MFMA phase on the same inputs: %zu of %zu elements differ\n", diff, o0.size());
    }
  }
  // checksum of the two outputs on the same inputs
  a.dZ = dZ[0]; a.Y = Y[0]; a.X = X[0]; a.OUT = OUT[0]; launch_dgrad_dw_v6(a, 256, 0, false);
  a.OUT = OUT[1]; launch_dgrad_dw_v6(a, 256, 0, true);
  CK(hipDeviceSynchronize());
  std::vector<unsigned short> o0((size_t)M * C), o1((size_t)M * C);
  CK(hipMemcpy(o0.data(), OUT[0], o0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), OUT[1], o1.size() * 2, hipMemcpyDeviceToHost));
  size_t diff = 0; for (size_t i = 0; i < o0.size(); ++i) diff += o0[i] != o1[i];
  printf("three-barrier vs two-barrier schedule on the same inputs: %zu of %zu elements differ\n", diff, o0.size());
  return 0;
}
