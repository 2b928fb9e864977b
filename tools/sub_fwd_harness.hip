// Timing harness for sub_fwd_v5 (tuning tool): 64-row vs 32-row tiles, with / without the kept depthwise output; rotating
// buffer sets (cold).  build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/sub_fwd_harness.hip -o tools/sub_fwd_harness
#include "../titanet_amd/csrc/tn_v2_kernels.h"
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s\n", (int)e, #x); return 1; } } while (0)
int main(int argc, char** argv) {
  const int M = 256 * 300, C = 256, T = 300, NSET = 8;
  std::vector<bf16_t*> X(NSET), Y(NSET), Q(NSET);
  bf16_t* W; float *stats, *ostats, *gamma, *beta, *wdw, *bdw, *bias;
  std::vector<unsigned short> hx((size_t)M * C);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (unsigned short)((0x3c00 + (i * 7919u) % 0x300) ^ ((i & 1) << 15));
  for (int s = 0; s < NSET; ++s) {
    CK(hipMalloc(&X[s], (size_t)M * C * 2)); CK(hipMalloc(&Y[s], (size_t)M * C * 2)); CK(hipMalloc(&Q[s], (size_t)M * C * 2));
    CK(hipMemcpy(X[s], hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&W, C * C * 2)); CK(hipMemcpy(W, hx.data(), C * C * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&stats, 8 * 2 * C * 4)); CK(hipMalloc(&ostats, 8 * 2 * C * 4)); CK(hipMalloc(&gamma, C * 4)); CK(hipMalloc(&beta, C * 4));
  CK(hipMalloc(&wdw, C * 3 * 4)); CK(hipMalloc(&bdw, C * 4)); CK(hipMalloc(&bias, C * 4));
  std::vector<float> ones(C * 3, 0.3f);
  CK(hipMemcpy(gamma, ones.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(beta, ones.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wdw, ones.data(), C * 3 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bdw, ones.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, ones.data(), C * 4, hipMemcpyHostToDevice));
  { std::vector<float> hs(8 * 2 * C, 0.f); for (int c = 0; c < C; ++c) { hs[c] = 0.1f * M; hs[C + c] = 1.5f * M; } CK(hipMemcpy(stats, hs.data(), hs.size() * 4, hipMemcpyHostToDevice)); }
  CK(hipMemset(ostats, 0, 8 * 2 * C * 4));
  uint4* swz; CK(hipMalloc(&swz, C * C * 2));
  { SwzDesc hd{W, swz, C, C}; SwzDesc* dd; CK(hipMalloc(&dd, sizeof(hd))); CK(hipMemcpy(dd, &hd, sizeof(hd), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(swizzle256_kernel<0>, dim3(16, 1), dim3(256), 0, 0, dd); }
  SubFwdV2Args a; memset(&a, 0, sizeof(a));
  a.act.mode = 1; a.act.stats = stats; a.act.gamma = gamma; a.act.beta = beta; a.act.inv_n = 1.f / M; a.act.eps = 1e-5f; a.act.relu = 1;
  a.act.drop_thr = 6554; a.act.inv_keep = 1.f / 0.9f; a.act.drop_key = 12345;
  a.wdw = wdw; a.bdw = bdw; a.W = W; a.bias = bias; a.stats = ostats; a.M = M; a.T = T; a.Wswz = swz;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  if (argc > 1 && !strcmp(argv[1], "sweep")) {
    // fixed cost per launch: the shipped form (32-row tiles, depthwise output kept) over the row count, down to one tile per launch
    for (int m : {32, 2400, 9600, 19200, 38400, 76800}) {
      a.M = m;
      auto go = [&](int it) { const int s = it % NSET; a.X = X[s]; a.Y = Y[s]; a.Q = Q[s]; return launch_sub_fwd_v5<3, true, 32>(a, 256, 0); };
      for (int it = 0; it < 4; ++it) go(it);
      CK(hipDeviceSynchronize());
      hipEventRecord(e0, 0);
      for (int it = 0; it < 40; ++it) go(it);
      hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("sub_fwd_v5<3,true,7,32> M = %6d rows: %.2f us per launch\n", m, ms * 1e3f / 40);
    }
    return 0;
  }
  for (int rep = 0; rep < 2; ++rep)
    for (int variant = 0; variant < 4; ++variant) {
      const bool r32 = variant & 1, keepq = !(variant & 2);
      auto go = [&](int it) { const int s = it % NSET; a.X = X[s]; a.Y = Y[s]; a.Q = keepq ? Q[s] : nullptr;
                              return r32 ? launch_sub_fwd_v5<3, true, 32>(a, 256, 0) : launch_sub_fwd_v5<3, true, 64>(a, 256, 0); };
      for (int it = 0; it < 4; ++it) { int rc = go(it); if (rc) { printf("launch rc %d\n", rc); return 1; } }
      CK(hipDeviceSynchronize());
      hipEventRecord(e0, 0);
      for (int it = 0; it < 40; ++it) go(it);
      hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("sub_fwd_v5<3,true,7> rows/tile %d, depthwise output %s: %.2f us per launch\n", r32 ? 32 : 64, keepq ? "kept" : "not kept", ms * 1e3f / 40);
    }
  // the skip conv (plain 1x1 on a stored block input): sub_fwd_v4 vs the producer / consumer kernel at both tile heights
  {
    SubFwdV2Args b = a; b.act = BnAct{}; b.wdw = nullptr; b.bdw = nullptr; b.Q = nullptr;
    for (int rep = 0; rep < 2; ++rep)
      for (int variant = 0; variant < 3; ++variant) {
        auto go = [&](int it) { const int s = it % NSET; b.X = X[s]; b.Y = Y[s];
                                return variant == 0 ? launch_sub_fwd_v4<1, false>(b, 256, 0) : variant == 1 ? launch_sub_fwd_v5<1, false, 64>(b, 256, 0) : launch_sub_fwd_v5<1, false, 32>(b, 256, 0); };
        for (int it = 0; it < 4; ++it) { int rc = go(it); if (rc) { printf("launch rc %d\n", rc); return 1; } }
        CK(hipDeviceSynchronize());
        hipEventRecord(e0, 0);
        for (int it = 0; it < 40; ++it) go(it);
        hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("skip conv 1x1: %s: %.2f us per launch\n", variant == 0 ? "sub_fwd_v4" : variant == 1 ? "sub_fwd_v5, 64-row tiles" : "sub_fwd_v5, 32-row tiles", ms * 1e3f / 40);
      }
  }
  // same result from both tile shapes?
  a.X = X[0]; a.Q = Q[0]; a.Y = Y[0]; launch_sub_fwd_v5<3, true, 64>(a, 256, 0);
  a.Q = Q[1]; a.Y = Y[1]; launch_sub_fwd_v5<3, true, 32>(a, 256, 0);
  CK(hipDeviceSynchronize());
  std::vector<unsigned short> o0((size_t)M * C), o1((size_t)M * C);
  CK(hipMemcpy(o0.data(), Y[0], o0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), Y[1], o1.size() * 2, hipMemcpyDeviceToHost));
  size_t diff = 0; for (size_t i = 0; i < o0.size(); ++i) diff += o0[i] != o1[i];
  CK(hipMemcpy(o0.data(), Q[0], o0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), Q[1], o1.size() * 2, hipMemcpyDeviceToHost));
  size_t diffq = 0; for (size_t i = 0; i < o0.size(); ++i) diffq += o0[i] != o1[i];
  printf("64-row vs 32-row tiles: %zu outputs, %zu kept depthwise outputs differ (of %zu)\n", diff, diffq, o0.size());
  return 0;
}
