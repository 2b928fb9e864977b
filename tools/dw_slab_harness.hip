// Timing harness for the depthwise slab kernels of the wide models (tuning tool): dw_fwd_slab / dw_bwd_slab at the
// TitaNet-L (C 1024, K 11) and TitaNet-M (C 512, K 7) layer shapes, batch 256 x 300 frames.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/dw_slab_harness.hip -o tools/dw_slab_harness
#include "../titanet_amd/csrc/tn_fwd_kernels.h"
#include "../titanet_amd/csrc/tn_v2_bwd_kernels.h"
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
int main(int argc, char** argv) {
  const int M = 256 * 300, T = 300, CMAX = 1024, NSET = 3;
  const int BWGS = argc > 1 ? atoi(argv[1]) : 256;      // persistent workgroups of the backward kernel (forward: TN_DWF_WGS)
  std::vector<bf16_t*> D(NSET), X(NSET), O(NSET), ADD(NSET);
  std::vector<unsigned short> hx((size_t)M * CMAX);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (unsigned short)((0x3c00 + (i * 7919u) % 0x300) ^ ((i & 1) << 15));
  for (int s = 0; s < NSET; ++s) {
    CK(hipMalloc(&D[s], hx.size() * 2)); CK(hipMalloc(&X[s], hx.size() * 2)); CK(hipMalloc(&O[s], hx.size() * 2)); CK(hipMalloc(&ADD[s], hx.size() * 2));
    CK(hipMemcpy(D[s], hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(X[s], hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(ADD[s], hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  }
  float *stats, *bs, *gamma, *beta, *wdw, *bdw, *gw, *gb;
  CK(hipMalloc(&stats, 8 * 2 * CMAX * 4)); CK(hipMalloc(&bs, 8 * 2 * CMAX * 4)); CK(hipMalloc(&gamma, CMAX * 4)); CK(hipMalloc(&beta, CMAX * 4));
  CK(hipMalloc(&wdw, CMAX * 16 * 4)); CK(hipMalloc(&bdw, CMAX * 4)); CK(hipMalloc(&gw, CMAX * 16 * 4)); CK(hipMalloc(&gb, CMAX * 4));
  std::vector<float> ones(CMAX * 16, 0.3f);
  CK(hipMemcpy(gamma, ones.data(), CMAX * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(beta, ones.data(), CMAX * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wdw, ones.data(), CMAX * 16 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bdw, ones.data(), CMAX * 4, hipMemcpyHostToDevice));
  CK(hipMemset(bs, 0, 8 * 2 * CMAX * 4)); CK(hipMemset(gw, 0, CMAX * 16 * 4)); CK(hipMemset(gb, 0, CMAX * 4));
  for (int cfg = 0; cfg < 2; ++cfg) {
    const int C = cfg == 0 ? 1024 : 512, K = cfg == 0 ? 11 : 7;
    { std::vector<float> hs(8 * 2 * C, 0.f); for (int c = 0; c < C; ++c) { hs[c] = 0.1f * M; hs[C + c] = 1.5f * M; } CK(hipMemcpy(stats, hs.data(), hs.size() * 4, hipMemcpyHostToDevice)); }
    BnAct act; memset(&act, 0, sizeof(act));
    act.mode = 1; act.stats = stats; act.gamma = gamma; act.beta = beta; act.inv_n = 1.f / M; act.eps = 1e-5f; act.relu = 1;
    act.drop_thr = 6554; act.drop_key = 12345u; act.inv_keep = 1.f / 0.9f;
    const double t = (double)M * C * 2;
    {
      DwFwdSlabArgs a; memset(&a, 0, sizeof(a));
      a.act = act; a.wdw = wdw; a.bdw = bdw; a.M = M; a.T = T; a.C = C;
      int rc = 0;
      float us = timeit([&](int i) { a.X = X[i % NSET]; a.Q = O[i % NSET]; rc |= launch_dw_fwd_slab(a, K, 0); });
      printf("C %4d K %2d dw_fwd_slab<BN relu drop>   : %7.2f us  (2t = %.0f MB -> %.2f TB/s) rc %d\n", C, K, us, 2 * t / 1e6, 2 * t / us / 1e6, rc);
    }
    {
      DwBwdSlabArgs a; memset(&a, 0, sizeof(a));
      a.actX = act; a.wdw = wdw; a.g_wdw = gw; a.g_bdw = gb; a.bsumsX = bs; a.M = M; a.T = T; a.C = C;
      int rc = 0;
      float us = timeit([&](int i) { a.dD = D[i % NSET]; a.X = X[i % NSET]; a.OUT = O[i % NSET];
                                     rc |= K == 7 ? launch_dw_bwd_slab<7>(a, BWGS, 0) : launch_dw_bwd_slab<11>(a, BWGS, 0); });
      printf("C %4d K %2d dw_bwd_slab<BN relu drop>   : %7.2f us  (3t = %.0f MB -> %.2f TB/s) rc %d\n", C, K, us, 3 * t / 1e6, 3 * t / us / 1e6, rc);
    }
  }
  return 0;
}
