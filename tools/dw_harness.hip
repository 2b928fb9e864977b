// standalone timing harness for dw_bwd_v3_kernel / dw_bwd_v4_kernel (tuning only)
#include "../titanet_amd/csrc/tn_v2_bwd_kernels.h"
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s\n", (int)e, #x); return 1; } } while (0)
int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 512;
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  const int M = 256 * 300, T = 300, C = 256;
  bf16_t *dD, *X, *ADD, *OUT; float *stats, *gamma, *beta, *wdw, *gw, *gb, *bs;
  CK(hipMalloc(&dD, (size_t)M * C * 2)); CK(hipMalloc(&X, (size_t)M * C * 2)); CK(hipMalloc(&ADD, (size_t)M * C * 2)); CK(hipMalloc(&OUT, (size_t)M * C * 2));
  CK(hipMalloc(&stats, 8 * 2 * C * 4)); CK(hipMalloc(&gamma, C * 4)); CK(hipMalloc(&beta, C * 4)); CK(hipMalloc(&wdw, C * 3 * 4));
  CK(hipMalloc(&gw, C * 3 * 4)); CK(hipMalloc(&gb, C * 4)); CK(hipMalloc(&bs, 8 * 2 * C * 4)); float* gacc; CK(hipMalloc(&gacc, 8 * 4 * C * 4)); CK(hipMemset(gacc, 0, 8 * 4 * C * 4));
  { std::vector<unsigned short> hx((size_t)M * C); for (size_t i = 0; i < hx.size(); ++i) hx[i] = (unsigned short)((0x3c00 + (i * 7919u) % 0x300) ^ ((i & 1) << 15));
    CK(hipMemcpy(X, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dD, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(ADD, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); }
  std::vector<float> ones(C * 3, 0.3f);
  CK(hipMemcpy(gamma, ones.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wdw, ones.data(), C * 3 * 4, hipMemcpyHostToDevice)); CK(hipMemset(beta, 0, C * 4));
  { std::vector<float> hs(8 * 2 * C, 0.f); for (int c = 0; c < C; ++c) { hs[c] = 0.1f * M; hs[C + c] = 1.5f * M; } CK(hipMemcpy(stats, hs.data(), hs.size() * 4, hipMemcpyHostToDevice)); }
  CK(hipMemset(gw, 0, C * 3 * 4)); CK(hipMemset(gb, 0, C * 4)); CK(hipMemset(bs, 0, 8 * 2 * C * 4));
  DwBwdV3Args a; memset(&a, 0, sizeof(a));
  a.dD = dD; a.X = X; a.ADD = nullptr; a.OUT = OUT; a.wdw = wdw; a.gacc = gacc; a.bsumsX = bs; a.M = M; a.T = T;
  a.actX.stats = stats; a.actX.gamma = gamma; a.actX.beta = beta; a.actX.inv_n = 1.f / M; a.actX.eps = 1e-5f; a.actX.mode = 1; a.actX.relu = 1;
  a.actX.drop_thr = 6554; a.actX.drop_key = 12345; a.actX.inv_keep = 1.f / 0.9f;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 3; ++it) launch_dw_bwd_v3<3>(a, grid, 0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < iters; ++it) launch_dw_bwd_v3<3>(a, grid, 0);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("dw_bwd_v3 grid=%d: %.2f us/launch (%.2f TB/s algorithmic 3t)\n", grid, ms * 1e3 / iters, 3.0 * M * C * 2 / (ms * 1e-3 / iters) / 1e12);
  for (int it = 0; it < 3; ++it) launch_dw_bwd_v4<3>(a, grid, 0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < iters; ++it) launch_dw_bwd_v4<3>(a, grid, 0);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("dw_bwd_v4 grid=%d: %.2f us/launch (%.2f TB/s algorithmic 3t)\n", grid, ms * 1e3 / iters, 3.0 * M * C * 2 / (ms * 1e-3 / iters) / 1e12);
  a.ADD = ADD; a.actX.mode = 0; a.actX.relu = 0; a.actX.drop_thr = 0;
  for (int it = 0; it < 3; ++it) launch_dw_bwd_v4<3>(a, grid, 0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < iters; ++it) launch_dw_bwd_v4<3>(a, grid, 0);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("dw_bwd_v4 (first sub-block: +ADD, identity act) grid=%d: %.2f us/launch\n", grid, ms * 1e3 / iters);
  return 0;
}
