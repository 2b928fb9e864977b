// standalone timing harness for the v2 kernels (debug / tuning; not part of the product)
#include "../titanet_amd/csrc/tn_v2_kernels.h"
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s\n", (int)e, #x); return 1; } } while (0)
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256;
  const int grid = argc > 2 ? atoi(argv[2]) : 256;
  const int M = B * 300, T = 300, C = 256;
  bf16_t *X, *W, *Y; float *stats, *gamma, *beta, *bias, *ostats, *wdw, *bdw;
  CK(hipMalloc(&X, (size_t)M * C * 2)); CK(hipMalloc(&Y, (size_t)M * C * 2)); CK(hipMalloc(&W, C * C * 2));
  CK(hipMalloc(&stats, 8 * 2 * C * 4)); CK(hipMalloc(&ostats, 8 * 2 * C * 4));
  CK(hipMalloc(&gamma, C * 4)); CK(hipMalloc(&beta, C * 4)); CK(hipMalloc(&bias, C * 4)); CK(hipMalloc(&wdw, C * 3 * 4)); CK(hipMalloc(&bdw, C * 4));
  { std::vector<unsigned short> hx((size_t)M * C); for (size_t i = 0; i < hx.size(); ++i) hx[i] = (unsigned short)((0x3c00 + (i * 7919u) % 0x300) ^ ((i & 1) << 15)); CK(hipMemcpy(X, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); }
  { std::vector<unsigned short> hw(C * C); for (size_t i = 0; i < hw.size(); ++i) hw[i] = (unsigned short)((0x3a00 + (i * 104729u) % 0x200) ^ ((i & 2) << 14)); CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); }
  std::vector<float> ones(C * 3, 0.3f);
  CK(hipMemcpy(gamma, ones.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wdw, ones.data(), C * 3 * 4, hipMemcpyHostToDevice));
  CK(hipMemset(beta, 0, C * 4)); CK(hipMemset(bias, 0, C * 4)); CK(hipMemset(bdw, 0, C * 4));
  { std::vector<float> hs(8 * 2 * C, 0.f); for (int c = 0; c < C; ++c) { hs[c] = 0.1f * M; hs[C + c] = 1.5f * M; } CK(hipMemcpy(stats, hs.data(), hs.size() * 4, hipMemcpyHostToDevice)); }
  CK(hipMemset(ostats, 0, 8 * 2 * C * 4));
  SubFwdV2Args a; memset(&a, 0, sizeof(a));
  a.X = X; a.W = W; a.Y = Y; a.bias = bias; a.stats = ostats; a.M = M; a.T = T; a.wdw = wdw; a.bdw = bdw;
  a.act.stats = stats; a.act.gamma = gamma; a.act.beta = beta; a.act.inv_n = 1.f / M; a.act.eps = 1e-5f; a.act.mode = 1; a.act.relu = 1;
  a.act.drop_thr = 6554; a.act.drop_key = 12345; a.act.inv_keep = 1.f / 0.9f;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 3; ++it) launch_sub_fwd_v2<3, true>(a, grid, 0);
  CK(hipDeviceSynchronize());
  const int N = 20;
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < N; ++it) launch_sub_fwd_v2<3, true>(a, grid, 0);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("sub_fwd_v2<3,dw> B=%d grid=%d: %.2f us/launch  (%.2f TB/s algorithmic)\n", B, grid, ms * 1e3 / N, 2.0 * M * C * 2 / (ms * 1e-3 / N) / 1e12);
  for (int it = 0; it < 3; ++it) launch_sub_fwd_v4<3, true>(a, grid, 0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < N; ++it) launch_sub_fwd_v4<3, true>(a, grid, 0);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("sub_fwd_v4<3,dw> B=%d grid=%d: %.2f us/launch  (%.2f TB/s algorithmic)\n", B, grid, ms * 1e3 / N, 2.0 * M * C * 2 / (ms * 1e-3 / N) / 1e12);
  for (int it = 0; it < 3; ++it) launch_sub_fwd_v5<3, true>(a, grid, 0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < N; ++it) launch_sub_fwd_v5<3, true>(a, grid, 0);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("sub_fwd_v5<3,dw> B=%d grid=%d: %.2f us/launch  (%.2f TB/s algorithmic)\n", B, grid, ms * 1e3 / N, 2.0 * M * C * 2 / (ms * 1e-3 / N) / 1e12);
  a.act.mode = 0; a.act.relu = 0; a.act.drop_thr = 0;
  for (int it = 0; it < 3; ++it) launch_sub_fwd_v4<1, false>(a, grid, 0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < N; ++it) launch_sub_fwd_v4<1, false>(a, grid, 0);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("sub_fwd_v4<1,plain> identity act B=%d grid=%d: %.2f us/launch\n", B, grid, ms * 1e3 / N);
  for (int it = 0; it < 3; ++it) launch_sub_fwd_v5<1, false>(a, grid, 0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < N; ++it) launch_sub_fwd_v5<1, false>(a, grid, 0);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("sub_fwd_v5<1,plain> identity act B=%d grid=%d: %.2f us/launch\n", B, grid, ms * 1e3 / N);
  for (int it = 0; it < 3; ++it) launch_sub_fwd_v2<1, false>(a, grid, 0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < N; ++it) launch_sub_fwd_v2<1, false>(a, grid, 0);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("sub_fwd_v2<1,plain> identity act B=%d grid=%d: %.2f us/launch\n", B, grid, ms * 1e3 / N);
  return 0;
}
