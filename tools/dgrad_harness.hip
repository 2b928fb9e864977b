#include "../titanet_amd/csrc/tn_v2_bwd_kernels.h"
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s\n", (int)e, #x); return 1; } } while (0)
template <int R> float run(DgradV2Args a, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) launch_dgrad_v2<R>(a, grid, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int it = 0; it < 20; ++it) launch_dgrad_v2<R>(a, grid, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / 20;
}
int main(int argc, char** argv) {
  const int M = 256 * 300, C = 256;
  bf16_t *dZ, *Y, *W, *OUT; float *stats, *bs, *gamma;
  CK(hipMalloc(&dZ, (size_t)M * C * 2)); CK(hipMalloc(&Y, (size_t)M * C * 2)); CK(hipMalloc(&OUT, (size_t)M * C * 2)); CK(hipMalloc(&W, C * C * 2));
  CK(hipMalloc(&stats, 8 * 2 * C * 4)); CK(hipMalloc(&bs, 8 * 2 * C * 4)); CK(hipMalloc(&gamma, C * 4));
  { std::vector<unsigned short> hx((size_t)M * C); for (size_t i = 0; i < hx.size(); ++i) hx[i] = (unsigned short)((0x3c00 + (i * 7919u) % 0x300) ^ ((i & 1) << 15));
    CK(hipMemcpy(dZ, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(Y, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hx.data(), C * C * 2, hipMemcpyHostToDevice)); }
  std::vector<float> ones(C, 0.3f); CK(hipMemcpy(gamma, ones.data(), C * 4, hipMemcpyHostToDevice));
  { std::vector<float> hs(8 * 2 * C, 0.f); for (int c = 0; c < C; ++c) { hs[c] = 0.1f * M; hs[C + c] = 1.5f * M; } CK(hipMemcpy(stats, hs.data(), hs.size() * 4, hipMemcpyHostToDevice)); }
  CK(hipMemset(bs, 0, 8 * 2 * C * 4));
  DgradV2Args a; memset(&a, 0, sizeof(a));
  a.dZ = dZ; a.Y = Y; a.Wt = W; a.OUT = OUT; a.M = M; a.bn.fstats = stats; a.bn.bsums = bs; a.bn.gamma = gamma; a.bn.inv_n = 1.f / M; a.bn.eps = 1e-5f; a.bn.batch = 1.f;
  for (int g : {256, 512, 768, 1024}) printf("dgrad_v2<32> grid=%4d: %.2f us\n", g, run<32>(a, g));
  for (int g : {256, 512}) printf("dgrad_v2<64> grid=%4d: %.2f us\n", g, run<64>(a, g));
  uint4* swz; CK(hipMalloc(&swz, C * C * 2));
  hipLaunchKernelGGL(dgrad_wide_swizzle_kernel, dim3(32), dim3(256), 0, 0, W, C, swz);
  a.Wswz = swz;
  for (int g : {256}) printf("dgrad_v2<64> fragment-ordered weights grid=%4d: %.2f us\n", g, run<64>(a, g));
  for (int g : {256, 512, 768}) printf("dgrad_v2<32> fragment-ordered weights grid=%4d: %.2f us\n", g, run<32>(a, g));
  a.Wswz = nullptr;
  for (int g : {256}) printf("dgrad_v2<64> row-major weights grid=%4d: %.2f us\n", g, run<64>(a, g));
  return 0;
}
