// Timing harness for the 1536-wide decoder-side kernels (tuning tool): wide_in_v2<0/1>, wide_out_v2<256,0>, <128,0>, <128,2>.
#include "../titanet_amd/csrc/tn_v2_wide_kernels.h"
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
  const int M = 256 * 300, D = 1536, H = 256, A = 128, NSET = 3;
  std::vector<bf16_t*> E(NSET), E2(NSET), X(NSET), HID(NSET), O(NSET);
  std::vector<unsigned short> hx((size_t)M * D);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (unsigned short)((0x3c00 + (i * 7919u) % 0x300) ^ ((i & 1) << 15));
  for (int s = 0; s < NSET; ++s) {
    CK(hipMalloc(&E[s], (size_t)M * D * 2)); CK(hipMalloc(&E2[s], (size_t)M * D * 2)); CK(hipMalloc(&X[s], (size_t)M * H * 2));
    CK(hipMalloc(&HID[s], (size_t)M * A * 2 + 512)); CK(hipMalloc(&O[s], (size_t)M * H * 2));
    CK(hipMemcpy(E[s], hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(E2[s], hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(X[s], hx.data(), (size_t)M * H * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(HID[s], hx.data(), (size_t)M * A * 2, hipMemcpyHostToDevice));
  }
  bf16_t* W; CK(hipMalloc(&W, (size_t)D * H * 2)); CK(hipMemcpy(W, hx.data(), (size_t)D * H * 2, hipMemcpyHostToDevice));
  float *stats, *bs, *gamma, *beta, *bias, *colsum;
  CK(hipMalloc(&stats, 8 * 2 * D * 4)); CK(hipMalloc(&bs, 8 * 2 * D * 4)); CK(hipMalloc(&gamma, D * 4)); CK(hipMalloc(&beta, D * 4)); CK(hipMalloc(&bias, D * 4)); CK(hipMalloc(&colsum, D * 4));
  std::vector<float> ones(D, 0.3f); CK(hipMemcpy(gamma, ones.data(), D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(beta, ones.data(), D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, ones.data(), D * 4, hipMemcpyHostToDevice));
  { std::vector<float> hs(8 * 2 * D, 0.f); for (int c = 0; c < D; ++c) { hs[c] = 0.1f * M; hs[D + c] = 1.5f * M; } CK(hipMemcpy(stats, hs.data(), hs.size() * 4, hipMemcpyHostToDevice)); }
  CK(hipMemset(bs, 0, 8 * 2 * D * 4)); CK(hipMemset(colsum, 0, D * 4));
  BnAct act; memset(&act, 0, sizeof(act));
  act.mode = 1; act.stats = stats; act.gamma = gamma; act.beta = beta; act.inv_n = 1.f / M; act.eps = 1e-5f; act.relu = 1;
  const double t = (double)M * H * 2;     // one 256-wide tensor
  {
    WideInArgs a; memset(&a, 0, sizeof(a));
    a.act = act; a.W = W; a.bias = bias; a.M = M; a.KW = D;
    float us = timeit([&](int i) { a.A = E[i % NSET]; a.Y = HID[i % NSET]; launch_wide_in_v2<0>(a, 256, 0); });
    printf("wide_in_v2<0>      : %7.2f us  (6.5t = %.0f MB -> %.2f TB/s)\n", us, 6.5 * t / 1e6, 6.5 * t / us / 1e6);
    a.act = BnAct{}; a.H = HID[0]; a.colsum = colsum; a.bias = nullptr;
    us = timeit([&](int i) { a.A = E[i % NSET]; a.Y = HID[(i + 1) % NSET]; a.H = HID[i % NSET]; launch_wide_in_v2<1>(a, 256, 0); });
    printf("wide_in_v2<1>      : %7.2f us  (7t = %.0f MB -> %.2f TB/s)\n", us, 7 * t / 1e6, 7 * t / us / 1e6);
  }
  {
    WideOutArgs a; memset(&a, 0, sizeof(a));
    a.W = W; a.bias = bias; a.stats = bs; a.M = M; a.N = D;
    float us = timeit([&](int i) { a.X = X[i % NSET]; a.Y = E[i % NSET]; launch_wide_out_v2<256, 0>(a, 256, 0); });
    printf("wide_out_v2<256,0> : %7.2f us  (7t = %.0f MB -> %.2f TB/s)\n", us, 7 * t / 1e6, 7 * t / us / 1e6);
    a.stats = nullptr;
    us = timeit([&](int i) { a.X = HID[i % NSET]; a.Y = E[i % NSET]; launch_wide_out_v2<128, 0>(a, 256, 0); });
    printf("wide_out_v2<128,0> : %7.2f us  (6.5t = %.0f MB -> %.2f TB/s)\n", us, 6.5 * t / 1e6, 6.5 * t / us / 1e6);
    a.RAW = E2[0]; a.actR = act; a.bsums = bs; a.bias = nullptr;
    us = timeit([&](int i) { a.X = HID[i % NSET]; a.Y = E[i % NSET]; a.RAW = E2[i % NSET]; launch_wide_out_v2<128, 2>(a, 256, 0); });
    printf("wide_out_v2<128,2> : %7.2f us  (18.5t = %.0f MB -> %.2f TB/s)\n", us, 18.5 * t / 1e6, 18.5 * t / us / 1e6);
  }
  return 0;
}
