// One-block (and N-block) harness of the utterance-resident forward chain (tools/chain_fwd.h) against the per-layer launches it
// replaces (sub_fwd_v4 skip conv + 3 x sub_fwd_v5 + se_combine_fwd_v3), headline shape: batch 256 x 300 frames x 256 channels.
//   tools/chain_harness [nblocks=1] [reps=20] [B=256] [T=300] [drop=1]
// Prints: per-block time of both paths (cold: NSET rotating buffer sets), the chain's per-phase wall-clock breakdown of
// workgroup 0, the grid barrier alone (probe kernel: statistics atomics + barrier, nothing else), and the comparison of every
// tensor backward reads (S, Y1..3, Q1..3, OUT, SE mean / hidden / gate, BatchNorm sums).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "chain_fwd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) at %s\n", (int)e_, hipGetErrorString(e_), #x); return 1; } } while (0)

static uint32_t rng_state = 12345u;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return (float)(rng_state >> 8) * (1.f / 16777216.f); }
static unsigned short f2bf_host(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float bf2f_host(unsigned short h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// the barrier alone: every workgroup adds 512 floats to a statistics buffer (as a pass's epilogue does), then meets the others
__global__ __launch_bounds__(512) void barrier_probe_kernel(unsigned* bar, float* stats, int nbar, int B, unsigned long long* stamps) {
  __shared__ unsigned flag;
  const int tid = threadIdx.x, b = blockIdx.x;
  for (int e = 1; e <= nbar; ++e) {
    if (b == 0 && tid == 0 && stamps) stamps[e - 1] = wall_clock64();
    atomic_add_f32(&stats[(size_t)(e - 1) * 4096 + (size_t)((b % TN_NREP) * 2 + (tid >> 8)) * 256 + (tid & 255)], 1.f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) flag = ch_grid_barrier(bar, (unsigned)e, b, B) ? 1u : 0u;
    __syncthreads();
    if (!flag) return;
    // what every workgroup does behind the barrier: read the sums of its channel
    float s = 0.f;
    for (int r = 0; r < TN_NREP; ++r) s += stats[(size_t)(e - 1) * 4096 + (size_t)(r * 2 + (tid >> 8)) * 256 + (tid & 255)];
    if (s != (float)B && b == 0 && stamps) stamps[nbar + 1] = 0xdeadull;      // a stale read
  }
  if (b == 0 && tid == 0 && stamps) stamps[nbar] = wall_clock64();
}

struct LayerBufs { float *bias, *gamma, *beta, *wdw, *bdw, *stats[2]; bf16_t* W; uint4* Wswz; bf16_t *Y[2], *Q[2]; };
struct BlockBufs { LayerBufs skip, sub[3]; float *w1, *w2, *m[2], *h[2], *g[2]; bf16_t* OUT[2]; };

int main(int argc, char** argv) {
  const int NB = argc > 1 ? atoi(argv[1]) : 1, REPS = argc > 2 ? atoi(argv[2]) : 20;
  const int B = argc > 3 ? atoi(argv[3]) : 256, T = argc > 4 ? atoi(argv[4]) : 300;
  const bool drop = argc > 5 ? atoi(argv[5]) != 0 : true;
  const int C = 256, M = B * T;
  const size_t tsz = (size_t)M * C * 2;
  printf("chain harness: %d block(s), B = %d, T = %d, dropout %s, reps %d\n", NB, B, T, drop ? "0.1" : "off", REPS);
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  if (B > prop.multiProcessorCount) { printf("batch exceeds the CU count: the chain needs every workgroup resident\n"); return 1; }

  auto dev_f = [&](size_t n, float lo, float hi, float** out) -> int {
    std::vector<float> h(n);
    for (auto& v : h) v = lo + (hi - lo) * frand();
    CK(hipMalloc(out, n * 4)); CK(hipMemcpy(*out, h.data(), n * 4, hipMemcpyHostToDevice));
    return 0;
  };
  bf16_t* X0;
  {
    std::vector<unsigned short> h((size_t)M * C);
    for (auto& v : h) { const float r = frand(); v = r < 0.45f ? 0 : f2bf_host((r - 0.45f) * 2.f); }     // like a block output: ReLU + dropout
    CK(hipMalloc(&X0, tsz)); CK(hipMemcpy(X0, h.data(), tsz, hipMemcpyHostToDevice));
  }
  std::vector<BlockBufs> blk(NB);
  std::vector<SwzDesc> swz;
  // one arena for all statistics so that one memset clears them: [path][block][4 layers][8][2][256]
  float* stats_arena; const size_t stats_per_layer = (size_t)TN_NREP * 2 * C;
  CK(hipMalloc(&stats_arena, (size_t)2 * NB * 4 * stats_per_layer * 4));
  auto make_layer = [&](LayerBufs& L, bool dw, int bi, int li) -> int {
    if (dev_f(C, -0.1f, 0.1f, &L.bias)) return 1;
    if (dev_f(C, 0.8f, 1.2f, &L.gamma)) return 1;
    if (dev_f(C, -0.2f, 0.2f, &L.beta)) return 1;
    if (dw) { if (dev_f((size_t)C * 3, -0.6f, 0.6f, &L.wdw)) return 1; if (dev_f(C, -0.1f, 0.1f, &L.bdw)) return 1; } else { L.wdw = nullptr; L.bdw = nullptr; }
    std::vector<unsigned short> hw((size_t)C * C);
    for (auto& v : hw) v = f2bf_host((frand() - 0.5f) * 0.25f);
    CK(hipMalloc(&L.W, (size_t)C * C * 2)); CK(hipMemcpy(L.W, hw.data(), (size_t)C * C * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&L.Wswz, (size_t)C * C * 2));
    swz.push_back(SwzDesc{L.W, L.Wswz, C, C});
    for (int p = 0; p < 2; ++p) {
      L.stats[p] = stats_arena + ((size_t)(p * NB + bi) * 4 + li) * stats_per_layer;
      CK(hipMalloc(&L.Y[p], tsz));
      if (dw) CK(hipMalloc(&L.Q[p], tsz)); else L.Q[p] = nullptr;
    }
    return 0;
  };
  for (int i = 0; i < NB; ++i) {
    if (make_layer(blk[i].skip, false, i, 0)) return 1;
    for (int j = 0; j < 3; ++j) if (make_layer(blk[i].sub[j], true, i, 1 + j)) return 1;
    if (dev_f((size_t)16 * C, -0.15f, 0.15f, &blk[i].w1)) return 1;
    if (dev_f((size_t)C * 16, -0.5f, 0.5f, &blk[i].w2)) return 1;
    for (int p = 0; p < 2; ++p) {
      CK(hipMalloc(&blk[i].m[p], (size_t)B * C * 4)); CK(hipMalloc(&blk[i].h[p], (size_t)B * 16 * 4)); CK(hipMalloc(&blk[i].g[p], (size_t)B * C * 4));
      CK(hipMalloc(&blk[i].OUT[p], tsz));
    }
  }
  {
    SwzDesc* dd; CK(hipMalloc(&dd, swz.size() * sizeof(SwzDesc))); CK(hipMemcpy(dd, swz.data(), swz.size() * sizeof(SwzDesc), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(swizzle256_kernel<0>, dim3(16, (unsigned)swz.size()), dim3(256), 0, 0, dd);
    CK(hipDeviceSynchronize());
  }
  const float pd = drop ? 0.1f : 0.f;
  const uint32_t thr = drop ? (uint32_t)lrintf(pd * 65536.f) : 0u;
  const float inv_keep = 1.f / (1.f - pd);
  const uint64_t seed = 0x1234567ull;
  auto mk_act = [&](const LayerBufs& L, int p, int relu, bool dr, int layer) {
    BnAct a; memset(&a, 0, sizeof(a));
    a.stats = L.stats[p]; a.gamma = L.gamma; a.beta = L.beta; a.inv_n = 1.f / (float)M; a.eps = 1e-5f; a.mode = 1; a.relu = relu;
    if (dr && drop) { a.drop_thr = thr; a.drop_key = tn_layer_key(seed, (uint32_t)layer); a.inv_keep = inv_keep; }
    return a;
  };
  // ---- the per-layer launches (what tn_api.hip's forward does per mega block)
  auto run_ref = [&](hipStream_t st) -> int {
    const bf16_t* xin = X0;
    for (int i = 0; i < NB; ++i) {
      BlockBufs& K = blk[i];
      BnAct ident; memset(&ident, 0, sizeof(ident));
      {
        SubFwdV2Args va{xin, ident, nullptr, nullptr, K.skip.W, K.skip.bias, K.skip.Y[0], K.skip.stats[0], M, T, 0, K.skip.Wswz, nullptr};
        int rc = launch_sub_fwd_v4<1, false>(va, 256, st); if (rc) return rc;
      }
      const bf16_t* cur = xin; BnAct acur = ident;
      for (int j = 0; j < 3; ++j) {
        SubFwdV2Args va{cur, acur, K.sub[j].wdw, K.sub[j].bdw, K.sub[j].W, K.sub[j].bias, K.sub[j].Y[0], K.sub[j].stats[0], M, T, 0, K.sub[j].Wswz, K.sub[j].Q[0]};
        int rc = launch_sub_fwd_v5<3, true, 32>(va, 256, st); if (rc) return rc;
        cur = K.sub[j].Y[0]; acur = mk_act(K.sub[j], 0, 1, true, i * 4 + j);
      }
      SeCombineV3Args fa; memset(&fa, 0, sizeof(fa));
      fa.se.Y = cur; fa.se.act = acur; fa.se.W1 = K.w1; fa.se.W2 = K.w2; fa.se.m_out = K.m[0]; fa.se.h_out = K.h[0]; fa.se.g_out = K.g[0]; fa.se.T = T; fa.se.len = nullptr;
      fa.S = K.skip.Y[0]; fa.actS = mk_act(K.skip, 0, 0, false, 0); fa.OUT = K.OUT[0];
      if (drop) { fa.drop_thr = thr; fa.drop_key = tn_layer_key(seed, (uint32_t)(i * 4 + 3)); fa.inv_keep = inv_keep; } else fa.inv_keep = 1.f;
      int rc = launch_se_combine_fwd_v3(fa, B, st); if (rc) return rc;
      xin = K.OUT[0];
    }
    return 0;
  };
  // ---- the chain
  std::vector<ChainBlock> hb(NB);
  for (int i = 0; i < NB; ++i) {
    auto fill = [&](ChainLayer& L, const LayerBufs& s, int layer) {
      L.Wswz = s.Wswz; L.bias = s.bias; L.Y = s.Y[1]; L.stats = s.stats[1]; L.gamma = s.gamma; L.beta = s.beta; L.wdw = s.wdw; L.bdw = s.bdw; L.Q = s.Q[1];
      L.drop_key = tn_layer_key(seed, (uint32_t)layer); L.pad_ = 0;
    };
    fill(hb[i].skip, blk[i].skip, 0);
    for (int j = 0; j < 3; ++j) fill(hb[i].sub[j], blk[i].sub[j], i * 4 + j);
    hb[i].se_w1 = blk[i].w1; hb[i].se_w2 = blk[i].w2; hb[i].m_out = blk[i].m[1]; hb[i].h_out = blk[i].h[1]; hb[i].g_out = blk[i].g[1];
    hb[i].OUT = blk[i].OUT[1]; hb[i].out_key = tn_layer_key(seed, (uint32_t)(i * 4 + 3)); hb[i].pad_ = 0;
  }
  ChainBlock* dblocks; CK(hipMalloc(&dblocks, NB * sizeof(ChainBlock))); CK(hipMemcpy(dblocks, hb.data(), NB * sizeof(ChainBlock), hipMemcpyHostToDevice));
  unsigned* bar; CK(hipMalloc(&bar, CH_BAR_WORDS * 4));
  const size_t nstamps = (size_t)NB * 16 + 4 * 2 * 12 * 4;
  unsigned long long* stamps; CK(hipMalloc(&stamps, nstamps * 8)); CK(hipMemset(stamps, 0, nstamps * 8));
  ChainArgs ca; memset(&ca, 0, sizeof(ca));
  ca.X0 = X0; ca.x0_mode = 0; ca.blocks = dblocks; ca.nblocks = NB; ca.B = B; ca.T = T; ca.inv_n = 1.f / (float)M; ca.eps = 1e-5f; ca.inv_keep = inv_keep;
  ca.drop_thr = thr; ca.key_add = nullptr; ca.bar = bar; ca.stamps = stamps;
  auto run_chain = [&](hipStream_t st, bool stamp) -> int {
    if (hipMemsetAsync(bar, 0, CH_BAR_WORDS * 4, st) != hipSuccess) return 1;
    return launch_chain_fwd(ca, stamp, st);
  };
  auto zero_stats = [&](int p) { return hipMemsetAsync(stats_arena + (size_t)p * NB * 4 * stats_per_layer, 0, (size_t)NB * 4 * stats_per_layer * 4, 0); };

  // ---- correctness first
  CK(zero_stats(0)); { int rc = run_ref(0); if (rc) { printf("reference launch rc %d\n", rc); return 1; } }
  CK(hipDeviceSynchronize());
  CK(zero_stats(1)); { int rc = run_chain(0, true); if (rc) { printf("chain launch rc %d\n", rc); return 1; } }
  { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { printf("chain kernel failed: %s\n", hipGetErrorString(e)); return 1; } }
  {
    unsigned hbar[CH_BAR_WORDS]; CK(hipMemcpy(hbar, bar, sizeof(hbar), hipMemcpyDeviceToHost));
    printf("barrier error word: 0x%x (0 = every barrier completed), top counter %u (expected %u)\n", hbar[CH_BAR_ERR], hbar[CH_BAR_TOP], (unsigned)((B < 8 ? B : 8) * 3 * NB));
    if (hbar[CH_BAR_ERR]) return 1;
  }
  auto cmp_bf = [&](const char* name, const bf16_t* r, const bf16_t* c) -> int {
    std::vector<unsigned short> hr((size_t)M * C), hc((size_t)M * C);
    CK(hipMemcpy(hr.data(), r, tsz, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), c, tsz, hipMemcpyDeviceToHost));
    size_t ndiff = 0, nbig = 0; double maxd = 0, maxv = 0; size_t first = (size_t)-1;
    for (size_t i = 0; i < hr.size(); ++i) {
      if (hr[i] != hc[i]) {
        ++ndiff;
        if (first == (size_t)-1) first = i;
        const double a = bf2f_host(hr[i]), b2 = bf2f_host(hc[i]), d = fabs(a - b2);
        maxd = std::max(maxd, d);
        if (d > 0.0079 * std::max(fabs(a), fabs(b2)) + 1e-6) ++nbig;      // more than one bf16 ulp
      }
      maxv = std::max(maxv, (double)fabs(bf2f_host(hr[i])));
    }
    printf("  %-8s %9zu of %zu elements differ (%zu by more than 1 bf16 ulp), max |diff| %.3g, max |value| %.3g", name, ndiff, hr.size(), nbig, maxd, maxv);
    if (ndiff) printf(", first at row %zu ch %zu", first / C, first % C);
    printf("\n");
    return 0;
  };
  auto cmp_f = [&](const char* name, const float* r, const float* c, size_t n) -> int {
    std::vector<float> hr(n), hc(n);
    CK(hipMemcpy(hr.data(), r, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), c, n * 4, hipMemcpyDeviceToHost));
    double maxrel = 0, maxd = 0;
    for (size_t i = 0; i < n; ++i) { const double d = fabs((double)hr[i] - hc[i]); maxd = std::max(maxd, d); maxrel = std::max(maxrel, d / (fabs((double)hr[i]) + 1e-6)); }
    printf("  %-8s max |diff| %.3g, max relative %.3g over %zu values\n", name, maxd, maxrel, n);
    return 0;
  };
  for (int i = 0; i < NB; ++i) {
    printf("block %d: chain vs per-layer launches\n", i);
    if (cmp_bf("S", blk[i].skip.Y[0], blk[i].skip.Y[1])) return 1;
    for (int j = 0; j < 3; ++j) {
      char nm[16];
      snprintf(nm, sizeof(nm), "Q%d", j + 1); if (cmp_bf(nm, blk[i].sub[j].Q[0], blk[i].sub[j].Q[1])) return 1;
      snprintf(nm, sizeof(nm), "Y%d", j + 1); if (cmp_bf(nm, blk[i].sub[j].Y[0], blk[i].sub[j].Y[1])) return 1;
    }
    if (cmp_bf("OUT", blk[i].OUT[0], blk[i].OUT[1])) return 1;
    if (cmp_f("SE mean", blk[i].m[0], blk[i].m[1], (size_t)B * C)) return 1;
    if (cmp_f("SE hid", blk[i].h[0], blk[i].h[1], (size_t)B * 16)) return 1;
    if (cmp_f("SE gate", blk[i].g[0], blk[i].g[1], (size_t)B * C)) return 1;
    {
      // BatchNorm sums: compare the replica totals
      std::vector<float> hs((size_t)2 * NB * 4 * stats_per_layer);
      CK(hipMemcpy(hs.data(), stats_arena, hs.size() * 4, hipMemcpyDeviceToHost));
      double worst = 0;
      for (int li = 0; li < 4; ++li)
        for (int w = 0; w < 2; ++w)
          for (int c = 0; c < C; ++c) {
            double s0 = 0, s1 = 0;
            for (int r = 0; r < TN_NREP; ++r) {
              s0 += hs[((size_t)(0 * NB + i) * 4 + li) * stats_per_layer + (size_t)(r * 2 + w) * C + c];
              s1 += hs[((size_t)(1 * NB + i) * 4 + li) * stats_per_layer + (size_t)(r * 2 + w) * C + c];
            }
            worst = std::max(worst, fabs(s0 - s1) / (fabs(s0) + 1.0));
          }
      printf("  BN sums  max relative difference of the replica totals %.3g\n", worst);
    }
  }
  {
    std::vector<unsigned long long> hs((size_t)NB * 16); CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    const char* names[9] = {"skip pass", "sub-block 1 pass", "grid barrier 1", "sub-block 2 pass (act on load)", "grid barrier 2", "sub-block 3 pass", "grid barrier 3",
                            "SE sums + mat-vecs", "combine"};
    printf("chain, workgroup 0, per phase (us; 100 MHz wall clock, first run = cold instruction cache):\n");
    for (int i = 0; i < NB; ++i) {
      printf("  block %d:", i);
      for (int k = 0; k < 9; ++k) printf(" %s %.1f |", names[k], (double)(hs[i * 16 + k + 1] - hs[i * 16 + k]) * 0.01);
      printf(" total %.1f\n", (double)(hs[i * 16 + 9] - hs[i * 16]) * 0.01);
    }
  }
  // ---- the noise floor: the per-layer launches against THEMSELVES (a second run into the chain's buffers: only the order of the
  // float atomics of the BatchNorm sums differs between two runs)
  {
    std::vector<unsigned short> keep((size_t)M * C);
    CK(hipMemcpy(keep.data(), blk[NB - 1].OUT[0], tsz, hipMemcpyDeviceToHost));
    CK(zero_stats(0)); { int rc = run_ref(0); if (rc) return 1; } CK(hipDeviceSynchronize());
    std::vector<unsigned short> again((size_t)M * C);
    CK(hipMemcpy(again.data(), blk[NB - 1].OUT[0], tsz, hipMemcpyDeviceToHost));
    size_t nd = 0, nbig = 0; double maxd = 0;
    for (size_t i = 0; i < keep.size(); ++i) if (keep[i] != again[i]) {
      ++nd; const double x = bf2f_host(keep[i]), y = bf2f_host(again[i]), d = fabs(x - y); maxd = std::max(maxd, d);
      if (d > 0.0079 * std::max(fabs(x), fabs(y)) + 1e-6) ++nbig;
    }
    printf("noise floor: two runs of the per-layer launches, OUT of block %d: %zu elements differ (%zu by more than 1 bf16 ulp), max |diff| %.3g\n", NB - 1, nd, nbig, maxd);
  }

  // ---- timing: alternating, events around REPS runs each
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int round = 0; round < 3; ++round) {
    for (int which = 0; which < 2; ++which) {
      for (int it = 0; it < 3; ++it) { CK(zero_stats(which)); int rc = which ? run_chain(0, false) : run_ref(0); if (rc) { printf("rc %d\n", rc); return 1; } }
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int it = 0; it < REPS; ++it) { CK(zero_stats(which)); int rc = which ? run_chain(0, false) : run_ref(0); if (rc) { printf("rc %d\n", rc); return 1; } }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s: %.1f us per mega block (%.1f us per run of %d block(s))\n", which ? "chain (one launch)        " : "per-layer launches (5/blk)", ms * 1e3f / REPS / NB, ms * 1e3f / REPS, NB);
    }
  }
  // stamped run after warm-up: the per-phase breakdown the timing above corresponds to
  {
    CK(zero_stats(1)); int rc = run_chain(0, true); if (rc) return 1; CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs((size_t)NB * 16); CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    const char* names[9] = {"skip", "sub1", "bar1", "sub2", "bar2", "sub3", "bar3", "se", "combine"};
    printf("chain, workgroup 0, per phase, warm (us):\n");
    for (int i = 0; i < NB; ++i) {
      printf("  block %d:", i);
      for (int k = 0; k < 9; ++k) printf(" %s %.1f", names[k], (double)(hs[i * 16 + k + 1] - hs[i * 16 + k]) * 0.01);
      printf(" | total %.1f\n", (double)(hs[i * 16 + 9] - hs[i * 16]) * 0.01);
    }
  }
  // per-iteration stamps of block 0's passes (same stamped run): what each role does between the workgroup barriers
  {
    std::vector<unsigned long long> hs(nstamps); CK(hipMemcpy(hs.data(), stamps, nstamps * 8, hipMemcpyDeviceToHost));
    const char* pn[4] = {"skip", "sub1", "sub2", "sub3"};
    for (int ps = 0; ps < 4; ++ps) {
      printf("block 0 %s pass, per iteration (us): producer work / wait at the barrier || consumer work / wait\n", pn[ps]);
      for (int it = 0; it < 12; ++it) {
        printf("   it %2d:", it);
        for (int role = 0; role < 2; ++role) {
          const unsigned long long* q = &hs[(size_t)NB * 16 + (((ps * 2 + role) * 12 + it) * 4)];
          printf(" %5.2f %5.2f %s", (double)(q[1] - q[0]) * 0.01, (double)(q[2] - q[1]) * 0.01, role == 0 ? "||" : "");
        }
        printf("\n");
      }
    }
  }
  // ---- the barrier alone
  {
    const int nbar = 51;
    float* pst; CK(hipMalloc(&pst, (size_t)nbar * 4096 * 4));
    unsigned long long* pstamps; CK(hipMalloc(&pstamps, (nbar + 2) * 8));
    for (int it = 0; it < 3; ++it) {
      CK(hipMemset(pst, 0, (size_t)nbar * 4096 * 4)); CK(hipMemset(bar, 0, CH_BAR_WORDS * 4)); CK(hipMemset(pstamps, 0, (nbar + 2) * 8));
      hipLaunchKernelGGL(barrier_probe_kernel, dim3(B), dim3(512), 0, 0, bar, pst, nbar, B, pstamps);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> hs(nbar + 2); CK(hipMemcpy(hs.data(), pstamps, hs.size() * 8, hipMemcpyDeviceToHost));
      unsigned herr; CK(hipMemcpy(&herr, bar + CH_BAR_ERR, 4, hipMemcpyDeviceToHost));
      printf("barrier probe: %d x (512 statistics atomics per workgroup + grid barrier + re-read), %d workgroups: %.2f us each (error word 0x%x, stale reads %s)\n",
             nbar, B, (double)(hs[nbar] - hs[0]) * 0.01 / nbar, herr, hs[nbar + 1] == 0xdeadull ? "SEEN" : "none");
    }
  }
  return 0;
}
