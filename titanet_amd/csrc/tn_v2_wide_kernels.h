// titanet_amd — "wide" pointwise kernels of the decoder side of TitaNet-S (bf16): GEMMs whose OUTPUT is the
// 1536-channel encoder width D (epilog 1x1 conv, ASP energies, d(epilog BN output)) with K = 256 or 128.
//
// MI355X-first layout of the work: M = B*T rows are cut into ONE contiguous row range per workgroup (300 rows at
// B = 256, T = 300, grid = 256 = one workgroup per CU).  The workgroup stages its rows of the A operand in LDS
// ONCE (160 KB LDS per CU: 160 x 264 or 320 x 136 bf16), then walks the N / 256 output slabs: the 256 x K weight
// slab lives in registers as MFMA A fragments (prefetched one slab ahead), the activation rows are the B operand,
// so the A operand is read from HBM once, weights stream from L2 once per (workgroup, slab), and the only HBM
// stream that scales with N is the output itself.  Per-channel reductions (BatchNorm statistics / BatchNorm
// backward sums) stay in registers for a whole slab and cost one replicated atomic per (workgroup, channel).
#pragma once
#include "tn_gemm.h"
#include "tn_v2_kernels.h"

struct WideOutArgs {
  const bf16_t* X;      // [M][K] A operand (stored final: no activation on load)
  const bf16_t* W;      // [N][K] bf16, row = output channel
  const uint4* Wswz;    // optional: W in MFMA-fragment order (swizzle256_kernel)
  const float* bias;    // [N] or null
  bf16_t* Y;            // [M][N]
  float* stats;         // EPI 0: [TN_NREP][2][N] sum / sum of squares of Y, or null
  const bf16_t* RAW;    // EPI 2: [M][N] raw forward output of the layer whose BN + ReLU the gradient passes through
  BnAct actR;           // EPI 2: that BatchNorm (+ ReLU)
  float* bsums;         // EPI 2: [TN_NREP][2][N] BatchNorm backward sums
  int M, N, rows_per_wg;
};

// EPI 0: Y = acc + bias (+ statistics).   EPI 2: Y = (acc + Y) * [BN(RAW) > 0];  sums of Y and Y * xhat(RAW).
template <int K, int EPI>
__global__ __launch_bounds__(V2_NT, 2) void wide_out_v2_kernel(WideOutArgs a) {
  constexpr int AP = K + 8;                        // LDS row pitch of the A rows (bank spread)
  constexpr int PR = K == 256 ? 160 : 320;         // rows staged per pass
  constexpr int KS = K / 16;                       // MFMA k-steps
  constexpr int VPR = K / 8;                       // 16-byte vectors per A row
  constexpr int NSTG = PR * VPR / V2_NT;           // staging vectors per thread per pass (10)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);                    // [PR][AP]
  bf16_t* Cs = As + PR * AP;                                       // [64][264] output staging; also the reduction scratch
  float* par = reinterpret_cast<float*>(Cs + V2_R * V2_AP);        // EPI 2: sc, sh, mean*rstd, rstd : [4][N]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int vc = tid & 31, rq = tid >> 5, c0 = vc * 8;
  const int nslabs = a.N / V2_C;
  const int r_begin = blockIdx.x * a.rows_per_wg;
  const int r_end = min(a.M, r_begin + a.rows_per_wg);
  if (r_begin >= r_end) return;

  bf16x8_t wf[KS], wfn[KS];
  float biasn[16];
  // fragment-order weights through buffer loads: one per-lane offset register, the (slab, wave, k-step) part is scalar (flat
  // addressing kept 16 per-lane 64-bit addresses alive next to 128 fragment registers: the <256, 0> instance spilled)
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.Wswz), 0, a.Wswz ? a.N * K * (int)sizeof(bf16_t) : 0, 0x00020000);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto fetch_w = [&](int slab) {
    const int co = slab * V2_C + wave * 32 + (lane & 31);
    if (a.Wswz) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        wfn[ks] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(srdW, lane * 16, ((slab * 8 + wave_u) * KS + ks) * 1024, 0));
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wfn[ks] = *reinterpret_cast<const bf16x8_t*>(a.W + (size_t)co * K + ks * 16 + half * 8);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) biasn[4 * g + j] = a.bias ? a.bias[slab * V2_C + wave * 32 + 8 * g + 4 * half + j] : 0.f;
  };
  fetch_w(0);
  if (EPI == 2) {
    for (int c = tid; c < a.N; c += V2_NT) {
      float s, h, mean, rstd;
      bn_scale_shift(a.actR, a.N, c, s, h);
      bn_mean_rstd(a.actR, a.N, c, mean, rstd);
      par[c] = s; par[a.N + c] = h; par[2 * a.N + c] = mean * rstd; par[3 * a.N + c] = rstd;
    }
  }

  for (int p0 = r_begin; p0 < r_end; p0 += PR) {
    const int prow = min(PR, r_end - p0);
    __syncthreads();                                // previous pass done with As
    {
      uint4 st[NSTG];
#pragma unroll
      for (int q = 0; q < NSTG; ++q) {
        const int v = tid + q * V2_NT, r = v / VPR, cv = v % VPR;
        st[q] = r < prow ? *reinterpret_cast<const uint4*>(a.X + (size_t)(p0 + r) * K + cv * 8) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < NSTG; ++q) {
        const int v = tid + q * V2_NT, r = v / VPR, cv = v % VPR;
        *reinterpret_cast<uint4*>(As + r * AP + cv * 8) = st[q];
      }
    }
    __syncthreads();
    for (int slab = 0; slab < nslabs; ++slab) {
      float biasr[16];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wf[ks] = wfn[ks];
#pragma unroll
      for (int r = 0; r < 16; ++r) biasr[r] = biasn[r];
      {
        // next slab of this pass, or slab 0 of the next pass: in flight during this slab's tiles
        const int nxt = slab + 1 < nslabs ? slab + 1 : 0;
        if (slab + 1 < nslabs || p0 + PR < r_end) fetch_w(nxt);
      }
      float s1[8], s2[8], psc[8], psh[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
      if (EPI == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { psc[i] = par[slab * V2_C + c0 + i]; psh[i] = par[a.N + slab * V2_C + c0 + i]; }
      }
      for (int tt = 0; tt < prow; tt += V2_R) {
        f32x16_t acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = biasr[r]; acc1[r] = biasr[r]; }
        const bf16_t* brow = As + (tt + (lane & 31)) * AP + half * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(brow + ks * 16);
          const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(brow + 32 * AP + ks * 16);   // rows past the pass: unused results
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b1, acc1, 0, 0, 0);
        }
        __syncthreads();                            // previous tile's staging has been consumed
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = wave * 32 + 8 * g + 4 * half;
          uint2 w0, w1;
          w0.x = f2bf_pk(acc0[4 * g], acc0[4 * g + 1]); w0.y = f2bf_pk(acc0[4 * g + 2], acc0[4 * g + 3]);
          w1.x = f2bf_pk(acc1[4 * g], acc1[4 * g + 1]); w1.y = f2bf_pk(acc1[4 * g + 2], acc1[4 * g + 3]);
          *reinterpret_cast<uint2*>(Cs + (lane & 31) * V2_AP + co) = w0;
          *reinterpret_cast<uint2*>(Cs + (32 + (lane & 31)) * V2_AP + co) = w1;
        }
        __syncthreads();
        if (EPI == 2) {
          uint4 rd[4], ry[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int o = tt + rq + 16 * q;
            if (o < prow) {
              const size_t off = (size_t)(p0 + o) * a.N + slab * V2_C + c0;
              rd[q] = *reinterpret_cast<const uint4*>(a.Y + off);
              ry[q] = *reinterpret_cast<const uint4*>(a.RAW + off);
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int o = tt + rq + 16 * q;
            if (o < prow) {
              float g8[8], d[8], y[8];
              unpack8(*reinterpret_cast<const uint4*>(Cs + (rq + 16 * q) * V2_AP + c0), g8);
              unpack8(rd[q], d);
              unpack8(ry[q], y);
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const float z = fmaf(y[i], psc[i], psh[i]);
                const float v = (z > 0.f) ? (g8[i] + d[i]) : 0.f;
                d[i] = v;
                s1[i] += v;
                s2[i] = fmaf(v, y[i], s2[i]);        // against the raw y; converted to xhat at the end of the slab
              }
              store8(a.Y + (size_t)(p0 + o) * a.N + slab * V2_C + c0, d);
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int o = tt + rq + 16 * q;
            if (o < prow) {
              const uint4 raw = *reinterpret_cast<const uint4*>(Cs + (rq + 16 * q) * V2_AP + c0);
              *reinterpret_cast<uint4*>(a.Y + (size_t)(p0 + o) * a.N + slab * V2_C + c0) = raw;
              if (a.stats) {
                float y[8];
                unpack8(raw, y);
#pragma unroll
                for (int i = 0; i < 8; ++i) { s1[i] += y[i]; s2[i] = fmaf(y[i], y[i], s2[i]); }
              }
            }
          }
        }
      }
      // ---- per-slab channel sums: 16 row phases -> LDS -> one replicated atomic per (workgroup, channel)
      float* dst = EPI == 2 ? a.bsums : a.stats;
      if (dst) {
        if (EPI == 2) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            s2[i] = par[3 * a.N + slab * V2_C + c0 + i] * s2[i] - par[2 * a.N + slab * V2_C + c0 + i] * s1[i];
        }
        __syncthreads();
        float* red = reinterpret_cast<float*>(Cs);   // [16][2][256] = 32 KB <= staging tile
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          red[(rq * 2 + 0) * V2_C + c0 + i] = s1[i];
          red[(rq * 2 + 1) * V2_C + c0 + i] = s2[i];
        }
        __syncthreads();
        const int which = tid >> 8, c = tid & 255;
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) v += red[(r * 2 + which) * V2_C + c];
        atomic_add_f32(&dst[(size_t)((blockIdx.x % TN_NREP) * 2 + which) * a.N + slab * V2_C + c], v);
      }
    }
  }
}

template <int K, int EPI>
inline int launch_wide_out_v2(WideOutArgs a, int max_wgs, hipStream_t st) {
  constexpr int AP = K + 8, PR = K == 256 ? 160 : 320;
  int grid = (a.M + V2_R - 1) / V2_R;
  if (grid > max_wgs) grid = max_wgs;
  a.rows_per_wg = (a.M + grid - 1) / grid;
  grid = (a.M + a.rows_per_wg - 1) / a.rows_per_wg;
  const size_t smem = (size_t)(PR * AP + V2_R * V2_AP) * sizeof(bf16_t) + (EPI == 2 ? (size_t)4 * a.N * sizeof(float) : 0);
  auto kern = wide_out_v2_kernel<K, EPI>;
  TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(V2_NT), smem, st, a);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// wide_in_v2: GEMMs that CONSUME the 1536-channel width:  out[M][128] = f(A[M][1536]) * W^T   (K = 1536, N = 128)
//   MODE 0 (ASP attention hidden layer, reference src/models.py:562-565): A = relu(BN(E)), out = tanh(acc + bias)
//   MODE 1 (its backward): A = dEN, out = acc * (1 - hid^2), column sums of out -> d bias
// One contiguous row range per workgroup (<= 320 rows per group = 5 tiles of 64); the K loop is OUTERMOST: the
// 128 x 256 weight chunk is resident in registers (prefetched through LDS one chunk ahead) while ALL the
// accumulators of the row group (5 tiles x 16 VGPRs) stay in registers — so both the activations (HBM, once) and
// the weights (L2, once per workgroup) are read exactly once, and nothing but the 128-wide result is written.
// Waves: 4 channel blocks (32 outputs) x 2 row halves of the 64-row tile.
// ------------------------------------------------------------------------------------------
struct WideInArgs {
  const bf16_t* A;      // [M][KW]
  BnAct act;            // MODE 0: BatchNorm (+ ReLU) applied to A on load
  const bf16_t* W;      // [128][KW] bf16, row = output channel
  const float* bias;    // MODE 0: [128]
  const bf16_t* H;      // MODE 1: [M][128] tanh outputs
  float* colsum;        // MODE 1: [128] += column sums (atomic) or null
  bf16_t* Y;            // [M][128]
  int M, KW, rows_per_wg;
};
template <int MODE>
__global__ __launch_bounds__(V2_NT, 2) void wide_in_v2_kernel(WideInArgs a) {
  constexpr int NO = 128, NOP = NO + 8, GT = 5, GR = GT * V2_R;       // outputs, staging pitch, tiles / rows per group
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);                       // [64][264] transformed A chunk tile; epilogue staging [64][136]
  bf16_t* Wl = As + V2_R * V2_AP;                                     // [128][264] next weight chunk
  float* par = reinterpret_cast<float*>(Wl + NO * V2_AP);             // MODE 0: sc, sh [2][KW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int cb = wave & 3, rh = wave >> 2;
  const int vc = tid & 31, rq = tid >> 5, c0 = vc * 8;
  const int nchunks = a.KW / V2_C;
  const int r_begin = blockIdx.x * a.rows_per_wg;
  const int r_end = min(a.M, r_begin + a.rows_per_wg);
  if (r_begin >= r_end) return;
  if (MODE == 0) {
    for (int c = tid; c < a.KW; c += V2_NT) {
      float s, h;
      bn_scale_shift(a.act, a.KW, c, s, h);
      par[c] = s; par[a.KW + c] = h;
    }
  }
  // weight chunk kc -> LDS in 8 pieces of 16 bytes per thread.  The NEXT chunk travels two pieces per tile step (loads issued
  // at the top of the step, stored behind its MFMAs): the 8-piece register prefetch of round 2 (32 VGPRs next to 80
  // accumulator + 64 fragment registers) lived in scratch — hipcc parked every piece there with a wait right behind its load,
  // so each chunk began with an exposed L2 round trip and the activation prefetch was drained with it (round 4).
  // Buffer loads with ONE per-lane offset register per stream and a scalar offset per (chunk, tile, piece): with flat
  // addresses hipcc hoisted the 20 + 8 per-lane 64-bit addresses of the unrolled steps out of the chunk loop and spilled them.
  // Rows at or beyond r_end read as zeros (num_records).
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W), 0, NO * a.KW * (int)sizeof(bf16_t), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.A), 0, (int)((size_t)r_end * a.KW * sizeof(bf16_t)), 0x00020000);
  const int voffW = ((tid >> 5) * a.KW + (tid & 31) * 8) * (int)sizeof(bf16_t);
  const int voffA = (rq * a.KW + c0) * (int)sizeof(bf16_t);
  auto w_load2 = [&](int kc, int q0, uint4 (&wr)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const u32x4_t r = __builtin_amdgcn_raw_buffer_load_b128(srdW, voffW, ((q0 + q) * 16 * a.KW + kc * V2_C) * (int)sizeof(bf16_t), 0);
      wr[q] = make_uint4(r[0], r[1], r[2], r[3]);
    }
  };
  auto w_store2 = [&](int q0, const uint4 (&wr)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int v = tid + (q0 + q) * V2_NT, r = v >> 5, cv = v & 31;
      *reinterpret_cast<uint4*>(Wl + r * V2_AP + cv * 8) = wr[q];
    }
  };
  for (int g0 = r_begin; g0 < r_end; g0 += GR) {
    const int grow = min(GR, r_end - g0);
    const int ntile = (grow + V2_R - 1) / V2_R;
    f32x16_t acc[GT];
#pragma unroll
    for (int t = 0; t < GT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    __syncthreads();
#pragma unroll
    for (int q0 = 0; q0 < 8; q0 += 2) {
      uint4 wr[2];
      w_load2(0, q0, wr);
      w_store2(q0, wr);
    }
    uint4 pa[4];
    auto a_fetch = [&](int kc, int t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u32x4_t r = __builtin_amdgcn_raw_buffer_load_b128(srdA, voffA, ((g0 + t * V2_R + 16 * q) * a.KW + kc * V2_C) * (int)sizeof(bf16_t), 0);
        pa[q] = make_uint4(r[0], r[1], r[2], r[3]);
      }
    };
    a_fetch(0, 0);
    for (int kc = 0; kc < nchunks; ++kc) {
      __syncthreads();                              // Wl holds chunk kc (the pieces stored during the previous chunk's steps)
      bf16x8_t wf[16];
      {
        const bf16_t* wrow = Wl + (cb * 32 + (lane & 31)) * V2_AP + half * 8;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) wf[ks] = *reinterpret_cast<const bf16x8_t*>(wrow + ks * 16);
      }
      const bool more_w = kc + 1 < nchunks;
#pragma unroll
      for (int t = 0; t < GT; ++t) {
        if (t < ntile) {
          uint4 wp[2];
          if (t < 4 && more_w) w_load2(kc + 1, 2 * t, wp);
          __syncthreads();                          // previous MFMA done with As (and, for t == 0, every wave holds its wf)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v[8];
            unpack8(pa[q], v);
            if (MODE == 0) {
              float sc[8], sh[8];
              *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(par + kc * V2_C + c0);
              *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(par + kc * V2_C + c0 + 4);
              *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(par + a.KW + kc * V2_C + c0);
              *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(par + a.KW + kc * V2_C + c0 + 4);
#pragma unroll
              for (int i = 0; i < 8; ++i) v[i] = fmaxf(fmaf(v[i], sc[i], sh[i]), 0.f);
            }
            store8(As + (rq + 16 * q) * V2_AP + c0, v);
          }
          // next tile of this chunk, or the first tile of the next chunk
          if (t + 1 < ntile) a_fetch(kc, t + 1);
          else if (kc + 1 < nchunks) a_fetch(kc + 1, 0);
          __syncthreads();
          const bf16_t* brow = As + (rh * 32 + (lane & 31)) * V2_AP + half * 8;
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) {
            const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(brow + ks * 16);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b, acc[t], 0, 0, 0);
          }
          if (t < 4 && more_w) w_store2(2 * t, wp);
        }
      }
      // row groups of fewer than 4 tiles: the pieces no tile step moved
      if (more_w && ntile < 4) {
        for (int t = ntile; t < 4; ++t) {
          uint4 wp[2];
          w_load2(kc + 1, 2 * t, wp);
          w_store2(2 * t, wp);
        }
      }
    }
    // ---- epilogue: lane = row (rh*32 + lane&31), regs 4g..4g+3 = channels cb*32 + 8g + 4*half + 0..3
    float biasr[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) biasr[4 * g + j] = (MODE == 0 && a.bias) ? a.bias[cb * 32 + 8 * g + 4 * half + j] : 0.f;
    float cs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cs[i] = 0.f;
    bf16_t* Cs = As;                                 // [64][136]
    const int ev = tid & 15, er = tid >> 4;          // coalesced phase: 16 vectors per 128-wide row, 32 rows per sweep
#pragma unroll
    for (int t = 0; t < GT; ++t) {
      if (t < ntile) {
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] = acc[t][4 * g + j] + biasr[4 * g + j];
            if (MODE == 0) o[j] = fast_tanh(o[j]);
          }
          uint2 w;
          w.x = f2bf_pk(o[0], o[1]); w.y = f2bf_pk(o[2], o[3]);
          *reinterpret_cast<uint2*>(Cs + (rh * 32 + (lane & 31)) * NOP + cb * 32 + 8 * g + 4 * half) = w;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int o = t * V2_R + er + 32 * q;
          if (o < grow) {
            uint4 raw = *reinterpret_cast<const uint4*>(Cs + (er + 32 * q) * NOP + ev * 8);
            if (MODE == 1) {
              float gq[8], h[8];
              unpack8(raw, gq);
              unpack8(*reinterpret_cast<const uint4*>(a.H + (size_t)(g0 + o) * NO + ev * 8), h);
#pragma unroll
              for (int i = 0; i < 8; ++i) { gq[i] *= (1.f - h[i] * h[i]); cs[i] += gq[i]; }
              store8(a.Y + (size_t)(g0 + o) * NO + ev * 8, gq);
            } else {
              *reinterpret_cast<uint4*>(a.Y + (size_t)(g0 + o) * NO + ev * 8) = raw;
            }
          }
        }
      }
    }
    if (MODE == 1 && a.colsum) {
      __syncthreads();
      float* red = reinterpret_cast<float*>(As);     // [32][128]
#pragma unroll
      for (int i = 0; i < 8; ++i) red[er * NO + ev * 8 + i] = cs[i];
      __syncthreads();
      if (tid < NO) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) v += red[r * NO + tid];
        atomic_add_f32(&a.colsum[tid], v);
      }
    }
  }
}

template <int MODE>
inline int launch_wide_in_v2(WideInArgs a, int max_wgs, hipStream_t st) {
  if ((size_t)a.M * a.KW * sizeof(bf16_t) >= ((size_t)1 << 31)) return -1000;      // 32-bit buffer offsets
  int grid = (a.M + V2_R - 1) / V2_R;
  if (grid > max_wgs) grid = max_wgs;
  a.rows_per_wg = (a.M + grid - 1) / grid;
  grid = (a.M + a.rows_per_wg - 1) / a.rows_per_wg;
  const size_t smem = (size_t)(V2_R * V2_AP + 128 * V2_AP) * sizeof(bf16_t) + (MODE == 0 ? (size_t)2 * a.KW * sizeof(float) : 0);
  auto kern = wide_in_v2_kernel<MODE>;
  TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(V2_NT), smem, st, a);
  return (int)hipGetLastError();
}

