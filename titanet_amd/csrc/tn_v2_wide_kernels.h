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
#include <type_traits>
#include "tn_gemm.h"
#include "tn_v2_kernels.h"
#include "tn_pgemm.h"

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
      for (int j = 0; j < 4; ++j) biasn[4 * g + j] = (EPI != 2 && a.bias) ? a.bias[slab * V2_C + wave * 32 + 8 * g + 4 * half + j] : 0.f;      // (EPI 2 has no bias: 32 registers)
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

  // EPI 2: the two element-wise operands of a tile (the direct gradient in Y, the raw forward output) are PREFETCHED — issued at
  // the end of the previous tile's epilogue, unconditional (rows clamped): loaded behind the staging barrier and used at once,
  // every tile exposed a full HBM round trip to all eight waves at the same moment (162 us for 708 MB)
  const __amdgpu_buffer_rsrc_t srdY = __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, EPI == 2 ? __builtin_amdgcn_readfirstlane((int)((size_t)a.M * a.N * sizeof(bf16_t))) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdR = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.RAW), 0, EPI == 2 ? __builtin_amdgcn_readfirstlane((int)((size_t)a.M * a.N * sizeof(bf16_t))) : 0, 0x00020000);
  uint4 rdc[4], ryc[4];
  auto epi_fetch = [&](int p0_, int prow_, int slab_, int tt_, uint4 (&rd_)[4], uint4 (&ry_)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = min(tt_ + rq + 16 * q, prow_ - 1);
      const int voff = (((p0_ + o) * a.N) + c0) * (int)sizeof(bf16_t);
      const u32x4_t d = __builtin_amdgcn_raw_buffer_load_b128(srdY, voff, slab_ * V2_C * (int)sizeof(bf16_t), 0);
      const u32x4_t y = __builtin_amdgcn_raw_buffer_load_b128(srdR, voff, slab_ * V2_C * (int)sizeof(bf16_t), 0);
      rd_[q] = make_uint4(d[0], d[1], d[2], d[3]);
      ry_[q] = make_uint4(y[0], y[1], y[2], y[3]);
    }
  };
  for (int p0 = r_begin; p0 < r_end; p0 += PR) {
    const int prow = min(PR, r_end - p0);
    if (EPI == 2) epi_fetch(p0, prow, 0, 0, rdc, ryc);
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
      float s1[8], s2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
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
          // BatchNorm scale / shift of the thread's 8 channels, read per tile (an opaque offset: as loop invariants they would
          // hold 16 registers next to two tiles of prefetched operands and the weight double buffer — spills)
          int pco = slab * V2_C + c0;
          asm volatile("" : "+v"(pco));
          float psc[8], psh[8];
          *reinterpret_cast<float4*>(psc) = *reinterpret_cast<const float4*>(par + pco);
          *reinterpret_cast<float4*>(psc + 4) = *reinterpret_cast<const float4*>(par + pco + 4);
          *reinterpret_cast<float4*>(psh) = *reinterpret_cast<const float4*>(par + a.N + pco);
          *reinterpret_cast<float4*>(psh + 4) = *reinterpret_cast<const float4*>(par + a.N + pco + 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int o = tt + rq + 16 * q;
            const bool ok = o < prow;
            float g8[8], d[8], y[8];
            unpack8(*reinterpret_cast<const uint4*>(Cs + (rq + 16 * q) * V2_AP + c0), g8);
            unpack8(rdc[q], d);
            unpack8(ryc[q], y);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float z = fmaf(y[i], psc[i], psh[i]);
              const float v = (ok && z > 0.f) ? (g8[i] + d[i]) : 0.f;
              d[i] = v;
              s1[i] += v;
              s2[i] = fmaf(v, y[i], s2[i]);        // against the raw y; converted to xhat at the end of the slab
            }
            uint4 w;
            w.x = f2bf_pk(d[0], d[1]); w.y = f2bf_pk(d[2], d[3]); w.z = f2bf_pk(d[4], d[5]); w.w = f2bf_pk(d[6], d[7]);
            const u32x4_t wv = {w.x, w.y, w.z, w.w};
            // (scalar offset 0, the slab in the vector offset: a 16-byte store with an SGPR scalar offset gets no hazard slots from
            //  hipcc before a VALU write of its data registers, and on gfx950 the store then reads the new contents — see
            //  se_combine_fwd_v3_kernel, tools/check_asm_hazards.py check_store_data)
            if (ok) __builtin_amdgcn_raw_buffer_store_b128(wv, srdY, (((p0 + o) * a.N) + c0 + slab * V2_C) * (int)sizeof(bf16_t), 0, 0);
          }
          // the next tile's operands (next tile of this slab, first tile of the next slab, or — last tile of the pass — this tile
          // again), issued BEHIND this tile's stores into the registers just consumed: they have the next tile's MFMA phase
          // and two barriers to land.  (Issued at the top of the step into a second register set, the wait for them stood in
          // front of a copy at the end of the step with this tile's stores in between — and `vmcnt` is in order only among
          // loads: with the stores made unconditional through out-of-range offsets, a store instruction whose every lane is
          // out of range retires at once, vmcnt(4) was satisfied with loads still in flight, and a few gradients per launch
          // were garbage.)
          const bool more_t = tt + V2_R < prow, more_s = slab + 1 < nslabs;
          epi_fetch(p0, prow, more_t ? slab : (more_s ? slab + 1 : slab), more_t ? tt + V2_R : (more_s ? 0 : tt), rdc, ryc);
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
  if (EPI == 2 && (size_t)a.M * a.N * sizeof(bf16_t) >= ((size_t)1 << 31)) return -1000;      // 32-bit buffer offsets
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
  // (the record count goes through readfirstlane: computed in vector registers it made the descriptor "divergent" and hipcc
  //  wrapped every load of the stream in a waterfall loop)
  const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W), 0, NO * a.KW * (int)sizeof(bf16_t), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.A), 0, __builtin_amdgcn_readfirstlane((int)((size_t)r_end * a.KW * sizeof(bf16_t))), 0x00020000);
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
          // Order and form of the loads matter to where hipcc waits (vmcnt is in order, and a load under a condition makes it
          // wait for everything): the next weight pieces go out AFTER the prefetched A tile has been consumed (issued in front
          // of it, the A tile's vmcnt(0) also waited for them: an exposed L2 round trip per tile step) and BEFORE the next A
          // tile (their store behind the MFMAs then waits with vmcnt(4), the A tile stays in flight); both are unconditional —
          // past the end they re-read the current chunk / tile
          if (t < 4) w_load2(more_w ? kc + 1 : kc, 2 * t, wp);
          {
            const bool nt_ = t + 1 < ntile;
            const int kcn = nt_ ? kc : (kc + 1 < nchunks ? kc + 1 : kc), tn_ = nt_ ? t + 1 : (kc + 1 < nchunks ? 0 : t);
            a_fetch(kcn, tn_);
          }
          __syncthreads();
          const bf16_t* brow = As + (rh * 32 + (lane & 31)) * V2_AP + half * 8;
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) {
            const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(brow + ks * 16);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b, acc[t], 0, 0, 0);
          }
          if (t < 4 && more_w) w_store2(2 * t, wp);       // (the last chunk's re-read pieces are dropped)
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

// ------------------------------------------------------------------------------------------
// asp_v2: attentive statistics pooling WITHOUT a stored energy tensor (round 5).  The energies e = W_out hid + b_out
// (reference src/models.py:562-567; K = 128, [M][1536]) used to be written by wide_out_v2<128, 0> and read back by
// asp_pool_fwd_kernel / asp_bwd_de_kernel: 3 passes over an A_D-sized tensor (236 MB at 256 x 300) that exist only to carry a
// K = 128 product between launches.  Here both consumers recompute them: one workgroup per (utterance, group of 256-channel
// slabs) stages the utterance's tanh outputs ([T <= 320][128] bf16, 82 KB) in LDS once, keeps the slab's 256 x 128 weights in
// registers as MFMA A fragments, and works on the f32 accumulators IN THEIR MFMA LAYOUT (lane = frame, 16 registers = 16
// channels): the epilog output E is loaded in that layout, so nothing is staged through LDS and the waves never meet after
// the staging barrier.  The weight ROWS of a wave's 32-channel block are permuted (MFMA row 8g + 4h + j holds channel
// 16 (r >> 3) + 8 h + (r & 7), r = 4g + j) so that a lane's 16 accumulator registers are two runs of 8 CONSECUTIVE channels:
// 16-byte loads / stores, 32 contiguous bytes per frame and instruction (with the natural order a lane owns 4-channel pieces:
// 8-byte accesses that cover half a 32-byte sector each — the backward kernel wrote its two tensors at 2.3 TB/s).
// Reductions over time are lane-local until the end of a slab (one 32-lane butterfly per statistic).
//   MODE 0 (forward): pass 1 = column maxima of the energies (MFMA only), pass 2 = exp / weighted sums -> mean, std, and what
//           the backward needs (max, 1 / sum, q = sum alpha x^2) + the BatchNorm1d(2D) batch statistics of the pooled vector.
//   MODE 1 (backward, element-wise part; asp_bwd_de_kernel's math): d e -> dEN, direct d x -> DXD, column sums -> d b_out.
// Both directions compute the energies with the same MFMA sequence: the softmax weights of the backward are bit-identical to
// the forward's (with the stored tensor they agreed through the same bf16 rounding).
// ------------------------------------------------------------------------------------------
struct AspV2Args {
  const bf16_t* HID;    // [M][128] tanh outputs
  const bf16_t* W;      // [D][128] W_out, row = channel
  const float* bias;    // [D]
  const bf16_t* E;      // [M][D] raw epilog output
  BnAct actE;           // its BatchNorm (+ ReLU); rm.len = valid frames per utterance or null
  float* pooled;        // [B][2D] mean | std            (MODE 1: input)
  float* smax;          // [B][D] softmax maxima         (MODE 1: input)
  float* sinv;          // [B][D] 1 / softmax sums       (MODE 1: input)
  float* qv;            // [B][D] sum alpha x^2          (MODE 1: input)
  float* stats;         // MODE 0: [TN_NREP][2][2D] batch statistics of `pooled`, or null
  const float* dpooled; // MODE 1: [B][2D]
  bf16_t* dEN;          // MODE 1: [M][D] gradient wrt the energies
  bf16_t* DXD;          // MODE 1: [M][D] direct gradient wrt x = act(E)
  float* g_bout;        // MODE 1: [D] += column sums of dEN
  int B, T, D, spw;     // spw: 256-channel slabs per workgroup (divides D / 256)
  float eps;
  int force_exact;      // MODE 0 test hook (TN_ASP_EXACT=1): always take the exact-maxima pass
};
#ifndef TN_ASP_ST_AUX
#define TN_ASP_ST_AUX 0      // cache policy bits of the backward kernel's stores.  Measured: 2 / 3 (nt) 203 -> 550 us — a store
                             // instruction covers 32 bytes of 32 different lines; only the L2 merges them into whole lines
#endif
constexpr int ASPV2_PR = 320;         // frames staged per utterance (T <= ASPV2_PR)
template <int MODE>
__global__ __launch_bounds__(V2_NT, 2) void asp_v2_kernel(AspV2Args a) {
  constexpr int K = 128, AP = K + 8, KS = K / 16, PR = ASPV2_PR, NK = MODE == 0 ? 3 : 6;
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);                    // [PR][AP] tanh outputs of the utterance
  float* par = reinterpret_cast<float*>(As + PR * AP);             // [spw][NK][256] per-channel constants
  __shared__ int wide_range;                                       // MODE 0: some channel's energies may span more than e^64
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int nsg = (a.D / V2_C) / a.spw;
  const int b = (int)blockIdx.x / nsg, slab0 = ((int)blockIdx.x % nsg) * a.spw;
  const int L = a.actE.rm.len ? tn_sload_i32(a.actE.rm.len, b) : a.T;       // softmax over the valid frames only
  const int row0 = b * a.T;
  if (tid == 0) wide_range = a.force_exact;
  {
    constexpr int NSTG = PR * (K / 8) / V2_NT;       // 10
    uint4 st[NSTG];
#pragma unroll
    for (int q = 0; q < NSTG; ++q) {
      const int v = tid + q * V2_NT, r = v >> 4, cv = v & 15;
      st[q] = r < L ? *reinterpret_cast<const uint4*>(a.HID + (size_t)(row0 + r) * K + cv * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < NSTG; ++q) {
      const int v = tid + q * V2_NT, r = v >> 4, cv = v & 15;
      *reinterpret_cast<uint4*>(As + r * AP + cv * 8) = st[q];
    }
  }
  __syncthreads();
  for (int i = tid; i < a.spw * V2_C; i += V2_NT) {
    const int si = i >> 8, cl = i & 255, c = (slab0 + si) * V2_C + cl;
    float* pk = par + (size_t)si * NK * V2_C + cl;
    float sc, sh;
    bn_scale_shift(a.actE, a.D, c, sc, sh);
    pk[0] = sc; pk[V2_C] = sh;
    if (MODE == 0) {
      // |tanh| <= 1: the channel's energies lie in b_c +- R_c, R_c = sum_a |W[c][a]| — a softmax shift that needs no pass over
      // the data (the result does not depend on the shift; the backward reads the one used from `smax`).  exp(e - shift) then
      // spans [e^(-2 R_c), 1]: kept while 2 R_c <= 64 (f32 has e^-87), else the workgroup takes the exact maxima (pass 1 below)
      float R = 0.f;
#pragma unroll
      for (int v = 0; v < K / 8; ++v) {
        float w[8];
        unpack8(*reinterpret_cast<const uint4*>(a.W + (size_t)c * K + v * 8), w);
#pragma unroll
        for (int u = 0; u < 8; ++u) R += fabsf(w[u]);
      }
      if (R > 32.f) wide_range = 1;
      pk[2 * V2_C] = R * LOG2E;                     // (shift of the bias-free energies: the accumulators start from zero)
    } else {
      const size_t o = (size_t)b * a.D + c;
      const float mu = a.pooled[(size_t)b * 2 * a.D + c], sg = a.pooled[(size_t)b * 2 * a.D + a.D + c];
      const float q = a.qv[o];
      const float gmu = a.dpooled[(size_t)b * 2 * a.D + c], gsg = a.dpooled[(size_t)b * 2 * a.D + a.D + c];
      const float dr = (q - mu * mu > a.eps) ? gsg / (2.f * sg) : 0.f;
      const float dmu = gmu - 2.f * mu * dr;
      const float iv = a.sinv[o];
      pk[2 * V2_C] = (a.smax[o] - (a.bias ? a.bias[c] : 0.f)) * LOG2E;      // alpha = exp2(acc log2e - this) * iv  (bias-free accumulators)
      pk[3 * V2_C] = iv * (dmu * mu + dr * q);      // d e = p (A1 x + A2 x^2 - A0),  d x = p (A1 + 2 A2 x),  p = alpha / iv
      pk[4 * V2_C] = iv * dmu;
      pk[5 * V2_C] = iv * dr;
    }
  }
  __syncthreads();
  const bool exact_max = MODE == 0 && wide_range != 0;
  const int ntile = ((MODE == 0 ? L : a.T) + V2_R - 1) / V2_R;
  const int M = a.B * a.T;
  const pg_i32x4_t srdE = pg_make_srd(a.E, (unsigned)((size_t)M * a.D * sizeof(bf16_t)));
  const __amdgpu_buffer_rsrc_t srdN = __builtin_amdgcn_make_buffer_rsrc(a.dEN, 0, MODE == 1 ? (int)((size_t)M * a.D * sizeof(bf16_t)) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc(a.DXD, 0, MODE == 1 ? (int)((size_t)M * a.D * sizeof(bf16_t)) : 0, 0x00020000);
  const int chl = wave * 32 + 8 * half;              // the lane's channels inside a slab: chl + 16 q + k  (register r = 8 q + k)
  const bool relu = a.actE.relu != 0;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  // E in the accumulator layout: frame tt + 32 h + l31 (clamped to the last valid frame: never used beyond it), 2 x 16 bytes.
  // The loads are inline asm with a HAND-COUNTED wait at the end of the step that issues them: left to hipcc, (i) a
  // conditional prefetch or store makes it wait for everything in flight (vmcnt(0) per tile: 246 us for the backward kernel),
  // (ii) a copy cur = nxt at the end of the step waits for the prefetch inside the step, (iii) the unpacking of the next
  // step's pieces floats up into this step and takes its wait along, (iv) at a loop header it merges the entry state (no
  // stores behind the first loads) with the back edge's (8 stores) and drains the stores every iteration.  The queue at the
  // end of a step: [the next step's 4 loads, issued at its top] [this step's NST stores] — vmcnt(NST) is exact; every
  // memory instruction of the loop is unconditional (clamped tile / out-of-range offset instead of a branch).
  constexpr int NST = MODE == 1 ? 8 : 0;
  auto e_load = [&](int slab, int tt, u32x4_t (&ex)[2][2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int r = tt + 32 * h + l31;
      r = r < L ? r : (L > 0 ? L - 1 : 0);
      const unsigned voff = (unsigned)(((row0 + r) * a.D + chl) * (int)sizeof(bf16_t));
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((slab * V2_C + 16 * q) * (int)sizeof(bf16_t));
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ex[h][q]) : "v"(voff), "s"(srdE), "s"(soff) : "memory");
      }
    }
  };
  auto tile_mfma = [&](const bf16x8_t (&wf)[KS], int tt, f32x16_t& acc0, f32x16_t& acc1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }      // (the bias lives in the softmax shift)
    const bf16_t* brow = As + (tt + l31) * AP + half * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(brow + ks * 16);
      const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(brow + 32 * AP + ks * 16);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b1, acc1, 0, 0, 0);
    }
  };
  // MFMA row m = l31 of the wave's block <- channel pi(m)
  const int pim = [&] { const int g = l31 >> 3, hh = (l31 >> 2) & 1, j = l31 & 3, r = 4 * g + j; return 16 * (r >> 3) + 8 * hh + (r & 7); }();
  for (int si = 0; si < a.spw; ++si) {
    const int slab = slab0 + si;
    const float* pk = par + (size_t)si * NK * V2_C;
    bf16x8_t wf[KS];
    {
      const int co = slab * V2_C + wave * 32 + pim;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wf[ks] = *reinterpret_cast<const bf16x8_t*>(a.W + (size_t)co * K + ks * 16 + half * 8);
    }
    u32x4_t ec[2][2], en[2][2] = {};
    e_load(slab, 0, ec);
    if (MODE == 0) {
      if (exact_max) {
        // ---- pass 1 (rare: see wide_range): column maxima of the energies over the valid frames
        float mx[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) mx[r] = -INFINITY;
        for (int t = 0; t < ntile; ++t) {
          f32x16_t acc0, acc1;
          tile_mfma(wf, t * V2_R, acc0, acc1);
          const bool v0 = t * V2_R + l31 < L, v1 = t * V2_R + 32 + l31 < L;
#pragma unroll
          for (int r = 0; r < 16; ++r) mx[r] = fmaxf(mx[r], fmaxf(v0 ? acc0[r] : -INFINITY, v1 ? acc1[r] : -INFINITY));
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o, 64));
        // the wave's own 32 channels of the shift table (no other wave reads them; LDS operations of a wave are in order)
        if (l31 == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) par[(size_t)si * NK * V2_C + 2 * V2_C + chl + 16 * (r >> 3) + (r & 7)] = mx[r] * LOG2E;
        }
      }
    }
    // ---- the pass over E
    float s0[16], s1[16], s2[16];      // MODE 0: sum p, sum p x, sum p x^2;  MODE 1: s0 = column sums of d e
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; s2[r] = 0.f; }
    // one tile; `cur` holds its E pieces (loaded a tile ago), the next tile's go to `nxt`.  The two register sets swap roles
    // in a loop unrolled by two: a copy cur = nxt at the end of the step made hipcc wait for the prefetch inside the step
    auto step = [&](auto pf, int t, u32x4_t (&cur)[2][2], u32x4_t (&nxt)[2][2]) {
      const int tt = t * V2_R;
      if constexpr (decltype(pf)::value) e_load(slab, (t + 1) * V2_R, nxt);      // (a step with prefetch is never the last tile)
      f32x16_t acc[2];
      tile_mfma(wf, tt, acc[0], acc[1]);
      // (scheduling fences: left alone hipcc unpacks all 64 E values of the step among the MFMAs — 64 more live registers,
      //  spills, and with a spill reload in the loop a vmcnt(0) that also drains the prefetch; the other wave of the SIMD fills
      //  the MFMA shadow instead)
      __builtin_amdgcn_sched_barrier(0);
      // the per-channel constants are read from LDS where they are used: hoisted out of the tile loop (they are invariant)
      // they would occupy 32 (forward) / 96 (backward) registers next to the weight fragments — an opaque offset per tile
      int kch = chl;
      asm volatile("" : "+v"(kch));
      u32x4_t on[2][2], ox[2][2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (q) __builtin_amdgcn_sched_barrier(0);
        float sc[8], sh[8], mv[8], a0[8], a1[8], a2[8];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          *reinterpret_cast<float4*>(sc + 4 * w) = *reinterpret_cast<const float4*>(pk + kch + 16 * q + 4 * w);
          *reinterpret_cast<float4*>(sh + 4 * w) = *reinterpret_cast<const float4*>(pk + V2_C + kch + 16 * q + 4 * w);
          *reinterpret_cast<float4*>(mv + 4 * w) = *reinterpret_cast<const float4*>(pk + 2 * V2_C + kch + 16 * q + 4 * w);
          if (MODE == 1) {
            *reinterpret_cast<float4*>(a0 + 4 * w) = *reinterpret_cast<const float4*>(pk + 3 * V2_C + kch + 16 * q + 4 * w);
            *reinterpret_cast<float4*>(a1 + 4 * w) = *reinterpret_cast<const float4*>(pk + 4 * V2_C + kch + 16 * q + 4 * w);
            *reinterpret_cast<float4*>(a2 + 4 * w) = *reinterpret_cast<const float4*>(pk + 5 * V2_C + kch + 16 * q + 4 * w);
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool valid = tt + 32 * h + l31 < L;
          const float kill = valid ? 0.f : -1000.f;        // exp2(-1000) = 0: padded frames carry no weight
          float x[8];
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            x[2 * w] = __uint_as_float(cur[h][q][w] << 16);
            x[2 * w + 1] = __uint_as_float(cur[h][q][w] & 0xffff0000u);
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            x[k] = fmaf(x[k], sc[k], sh[k]);
            x[k] = relu ? fmaxf(x[k], 0.f) : x[k];
          }
          if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const int r = 8 * q + k;
              const float p = __builtin_amdgcn_exp2f(fmaf(acc[h][r], LOG2E, kill - mv[k]));
              s0[r] += p;
              const float px = p * x[k];
              s1[r] += px;
              s2[r] = fmaf(px, x[k], s2[r]);
            }
          } else {
            float de[8], dx[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float p = __builtin_amdgcn_exp2f(fmaf(acc[h][8 * q + k], LOG2E, kill - mv[k]));
              const float ax = a2[k] * x[k];
              de[k] = p * (fmaf(a1[k] + ax, x[k], -a0[k]));
              dx[k] = p * (a1[k] + 2.f * ax);
              s0[8 * q + k] += de[k];
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) { on[h][q][w] = f2bf_pk(de[2 * w], de[2 * w + 1]); ox[h][q][w] = f2bf_pk(dx[2 * w], dx[2 * w + 1]); }
          }
        }
      }
      if (MODE == 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          // frames [L, T): zeros (p = 0).  Frames >= T belong to the next utterance: an offset beyond the buffer drops the store
          const int fr = tt + 32 * h + l31;
          const int voff = fr < a.T ? ((row0 + fr) * a.D + chl) * (int)sizeof(bf16_t) : 0x7ffffff0;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            // (scalar offset 0: see the store-data hazard note in wide_out_v2_kernel; the out-of-range sentinel stays out of range)
            const int vo = voff + (slab * V2_C + 16 * q) * (int)sizeof(bf16_t);
            __builtin_amdgcn_raw_buffer_store_b128(on[h][q], srdN, vo, 0, TN_ASP_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(ox[h][q], srdX, vo, 0, TN_ASP_ST_AUX);
          }
        }
      }
      // the prefetch has had the whole step to land; it is awaited HERE, at the end of the step that issued it, not at the
      // top of the one that consumes it: between the two hipcc may COPY the registers (it did, on the way into the peeled last
      // step: v_mov of pieces still in flight — utterances with garbage statistics, a few per launch)
      // vmcnt counts loads in order among loads and stores among stores — NOT across the two: a store instruction whose every
      // lane is out of range (frames >= T: only in an utterance's LAST tile) retires at once, and vmcnt(8) would then be
      // satisfied with loads still in flight (found in wide_out_v2's EPI 2, same trick).  Hence the loop structure below: the
      // last tile is always a step without prefetch and without a counted wait.
      if constexpr (decltype(pf)::value) {
        if (NST == 8) asm volatile("s_waitcnt vmcnt(8)" : "+v"(nxt[0][0]), "+v"(nxt[0][1]), "+v"(nxt[1][0]), "+v"(nxt[1][1]) : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt[0][0]), "+v"(nxt[0][1]), "+v"(nxt[1][0]), "+v"(nxt[1][1]) : : "memory");
      }
    };
    {
      // (the first step's loads have no stores behind them: drained here, behind the weights / bias / pass-1 work)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(ec[0][0]), "+v"(ec[0][1]), "+v"(ec[1][0]), "+v"(ec[1][1]) : : "memory");
      // ... and the COMPILER must see the weight fragments as arrived here: it puts its wait in front of the first use of a
      // loaded register, which would be the MFMAs inside the loop — vmcnt(7 - ks), i.e. a drain of the prefetch per tile
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(wf[ks]));
      int t = 0;
      for (; t + 2 < ntile; t += 2) { step(std::true_type{}, t, ec, en); step(std::true_type{}, t + 1, en, ec); }
      // the last one or two tiles; the last without prefetch (a load nobody awaits would land in registers the compiler has
      // handed to other values by then)
      if (t + 1 < ntile) { step(std::true_type{}, t, ec, en); step(std::false_type{}, t + 1, en, ec); }
      else if (t < ntile) step(std::false_type{}, t, ec, en);
    }
    // ---- end of the slab: the 32 frame lanes of each half add up, lane r of a half then owns channel register r
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] += __shfl_xor(s0[r], o, 64);
        if (MODE == 0) { s1[r] += __shfl_xor(s1[r], o, 64); s2[r] += __shfl_xor(s2[r], o, 64); }
      }
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, tm = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool me = l31 == r;
      t0 = me ? s0[r] : t0;
      if (MODE == 0) { t1 = me ? s1[r] : t1; t2 = me ? s2[r] : t2; }
    }
    if (l31 < 16) {
      const int cg = slab * V2_C + chl + 16 * (l31 >> 3) + (l31 & 7);
      if (MODE == 0) {
        tm = pk[2 * V2_C + chl + 16 * (l31 >> 3) + (l31 & 7)];
        const float inv = 1.f / t0;
        const float mu = t1 * inv, q = t2 * inv;
        const float sd = sqrtf(fmaxf(q - mu * mu, a.eps));
        a.pooled[(size_t)b * 2 * a.D + cg] = mu;
        a.pooled[(size_t)b * 2 * a.D + a.D + cg] = sd;
        a.smax[(size_t)b * a.D + cg] = tm * (1.f / LOG2E) + (a.bias ? a.bias[cg] : 0.f);
        a.sinv[(size_t)b * a.D + cg] = inv;
        a.qv[(size_t)b * a.D + cg] = q;
        if (a.stats) {
          const int rep = b % TN_NREP;
          atomic_add_f32(&a.stats[(size_t)(rep * 2 + 0) * 2 * a.D + cg], mu);
          atomic_add_f32(&a.stats[(size_t)(rep * 2 + 1) * 2 * a.D + cg], mu * mu);
          atomic_add_f32(&a.stats[(size_t)(rep * 2 + 0) * 2 * a.D + a.D + cg], sd);
          atomic_add_f32(&a.stats[(size_t)(rep * 2 + 1) * 2 * a.D + a.D + cg], sd * sd);
        }
      } else {
        atomic_add_f32(&a.g_bout[cg], t0);
      }
    }
  }
}

// -1000: shape outside the kernel (the caller runs the stored-energies path)
template <int MODE>
inline int launch_asp_v2(AspV2Args a, hipStream_t st) {
  const int nslab = a.D / V2_C;
  if (a.D % V2_C || a.T > ASPV2_PR || a.actE.drop_thr || (size_t)a.B * a.T * a.D * sizeof(bf16_t) >= ((size_t)1 << 31)) return -1000;
  // slabs per workgroup: the fewest workgroups that still give every CU two rounds (the staging of an utterance's tanh
  // outputs is paid once per workgroup)
  int spw = 1;
  for (int s = nslab; s >= 1; --s)
    if (nslab % s == 0 && (long)a.B * (nslab / s) >= 512) { spw = s; break; }
  a.spw = spw;
  { const char* e = getenv("TN_ASP_EXACT"); a.force_exact = e ? atoi(e) : 0; }
  { const char* es = getenv("TN_ASP_SPW"); if (es && atoi(es) > 0 && nslab % atoi(es) == 0) a.spw = spw = atoi(es); }      // (debug)
  const size_t smem = (size_t)ASPV2_PR * (128 + 8) * sizeof(bf16_t) + (size_t)spw * (MODE == 0 ? 3 : 6) * V2_C * sizeof(float);
  if (smem > 160 * 1024) return -1000;
  auto kern = asp_v2_kernel<MODE>;
  TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kern, dim3(a.B * (nslab / spw)), dim3(V2_NT), smem, st, a);
  return (int)hipGetLastError();
}
