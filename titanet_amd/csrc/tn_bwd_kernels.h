// titanet_amd — backward-pass kernels.  The reference defines no backward (it is PyTorch autograd
// of src/models.py / src/modules.py / src/losses.py, reference src/learn.py:117); the formulas
// below are the analytic gradients of those forward definitions.
#pragma once
#ifndef TN_NT_F8C
#define TN_NT_F8C 0      // tuning only: fp8 plans store the per-column-scaled e4m3 copy of dS non-temporal (measured: no effect on L/5 fp8)
#endif
#include <algorithm>

#include "../../include/titanet_amd.h"
#include "tn_common.h"
#include "tn_gemm.h"
#include "tn_internal.h"

// ------------------------------------------------------------------------------------------
// BatchNorm backward "on load": a stored tensor dz = d loss / d BN-output becomes
//   dy = gamma*rstd*(dz - mean_r(dz) - yhat*mean_r(dz*yhat)) = k0*dz + k1*y + k2      (train)
//   dy = gamma*rstd*dz                                                                 (eval)
// per channel, from the forward sums and the backward sums (sum dz, sum dz*yhat) that the
// kernel PRODUCING dz accumulated.  d gamma = sum dz*yhat, d beta = sum dz.
// ------------------------------------------------------------------------------------------
struct BnBwd {
  const float* fstats;   // forward  [TN_NREP][2][C] (eval: equivalent sums of the running statistics)
  const float* bsums;    // backward [TN_NREP][2][C]
  const float* gamma;
  float inv_n, eps;
  float batch;           // 1.f: batch statistics (train) -> mean/variance terms; 0.f: fixed statistics (eval)
  RowMask rm;            // padded rows have no gradient (dy = 0 there, not k1*y + k2)
};

__device__ __forceinline__ void bn_fwd_mean_rstd(const BnBwd& b, int C, int c, float& mean, float& rstd) {
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int r = 0; r < TN_NREP; ++r) { s += b.fstats[(r * 2 + 0) * C + c]; q += b.fstats[(r * 2 + 1) * C + c]; }
  mean = s * b.inv_n;
  rstd = rsqrtf(fmaxf(q * b.inv_n - mean * mean, 0.f) + b.eps);
}
__device__ __forceinline__ void bn_bwd_coefs(const BnBwd& b, int C, int c, float& k0, float& k1, float& k2) {
  float mean, rstd;
  bn_fwd_mean_rstd(b, C, c, mean, rstd);
  k0 = b.gamma[c] * rstd;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int r = 0; r < TN_NREP; ++r) { s1 += b.bsums[(r * 2 + 0) * C + c]; s2 += b.bsums[(r * 2 + 1) * C + c]; }
  const float c1 = s1 * b.inv_n * b.batch, c2 = s2 * b.inv_n * b.batch;   // branch-free: zero in eval mode
  k1 = -k0 * c2 * rstd;
  k2 = k0 * (c2 * rstd * mean - c1);
}

// P_DY: A[r][c] = k0[c]*dZ[r][c] + k1[c]*Y[r][c] + k2[c]
struct ProdDy {
  struct Args {
    const void* dZ;   // [M][ld]
    const void* Y;    // [M][ld] raw forward output of the same layer
    int ld;
    BnBwd bn;
  };
  float* k0;
  float* k1;
  float* k2;
  __host__ __device__ static size_t scratch_bytes(int K, int, int, int, size_t) { return (size_t)3 * K * sizeof(float); }
  template <typename AT, int NT, int CW>
  __device__ __forceinline__ void init(const Args& a, int M, int K, char* scratch, int tid) {
    k0 = reinterpret_cast<float*>(scratch);
    k1 = k0 + K;
    k2 = k1 + K;
    for (int c = tid; c < K; c += NT) bn_bwd_coefs(a.bn, K, c, k0[c], k1[c], k2[c]);
    __syncthreads();
  }
  // bf16 pipelined loop of gemm_nt_kernel (tn_gemm.h); two input streams: 128 x 512 workgroup tiles for wide outputs
  static constexpr bool kRaw = true, kWideCols = true;
  template <int N> struct Regs { uint4 z[N]; uint4 y[N]; };
  template <int ROWS, int NT, int CW, int N>
  __device__ __forceinline__ void load_raw(Regs<N>& rg, const Args& a, int M, int K, int tid, int r0, int kc) {
    constexpr int VC = CW / 8, RL = NT / VC;
    const int vc = tid % VC, rl = tid / VC, k = kc + vc * 8;
    const bf16_t* dZ = reinterpret_cast<const bf16_t*>(a.dZ);
    const bf16_t* Y = reinterpret_cast<const bf16_t*>(a.Y);
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const int gr = r0 + rl + q * RL;
      const bool ok = gr < M && k < K;
      const size_t o = (size_t)gr * a.ld + k;
      rg.z[q] = ok ? *reinterpret_cast<const uint4*>(dZ + o) : make_uint4(0, 0, 0, 0);
      rg.y[q] = ok ? *reinterpret_cast<const uint4*>(Y + o) : make_uint4(0, 0, 0, 0);
    }
  }
  template <int ROWS, int NT, int CW, int PITCH, int N>
  __device__ __forceinline__ void commit_raw(bf16_t* As, const Regs<N>& rg, const Args& a, int M, int K, int tid, int r0, int kc) {
    constexpr int VC = CW / 8, RL = NT / VC;
    const int vc = tid % VC, rl = tid / VC, k = kc + vc * 8;
    // coefficients of the thread's 8 columns: unconditional 16-byte LDS reads (K is a multiple of 8 on this path)
    float c0[8], c1[8], c2[8];
    const int kk = k < K ? k : 0;
    *reinterpret_cast<float4*>(c0) = *reinterpret_cast<const float4*>(k0 + kk); *reinterpret_cast<float4*>(c0 + 4) = *reinterpret_cast<const float4*>(k0 + kk + 4);
    *reinterpret_cast<float4*>(c1) = *reinterpret_cast<const float4*>(k1 + kk); *reinterpret_cast<float4*>(c1 + 4) = *reinterpret_cast<const float4*>(k1 + kk + 4);
    *reinterpret_cast<float4*>(c2) = *reinterpret_cast<const float4*>(k2 + kk); *reinterpret_cast<float4*>(c2 + 4) = *reinterpret_cast<const float4*>(k2 + kk + 4);
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const int r = rl + q * RL, gr = r0 + r;
      const bool ok = gr < M && k < K && tn_row_valid(a.bn.rm, (uint32_t)gr);
      const uint4 z = rg.z[q], y = rg.y[q];
      const uint32_t zw[4] = {z.x, z.y, z.z, z.w}, yw[4] = {y.x, y.y, y.z, y.w};
      float v[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a0 = fmaf(c0[2 * i], __uint_as_float(zw[i] << 16), fmaf(c1[2 * i], __uint_as_float(yw[i] << 16), c2[2 * i]));
        const float a1 = fmaf(c0[2 * i + 1], __uint_as_float(zw[i] & 0xffff0000u), fmaf(c1[2 * i + 1], __uint_as_float(yw[i] & 0xffff0000u), c2[2 * i + 1]));
        v[2 * i] = ok ? a0 : 0.f; v[2 * i + 1] = ok ? a1 : 0.f;
      }
      store8(As + r * PITCH + vc * 8, v);
    }
  }
  template <typename AT, int ROWS, int NT, int CW, int PITCH>
  __device__ __forceinline__ void fill(AT* As, const Args& a, int M, int K, int tid, int r0, int kc) {
    constexpr int VC = CW / 8, RL = NT / VC;
    const int vc = tid % VC, rl = tid / VC;
    const int k = kc + vc * 8;
    const AT* dZ = reinterpret_cast<const AT*>(a.dZ);
    const AT* Y = reinterpret_cast<const AT*>(a.Y);
    for (int r = rl; r < ROWS; r += RL) {
      float v[8];
      const int gr = r0 + r;
      if (gr < M && k < K && tn_row_valid(a.bn.rm, (uint32_t)gr)) {
        float y[8];
        load8(dZ + (size_t)gr * a.ld + k, v);
        load8(Y + (size_t)gr * a.ld + k, y);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = k0[k + i] * v[i] + k1[k + i] * y[i] + k2[k + i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
      }
      store8_lds(As + r * PITCH + vc * 8, v);
    }
  }
};

// ------------------------------------------------------------------------------------------
// dgrad epilogues
// ------------------------------------------------------------------------------------------
// d hidden_pre = acc * (1 - h^2)  (tanh backward, reference src/models.py:565), column sums -> d b_in
struct EpiTanhBwd {
  struct Args {
    void* Y;            // [M][ldy] AT out
    int ldy;
    const void* H;      // [M][ldy] AT tanh outputs
    float* colsum;      // [N] float, atomically accumulated (d bias) or null
  };
  template <typename AT, int WM, int WN>
  static constexpr size_t lds_bytes() { return EpiStore::lds_bytes<AT, WM, WN>(); }
  template <typename AT, int WM, int WN>
  __device__ static __forceinline__ void run(f32x16_t (&acc)[2][2], const Args& e, const GemmShape& g, char* smem,
                                             int tid, int r0, int n0) {
    constexpr int BN = WN * 64, NT = WM * WN * 64, CSP = BN + 8;
    AT* Cs = reinterpret_cast<AT*>(smem);
    float* colsum = reinterpret_cast<float*>(smem + (size_t)64 * CSP * sizeof(AT));
    const int lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
    for (int i = tid; i < BN; i += NT) colsum[i] = 0.f;
    __syncthreads();
    const AT* H = reinterpret_cast<const AT*>(e.H);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int n = wn * 64 + nt * 32 + (lane & 31);
      float s = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r0 + wm * 64 + mt * 32 + cd_row(r, lane);
          float y = 0.f;
          if (row < g.M && n0 + n < g.N) {
            const float h = Elem<AT>::to_f(H[(size_t)row * e.ldy + n0 + n]);
            y = acc[mt][nt][r] * (1.f - h * h);
            s += y;
          }
          acc[mt][nt][r] = y;
        }
      s += __shfl_xor(s, 32, 64);
      if (lane < 32) atomicAdd(&colsum[n], s);
    }
    AT* Y = reinterpret_cast<AT*>(e.Y);
    for (int pass = 0; pass < WM; ++pass) {
      __syncthreads();
      if (wm == pass) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              Cs[(mt * 32 + cd_row(r, lane)) * CSP + wn * 64 + nt * 32 + (lane & 31)] = Elem<AT>::from_f(acc[mt][nt][r]);
      }
      __syncthreads();
      constexpr int VCN = BN / 8, RL = NT / VCN;
      const int vc = tid % VCN, rl = tid / VCN;
      for (int r = rl; r < 64; r += RL) {
        const int row = r0 + pass * 64 + r, n = n0 + vc * 8;
        if (row < g.M && n < g.N) {
          const uint4* src = reinterpret_cast<const uint4*>(Cs + r * CSP + vc * 8);
          uint4* dst = reinterpret_cast<uint4*>(Y + (size_t)row * e.ldy + n);
          dst[0] = src[0];
          if (sizeof(AT) == 4) dst[1] = src[1];
        }
      }
    }
    if (e.colsum) {
      for (int i = tid; i < BN; i += NT)
        if (n0 + i < g.N) atomic_add_f32(&e.colsum[n0 + i], colsum[i]);
    }
  }
};

// out = (acc + ADD[row][n]) * mask(RAW[row][n]);  accumulates sum(out), sum(out * yhat) for the
// BatchNorm backward of the layer that produced RAW.  Used for d(epilog BN output): the direct
// pooling gradient (ADD, written by asp_bwd_de_kernel into the output buffer itself) plus the
// attention-path gradient (this GEMM), through the epilog ReLU.
struct EpiAddMaskStore {
  struct Args {
    void* Y;            // [M][ldy] AT: in = addend, out = result (same buffer)
    int ldy;
    const void* RAW;    // [M][ldy] AT raw forward output (for the relu mask and yhat)
    BnAct act;          // activation of RAW (BN + relu)
    float* bsums;       // [TN_NREP][2][N]
  };
  template <typename AT, int WM, int WN>
  static constexpr size_t lds_bytes() {
    return (size_t)64 * (WN * 64 + 8) * sizeof(AT) + (size_t)6 * WN * 64 * sizeof(float);
  }
  template <typename AT, int WM, int WN>
  __device__ static __forceinline__ void run(f32x16_t (&acc)[2][2], const Args& e, const GemmShape& g, char* smem,
                                             int tid, int r0, int n0) {
    constexpr int BN = WN * 64, NT = WM * WN * 64, CSP = BN + 8;
    AT* Cs = reinterpret_cast<AT*>(smem);
    float* colsum = reinterpret_cast<float*>(smem + (size_t)64 * CSP * sizeof(AT));   // [2][BN]
    float* par = colsum + 2 * BN;                                                        // sc, sh, mean, rstd [4][BN]
    const int lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
    for (int i = tid; i < 2 * BN; i += NT) colsum[i] = 0.f;
    for (int i = tid; i < BN; i += NT) {
      float sc = 1.f, sh = 0.f, mean = 0.f, rstd = 1.f;
      if (n0 + i < g.N) {
        bn_scale_shift(e.act, g.N, n0 + i, sc, sh);
        bn_mean_rstd(e.act, g.N, n0 + i, mean, rstd);
      }
      par[i] = sc; par[BN + i] = sh; par[2 * BN + i] = mean; par[3 * BN + i] = rstd;
    }
    __syncthreads();
    AT* Y = reinterpret_cast<AT*>(e.Y);
    const AT* RAW = reinterpret_cast<const AT*>(e.RAW);
    // stage the accumulators through LDS so the add / mask / sums run on coalesced 8-channel vectors
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
    constexpr int VCN = BN / 8, RL = NT / VCN;
    const int vc = tid % VCN, rl = tid / VCN;
    for (int pass = 0; pass < WM; ++pass) {
      __syncthreads();
      if (wm == pass) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              Cs[(mt * 32 + cd_row(r, lane)) * CSP + wn * 64 + nt * 32 + (lane & 31)] = Elem<AT>::from_f(acc[mt][nt][r]);
      }
      __syncthreads();
      for (int r = rl; r < 64; r += RL) {
        const int row = r0 + pass * 64 + r, n = n0 + vc * 8;
        if (row < g.M && n < g.N) {
          float a[8], d[8], y[8];
          load8_lds(Cs + r * CSP + vc * 8, a);
          load8(Y + (size_t)row * e.ldy + n, d);
          load8(RAW + (size_t)row * e.ldy + n, y);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int c = vc * 8 + i;
            const float z = y[i] * par[c] + par[BN + c];
            const float o = (z > 0.f) ? (a[i] + d[i]) : 0.f;
            const float yh = (y[i] - par[2 * BN + c]) * par[3 * BN + c];
            s1[i] += o; s2[i] += o * yh;
            d[i] = o;
          }
          store8(Y + (size_t)row * e.ldy + n, d);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      atomicAdd(&colsum[vc * 8 + i], s1[i]);
      atomicAdd(&colsum[BN + vc * 8 + i], s2[i]);
    }
    __syncthreads();
    const int rep = blockIdx.x % TN_NREP;
    for (int i = tid; i < 2 * BN; i += NT) {
      const int which = i / BN, n = i % BN;
      if (n0 + n < g.N) atomic_add_f32(&e.bsums[(size_t)(rep * 2 + which) * g.N + n0 + n], colsum[i]);
    }
  }
};

// ------------------------------------------------------------------------------------------
// Weight-gradient ("TN") GEMM:  OUT[ca][cb] = sum over rows r of P[r][ca] * Q[r][cb].
// Both operand tiles are PRODUCED into LDS as row-major [rows][channels] (the same producers as the
// forward GEMM: BN-backward-on-load for P, activation (+ depthwise stencil / im2col) recompute for
// Q).  The contraction runs over rows, i.e. across the LDS rows: the bf16 MFMA fragments (8
// row-consecutive values per lane) are gathered with the gfx950 transpose read ds_read_b64_tr_b16;
// the f32 MFMA takes one value per lane and reads the row-major tile directly.
// Split-K over row ranges: every workgroup writes its partial 128x128 tile into its own slab, a
// second kernel adds the slabs in a fixed order (deterministic, no atomics).
// ------------------------------------------------------------------------------------------
template <typename AT> struct WgTile;
template <> struct WgTile<bf16_t> { static constexpr int RK = 32, PAD = 32; };   // 320-byte rows: 4 tr-read rows on distinct banks
template <> struct WgTile<float> { static constexpr int RK = 16, PAD = 4; };

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __forceinline__ bf16x8_t wg_frag(const bf16_t* tile, int pitch, int k0, int cbase, int lane) {
  // lane (16-lane group g = lane>>4, i = lane&15) addresses 4 bf16 of row k0 + 8*(lane>>5) + (i>>2) [+4],
  // columns cbase + 16*(g&1) + 4*(i&3); the transpose read returns column (cbase + 16*(g&1) + i), rows +0..3.
  const int i = lane & 15;
  const bf16_t* p = tile + (size_t)(k0 + 8 * (lane >> 5) + (i >> 2)) * pitch + cbase + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 4 * pitch));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  s16x8_t v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ float wg_frag(const float* tile, int pitch, int k0, int cbase, int lane) {
  return tile[(size_t)(k0 + (lane >> 5)) * pitch + cbase + (lane & 31)];
}

struct WgradShape {
  int M;        // rows (contraction length)
  int CA, CB;   // output [CA][CB]
  int rows_per_split;
  float* slabs; // [splits][CA][CB]
};

template <typename AT, typename ProdP, typename ProdQ>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradShape g, typename ProdP::Args pa, typename ProdQ::Args qa) {
  constexpr int TA = 128, TB = 128, NT = 256, RK = WgTile<AT>::RK, PITCH = 128 + WgTile<AT>::PAD, KM = Elem<AT>::KM;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AT* Pt = reinterpret_cast<AT*>(smem);
  AT* Qt = Pt + RK * PITCH;
  char* scratch = reinterpret_cast<char*>(Qt + RK * PITCH);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wa = wave >> 1, wb = wave & 1;
  const int tiles_b = (g.CB + TB - 1) / TB;
  const int ta = blockIdx.x / tiles_b, tb = blockIdx.x % tiles_b;
  const int split = blockIdx.y;
  const int row_begin = split * g.rows_per_split;
  const int row_end = min(g.M, row_begin + g.rows_per_split);

  ProdP pp;
  ProdQ pq;
  const size_t p_scratch = (ProdP::scratch_bytes(g.CA, 0, RK, TA, sizeof(AT)) + 15) & ~(size_t)15;
  pp.template init<AT, NT, TA>(pa, g.M, g.CA, scratch, tid);
  pq.template init<AT, NT, TB>(qa, g.M, g.CB, scratch + p_scratch, tid);

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int r0 = row_begin; r0 < row_end; r0 += RK) {
    // rows beyond row_end must not contribute: producers zero rows >= M; clip the split tail here
    const int mlim = row_end;
    pp.template fill<AT, RK, NT, TA, PITCH>(Pt, pa, mlim, g.CA, tid, r0, ta * TA);
    pq.template fill<AT, RK, NT, TB, PITCH>(Qt, qa, g.M, g.CB, tid, r0, tb * TB);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < RK / KM; ++ks) {
      auto a0 = wg_frag(Pt, PITCH, ks * KM, wa * 64, lane);
      auto a1 = wg_frag(Pt, PITCH, ks * KM, wa * 64 + 32, lane);
      auto b0 = wg_frag(Qt, PITCH, ks * KM, wb * 64, lane);
      auto b1 = wg_frag(Qt, PITCH, ks * KM, wb * 64 + 32, lane);
      acc[0][0] = Mma<AT>::mma(a0, b0, acc[0][0]);
      acc[0][1] = Mma<AT>::mma(a0, b1, acc[0][1]);
      acc[1][0] = Mma<AT>::mma(a1, b0, acc[1][0]);
      acc[1][1] = Mma<AT>::mma(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
  float* slab = g.slabs + (size_t)split * g.CA * g.CB;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ca = ta * TA + wa * 64 + mt * 32 + cd_row(r, lane);
        const int cb = tb * TB + wb * 64 + nt * 32 + (lane & 31);
        if (ca < g.CA && cb < g.CB) slab[(size_t)ca * g.CB + cb] = acc[mt][nt][r];
      }
}

// (8 slab loads in flight per thread: with a run-time trip count hipcc emits one load -> wait -> add per split, and the ~48
//  splits of the prolog weight gradient then cost 48 dependent L2 round trips: 37 us for 12 MB)
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, int splits, int64_t n, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k0 = 0; k0 < splits; k0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (k0 + u < splits) ? slabs[(size_t)(k0 + u) * n + i] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    out[i] = s;
  }
}

// gradient of the prolog weight computed in [out][tap][ci] order (ProdTaps) -> the reference layout [out][ci][tap]
__global__ void prolog_wgrad_untap_kernel(const float* __restrict__ g_taps, int H, int C, int KP, float* __restrict__ g_w) {
  const int n = H * C * KP;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int h = i / (C * KP), r = i - h * C * KP, ci = r / KP, j = r - ci * KP;   // i indexes the reference layout [h][ci][j]
    g_w[i] = g_taps[(size_t)h * C * KP + j * C + ci];
  }
}

template <typename AT, typename ProdP, typename ProdQ>
inline int launch_wgrad(int M, int CA, int CB, const typename ProdP::Args& pa, const typename ProdQ::Args& qa, int KD,
                        float* slabs, size_t slab_bytes, float* out, hipStream_t st, tn_plan* prof_plan = nullptr,
                        int prof_cls = 0) {
  constexpr int RK = WgTile<AT>::RK, PITCH = 128 + WgTile<AT>::PAD;
  const int tiles = ((CA + 127) / 128) * ((CB + 127) / 128);
  // aim at ~4 resident workgroups per CU (the kernel is latency-bound per workgroup: produce -> barrier
  // -> MFMA), bounded by the slab buffer
  int splits = (1024 + tiles - 1) / tiles;
  if (splits > 128) splits = 128;
  const int max_by_rows = (M + RK - 1) / RK;
  if (splits > max_by_rows) splits = max_by_rows;
  while ((size_t)splits * CA * CB * sizeof(float) > slab_bytes && splits > 1) --splits;
  int rps = (M + splits - 1) / splits;
  rps = ((rps + RK - 1) / RK) * RK;
  splits = (M + rps - 1) / rps;
  WgradShape g{M, CA, CB, rps, slabs};
  size_t smem = (size_t)2 * RK * PITCH * sizeof(AT);
  smem += (ProdP::scratch_bytes(CA, 0, RK, 128, sizeof(AT)) + 15) & ~(size_t)15;
  smem += ProdQ::scratch_bytes(CB, KD, RK, 128, sizeof(AT));
  smem = (smem + 15) & ~(size_t)15;
  auto kern = wgrad_kernel<AT, ProdP, ProdQ>;
  if (smem > 64 * 1024) {
    TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  }
  {
    ProfScope ps(prof_plan, prof_cls, st);
    hipLaunchKernelGGL(kern, dim3(tiles, splits), dim3(256), smem, st, g, pa, qa);
  }
  const int64_t n = (int64_t)CA * CB;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0, st, slabs, splits, n, out);
  return (int)hipGetLastError();
}

// ==========================================================================================
// element-wise / reduction kernels of the backward pass
// ==========================================================================================

// ------------------------------------------------------------------------------------------
// Mega-block tail backward, pass 1 (one workgroup per utterance):
//   dZ = dOUT * [OUT > 0] * inv_keep            (relu + dropout of reference src/models.py:467-472;
//                                                OUT > 0 <=> kept AND pre-activation > 0)
//   dgate[b][c] = sum_t dZ * A3                 (A3 = act3(Y3), the SE input)
//   skip-BN backward sums: sum dZ, sum dZ * shat
// The mask is RECOMPUTED from what this pass reads anyway (Y3 and S, + the SE gate and the dropout hash of the block
// output) instead of reading the block output back: one tensor pass less per mega block.
// ------------------------------------------------------------------------------------------
template <typename AT>
__global__ __launch_bounds__(512) void combine_bwd1_kernel(const AT* __restrict__ dOUT, const float* __restrict__ gate,
                                                           const AT* __restrict__ Y3, BnAct act3,
                                                           const AT* __restrict__ S, BnAct actS, int T, int C,
                                                           float inv_keep, uint32_t drop_thr, uint32_t drop_key, const uint32_t* key_add,
                                                           AT* __restrict__ dZ, float* __restrict__ dgate,
                                                           float* __restrict__ bsumsS) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sc3 = reinterpret_cast<float*>(smem);
  float* sh3 = sc3 + C;
  float* mS = sh3 + C;
  float* rS = mS + C;
  float* scS = rS + C;
  float* shS = scS + C;
  float* gS = shS + C;
  float* part = gS + C;   // [TG][3][C]
  const int tid = threadIdx.x, NT = blockDim.x, b = blockIdx.x;
  const int CV = C / 8, TG = NT / CV;
  act3 = tn_resolve_key(act3);
  for (int c = tid; c < C; c += NT) {
    bn_scale_shift(act3, C, c, sc3[c], sh3[c]);
    bn_mean_rstd(actS, C, c, mS[c], rS[c]);
    bn_scale_shift(actS, C, c, scS[c], shS[c]);
    gS[c] = gate[(size_t)b * C + c];
  }
  __syncthreads();
  const uint32_t okey = key_add ? drop_key + *key_add : drop_key;
  const int vc = tid % CV, tg = tid / CV, c0 = vc * 8;
  // gridDim.y workgroups per utterance (small batches of long utterances), each a range of frames; frames >= len[b] are
  // padding: not read, their gradient written as zero.  With more than one part dgate is ACCUMULATED (pre-zeroed buffer)
  const int L = act3.rm.len ? act3.rm.len[b] : T;
  const int per = (T + (int)gridDim.y - 1) / (int)gridDim.y;
  const int t_lo = (int)blockIdx.y * per, t_end = min(T, t_lo + per), t_hi = min(L, t_end);
  if (tg < TG) {
    float dg[8], s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { dg[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f; }
    // the thread's 8 channels are fixed: their constants in registers (40 scalar LDS reads per row otherwise)
    float kS[8], hS[8], g8[8], m8[8], r8[8], k3[8], h3[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      kS[i] = scS[c0 + i]; hS[i] = shS[c0 + i]; g8[i] = gS[c0 + i]; m8[i] = mS[c0 + i]; r8[i] = rS[c0 + i];
      k3[i] = sc3[c0 + i]; h3[i] = sh3[c0 + i];
    }
    constexpr int U = 2;
    for (int t0 = t_lo + tg; t0 < t_hi; t0 += TG * U) {
      float d[U][8], y[U][8], sv[U][8];
#pragma unroll
      for (int q = 0; q < U; ++q) {                  // (branch-free loads, see combine_bwd2_kernel)
        const int t = min(t0 + q * TG, t_hi - 1);
        const size_t o = ((size_t)b * T + t) * C + c0;
        load8(dOUT + o, d[q]);
        load8(Y3 + o, y[q]);
        load8(S + o, sv[q]);
      }
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int t = t0 + q * TG;
        if (t < t_hi) {
          const uint32_t row = (uint32_t)b * T + t;
          float u[8];
          act8(y[q], k3, h3, act3, row, C, c0);
          // u = 1 where the block output was positive: pre-activation > 0 (the forward's expression, up to its positive
          // 1/(1-p) factor), the element survived the block's dropout, and the row is a valid frame
#pragma unroll
          for (int i = 0; i < 8; ++i) u[i] = (sv[q][i] * kS[i] + hS[i] + g8[i] * y[q][i] > 0.f) ? 1.f : 0.f;
          if (drop_thr) tn_drop8(u, (row * (uint32_t)C + (uint32_t)c0) >> 3, okey, drop_thr);
          if (act3.rm.len && !tn_row_valid(act3.rm, row)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) u[i] = 0.f;
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float z = (u[i] > 0.f) ? d[q][i] * inv_keep : 0.f;
            d[q][i] = z;
            dg[i] += z * y[q][i];
            s1[i] += z;
            s2[i] += z * (sv[q][i] - m8[i]) * r8[i];
          }
          store8(dZ + (size_t)row * C + c0, d[q]);
        }
      }
    }
    // padding frames of this workgroup's range: no gradient (written, not read)
    {
      float z[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) z[i] = 0.f;
      for (int t = max(t_lo, L) + tg; t < t_end; t += TG) store8(dZ + ((size_t)b * T + t) * C + c0, z);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      part[(tg * 3 + 0) * C + c0 + i] = dg[i];
      part[(tg * 3 + 1) * C + c0 + i] = s1[i];
      part[(tg * 3 + 2) * C + c0 + i] = s2[i];
    }
  }
  __syncthreads();
  const int rep = b % TN_NREP;
  for (int c = tid; c < C; c += NT) {
    float a = 0.f, x1 = 0.f, x2 = 0.f;
    for (int k = 0; k < TG; ++k) { a += part[(k * 3 + 0) * C + c]; x1 += part[(k * 3 + 1) * C + c]; x2 += part[(k * 3 + 2) * C + c]; }
    if (gridDim.y > 1) atomic_add_f32(&dgate[(size_t)b * C + c], a);
    else dgate[(size_t)b * C + c] = a;
    atomic_add_f32(&bsumsS[(size_t)(rep * 2 + 0) * C + c], x1);
    atomic_add_f32(&bsumsS[(size_t)(rep * 2 + 1) * C + c], x2);
  }
}

// ------------------------------------------------------------------------------------------
// Mega-block tail backward, pass 2 (one workgroup per utterance): SE backward (tiny mat-vecs,
// reference src/modules.py:182-189) then
//   dA3 = dZ * g + dmean / T ;  dY3bn = dA3 * d act3 / d bn  ->  stored + BN3 backward sums.
// dgate_dpre2: in = dgate (from pass 1), out = d(pre-sigmoid) for the SE weight gradients.
// ------------------------------------------------------------------------------------------
template <typename AT>
__global__ __launch_bounds__(512) void combine_bwd2_kernel(const AT* __restrict__ dZ, const AT* __restrict__ Y3, BnAct act3,
                                                           const float* __restrict__ gate, const float* __restrict__ hid,
                                                           const float* dgate_in, float* dgate_dpre2, float* __restrict__ dpre1,
                                                           const float* __restrict__ W1, const float* __restrict__ W2,
                                                           int T, int C, int Hr, AT* __restrict__ dYbn,
                                                           float* __restrict__ bsums3) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sc3 = reinterpret_cast<float*>(smem);
  float* sh3 = sc3 + C;
  float* m3 = sh3 + C;
  float* r3 = m3 + C;
  float* gS = r3 + C;       // gate
  float* dmT = gS + C;      // dmean / T
  float* p2 = dmT + C;      // dpre2
  float* p1 = p2 + C;       // dpre1 [Hr]
  float* part = p1 + ((Hr + 3) & ~3);   // [TG][2][C]
  const int tid = threadIdx.x, NT = blockDim.x, b = blockIdx.x;
  const int CV = C / 8, TG = NT / CV;
  act3 = tn_resolve_key(act3);
  for (int c = tid; c < C; c += NT) {
    bn_scale_shift(act3, C, c, sc3[c], sh3[c]);
    bn_mean_rstd(act3, C, c, m3[c], r3[c]);
    const float g = gate[(size_t)b * C + c];
    gS[c] = g;
    // (dgate_in == dgate_dpre2: in place, one workgroup per utterance; with gridDim.y parts the two are different buffers
    //  — every part reads dgate, part 0 writes the per-utterance results)
    const float d2 = dgate_in[(size_t)b * C + c] * g * (1.f - g);
    p2[c] = d2;
    if (blockIdx.y == 0) dgate_dpre2[(size_t)b * C + c] = d2;
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, NW = NT >> 6;
  for (int j = wave; j < Hr; j += NW) {
    float s = lane < C ? tn_dot_batched(W2 + (size_t)lane * Hr + j, 64 * Hr, p2 + lane, 64, (C - lane + 63) / 64) : 0.f;
    s = wave_sum(s);
    if (lane == 0) {
      s = (hid[(size_t)b * Hr + j] > 0.f) ? s : 0.f;
      p1[j] = s;
      if (blockIdx.y == 0) dpre1[(size_t)b * Hr + j] = s;
    }
  }
  __syncthreads();
  const int L = act3.rm.len ? act3.rm.len[b] : T;
  const int per = (T + (int)gridDim.y - 1) / (int)gridDim.y;
  const int t_lo = (int)blockIdx.y * per, t_end = min(T, t_lo + per), t_hi = min(L, t_end);
  const float invT = 1.f / (float)max(L, 1);   // the SE mean ran over the valid frames
  for (int c = tid; c < C; c += NT) {
    const float s = tn_dot_batched(W1 + c, C, p1, 1, Hr);
    dmT[c] = s * invT;
  }
  __syncthreads();
  const int vc = tid % CV, tg = tid / CV, c0 = vc * 8;
  if (tg < TG) {
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
    // the thread's 8 channels are fixed: their constants live in registers (they were 32 scalar LDS reads per row), and 4
    // rows of both streams are in flight per thread (one workgroup per utterance: one row at a time is a round trip per row)
    float g8[8], dm8[8], m8[8], r8[8], k38[8], h38[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      g8[i] = gS[c0 + i]; dm8[i] = dmT[c0 + i]; m8[i] = m3[c0 + i]; r8[i] = r3[c0 + i]; k38[i] = sc3[c0 + i]; h38[i] = sh3[c0 + i];
    }
    constexpr int U = 4;
    for (int t0 = t_lo + tg; t0 < t_hi; t0 += TG * U) {
      float d[U][8], y[U][8];
      // (branch-free loads: rows past the end are clamped to the last one and weighted zero — predicated loads compile to one
      //  exec-masked block with its own wait per row)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = min(t0 + u * TG, t_hi - 1);
        const size_t o = ((size_t)b * T + t) * C + c0;
        load8(dZ + o, d[u]);
        load8(Y3 + o, y[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u * TG;
        const bool live = t < t_hi;
        const uint32_t row = (uint32_t)b * T + min(t, t_hi - 1);
        float m[8];
        act8_grad_mask(y[u], m, k38, h38, act3, row, C, c0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float v = live ? (d[u][i] * g8[i] + dm8[i]) * m[i] : 0.f;
          d[u][i] = v;
          s1[i] += v;
          s2[i] += v * (y[u][i] - m8[i]) * r8[i];
        }
        if (live) store8(dYbn + (size_t)row * C + c0, d[u]);
      }
    }
    {
      float z[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) z[i] = 0.f;
      for (int t = max(t_lo, L) + tg; t < t_end; t += TG) store8(dYbn + ((size_t)b * T + t) * C + c0, z);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      part[(tg * 2 + 0) * C + c0 + i] = s1[i];
      part[(tg * 2 + 1) * C + c0 + i] = s2[i];
    }
  }
  __syncthreads();
  const int rep = b % TN_NREP;
  for (int c = tid; c < C; c += NT) {
    float x1 = 0.f, x2 = 0.f;
    for (int k = 0; k < TG; ++k) { x1 += part[(k * 2 + 0) * C + c]; x2 += part[(k * 2 + 1) * C + c]; }
    atomic_add_f32(&bsums3[(size_t)(rep * 2 + 0) * C + c], x1);
    atomic_add_f32(&bsums3[(size_t)(rep * 2 + 1) * C + c], x2);
  }
}

// ------------------------------------------------------------------------------------------
// Depthwise conv backward (reference src/modules.py:65-75) fused with the backward of the
// activation that fed it:
//   dA[r]       = sum_k w[c][k] * dD[r - k + pad]                (transposed stencil over time)
//   dw[c][k]   += sum_r dD[r] * A[r + k - pad],  db[c] += sum_r dD[r]
//   out[r]      = (dA[r] + ADD[r]) * d act / d bn (XRAW[r])      -> stored; BN backward sums.
// LDS-tiled over time: dD and the re-activated input rows (with halo) are staged once per tile.
// ------------------------------------------------------------------------------------------
struct DwBwdArgs {
  const void* dD;        // [M][C] AT
  const void* XRAW;      // [M][C] AT raw input of the depthwise conv (before its activation)
  BnAct actX;            // activation applied to XRAW on load (identity for an already-activated tensor)
  const void* ADD;       // [M][C] AT or null (skip-connection data gradient)
  void* OUT;             // [M][C] AT
  const float* wdw;      // [C][KD]
  float* g_wdw;          // [C][KD] gradient (atomic accumulate, pre-zeroed)
  float* g_bdw;          // [C]
  float* bsumsX;         // [TN_NREP][2][C] or null (BN that produced XRAW)
  int M, T, C;
  int tiles_per_wg;
};

template <typename AT, int KD>
__global__ __launch_bounds__(256) void dw_bwd_kernel(DwBwdArgs a) {
  constexpr int RT = 64, CW = 64, NT = 256, PAD = (KD - 1) / 2, ROWS = RT + KD - 1, VC = CW / 8, RL = NT / VC;
  // + 8 floats per row: a wave covers 8 rows x 8 channel vectors; with a 256-byte row stride all 8 rows hit the same banks
  __shared__ __attribute__((aligned(16))) float dDs[ROWS][CW + 8];
  __shared__ __attribute__((aligned(16))) float As[ROWS][CW + 8];
  __shared__ float red[KD + 3][CW];
  __shared__ float par[4][CW];   // sc, sh, mean, rstd of actX
  __shared__ float wds[KD][CW];
  const int tid = threadIdx.x;
  const int cbase = blockIdx.y * CW;
  const int vc = tid % VC, rl = tid / VC, c0 = cbase + vc * 8;
  const AT* dD = reinterpret_cast<const AT*>(a.dD);
  const AT* XR = reinterpret_cast<const AT*>(a.XRAW);
  const AT* ADD = reinterpret_cast<const AT*>(a.ADD);
  AT* OUT = reinterpret_cast<AT*>(a.OUT);
  for (int i = tid; i < (KD + 3) * CW; i += NT) (&red[0][0])[i] = 0.f;
  for (int c = tid; c < CW; c += NT) {
    float sc = 1.f, sh = 0.f, mean = 0.f, rstd = 1.f;
    if (cbase + c < a.C && a.actX.mode != 0) {
      bn_scale_shift(a.actX, a.C, cbase + c, sc, sh);
      bn_mean_rstd(a.actX, a.C, cbase + c, mean, rstd);
    }
    par[0][c] = sc; par[1][c] = sh; par[2][c] = mean; par[3][c] = rstd;
  }
  for (int i = tid; i < KD * CW; i += NT) {
    const int k = i / CW, c = i % CW;
    wds[k][c] = (cbase + c < a.C) ? a.wdw[(size_t)(cbase + c) * KD + k] : 0.f;
  }
  float gw[KD][8], gb[8], s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    gb[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f;
#pragma unroll
    for (int k = 0; k < KD; ++k) gw[k][i] = 0.f;
  }
  const bool cvalid = c0 < a.C;
  for (int tile = 0; tile < a.tiles_per_wg; ++tile) {
    const int r0 = (blockIdx.x * a.tiles_per_wg + tile) * RT;
    if (r0 >= a.M) break;
    __syncthreads();
    for (int i = rl; i < ROWS; i += RL) {
      const int gr = r0 - PAD + i;
      float d[8], x[8];
      if (gr >= 0 && gr < a.M && cvalid) {
        load8(dD + (size_t)gr * a.C + c0, d);
        load8(XR + (size_t)gr * a.C + c0, x);
        act8(x, &par[0][vc * 8], &par[1][vc * 8], a.actX, (uint32_t)gr, a.C, c0);
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) { d[q] = 0.f; x[q] = 0.f; }
      }
      store8(&dDs[i][vc * 8], d);
      store8(&As[i][vc * 8], x);
    }
    __syncthreads();
    for (int r = rl; r < RT; r += RL) {
      const int gr = r0 + r;
      if (gr >= a.M || !cvalid) continue;
      const int t = gr % a.T;
      float dA[8], dc[8];
      load8(&dDs[r + PAD][vc * 8], dc);
#pragma unroll
      for (int q = 0; q < 8; ++q) { dA[q] = 0.f; gb[q] += dc[q]; }
#pragma unroll
      for (int k = 0; k < KD; ++k) {
        // data gradient: needs dD[r - k + pad]
        const int tb = t - k + PAD;
        if (tb >= 0 && tb < a.T) {
          float v[8];
          load8(&dDs[r + PAD - k + PAD][vc * 8], v);
#pragma unroll
          for (int q = 0; q < 8; ++q) dA[q] = fmaf(wds[k][vc * 8 + q], v[q], dA[q]);
        }
        // weight gradient: dD[r] * A[r + k - pad]
        const int tf = t + k - PAD;
        if (tf >= 0 && tf < a.T) {
          float v[8];
          load8(&As[r + k][vc * 8], v);
#pragma unroll
          for (int q = 0; q < 8; ++q) gw[k][q] = fmaf(dc[q], v[q], gw[k][q]);
        }
      }
      const size_t o = (size_t)gr * a.C + c0;
      if (ADD) {
        float ad[8];
        load8(ADD + o, ad);
#pragma unroll
        for (int q = 0; q < 8; ++q) dA[q] += ad[q];
      }
      if (a.actX.mode != 0 || a.actX.relu || a.actX.drop_thr || a.actX.rm.len) {
        float y[8], m[8];
        load8(XR + o, y);
        act8_grad_mask(y, m, &par[0][vc * 8], &par[1][vc * 8], a.actX, (uint32_t)gr, a.C, c0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          dA[q] *= m[q];
          s1[q] += dA[q];
          s2[q] += dA[q] * (y[q] - par[2][vc * 8 + q]) * par[3][vc * 8 + q];
        }
      }
      store8(OUT + o, dA);
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 8; ++q) {
#pragma unroll
    for (int k = 0; k < KD; ++k) atomicAdd(&red[k][vc * 8 + q], gw[k][q]);
    atomicAdd(&red[KD][vc * 8 + q], gb[q]);
    atomicAdd(&red[KD + 1][vc * 8 + q], s1[q]);
    atomicAdd(&red[KD + 2][vc * 8 + q], s2[q]);
  }
  __syncthreads();
  const int rep = blockIdx.x % TN_NREP;
  for (int i = tid; i < (KD + 3) * CW; i += NT) {
    const int k = i / CW, c = cbase + i % CW;
    if (c >= a.C) continue;
    const float v = red[k][i % CW];
    if (k < KD) atomic_add_f32(&a.g_wdw[(size_t)c * KD + k], v);
    else if (k == KD) atomic_add_f32(&a.g_bdw[c], v);
    else if (a.bsumsX) atomic_add_f32(&a.bsumsX[(size_t)(rep * 2 + (k - KD - 1)) * a.C + c], v);
  }
}

// ------------------------------------------------------------------------------------------
// Attentive statistics pooling backward, element-wise part (reference src/models.py:569-584):
// with alpha = softmax_T(e), mu = sum alpha x, q = sum alpha x^2, sigma = sqrt(clamp(q - mu^2, eps)):
//   dr = [q - mu^2 > eps] dsigma / (2 sigma);  dmu' = dmu - 2 mu dr;  dq = dr
//   d e   = alpha (dmu' x + dq x^2 - (dmu' mu + dq q))         -> dEN (grad wrt energies), sum -> d b_out
//   d x   = alpha (dmu' + 2 x dq)  (direct path)               -> DXD (added to the attention path later)
// ------------------------------------------------------------------------------------------
template <typename AT, int CVB = 64, int TG = 4>
__global__ __launch_bounds__(256) void asp_bwd_de_kernel(const AT* __restrict__ E, BnAct actE, const AT* __restrict__ EN,
                                                         int T, int D, float eps, const float* __restrict__ pooled,
                                                         const float* __restrict__ qv, const float* __restrict__ smax,
                                                         const float* __restrict__ sinv, const float* __restrict__ dpooled,
                                                         AT* __restrict__ dEN, AT* __restrict__ DXD,
                                                         float* __restrict__ g_bout) {
  static_assert(CVB * TG == 256, "256 threads");
  __shared__ float red[TG][CVB * 8];
  __shared__ float scs[CVB * 8], shs[CVB * 8];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int cbase = blockIdx.y * CVB * 8;
  for (int c = tid; c < CVB * 8; c += 256) {
    float sc = 1.f, sh = 0.f;
    if (cbase + c < D) bn_scale_shift(actE, D, cbase + c, sc, sh);
    scs[c] = sc; shs[c] = sh;
  }
  __syncthreads();
  const int vc = tid % CVB, tg = tid / CVB;
  const int c0 = cbase + vc * 8;
  float sb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) sb[i] = 0.f;
  if (c0 < D) {
    float dmu[8], dq[8], cst[8], mx[8], iv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const size_t o = (size_t)b * D + c0 + i;
      const float mu = pooled[(size_t)b * 2 * D + c0 + i], sg = pooled[(size_t)b * 2 * D + D + c0 + i];
      const float q = qv[o];
      const float gmu = dpooled[(size_t)b * 2 * D + c0 + i], gsg = dpooled[(size_t)b * 2 * D + D + c0 + i];
      const float dr = (q - mu * mu > eps) ? gsg / (2.f * sg) : 0.f;
      dmu[i] = gmu - 2.f * mu * dr;
      dq[i] = dr;
      cst[i] = dmu[i] * mu + dr * q;
      mx[i] = smax[o];
      iv[i] = sinv[o];
    }
    // gridDim.z = P workgroups share the utterance's frames (small batches of long utterances): element-wise work, the column
    // sums meet in the atomics below
    const int Lb = actE.rm.len ? actE.rm.len[b] : T;     // softmax over the valid frames only
    const int per = (Lb + (int)gridDim.z - 1) / (int)gridDim.z;
    const int t_lo = (int)blockIdx.z * per, L = min(Lb, t_lo + per);
    for (int t = Lb + tg + (int)blockIdx.z * TG; t < T; t += TG * (int)gridDim.z) {                 // padded frames: no gradient
      const size_t o = ((size_t)b * T + t) * D + c0;
      float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      store8(dEN + o, z);
      store8(DXD + o, z);
    }
    constexpr int U = 4;       // rows in flight per thread (asp_pool_fwd_kernel: one row per trip was a round trip per row)
    for (int t0 = t_lo + tg; t0 < L; t0 += TG * U) {
      float x[U][8], e[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t o = ((size_t)b * T + min(t0 + u * TG, L - 1)) * D + c0;
        load8(E + o, x[u]);
        load8(EN + o, e[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (t0 + u * TG < L) {
          const uint32_t row = (uint32_t)b * T + t0 + u * TG;
          const size_t o = (size_t)row * D + c0;
          float de[8], dx[8];
          act8(x[u], scs + vc * 8, shs + vc * 8, actE, row, D, c0);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float al = __expf(e[u][i] - mx[i]) * iv[i];
            de[i] = al * (dmu[i] * x[u][i] + dq[i] * x[u][i] * x[u][i] - cst[i]);
            dx[i] = al * (dmu[i] + 2.f * x[u][i] * dq[i]);
            sb[i] += de[i];
          }
          store8(dEN + o, de);
          store8(DXD + o, dx);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[tg][vc * 8 + i] = sb[i];
  __syncthreads();
  for (int c = tid; c < CVB * 8; c += 256) {
    if (cbase + c >= D) continue;
    float s = 0.f;
    for (int k = 0; k < TG; ++k) s += red[k][c];
    atomic_add_f32(&g_bout[cbase + c], s);
  }
}

// ------------------------------------------------------------------------------------------
// loss-head backward (reference src/losses.py:40-44, :85-132 via autograd)
// ------------------------------------------------------------------------------------------
struct HeadBwdArgs {
  int B, E, NC, loss_type;
  const float* dlogits;     // [B][NC], already / B
  const float* dscale;      // [B]
  const float* emb;         // [B][E] pre-normalisation embeddings
  const float* emb_norm;    // [B][E] returned embeddings (normalised)
  const float* W;           // [NC][E]
  float gs;                 // host grad scale
  const float* gs_dev;      // device grad scale or null
  const float* g_embnorm;   // [B][E] upstream grad on returned embeddings or null
  float* g_W;               // [NC][E]
  float* g_bias;            // [NC] or null
  float* demb;              // [B][E] out: grad wrt BN(lin) output
};

__global__ __launch_bounds__(256) void head_bwd_w_kernel(HeadBwdArgs a) {
  // one thread per (class, embedding dim): d W[c][e] = gs * sum_b dlogits[b][c] * x[b][e]
  const float gs = a.gs * (a.gs_dev ? *a.gs_dev : 1.f);
  const float* x = (a.loss_type == TN_LOSS_MARGIN) ? a.emb_norm : a.emb;
  const int n = a.NC * a.E;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c = i / a.E, e = i % a.E;
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sb = 0.f;      // 8 independent chains (see se_wgrad_kernel)
    int b = 0;
    for (; b + 8 <= a.B; b += 8) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float d = a.dlogits[(size_t)(b + q) * a.NC + c];
        s8[q] = fmaf(d, x[(size_t)(b + q) * a.E + e], s8[q]);
        sb += d;
      }
    }
    for (; b < a.B; ++b) {
      const float d = a.dlogits[(size_t)b * a.NC + c];
      s8[0] = fmaf(d, x[(size_t)b * a.E + e], s8[0]);
      sb += d;
    }
    const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    a.g_W[i] = s * gs;
    if (a.g_bias && e == 0) a.g_bias[c] = sb * gs;
  }
}

__global__ __launch_bounds__(256) void head_bwd_x_kernel(HeadBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* din = reinterpret_cast<float*>(smem);   // [E]
  __shared__ float red[4];
  const int tid = threadIdx.x, b = blockIdx.x;
  const float gs = a.gs * (a.gs_dev ? *a.gs_dev : 1.f);
  float nn = 0.f;
  for (int e = tid; e < a.E; e += 256) {
    float s = 0.f;
    if (a.loss_type != TN_LOSS_NONE) {
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
      int c = 0;
      for (; c + 4 <= a.NC; c += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s4[q] = fmaf(a.dlogits[(size_t)b * a.NC + c + q], a.W[(size_t)(c + q) * a.E + e], s4[q]);
      }
      for (; c < a.NC; ++c) s4[0] = fmaf(a.dlogits[(size_t)b * a.NC + c], a.W[(size_t)c * a.E + e], s4[0]);
      s = ((s4[0] + s4[1]) + (s4[2] + s4[3])) * gs;
    }
    din[e] = s;
    const float v = a.emb[(size_t)b * a.E + e];
    nn += v * v;
  }
  nn = block_sum_256(nn, red);
  const float norm = sqrtf(nn);
  if (a.loss_type == TN_LOSS_MARGIN) {
    // x_n = x / ||x|| feeds the fc AND is the returned embedding
    float dot = 0.f;
    for (int e = tid; e < a.E; e += 256) {
      float t = din[e] + (a.g_embnorm ? a.g_embnorm[(size_t)b * a.E + e] : 0.f);
      din[e] = t;
      dot += t * a.emb_norm[(size_t)b * a.E + e];
    }
    dot = block_sum_256(dot, red);
    const float ds = a.dscale[b] * gs;
    for (int e = tid; e < a.E; e += 256) {
      const float xn = a.emb_norm[(size_t)b * a.E + e];
      a.demb[(size_t)b * a.E + e] = (din[e] - xn * dot) / norm + ds * xn;
    }
  } else {
    // returned embedding = F.normalize(x) (eps 1e-12) — only an explicit upstream gradient flows through it
    float dot = 0.f;
    if (a.g_embnorm)
      for (int e = tid; e < a.E; e += 256) dot += a.g_embnorm[(size_t)b * a.E + e] * a.emb_norm[(size_t)b * a.E + e];
    dot = block_sum_256(dot, red);
    const float inv = 1.f / fmaxf(norm, 1e-12f);
    for (int e = tid; e < a.E; e += 256) {
      float t = din[e];
      if (a.g_embnorm) t += (a.g_embnorm[(size_t)b * a.E + e] - a.emb_norm[(size_t)b * a.E + e] * dot) * inv;
      a.demb[(size_t)b * a.E + e] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Decoder tail backward (reference src/models.py:506-513): small [B x features] matrices in fp32.
// ------------------------------------------------------------------------------------------
// column sums for a BN over rows: bsums += (sum_r dz, sum_r dz * yhat).  grid = (ceil(C/256), row groups of 16)
__global__ __launch_bounds__(256) void rows_bn_bwd_sums_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                               BnAct act, int R, int C, float* __restrict__ bsums) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, rstd;
  bn_mean_rstd(act, C, c, mean, rstd);
  float s1 = 0.f, s2 = 0.f;
  const int r0 = blockIdx.y * 16, r1 = min(R, r0 + 16);
  for (int r = r0; r < r1; ++r) {
    const float d = dz[(size_t)r * C + c];
    s1 += d;
    s2 += d * (y[(size_t)r * C + c] - mean) * rstd;
  }
  const int rep = blockIdx.y % TN_NREP;
  atomic_add_f32(&bsums[(size_t)(rep * 2 + 0) * C + c], s1);
  atomic_add_f32(&bsums[(size_t)(rep * 2 + 1) * C + c], s2);
}

// dy = k0 dz + k1 y + k2 for a [R][C] fp32 matrix
__global__ __launch_bounds__(256) void rows_bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                                BnBwd bn, int R, int C, float* __restrict__ dy) {
  const int n = R * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c = i % C;
    float k0, k1, k2;
    bn_bwd_coefs(bn, C, c, k0, k1, k2);
    dy[i] = k0 * dz[i] + k1 * y[i] + k2;
  }
}

// d W_lin[e][k] = sum_b dlin[b][e] * pbn[b][k],  pbn = BN(pooled).  grid = (ceil(K/64), ceil(E/8)): a workgroup owns 64 k and 8
// consecutive e; its 4 waves take a QUARTER of the utterances each and their partial sums are added in wave order (round 5: with one
// thread walking all B utterances the launch was 1152 waves of a 256-step load-to-use chain, 28 us; deterministic, the order of the
// additions is (q0 + q1) + (q2 + q3) instead of one running sum).  dlin comes from LDS as a broadcast.
__global__ __launch_bounds__(256) void tail_bwd_dw_kernel(const float* __restrict__ dlin, const float* __restrict__ pooled,
                                                          BnAct actP, int B, int K, int E, float* __restrict__ g_W) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* dl = reinterpret_cast<float*>(smem);   // [B][8]
  float* part = dl + (size_t)B * 8;             // [4][8][64]
  const int e0 = blockIdx.y * 8;
  for (int i = threadIdx.x; i < B * 8; i += 256) {
    const int b = i >> 3, j = i & 7;
    dl[i] = (e0 + j < E) ? dlin[(size_t)b * E + e0 + j] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  const int kc = min(k, K - 1);
  float sc, sh;
  bn_scale_shift(actP, K, kc, sc, sh);
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  const int per = (B + 3) / 4, b_lo = q * per, b_hi = min(B, b_lo + per);
  for (int b = b_lo; b < b_hi; ++b) {
    const float pv = pooled[(size_t)b * K + kc] * sc + sh;
    const float4 d0 = *reinterpret_cast<const float4*>(dl + b * 8), d1 = *reinterpret_cast<const float4*>(dl + b * 8 + 4);
    s[0] = fmaf(d0.x, pv, s[0]); s[1] = fmaf(d0.y, pv, s[1]); s[2] = fmaf(d0.z, pv, s[2]); s[3] = fmaf(d0.w, pv, s[3]);
    s[4] = fmaf(d1.x, pv, s[4]); s[5] = fmaf(d1.y, pv, s[5]); s[6] = fmaf(d1.z, pv, s[6]); s[7] = fmaf(d1.w, pv, s[7]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[(q * 8 + j) * 64 + lane] = s[j];
  __syncthreads();
  // 512 sums (8 e x 64 k) by 256 threads: two each
  for (int i = threadIdx.x; i < 8 * 64; i += 256) {
    const int j = i >> 6, l = i & 63, kk = blockIdx.x * 64 + l;
    const float v = (part[(0 * 8 + j) * 64 + l] + part[(1 * 8 + j) * 64 + l]) + (part[(2 * 8 + j) * 64 + l] + part[(3 * 8 + j) * 64 + l]);
    if (kk < K && e0 + j < E) g_W[(size_t)(e0 + j) * K + kk] = v;
  }
}

// d pbn[b][k] = sum_e dlin[b][e] * W[e][k].  grid = (ceil(K/256), ceil(B/4)): a thread owns one k and FOUR utterances (round 5: with
// one utterance per workgroup row W was read B times through L2 — 604 MB for a 2.4 MB matrix, 29 us); the sum over e of every
// (b, k) keeps its four interleaved chains, so the result is bit-identical
__global__ __launch_bounds__(256) void tail_bwd_dp_kernel(const float* __restrict__ dlin, const float* __restrict__ W, int B,
                                                          int K, int E, float* __restrict__ dpbn) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int b0 = blockIdx.y * 4;
  if (k >= K) return;
  float s4[4][4];
  const float* dl[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    dl[r] = dlin + (size_t)min(b0 + r, B - 1) * E;      // (rows past B: a clamped re-read, not stored)
#pragma unroll
    for (int q = 0; q < 4; ++q) s4[r][q] = 0.f;
  }
  int e = 0;
  for (; e + 4 <= E; e += 4) {
    float w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = W[(size_t)(e + q) * K + k];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) s4[r][q] = fmaf(dl[r][e + q], w[q], s4[r][q]);
  }
  for (; e < E; ++e) {
    const float w = W[(size_t)e * K + k];
#pragma unroll
    for (int r = 0; r < 4; ++r) s4[r][0] = fmaf(dl[r][e], w, s4[r][0]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (b0 + r < B) dpbn[(size_t)(b0 + r) * K + k] = (s4[r][0] + s4[r][1]) + (s4[r][2] + s4[r][3]);
}

// ------------------------------------------------------------------------------------------
// parameter gradients that are pure functions of the accumulated sums: BatchNorm gamma/beta and the
// bias of the conv/linear layer feeding each BatchNorm (d bias = sum_r dy = k0 s1 + k1 sum_y + k2 n,
// exactly zero in train mode up to rounding).  One launch for all layers.
// ------------------------------------------------------------------------------------------
struct BnGradDesc {
  BnBwd bn;
  float* g_gamma;
  float* g_beta;
  float* g_bias;    // bias of the producing conv / linear, or null
  int C, n;
};

__global__ void bn_param_grad_kernel(const BnGradDesc* descs) {
  const BnGradDesc d = descs[blockIdx.y];
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < d.C; c += gridDim.x * blockDim.x) {
    float s1 = 0.f, s2 = 0.f, sy = 0.f;
    for (int r = 0; r < TN_NREP; ++r) {
      s1 += d.bn.bsums[(r * 2 + 0) * d.C + c];
      s2 += d.bn.bsums[(r * 2 + 1) * d.C + c];
      sy += d.bn.fstats[(r * 2 + 0) * d.C + c];
    }
    d.g_gamma[c] = s2;
    d.g_beta[c] = s1;
    if (d.g_bias) {
      // train mode: sum_r dy = k0 s1 + k1 sum_y + k2 n == 0 identically (BatchNorm removes the mean, so
      // the bias of the layer in front of it has no effect on the loss); eval mode: k0 * sum dz.
      float k0, k1, k2;
      bn_bwd_coefs(d.bn, d.C, c, k0, k1, k2);
      d.g_bias[c] = (d.bn.batch != 0.f) ? 0.f : k0 * s1;
    }
  }
}

// SE weight gradients for all mega blocks in one launch (reference src/modules.py:166-171)
struct SeGradDesc {
  const float* dpre2;   // [B][C]
  const float* hid;     // [B][Hr]
  const float* dpre1;   // [B][Hr]
  const float* mean;    // [B][C]
  float* g_w1;          // [Hr][C]
  float* g_w2;          // [C][Hr]
};

__global__ void se_wgrad_kernel(const SeGradDesc* descs, int B, int C, int Hr) {
  const SeGradDesc d = descs[blockIdx.y];
  const int n = C * Hr;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += gridDim.x * blockDim.x) {
    if (i < n) {           // g_w2[c][j]
      const int c = i / Hr, j = i % Hr;
      // 8 independent partial sums: the loads of 8 batch rows are in flight together (a single dependent chain over B = 256
      // made this launch-for-all-blocks kernel latency-bound: 103 us for 2.3 MFLOP)
      float p8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int b = 0;
      for (; b + 8 <= B; b += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) p8[q] = fmaf(d.dpre2[(size_t)(b + q) * C + c], d.hid[(size_t)(b + q) * Hr + j], p8[q]);
      }
      for (; b < B; ++b) p8[0] = fmaf(d.dpre2[(size_t)b * C + c], d.hid[(size_t)b * Hr + j], p8[0]);
      d.g_w2[i] = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
    } else {               // g_w1[j][c]
      const int k = i - n, j = k / C, c = k % C;
      float p8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int b = 0;
      for (; b + 8 <= B; b += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) p8[q] = fmaf(d.dpre1[(size_t)(b + q) * Hr + j], d.mean[(size_t)(b + q) * C + c], p8[q]);
      }
      for (; b < B; ++b) p8[0] = fmaf(d.dpre1[(size_t)b * Hr + j], d.mean[(size_t)b * C + c], p8[0]);
      d.g_w1[k] = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
    }
  }
}

// d loss / d spectrograms through the prolog conv (only for utils.chart_dependencies-style checks):
// dx[b][ci][t] = sum_j sum_h dY0[b, t - j + pad][h] * W[h][ci][j]
template <typename AT>
__global__ __launch_bounds__(256) void prolog_input_grad_kernel(const AT* __restrict__ dZ, const AT* __restrict__ Y, BnBwd bn,
                                                                const float* __restrict__ W, int B, int n_mels, int T, int H,
                                                                int KP, float* __restrict__ dx) {
  const int pad = (KP - 1) / 2;
  const int n = B * n_mels * T;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int t = i % T, ci = (i / T) % n_mels, b = i / (T * n_mels);
    float s = 0.f;
    for (int j = 0; j < KP; ++j) {
      const int tt = t - j + pad;
      if (tt < 0 || tt >= T) continue;
      const size_t row = (size_t)b * T + tt;
      if (!tn_row_valid(bn.rm, (uint32_t)row)) continue;
      for (int h = 0; h < H; ++h) {
        float k0, k1, k2;
        bn_bwd_coefs(bn, H, h, k0, k1, k2);
        const float dy = k0 * Elem<AT>::to_f(dZ[row * H + h]) + k1 * Elem<AT>::to_f(Y[row * H + h]) + k2;
        s = fmaf(dy, W[((size_t)h * n_mels + ci) * KP + j], s);
      }
    }
    dx[i] = s;
  }
}


// ------------------------------------------------------------------------------------------
// Decoder(simple_pool=True) backward (reference src/models.py:497-502 under autograd)
// ------------------------------------------------------------------------------------------
// out[b][n] = sum_k in[b][k] * W[k][n]   (d mean = d pooled * W_pool: an "NN" product over B rows)
__global__ __launch_bounds__(256) void rows_matmul_nn_kernel(const float* __restrict__ in, const float* __restrict__ W, int B, int K,
                                                             int N, float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, b0 = blockIdx.y * 8;
  if (n >= N) return;
  float acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = 0.f;
  for (int k = 0; k < K; ++k) {
    const float w = W[(size_t)k * N + n];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (b0 + q < B) acc[q] = fmaf(in[(size_t)(b0 + q) * K + k], w, acc[q]);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (b0 + q < B) out[(size_t)(b0 + q) * N + n] = acc[q];
}
// out[n] = sum_b in[b][n]
__global__ void rows_colsum_kernel(const float* __restrict__ in, int B, int N, float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += in[(size_t)b * N + n];
  out[n] = s;
}
// d(epilog BN output)[b][t][c] = [BN(E) > 0] * dmu[b][c] / T, with the BatchNorm backward sums of the epilog BN
template <typename AT>
__global__ __launch_bounds__(256) void mean_pool_bwd_kernel(const float* __restrict__ dmu, const AT* __restrict__ E, BnAct actE,
                                                            int T, int D, AT* __restrict__ dEbn, float* __restrict__ bsums) {
  constexpr int CVB = 64, TG = 4;
  __shared__ float red[TG][2][CVB * 8];
  __shared__ float par[4][CVB * 8];      // sc, sh, mean, rstd
  const int tid = threadIdx.x, b = blockIdx.x, cbase = blockIdx.y * CVB * 8;
  for (int c = tid; c < CVB * 8; c += 256) {
    float sc = 1.f, sh = 0.f, mean = 0.f, rstd = 1.f;
    if (cbase + c < D) { bn_scale_shift(actE, D, cbase + c, sc, sh); bn_mean_rstd(actE, D, cbase + c, mean, rstd); }
    par[0][c] = sc; par[1][c] = sh; par[2][c] = mean; par[3][c] = rstd;
  }
  __syncthreads();
  const int vc = tid % CVB, tg = tid / CVB, c0 = cbase + vc * 8;
  float s1[8], s2[8], g[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; g[i] = 0.f; }
  if (c0 < D) {
    const float invT = 1.f / (float)T;
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = dmu[(size_t)b * D + c0 + i] * invT;
    for (int t = tg; t < T; t += TG) {
      const size_t o = ((size_t)b * T + t) * D + c0;
      float y[8], d[8];
      load8(E + o, y);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float z = y[i] * par[0][vc * 8 + i] + par[1][vc * 8 + i];
        const float v = (z > 0.f) ? g[i] : 0.f;
        d[i] = v;
        s1[i] += v;
        s2[i] += v * (y[i] - par[2][vc * 8 + i]) * par[3][vc * 8 + i];
      }
      store8(dEbn + o, d);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { red[tg][0][vc * 8 + i] = s1[i]; red[tg][1][vc * 8 + i] = s2[i]; }
  __syncthreads();
  const int rep = b % TN_NREP;
  for (int i = tid; i < 2 * CVB * 8; i += 256) {
    const int which = i / (CVB * 8), c = i % (CVB * 8);
    if (cbase + c < D)
      atomic_add_f32(&bsums[(size_t)(rep * 2 + which) * D + cbase + c],
                     red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c]);
  }
}

// ==========================================================================================
// dS = BatchNorm-backward(dZ, Y) = k0[c] dZ + k1[c] Y + k2[c], written IN PLACE over dZ (bf16): the operand both pipelined
// GEMMs of a layer's backward read as a stored matrix (data gradient: dS * W, weight gradient: dS^T * Q).  One streaming pass
// (2 reads + 1 write) instead of the transform inside two GEMM producers.
// ==========================================================================================
// EMU8 (experiment, TN_FP8_BWD_EMU=1 on an fp8 plan): dS is rounded through e4m3 with one power-of-two scale per ROW — what a
// data-gradient / weight-gradient GEMM on the f8f6f4 MFMA would read — so that the accuracy of an fp8 backward can be
// measured against the parity tests before its kernels exist (C = 512 / 1024 only: a row is one or two waves).
// MODE 2 (fp8 plans, hidden 512 / 1024): dS additionally as e4m3 bytes with one power-of-two scale per row + the E8M0 exponent
// bytes (Fp8Rows, tn_common.h) — the operand of the data gradient on the f8f6f4 MFMA; the bf16 dS stays for the weight gradient
// (a contraction over rows: per-row scales do not factor out of it).
// rowtiles (variable-length batches): the 256-row tiles with valid frames (PGemmNtArgs::rowtiles) — the pass then walks only
// those; null = all M rows.
// NV (round 5): 8-channel vectors per thread (vector v of lane l: channels 8 l + v * C / NV ..).  NV = 2 at 1024 channels with
// fp8 outputs: a row is then ONE wave and the row maximum a wave reduction (with one vector per thread it crossed two waves
// through LDS between two workgroup barriers — per row).
template <int MODE, int NV = 1>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(bf16_t* __restrict__ dZ, const bf16_t* __restrict__ Y, BnBwd bn, int M, int C, Fp8Rows f8,
                                                           const int* __restrict__ rowtiles, int n_rowtiles, Fp8Cols fc) {
  constexpr bool EMU8 = MODE == 1, OUT8 = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) float pg_k[];      // k0, k1, k2 : [3][C]  (+ the column scales [C] with OUT8)
  __shared__ float wmax[4];
  __shared__ float cmax[OUT8 ? 256 * 8 * NV : 1];                   // per-thread column maxima (fp8 weight gradient)
  const bool cols = OUT8 && fc.q != nullptr;
  float* csc_l = pg_k + 3 * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    bn_bwd_coefs(bn, C, c, pg_k[c], pg_k[C + c], pg_k[2 * C + c]);
    if (OUT8) {
      // fp8 weight gradient: the column's scale from the previous step's maximum, and the E8M0 byte that undoes it
      const float sc_ = cols ? tn_e4m3_col_scale(fc.amax_prev[c]) : 1.f;
      csc_l[c] = sc_;
      if (cols && blockIdx.x == 0) fc.cexp[c] = (uint8_t)(254u - ((__float_as_uint(sc_) >> 23) & 0xffu));
    }
  }
  __syncthreads();
  const int VC = C / 8 / NV;                 // threads per row
  const int VS = VC * 8;                     // channel distance between a thread's vectors
  // a thread keeps ONE set of columns for the whole loop (the grid stride is a multiple of VC): this step's maxima of them
  float cmx[NV][8];
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int u = 0; u < 8; ++u) cmx[v][u] = 0.f;
  const size_t nslot = (size_t)(rowtiles ? n_rowtiles * 256 : M) * VC;
  for (size_t iv = (size_t)blockIdx.x * 256 + threadIdx.x; iv < nslot; iv += (size_t)gridDim.x * 256) {
    const int cl = (int)(iv % VC) * 8;
    const uint32_t vr = (uint32_t)(iv / VC);
    const uint32_t row = rowtiles ? (uint32_t)rowtiles[vr >> 8] * 256u + (vr & 255u) : vr;
    const bool oob = row >= (uint32_t)M;                 // the last listed tile may reach past the tensor
    const bool padrow = oob || (bn.rm.len && !tn_row_valid(bn.rm, row));     // padding rows carry no gradient (uniform per row)
    float z[NV][8];
    if (MODE == 0 && padrow) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int u = 0; u < 8; ++u) z[v][u] = 0.f;
        if (!oob) store8(dZ + (size_t)row * C + cl + v * VS, z[v]);
      }
      continue;
    }
    float m = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c0 = cl + v * VS;
      if (!padrow) {
        float y[8];
        load8(dZ + (size_t)row * C + c0, z[v]);
        load8(Y + (size_t)row * C + c0, y);
#pragma unroll
        for (int u = 0; u < 8; ++u) z[v][u] = fmaf(pg_k[c0 + u], z[v][u], fmaf(pg_k[C + c0 + u], y[u], pg_k[2 * C + c0 + u]));
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) z[v][u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) m = fmaxf(m, fabsf(z[v][u]));
    }
    if (EMU8 || OUT8) {
      m = wave_max(m);
      if (VC == 128) {                      // a row is two waves (every thread of the workgroup is in this iteration: the slot
        const int w = threadIdx.x >> 6;     // count is a multiple of 256, launch_bn_bwd_apply)
        wmax[w] = m;
        __syncthreads();
        m = fmaxf(m, wmax[w ^ 1]);
        __syncthreads();
      }
      const float sc = tn_e4m3_row_scale(m);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c0 = cl + v * VS;
        if (EMU8) tn_e4m3_roundtrip8(z[v], sc, 1.f / sc);
        if (OUT8 && !oob) {
          *reinterpret_cast<uint2*>(f8.q + (size_t)row * C + c0) = tn_e4m3_pack8(z[v], 1.f / sc);
          if (cols) {
            float cs8[8];
            *reinterpret_cast<float4*>(cs8) = *reinterpret_cast<const float4*>(csc_l + c0);
            *reinterpret_cast<float4*>(cs8 + 4) = *reinterpret_cast<const float4*>(csc_l + c0 + 4);
#pragma unroll
            for (int u = 0; u < 8; ++u) cmx[v][u] = fmaxf(cmx[v][u], fabsf(z[v][u]));
            {
              // (the column-scaled copy is read by the weight-gradient launch at the end of backward only: non-temporal, TN_NT_F8C)
              typedef __attribute__((ext_vector_type(2))) unsigned int f8c_u2;
              const uint2 pk = tn_e4m3_pack8_cols(z[v], cs8);
              if (TN_NT_F8C) __builtin_nontemporal_store(f8c_u2{pk.x, pk.y}, reinterpret_cast<f8c_u2*>(fc.q + (size_t)row * C + c0));
              else *reinterpret_cast<uint2*>(fc.q + (size_t)row * C + c0) = pk;
            }
          }
        }
      }
      if (OUT8 && !oob && cl == 0) f8.rowexp[tn_rowexp_pos(row)] = tn_e8m0_of_pow2(sc);
    }
    if (!oob && !(OUT8 && cols && fc.skip_bf16)) {
#pragma unroll
      for (int v = 0; v < NV; ++v) store8(dZ + (size_t)row * C + cl + v * VS, z[v]);
    }
  }
  if (OUT8) {
    if (cols) {      // workgroup-uniform
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int u = 0; u < 8; ++u) cmax[(threadIdx.x * NV + v) * 8 + u] = cmx[v][u];
      __syncthreads();
      // column c = vector v (c / VS) of the row's thread (c % VS) / 8, held by the threads k VC + that one, k < 256 / VC
      for (int c = threadIdx.x; c < C; c += 256) {
        const int v = c / VS, ln = (c % VS) / 8;
        float mm = 0.f;
        for (int k = 0; k < 256 / VC; ++k) mm = fmaxf(mm, cmax[((k * VC + ln) * NV + v) * 8 + (c & 7)]);
        if (mm > 0.f) atomicMax(reinterpret_cast<unsigned int*>(fc.amax_cur) + c, __float_as_uint(mm));
      }
    }
  }
}
// emu8: the e4m3 round trip in place (experiment);  f8: also write the fp8 operand (q != null)
inline int launch_bn_bwd_apply(bf16_t* dZ, const bf16_t* Y, const BnBwd& bn, int M, int C, hipStream_t st, bool emu8 = false, Fp8Rows f8 = Fp8Rows{nullptr, nullptr},
                               const int* rowtiles = nullptr, int n_rowtiles = 0, Fp8Cols fc = Fp8Cols{nullptr, nullptr, nullptr, nullptr, 0}) {
  if (C % 8 || C > 4096) return TN_E_UNSUPPORTED;
  if (!bn.rm.len || n_rowtiles <= 0) { rowtiles = nullptr; n_rowtiles = 0; }
  const bool rows_ok = (C == 512 || C == 1024) && ((size_t)M * (C / 8)) % 256 == 0;
  if (f8.q && !rows_ok) return TN_E_UNSUPPORTED;
  const size_t smem = (size_t)(f8.q ? 4 : 3) * C * sizeof(float);
  if (f8.q && C == 1024)      // one wave per row: two vectors per thread
    hipLaunchKernelGGL((bn_bwd_apply_kernel<2, 2>), dim3(2048), dim3(256), smem, st, dZ, Y, bn, M, C, f8, rowtiles, n_rowtiles, fc);
  else if (f8.q)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<2, 1>), dim3(2048), dim3(256), smem, st, dZ, Y, bn, M, C, f8, rowtiles, n_rowtiles, fc);
  else if (emu8 && !bn.rm.len && rows_ok)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<1, 1>), dim3(2048), dim3(256), smem, st, dZ, Y, bn, M, C, f8, rowtiles, n_rowtiles, fc);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<0, 1>), dim3(2048), dim3(256), smem, st, dZ, Y, bn, M, C, f8, rowtiles, n_rowtiles, fc);
  return (int)hipGetLastError();
}
