// titanet_amd — generic "rows x channels" NT GEMM on MFMA with pluggable A-tile producers and
// epilogues.  C[M x N] = A[M x K] * W[N x K]^T, where the A tile is *produced* into LDS by a functor
// (activation-on-load, depthwise stencil, im2col, BatchNorm-backward-on-load ...) and never exists in
// HBM, and the epilogue functor consumes the accumulators (bias, BN statistics, tanh, stores ...).
//
// One workgroup = WM x WN waves, each wave owns a 64 x 64 output sub-tile as 2 x 2 MFMA 32x32 tiles:
//   bf16 path : v_mfma_f32_32x32x16_bf16   (8 K-elements per lane, ds_read_b128 fragments)
//   fp32 path : v_mfma_f32_32x32x2_f32     (exact f32 == fmaf chain; the parity path)
// Both share the C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma once
#include "tn_common.h"

struct GemmShape {
  int M, N, K;
  const void* W;   // [N][K], element type AT, K contiguous
};

template <typename AT> struct Mma;
template <> struct Mma<bf16_t> {
  using Frag = bf16x8_t;
  __device__ static __forceinline__ Frag load(const bf16_t* row, int ks, int half) {
    return *reinterpret_cast<const bf16x8_t*>(row + ks * 16 + half * 8);
  }
  __device__ static __forceinline__ f32x16_t mma(Frag a, Frag b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  using Frag = float;
  __device__ static __forceinline__ Frag load(const float* row, int ks, int half) { return row[ks * 2 + half]; }
  __device__ static __forceinline__ f32x16_t mma(Frag a, Frag b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};

__device__ __forceinline__ int cd_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Fill a [ROWS][BK] tile (row stride BKP) from a row-major global matrix src[nrows][ld], rows
// row0.., columns kc..kc+BK, zero outside [0,nrows) x [0,kmax).  NT threads, 8-element vectors.
template <typename AT, int ROWS, int NT>
__device__ __forceinline__ void fill_tile_plain(AT* dst, const AT* __restrict__ src, int nrows, int ld, int kmax,
                                                int row0, int kc, int tid) {
  constexpr int BK = Elem<AT>::BK, BKP = BK + Elem<AT>::PAD, VC = BK / 8, RL = NT / VC;
  const int vc = tid % VC, rl = tid / VC;
  const int k = kc + vc * 8;
  for (int r = rl; r < ROWS; r += RL) {
    float v[8];
    const int gr = row0 + r;
    if (gr < nrows && gr >= 0 && k < kmax) load8(src + (size_t)gr * ld + k, v);
    else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
    store8_lds(dst + r * BKP + vc * 8, v);
  }
}

// ------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------
// RH = row halves per wave: 1 -> a wave owns a 64 x 64 output tile (2 x 2 MFMA tiles); 2 -> 128 x 64 (rows wm*64 + i*32 of
// BOTH 128-row halves of a 256-row workgroup tile): 6 fragment reads per 8 MFMAs instead of 4 per 4 — the 64 x 64 wave
// tile is LDS-read-bound (1 KB of fragments per MFMA), and the epilogues run once per half unchanged
// CC = column halves per wave (with RH = 1): a wave owns 64 x 128 of a 128-row x 512-column workgroup tile — the A rows are
// then read once per 512 output columns instead of once per 256 (the pointwise GEMMs of the wide models re-read their
// row operand N / 256 times: at N = 1024 that re-read, not the MFMA rate, bounded them)
template <typename AT, int WM, int WN, typename Prod, typename Epi, int RH = 1, int CC = 1>
__global__ __launch_bounds__(WM* WN * 64, ((sizeof(AT) == 2 && Prod::kRaw) ? 2 : (WM * WN >= 8 ? 4 : 3))) void gemm_nt_kernel(GemmShape g, typename Prod::Args pa,
                                                               typename Epi::Args ea) {
  constexpr int BM = WM * 64 * RH, BN = WN * 64 * CC, BK = Elem<AT>::BK, BKP = BK + Elem<AT>::PAD, NT = WM * WN * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AT* As = reinterpret_cast<AT*>(smem);
  AT* Bs = As + BM * BKP;
  char* scratch = reinterpret_cast<char*>(As + (RH == 2 ? 2 : 1) * (BM + BN) * BKP);   // RH == 2: two tile buffers

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // RH == 2 (big problems): column tiles are the FAST grid index, so the workgroups that re-read the same A rows (one per
  // 256-column tile of the output) run at the same time and find them in L2 / Infinity Cache
  const int r0 = (RH == 2 ? blockIdx.y : blockIdx.x) * BM, n0 = (RH == 2 ? blockIdx.x : blockIdx.y) * BN;

  Prod prod;
  constexpr bool RAW = sizeof(AT) == 2 && Prod::kRaw;
  // raw path: the first K chunk is requested BEFORE the producer's init (BatchNorm coefficients: 16 - 32 dependent-latency
  // loads per channel, once per workgroup — 20 % of a K = 512 workgroup's life when it ran first)
  constexpr int RVC = BK / 8, RRL = NT / RVC, RNW = BN / RRL;
  uint4 wreg[RAW ? RNW : 1];
  typename Prod::template Regs<RAW ? BM / RRL : 1> areg;
  const AT* W = reinterpret_cast<const AT*>(g.W);
  auto load_w = [&](int kc) {
    const int k = kc + (tid % RVC) * 8, rl = tid / RVC;
#pragma unroll
    for (int q = 0; q < (RAW ? RNW : 1); ++q) {
      const int gn = n0 + rl + q * RRL;
      wreg[q] = (gn < g.N && k < g.K) ? *reinterpret_cast<const uint4*>(W + (size_t)gn * g.K + k) : make_uint4(0, 0, 0, 0);
    }
  };
  if constexpr (RAW) {
    load_w(0);
    prod.template load_raw<BM, NT, BK>(areg, pa, g.M, g.K, tid, r0, 0);
  }
  prod.template init<AT, NT, BK>(pa, g.M, g.K, scratch, tid);   // ends with __syncthreads()

  static_assert(RH == 1 || CC == 1, "one of the two");
  constexpr int NH = RH * CC;              // 64 x 64 sub-tiles per wave
  f32x16_t acc[NH][2][2];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;

  auto mma_tile = [&]() {
    const AT* arow0 = As + (wm * 64 + (lane & 31)) * BKP;
    const AT* brow0 = Bs + (wn * 64 + (lane & 31)) * BKP;
    const int half = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < BK / Elem<AT>::KM; ++ks) {
      if constexpr (CC == 2) {
        typename Mma<AT>::Frag a0 = Mma<AT>::load(arow0, ks, half);
        typename Mma<AT>::Frag a1 = Mma<AT>::load(arow0 + 32 * BKP, ks, half);
#pragma unroll
        for (int c = 0; c < CC; ++c) {
          typename Mma<AT>::Frag b0 = Mma<AT>::load(brow0 + c * WN * 64 * BKP, ks, half);
          typename Mma<AT>::Frag b1 = Mma<AT>::load(brow0 + (c * WN * 64 + 32) * BKP, ks, half);
          acc[c][0][0] = Mma<AT>::mma(a0, b0, acc[c][0][0]);
          acc[c][0][1] = Mma<AT>::mma(a0, b1, acc[c][0][1]);
          acc[c][1][0] = Mma<AT>::mma(a1, b0, acc[c][1][0]);
          acc[c][1][1] = Mma<AT>::mma(a1, b1, acc[c][1][1]);
        }
      } else {
        typename Mma<AT>::Frag b0 = Mma<AT>::load(brow0, ks, half);
        typename Mma<AT>::Frag b1 = Mma<AT>::load(brow0 + 32 * BKP, ks, half);
#pragma unroll
        for (int h = 0; h < RH; ++h) {
          typename Mma<AT>::Frag a0 = Mma<AT>::load(arow0 + h * WM * 64 * BKP, ks, half);
          typename Mma<AT>::Frag a1 = Mma<AT>::load(arow0 + (h * WM * 64 + 32) * BKP, ks, half);
          acc[h][0][0] = Mma<AT>::mma(a0, b0, acc[h][0][0]);
          acc[h][0][1] = Mma<AT>::mma(a0, b1, acc[h][0][1]);
          acc[h][1][0] = Mma<AT>::mma(a1, b0, acc[h][1][0]);
          acc[h][1][1] = Mma<AT>::mma(a1, b1, acc[h][1][1]);
        }
      }
    }
  };
  if constexpr (sizeof(AT) == 2 && Prod::kRaw) {
    // bf16 operands that are (functions of) stored matrices: the raw 16-byte vectors of the NEXT K chunk are requested
    // before this chunk's MFMAs and turned into LDS tiles after them (the weight tile is a straight copy) — the synchronous
    // loop below spent as many VALU cycles on bf16 -> f32 -> bf16 round trips and exposed loads as on the MFMAs
    // (TitaNet-L pointwise GEMMs: 354 / 435 us for 161 GFLOP)
    constexpr int VC = RVC, RL = RRL, NW = RNW;
    const int vc = tid % VC, rl = tid / VC;
    if constexpr (RH == 2) {
      // big problems: TWO LDS tile buffers, one barrier per K chunk — chunk k + 1 is written into the other buffer while
      // other waves still multiply chunk k (the launcher sizes the LDS for it)
      constexpr int TILE = (BM + BN) * BKP;                  // elements of one (A, W) tile pair
      auto commit = [&](int buf, int kc) {
        AT* Ab = As + buf * TILE;
        AT* Bb = Ab + BM * BKP;
#pragma unroll
        for (int q = 0; q < NW; ++q) *reinterpret_cast<uint4*>(Bb + (rl + q * RL) * BKP + vc * 8) = wreg[q];
        prod.template commit_raw<BM, NT, BK, BKP>(reinterpret_cast<bf16_t*>(Ab), areg, pa, g.M, g.K, tid, r0, kc);
      };
      commit(0, 0);
      if (BK < g.K) {
        load_w(BK);
        prod.template load_raw<BM, NT, BK>(areg, pa, g.M, g.K, tid, r0, BK);
      }
      __syncthreads();
      int buf = 0;
      for (int kc = 0; kc < g.K; kc += BK, buf ^= 1) {
        if (kc + BK < g.K) {
          commit(buf ^ 1, kc + BK);                          // the other buffer was last read before the previous barrier
          if (kc + 2 * BK < g.K) {
            load_w(kc + 2 * BK);
            prod.template load_raw<BM, NT, BK>(areg, pa, g.M, g.K, tid, r0, kc + 2 * BK);
          }
        }
        {
          const AT* arow0 = As + buf * TILE + (wm * 64 + (lane & 31)) * BKP;
          const AT* brow0 = As + buf * TILE + BM * BKP + (wn * 64 + (lane & 31)) * BKP;
          const int half = lane >> 5;
#pragma unroll
          for (int ks = 0; ks < BK / Elem<AT>::KM; ++ks) {
            typename Mma<AT>::Frag b0 = Mma<AT>::load(brow0, ks, half);
            typename Mma<AT>::Frag b1 = Mma<AT>::load(brow0 + 32 * BKP, ks, half);
#pragma unroll
            for (int h = 0; h < RH; ++h) {
              typename Mma<AT>::Frag a0 = Mma<AT>::load(arow0 + h * WM * 64 * BKP, ks, half);
              typename Mma<AT>::Frag a1 = Mma<AT>::load(arow0 + (h * WM * 64 + 32) * BKP, ks, half);
              acc[h][0][0] = Mma<AT>::mma(a0, b0, acc[h][0][0]);
              acc[h][0][1] = Mma<AT>::mma(a0, b1, acc[h][0][1]);
              acc[h][1][0] = Mma<AT>::mma(a1, b0, acc[h][1][0]);
              acc[h][1][1] = Mma<AT>::mma(a1, b1, acc[h][1][1]);
            }
          }
        }
        __syncthreads();
      }
    } else
    for (int kc = 0; kc < g.K; kc += BK) {
#pragma unroll
      for (int q = 0; q < NW; ++q) *reinterpret_cast<uint4*>(Bs + (rl + q * RL) * BKP + vc * 8) = wreg[q];
      prod.template commit_raw<BM, NT, BK, BKP>(reinterpret_cast<bf16_t*>(As), areg, pa, g.M, g.K, tid, r0, kc);
      __syncthreads();
      if (kc + BK < g.K) {
        load_w(kc + BK);
        prod.template load_raw<BM, NT, BK>(areg, pa, g.M, g.K, tid, r0, kc + BK);
      }
      mma_tile();
      __syncthreads();
    }
  } else {
    for (int kc = 0; kc < g.K; kc += BK) {
      fill_tile_plain<AT, BN, NT>(Bs, W, g.N, g.K, g.K, n0, kc, tid);
      prod.template fill<AT, BM, NT, BK, BKP>(As, pa, g.M, g.K, tid, r0, kc);   // result in As; may sync internally
      __syncthreads();
      mma_tile();
      __syncthreads();
    }
  }
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    if (h > 0) __syncthreads();                 // the previous half's staging has been stored
    Epi::template run<AT, WM, WN>(acc[h], ea, g, smem, tid, r0 + (RH == 2 ? h : 0) * WM * 64, n0 + (CC == 2 ? h : 0) * WN * 64);
  }
}

// LDS bytes needed by the main loop (without producer scratch)
template <typename AT, int WM, int WN>
constexpr size_t gemm_tile_bytes() {
  return (size_t)(WM * 64 + WN * 64) * (Elem<AT>::BK + Elem<AT>::PAD) * sizeof(AT);
}

// ------------------------------------------------------------------------------------------
// producers
// ------------------------------------------------------------------------------------------
// P_PLAIN: A[r][c] = act(X[r][c])
struct ProdPlain {
  struct Args {
    const void* X;   // [M][ldx]
    int ldx;
    BnAct act;
  };
  float* sc;
  float* sh;
  __host__ __device__ static size_t scratch_bytes(int K, int /*KD*/, int /*ROWS*/, int /*CW*/, size_t /*elem*/) {
    return (size_t)2 * K * sizeof(float);
  }
  template <typename AT, int NT, int CW>
  __device__ __forceinline__ void init(const Args& a, int M, int K, char* scratch, int tid) {
    sc = reinterpret_cast<float*>(scratch);
    sh = sc + K;
    if (a.act.mode != 0) {
      for (int c = tid; c < K; c += NT) bn_scale_shift(a.act, K, c, sc[c], sh[c]);
    }
    __syncthreads();
  }
  // bf16 pipelined loop of gemm_nt_kernel: raw vectors of a K chunk (requested early), then tile rows (after the MFMAs)
  static constexpr bool kRaw = true, kWideCols = false;
  template <int N> struct Regs { uint4 x[N]; };
  template <int ROWS, int NT, int CW, int N>
  __device__ __forceinline__ void load_raw(Regs<N>& rg, const Args& a, int M, int K, int tid, int r0, int kc) {
    constexpr int VC = CW / 8, RL = NT / VC;
    const int vc = tid % VC, rl = tid / VC, k = kc + vc * 8;
    const bf16_t* X = reinterpret_cast<const bf16_t*>(a.X);
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const int gr = r0 + rl + q * RL;
      rg.x[q] = (gr < M && k < K) ? *reinterpret_cast<const uint4*>(X + (size_t)gr * a.ldx + k) : make_uint4(0, 0, 0, 0);
    }
  }
  template <int ROWS, int NT, int CW, int PITCH, int N>
  __device__ __forceinline__ void commit_raw(bf16_t* As, const Regs<N>& rg, const Args& a, int M, int K, int tid, int r0, int kc) {
    constexpr int VC = CW / 8, RL = NT / VC;
    const int vc = tid % VC, rl = tid / VC, k = kc + vc * 8;
    const bool plain = a.act.mode == 0 && !a.act.relu && !a.act.drop_thr && !a.act.rm.len;     // uniform
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const int r = rl + q * RL, gr = r0 + r;
      if (plain) {
        *reinterpret_cast<uint4*>(As + r * PITCH + vc * 8) = rg.x[q];       // out-of-range vectors were loaded as zeros
      } else {
        float v[8];
        v[0] = __uint_as_float(rg.x[q].x << 16); v[1] = __uint_as_float(rg.x[q].x & 0xffff0000u);
        v[2] = __uint_as_float(rg.x[q].y << 16); v[3] = __uint_as_float(rg.x[q].y & 0xffff0000u);
        v[4] = __uint_as_float(rg.x[q].z << 16); v[5] = __uint_as_float(rg.x[q].z & 0xffff0000u);
        v[6] = __uint_as_float(rg.x[q].w << 16); v[7] = __uint_as_float(rg.x[q].w & 0xffff0000u);
        if (gr < M && k < K) act8(v, sc + k, sh + k, a.act, (uint32_t)gr, K, k);
        else {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = 0.f;
        }
        store8(As + r * PITCH + vc * 8, v);
      }
    }
  }
  template <typename AT, int ROWS, int NT, int CW, int PITCH>
  __device__ __forceinline__ void fill(AT* As, const Args& a, int M, int K, int tid, int r0, int kc) {
    constexpr int BKP = PITCH, VC = CW / 8, RL = NT / VC, BM = ROWS;
    const int vc = tid % VC, rl = tid / VC;
    const int k = kc + vc * 8;
    const AT* X = reinterpret_cast<const AT*>(a.X);
    for (int r = rl; r < BM; r += RL) {
      float v[8];
      const int gr = r0 + r;
      if (gr < M && k < K) {
        load8(X + (size_t)gr * a.ldx + k, v);
        act8(v, sc + k, sh + k, a.act, (uint32_t)gr, K, k);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
      }
      store8_lds(As + r * BKP + vc * 8, v);
    }
  }
};

// P_DW: A[r][c] = b_dw[c] + sum_j w_dw[c][j] * act(X)[r + j - pad][c]   (zero outside the utterance)
// — the depthwise conv of reference src/modules.py:65-75 fused as the prologue of its pointwise GEMM.
struct ProdDw {
  static constexpr bool kRaw = false, kWideCols = false;
  template <int N> struct Regs {};
  struct Args {
    const void* X;
    int ldx;
    BnAct act;
    const float* wdw;   // [C][KD]  (reference weight [C,1,KD])
    const float* bdw;   // [C]
    int KD;
    int T;              // frames per utterance
    void* Qout;         // [M][ldx] AT or null: the produced tile (the depthwise output) is also stored by the workgroups of
                        // the first output-column block, so that the weight-gradient pass reads it instead of recomputing
                        // activation + stencil (forward GEMM only; the wgrad producers pass null)
  };
  float* sc;
  float* sh;
  float* wd;    // [KD][BK]
  float* bd;    // [BK]
  char* xs;     // [(BM + KD - 1)][BK] AT
  __host__ __device__ static size_t scratch_bytes(int K, int KD, int ROWS, int CW, size_t elem) {
    return (size_t)2 * K * sizeof(float) + (size_t)(KD + 1) * CW * sizeof(float) + (size_t)(ROWS + KD - 1) * (CW + 8) * elem + 32;
  }
  template <typename AT, int NT, int CW>
  __device__ __forceinline__ void init(const Args& a, int M, int K, char* scratch, int tid) {
    sc = reinterpret_cast<float*>(scratch);
    sh = sc + K;
    wd = sh + K;
    bd = wd + a.KD * CW;
    size_t off = (size_t)(2 * K + (a.KD + 1) * CW) * sizeof(float);
    off = (off + 15) & ~(size_t)15;
    xs = scratch + off;
    if (a.act.mode != 0) {
      for (int c = tid; c < K; c += NT) bn_scale_shift(a.act, K, c, sc[c], sh[c]);
    }
    __syncthreads();
  }
  template <typename AT, int ROWS, int NT, int CW, int PITCH>
  __device__ __forceinline__ void fill(AT* As, const Args& a, int M, int K, int tid, int r0, int kc) {
    constexpr int BK = CW, BKP = PITCH, VC = BK / 8, RL = NT / VC, BM = ROWS;
    constexpr int XP = BK + 8;    // staged-row pitch: without the pad the 8 rows a wave touches share their LDS banks
    struct { int M, K; } g{M, K};
    const int vc = tid % VC, rl = tid / VC;
    const int k = kc + vc * 8;
    const int KD = a.KD, pad = (KD - 1) / 2;
    const AT* X = reinterpret_cast<const AT*>(a.X);
    AT* Xs = reinterpret_cast<AT*>(xs);
    // stage the activated input rows (with halo) once
    for (int i = rl; i < BM + KD - 1; i += RL) {
      float v[8];
      const int gr = r0 - pad + i;
      if (gr >= 0 && gr < g.M && k < g.K) {
        load8(X + (size_t)gr * a.ldx + k, v);
        act8(v, sc + k, sh + k, a.act, (uint32_t)gr, g.K, k);
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = 0.f;
      }
      store8(Xs + i * XP + vc * 8, v);
    }
    for (int i = tid; i < (KD + 1) * BK; i += NT) {
      const int j = i / BK, c = i % BK;
      float w = 0.f;
      if (kc + c < g.K) w = (j < KD) ? a.wdw[(size_t)(kc + c) * KD + j] : a.bdw[kc + c];
      wd[i] = w;   // row KD of wd == bd
    }
    __syncthreads();
    // stencil over time
    for (int r = rl; r < BM; r += RL) {
      const int gr = r0 + r;
      const int t = gr % a.T;
      float o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = bd[vc * 8 + q];
      for (int j = 0; j < KD; ++j) {
        const int tt = t + j - pad;
        if (tt >= 0 && tt < a.T) {
          float v[8];
          load8(Xs + (r + j) * XP + vc * 8, v);
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = fmaf(wd[j * BK + vc * 8 + q], v[q], o[q]);
        }
      }
      store8_lds(As + r * BKP + vc * 8, o);
      if (a.Qout && blockIdx.y == 0 && gr < g.M && k < g.K) store8(reinterpret_cast<AT*>(a.Qout) + (size_t)gr * a.ldx + k, o);
    }
  }
};

// P_IM2COL (prolog): A[r][ci*KP + j] = x[b][ci][t + j - pad]; x is the float [B, n_mels, T] input of
// TitaNet.forward (reference src/models.py:318-331, prolog conv :370).  K = n_mels * KP.
struct ProdIm2col {
  static constexpr bool kRaw = false, kWideCols = false;
  template <int N> struct Regs {};
  struct Args {
    const float* x;   // [B][n_mels][T] float32, T contiguous (reference layout)
    int n_mels, KP, T;
    const int* len;   // [B] valid frames (frames beyond them read as zero whatever the buffer holds) or null
  };
  __host__ __device__ static size_t scratch_bytes(int, int, int, int, size_t) { return 16; }
  template <typename AT, int NT, int CW>
  __device__ __forceinline__ void init(const Args&, int, int, char*, int) { __syncthreads(); }
  template <typename AT, int ROWS, int NT, int CW, int PITCH>
  __device__ __forceinline__ void fill(AT* As, const Args& a, int M, int K, int tid, int r0, int kc) {
    constexpr int BK = CW, BKP = PITCH, BM = ROWS;
    struct { int M, K; } g{M, K};
    const int pad = (a.KP - 1) / 2;
    // consecutive threads -> consecutive rows (t) for coalesced reads of the T-contiguous input.  The row of a
    // thread is the same for all its elements (NT is a multiple of BM), so the row -> (utterance, frame) division
    // happens once per call; k -> (mel, tap) uses a constant divisor for the reference's prolog kernel size 3.
    static_assert(NT % BM == 0, "thread count must be a multiple of the tile rows");
    const int r = tid % BM, gr = r0 + r;
    const bool row_ok = gr < g.M;
    const int b = row_ok ? gr / a.T : 0, t = row_ok ? gr - b * a.T : 0;
    const float* xb = a.x + (size_t)b * a.n_mels * a.T;
    const int L = (a.len && row_ok) ? a.len[b] : a.T;
    for (int kk = tid / BM; kk < BK; kk += NT / BM) {
      const int k = kc + kk;
      float v = 0.f;
      if (row_ok && k < g.K) {
        int ci, j;
        if (a.KP == 3) { ci = (int)((unsigned)k / 3u); j = k - 3 * ci; }
        else { ci = k / a.KP; j = k - ci * a.KP; }
        const int tt = t + j - pad;
        if (tt >= 0 && tt < L) v = xb[(size_t)ci * a.T + tt];
      }
      As[r * BKP + kk] = Elem<AT>::from_f(v);
    }
  }
};

// P_TAPS: the same im2col operand from a PACKED copy of the input: X0[row][C] (row = b*T + t, element type AT, written by
// prolog_pack_kernel) with the K axis in TAP-major order, A[r][j*C + ci] = X0[r + j - pad][ci] — i.e. row r of the operand
// is KP*C contiguous elements of X0 starting at row r - pad, so the fill is 16-byte vector loads like a plain operand
// (the weight is used in the matching [out][tap][ci] order).  Taps outside the utterance read as zero; frames beyond an
// utterance's valid length are already zero in X0.  C must be a multiple of 8.
struct ProdTaps {
  static constexpr bool kRaw = false, kWideCols = false;
  template <int N> struct Regs {};
  struct Args {
    const void* X0;   // [M][C]
    int C, KP, T;
  };
  __host__ __device__ static size_t scratch_bytes(int, int, int, int, size_t) { return 16; }
  template <typename AT, int NT, int CW>
  __device__ __forceinline__ void init(const Args&, int, int, char*, int) { __syncthreads(); }
  template <typename AT, int ROWS, int NT, int CW, int PITCH>
  __device__ __forceinline__ void fill(AT* As, const Args& a, int M, int K, int tid, int r0, int kc) {
    constexpr int BKP = PITCH, VC = CW / 8, RL = NT / VC, BM = ROWS;
    const int vc = tid % VC, rl = tid / VC;
    const int k = kc + vc * 8;
    const int j = k / a.C, ci = k - j * a.C, dj = j - (a.KP - 1) / 2;
    const AT* X0 = reinterpret_cast<const AT*>(a.X0);
    for (int r = rl; r < BM; r += RL) {
      float v[8];
      const int gr = r0 + r;
      bool ok = gr < M && k < K;
      if (ok) {
        const int tt = gr % a.T + dj;
        ok = tt >= 0 && tt < a.T;
      }
      if (ok) load8(X0 + (size_t)(gr + dj) * a.C + ci, v);
      else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
      }
      store8_lds(As + r * BKP + vc * 8, v);
    }
  }
};

// ------------------------------------------------------------------------------------------
// epilogues
// ------------------------------------------------------------------------------------------
// y = acc + bias [-> tanh]; store as AT through LDS (coalesced 16-byte rows); optional per-column
// sum / sum-of-squares of y over the valid rows -> replicated global atomics (BatchNorm statistics
// accumulated in the producing kernel's epilogue, SURVEY.md §2 "native_batch_norm" row).
struct EpiStoreArgs {
  void* Y;             // [M][ldy] AT
  int ldy;
  const float* bias;   // [N] or null
  float* stats;        // [TN_NREP][2][N] or null
  RowMask rm;          // rows left out of the statistics (variable-length batches); {null, 0}: none
  const float* colscale;   // [N] or null: y = acc * colscale[n] + bias[n] (fp8 weights: one scale per output channel)
};
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __expf(2.f * x);       // inf for large x -> 1, 0 for very negative x -> -1
  return 1.f - 2.f / (e + 1.f);
}
template <bool TANH>
struct EpiStoreT {
  using Args = EpiStoreArgs;
  template <typename AT, int WM, int WN>
  static constexpr size_t lds_bytes() {
    return (size_t)64 * (WN * 64 + 8) * sizeof(AT) + (size_t)2 * WN * 64 * sizeof(float);
  }
  template <typename AT, int WM, int WN>
  __device__ static __forceinline__ void run(f32x16_t (&acc)[2][2], const Args& e, const GemmShape& g, char* smem,
                                             int tid, int r0, int n0) {
    constexpr int BN = WN * 64, NT = WM * WN * 64, CSP = BN + 8;
    AT* Cs = reinterpret_cast<AT*>(smem);
    float* colsum = reinterpret_cast<float*>(smem + (size_t)64 * CSP * sizeof(AT));
    const int lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
    if (e.stats) {
      for (int i = tid; i < 2 * BN; i += NT) colsum[i] = 0.f;
    }
    __syncthreads();
    // bias (+tanh) and statistics in registers
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int n = wn * 64 + nt * 32 + (lane & 31);
      const float b = (e.bias && n0 + n < g.N) ? e.bias[n0 + n] : 0.f;
      const float cs = (e.colscale && n0 + n < g.N) ? e.colscale[n0 + n] : 1.f;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float y = fmaf(acc[mt][nt][r], cs, b);
          if (TANH) y = fast_tanh(y);
          acc[mt][nt][r] = y;
          const int row = r0 + wm * 64 + mt * 32 + cd_row(r, lane);
          if (row < g.M && tn_row_valid(e.rm, (uint32_t)row)) { s += y; q += y * y; }
        }
      }
      if (e.stats) {
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        if (lane < 32) {
          atomicAdd(&colsum[n], s);
          atomicAdd(&colsum[BN + n], q);
        }
      }
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
    if (e.stats) {
      // colsum complete: all LDS atomics happened before the first pass barrier
      const int rep = blockIdx.x % TN_NREP;
      for (int i = tid; i < 2 * BN; i += NT) {
        const int which = i / BN, n = i % BN;
        if (n0 + n < g.N) atomic_add_f32(&e.stats[(size_t)(rep * 2 + which) * g.N + n0 + n], colsum[i]);
      }
    }
  }
};

using EpiStore = EpiStoreT<false>;
using EpiStoreTanh = EpiStoreT<true>;

// ------------------------------------------------------------------------------------------
// host-side launcher
// ------------------------------------------------------------------------------------------
template <typename AT, int WM, int WN, typename Prod, typename Epi>
inline int launch_gemm(const GemmShape& g, const typename Prod::Args& pa, const typename Epi::Args& ea, int KD,
                       hipStream_t stream) {
  constexpr int BN = WN * 64;
  if constexpr (sizeof(AT) == 2 && Prod::kRaw && WM * WN == 8) {
    // big problems (>= 4 tiles per CU): 256-row workgroup tiles, 128 x 64 per wave
    const long tiles2 = (long)((g.M + 2 * WM * 64 - 1) / (2 * WM * 64)) * ((g.N + BN - 1) / BN);
    // wide outputs: 128 x 512 workgroup tiles (64 x 128 per wave), the row operand read once per 512 columns
    // (measured per producer: with two input streams — BatchNorm backward on load — the saved re-read wins, 343 -> 289 us at
    //  N = K = 1024; the single-stream producer is better off with the double-buffered 256 x 256 tile below, 241 vs 261 us)
    if (Prod::kWideCols && g.N >= 512 && (long)((g.M + WM * 64 - 1) / (WM * 64)) * ((g.N + 2 * BN - 1) / (2 * BN)) >= 512) {
      constexpr int BM = WM * 64, BNC = 2 * BN;
      size_t main_bytes = (size_t)(BM + BNC) * (Elem<AT>::BK + Elem<AT>::PAD) * sizeof(AT) + Prod::scratch_bytes(g.K, KD, BM, Elem<AT>::BK, sizeof(AT));
      size_t epi_bytes = Epi::template lds_bytes<AT, WM, WN>();
      size_t smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
      smem = (smem + 15) & ~(size_t)15;
      auto kern = gemm_nt_kernel<AT, WM, WN, Prod, Epi, 1, 2>;
      TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      dim3 grid((g.M + BM - 1) / BM, (g.N + BNC - 1) / BNC);
      hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, g, pa, ea);
      return (int)hipGetLastError();
    }
    constexpr int BM2 = WM * 64 * 2;
    const size_t main2 = (size_t)2 * (BM2 + BN) * (Elem<AT>::BK + Elem<AT>::PAD) * sizeof(AT) + Prod::scratch_bytes(g.K, KD, BM2, Elem<AT>::BK, sizeof(AT));
    if (tiles2 >= 1024 && main2 <= 160 * 1024) {          // two tile buffers + the producer's constants must fit the LDS
      constexpr int BM = BM2;
      size_t main_bytes = main2;
      size_t epi_bytes = Epi::template lds_bytes<AT, WM, WN>();
      size_t smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
      smem = (smem + 15) & ~(size_t)15;
      auto kern = gemm_nt_kernel<AT, WM, WN, Prod, Epi, 2>;
      TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM);          // x = column tile (see the kernel)
      hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, g, pa, ea);
      return (int)hipGetLastError();
    }
  }
  constexpr int BM = WM * 64;
  size_t main_bytes = gemm_tile_bytes<AT, WM, WN>() + Prod::scratch_bytes(g.K, KD, BM, Elem<AT>::BK, sizeof(AT));
  size_t epi_bytes = Epi::template lds_bytes<AT, WM, WN>();
  size_t smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  smem = (smem + 15) & ~(size_t)15;
  auto kern = gemm_nt_kernel<AT, WM, WN, Prod, Epi>;
  if (smem > 64 * 1024) {
    TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)smem));
  }
  dim3 grid((g.M + BM - 1) / BM, (g.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, g, pa, ea);
  return (int)hipGetLastError();
}


// ------------------------------------------------------------------------------------------
// fp8 variant (TN_PREC_FP8): C[M x N] = A8[M x K] * W8[N x K]^T on v_mfma_scale_f32_32x32x64_f8f6f4 (OCP e4m3 operands, unit
// block scales, f32 accumulate: twice the rate of the bf16 MFMA).
// Both operands are plain fp8 matrices in HBM (the depthwise producer of the wide models writes its output in e4m3
// beside the bf16 copy the weight gradients read; the weights are cast once per step with one scale per output row), so
// the tiles are straight 16-byte copies: half the operand bytes of the bf16 GEMM through HBM, L2 and LDS.  Same
// tiling, fragment layout (8 K-elements per lane) and epilogue as gemm_nt_kernel; the epilogue applies the row scales.
// ------------------------------------------------------------------------------------------
template <int WM, int WN, typename Epi, int RH = 1>
__global__ __launch_bounds__(WM* WN * 64, 2) void gemm_fp8_nt_kernel(GemmShape g, const uint8_t* __restrict__ A8, typename Epi::Args ea) {
  // same loop as the bf16 raw path of gemm_nt_kernel: next K chunk requested before the MFMAs, RH row halves per wave
  constexpr int BM = WM * 64 * RH, BN = WN * 64, BK = 128, BKP = BK + 16, NT = WM * WN * 64;
  constexpr int VC = BK / 16, RL = NT / VC, NA = BM / RL, NW = BN / RL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint8_t* As = reinterpret_cast<uint8_t*>(smem);
  uint8_t* Bs = As + BM * BKP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  f32x16_t acc[RH][2][2];
#pragma unroll
  for (int h = 0; h < RH; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;
  const uint8_t* W8 = reinterpret_cast<const uint8_t*>(g.W);
  const int vc = tid % VC, rl = tid / VC;
  uint4 areg[NA], wreg[NW];
  auto load = [&](int kc) {
    const int k = kc + vc * 16;
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int gr = r0 + rl + q * RL;
      areg[q] = (gr < g.M && k < g.K) ? *reinterpret_cast<const uint4*>(A8 + (size_t)gr * g.K + k) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const int gn = n0 + rl + q * RL;
      wreg[q] = (gn < g.N && k < g.K) ? *reinterpret_cast<const uint4*>(W8 + (size_t)gn * g.K + k) : make_uint4(0, 0, 0, 0);
    }
  };
  load(0);
  for (int kc = 0; kc < g.K; kc += BK) {
#pragma unroll
    for (int q = 0; q < NA; ++q) *reinterpret_cast<uint4*>(As + (rl + q * RL) * BKP + vc * 16) = areg[q];
#pragma unroll
    for (int q = 0; q < NW; ++q) *reinterpret_cast<uint4*>(Bs + (rl + q * RL) * BKP + vc * 16) = wreg[q];
    __syncthreads();
    if (kc + BK < g.K) load(kc + BK);
    // v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and unit block scales (E8M0 127): 64 k per instruction at twice the
    // rate of the 32x32x16 fp8 MFMA.  A lane feeds 32 bytes of its row; any assignment of k to (lane half, byte) works as long
    // as both operands use the same one (tools/f8_probe.hip): lane half h takes bytes [32 h, 32 h + 32) of the 64-k block.
    typedef __attribute__((ext_vector_type(8))) int i32x8_t;
    auto frag = [&](const uint8_t* ptr) {
      const uint4 lo = *reinterpret_cast<const uint4*>(ptr), hi = *reinterpret_cast<const uint4*>(ptr + 16);
      i32x8_t f;
      f[0] = (int)lo.x; f[1] = (int)lo.y; f[2] = (int)lo.z; f[3] = (int)lo.w; f[4] = (int)hi.x; f[5] = (int)hi.y; f[6] = (int)hi.z; f[7] = (int)hi.w;
      return f;
    };
    const uint8_t* arow0 = As + (wm * 64 + (lane & 31)) * BKP + (lane >> 5) * 32;
    const uint8_t* brow0 = Bs + (wn * 64 + (lane & 31)) * BKP + (lane >> 5) * 32;
    constexpr int SC1 = 0x7f7f7f7f;
#pragma unroll
    for (int ks = 0; ks < BK / 64; ++ks) {
      const i32x8_t b0 = frag(brow0 + ks * 64), b1 = frag(brow0 + 32 * BKP + ks * 64);
#pragma unroll
      for (int h = 0; h < RH; ++h) {
        const i32x8_t a0 = frag(arow0 + h * WM * 64 * BKP + ks * 64), a1 = frag(arow0 + (h * WM * 64 + 32) * BKP + ks * 64);
        acc[h][0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b0, acc[h][0][0], 0, 0, 0, SC1, 0, SC1);
        acc[h][0][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b1, acc[h][0][1], 0, 0, 0, SC1, 0, SC1);
        acc[h][1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b0, acc[h][1][0], 0, 0, 0, SC1, 0, SC1);
        acc[h][1][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b1, acc[h][1][1], 0, 0, 0, SC1, 0, SC1);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int h = 0; h < RH; ++h) {
    if (h > 0) __syncthreads();
    Epi::template run<bf16_t, WM, WN>(acc[h], ea, g, smem, tid, r0 + h * WM * 64, n0);
  }
}

template <typename Epi>
inline int launch_gemm_fp8(const GemmShape& g, const uint8_t* A8, const typename Epi::Args& ea, hipStream_t stream) {
  constexpr int WM = 2, WN = 4, BN = WN * 64;
  if (g.K % 64) return -2;                  // whole 64-k MFMA blocks
  const size_t epi = Epi::template lds_bytes<bf16_t, WM, WN>();
  const long tiles2 = (long)((g.M + 255) / 256) * ((g.N + BN - 1) / BN);
  if (tiles2 >= 1024) {                // big problems: 256-row workgroup tiles (see gemm_nt_kernel)
    constexpr int BM = 256;
    size_t smem = (size_t)(BM + BN) * (128 + 16);
    if (epi > smem) smem = epi;
    auto kern = gemm_fp8_nt_kernel<WM, WN, Epi, 2>;
    TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    dim3 grid((g.M + BM - 1) / BM, (g.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, g, A8, ea);
    return (int)hipGetLastError();
  }
  constexpr int BM = WM * 64;
  size_t smem = (size_t)(BM + BN) * (128 + 16);
  if (epi > smem) smem = epi;
  auto kern = gemm_fp8_nt_kernel<WM, WN, Epi, 1>;
  if (smem > 64 * 1024) TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  dim3 grid((g.M + BM - 1) / BM, (g.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, g, A8, ea);
  return (int)hipGetLastError();
}
