// titanet_amd — forward-pass kernels that are not GEMMs: SE squeeze + gate, residual combine,
// attentive-statistics pooling, decoder tail, loss heads, BatchNorm running-statistics update and the
// per-step weight cast/transposition.  All tensors are "rows x channels" (see tn_common.h).
#pragma once
#include <algorithm>

#include "tn_common.h"

// ------------------------------------------------------------------------------------------
// per-step weight preparation: fp32 master -> compute-precision copy and its transpose
// ------------------------------------------------------------------------------------------
struct CastDesc {
  const float* src;   // [R][C] fp32 master
  void* dst;          // [R][C] AT or null
  void* dstT;         // [C][R] AT or null
  int R, C;
};

// matrices of at least this many elements go through the LDS-tiled kernel below (round 5: 128 x 128 instead of 512 x 512 — the 68
// 256 x 256 pointwise weights of TitaNet-S spent 32 us per step in the element-wise kernel's strided 2-byte transposed stores)
#define TN_CAST_TILED_MIN (128 * 128)
template <typename AT>
__global__ void cast_params_kernel(const CastDesc* descs) {
  const CastDesc d = descs[blockIdx.y];
  const int n = d.R * d.C;
  if (sizeof(AT) == 2 && n >= TN_CAST_TILED_MIN && !(d.C & 3) && !(d.R & 3)) return;      // cast_params_tiled_kernel takes these
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float v = d.src[i];
    if (d.dst) reinterpret_cast<AT*>(d.dst)[i] = Elem<AT>::from_f(v);
    if (d.dstT) {
      const int r = i / d.C, c = i % d.C;
      reinterpret_cast<AT*>(d.dstT)[(size_t)c * d.R + r] = Elem<AT>::from_f(v);
    }
  }
}

// The same for matrices of >= TN_CAST_TILED_MIN elements (every pointwise weight, the epilog conv), 64 x 64 tiles through
// LDS: the transposed copy of cast_params_kernel is a 2-byte store per lane at a stride of a whole row — 64 cache lines per wave
// instruction, 218 us per step for the 24.7 M weights of TitaNet-L.  Here both copies are written as 8-byte row pieces.
// grid (tiles per launch, descriptors); descriptors with small matrices are left to cast_params_kernel (skipped here / there by
// the same size test).
template <typename AT>
__global__ __launch_bounds__(256) void cast_params_tiled_kernel(const CastDesc* descs) {
  const CastDesc d = descs[blockIdx.y];
  if (d.R * d.C < TN_CAST_TILED_MIN || sizeof(AT) != 2 || (d.C & 3) || (d.R & 3)) return;
  __shared__ float tile[64][65];
  const int tr = (d.R + 63) / 64, tc = (d.C + 63) / 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
    const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
    __syncthreads();
#pragma unroll
    for (int pss = 0; pss < 4; ++pss) {
      const int r = r0 + ty + 16 * pss, c = c0 + 4 * tx;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < d.R && c < d.C) v = *reinterpret_cast<const float4*>(d.src + (size_t)r * d.C + c);
      tile[ty + 16 * pss][4 * tx] = v.x; tile[ty + 16 * pss][4 * tx + 1] = v.y; tile[ty + 16 * pss][4 * tx + 2] = v.z; tile[ty + 16 * pss][4 * tx + 3] = v.w;
      if (d.dst && r < d.R && c < d.C) {
        uint2 o;
        o.x = f2bf_pk(v.x, v.y); o.y = f2bf_pk(v.z, v.w);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(d.dst) + (size_t)r * d.C + c) = o;
      }
    }
    __syncthreads();
    if (d.dstT) {
#pragma unroll
      for (int pss = 0; pss < 4; ++pss) {
        const int c = c0 + ty + 16 * pss, r = r0 + 4 * tx;      // row c of the transposed copy, its columns r .. r + 3
        if (c < d.C && r < d.R) {
          uint2 o;
          o.x = f2bf_pk(tile[4 * tx][ty + 16 * pss], tile[4 * tx + 1][ty + 16 * pss]);
          o.y = f2bf_pk(tile[4 * tx + 2][ty + 16 * pss], tile[4 * tx + 3][ty + 16 * pss]);
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(d.dstT) + (size_t)c * d.R + r) = o;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Packed prolog operand (ProdTaps, tn_gemm.h): the reference-layout input [B][C][T] float32 -> X0[B*T][C] AT (frames beyond
// an utterance's valid length written as zero), the prolog weight [H][C][KP] -> [H][KP][C] AT, and the way back for its
// gradient.  grid = (ceil(T / 64), B) x 256 threads; LDS tile [C][65] floats.
// ------------------------------------------------------------------------------------------
template <typename AT>
__global__ __launch_bounds__(256) void prolog_pack_kernel(const float* __restrict__ x, int C, int T, const int* __restrict__ len,
                                                          AT* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);      // [C][65]
  const int b = blockIdx.y, t0 = blockIdx.x * 64;
  const int L = len ? len[b] : T;
  const float* xb = x + (size_t)b * C * T;
  for (int i = threadIdx.x; i < C * 64; i += 256) {
    const int ci = i >> 6, tl = i & 63, t = t0 + tl;
    tile[ci * 65 + tl] = (t < L) ? xb[(size_t)ci * T + t] : 0.f;        // t < L <= T
  }
  __syncthreads();
  const int rows = min(64, T - t0);
  AT* ob = out + ((size_t)b * T + t0) * C;
  for (int i = threadIdx.x; i < rows * C; i += 256) {
    const int tl = i / C, ci = i - tl * C;
    ob[i] = Elem<AT>::from_f(tile[ci * 65 + tl]);
  }
}
template <typename AT>
__global__ void prolog_weight_taps_kernel(const float* __restrict__ w, int H, int C, int KP, AT* __restrict__ out) {
  const int n = H * C * KP;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int h = i / (C * KP), r = i - h * C * KP, j = r / C, ci = r - j * C;     // i indexes the OUTPUT [h][j][ci]
    out[i] = Elem<AT>::from_f(w[(size_t)h * C * KP + ci * KP + j]);
  }
}

// ------------------------------------------------------------------------------------------
// fp8 (OCP e4m3) helpers and the per-step weight cast of the TN_PREC_FP8 plan: W8[n][:] = e4m3(W[n][:] / s[n]) with
// s[n] = max|W[n][:]| / 448 (the largest finite e4m3), one workgroup (one wave) per output row.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2fp8x4(float a, float b, float c, float d) {
  uint32_t w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return w;
}
// src16 != null: the rows are read from a bf16 matrix (the transposed compute-precision copy W^T [ci][co] of a pointwise
// weight: its e4m3 rows with one scale per INPUT channel are the B operand of the fp8 data gradient dS * W)
struct Fp8CastDesc { const float* src; uint8_t* dst; float* scale; int R, C; const bf16_t* src16; };
__global__ __launch_bounds__(64) void cast_fp8_rows_kernel(const Fp8CastDesc* descs) {
  const Fp8CastDesc d = descs[blockIdx.y];
  const int lane = threadIdx.x;
  auto ld4 = [&](int r, int c) -> float4 {
    if (d.src16) {
      const uint2 w = *reinterpret_cast<const uint2*>(d.src16 + (size_t)r * d.C + c);
      return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4*>(d.src + (size_t)r * d.C + c);
  };
  for (int r = blockIdx.x; r < d.R; r += gridDim.x) {
    float m = 0.f;
    for (int c = lane * 4; c < d.C; c += 256) {
      const float4 v = ld4(r, c);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    m = wave_max(m);
    const float sc = m > 0.f ? m * (1.f / 448.f) : 1.f, inv = 1.f / sc;
    if (lane == 0) d.scale[r] = sc;
    for (int c = lane * 4; c < d.C; c += 256) {
      const float4 v = ld4(r, c);
      *reinterpret_cast<uint32_t*>(d.dst + (size_t)r * d.C + c) = f2fp8x4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Stand-alone depthwise producer for the WIDE models (hidden 1024, TitaNet-L): Q = dwconv_KD(act(X)) + b_dw
// (reference src/modules.py:65-75).  Fusing this producer into the pointwise GEMM (ProdDw) recomputes activation +
// stencil once per 256-column block of the output — 4 times at hidden 1024, with K = 11 taps — and made the forward GEMM
// VALU / LDS bound (1.3 ms per launch at B = 256); the depthwise output is kept for the weight gradients anyway, so here
// it is produced ONCE by a streaming kernel and the GEMM reads it as a plain operand.
// Persistent workgroups, each bound to one 256-channel slab and walking 64-row tiles (the per-channel constants — BatchNorm
// scale / shift from the replicated statistics, KD + 1 taps per channel — cost more than a tile if reloaded per tile): the
// activated input rows of a tile (+ halo) are staged in LDS with all their loads in flight at once, then every thread
// produces 8 output rows for its 8 channels, taps in registers.
// ------------------------------------------------------------------------------------------
template <typename AT, int KD>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const AT* __restrict__ X, BnAct act, const float* __restrict__ wdw,
                                                     const float* __restrict__ bdw, AT* __restrict__ Q, int M, int T, int C,
                                                     uint8_t* __restrict__ Q8) {
  constexpr int PAD = (KD - 1) / 2, RT = 64, ROWS = RT + KD - 1, CW = 256, XP = CW + 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AT* Xs = reinterpret_cast<AT*>(smem);                       // [ROWS][XP] activated rows
  float* cst = reinterpret_cast<float*>(Xs + ROWS * XP);      // sc, sh : [2][CW]
  const int tid = threadIdx.x, vc = tid & 31, rq = tid >> 5;
  const int cb = blockIdx.y * CW, c0 = cb + vc * 8;
  {
    float s = 1.f, h = 0.f;
    if (cb + tid < C) bn_scale_shift(act, C, cb + tid, s, h);
    cst[tid] = s; cst[CW + tid] = h;
  }
  __syncthreads();
  const bool cvalid = c0 < C;
  float bd[8], wd[KD][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    bd[i] = cvalid ? bdw[c0 + i] : 0.f;
#pragma unroll
    for (int k = 0; k < KD; ++k) wd[k][i] = cvalid ? wdw[(size_t)(c0 + i) * KD + k] : 0.f;
  }
  const int ntiles = (M + RT - 1) / RT;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const int r0 = tile * RT;
  __syncthreads();            // the previous tile's taps have been read
  for (int i = rq; i < ROWS; i += 8) {
    const int gr = r0 - PAD + i;
    float v[8];
    if (cvalid && gr >= 0 && gr < M) {
      load8(X + (size_t)gr * C + c0, v);
      act8(v, cst + vc * 8, cst + CW + vc * 8, act, (uint32_t)gr, C, c0);
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = 0.f;
    }
    store8(Xs + i * XP + vc * 8, v);
  }
  __syncthreads();
  if (cvalid)
  for (int r = rq; r < RT; r += 8) {
    const int gr = r0 + r;
    if (gr >= M) break;
    const int t = gr % T;
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = bd[i];
#pragma unroll
    for (int k = 0; k < KD; ++k) {
      const int tt = t + k - PAD;
      if (tt >= 0 && tt < T) {               // taps beyond the utterance read the zero padding
        float v[8];
        load8(Xs + (r + k) * XP + vc * 8, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = fmaf(wd[k][i], v[i], o[i]);
      }
    }
    store8(Q + (size_t)gr * C + c0, o);
    if (Q8) {       // TN_PREC_FP8: the same values in e4m3 for the fp8 pointwise GEMM (unit scale: the operand is O(1))
      uint2 w;
      w.x = f2fp8x4(o[0], o[1], o[2], o[3]);
      w.y = f2fp8x4(o[4], o[5], o[6], o[7]);
      *reinterpret_cast<uint2*>(Q8 + (size_t)gr * C + c0) = w;
    }
  }
  }
}
// ------------------------------------------------------------------------------------------
// dw_fwd_slab: the same depthwise producer for the wide bf16 models (C % 256 == 0, K = 7 / 11) in the structure of
// dw_bwd_slab (tn_v2_bwd_kernels.h): one 256-channel slab per workgroup, raw X tiles of 64 + K - 1 rows by LDS-DMA into two
// LDS buffers, a lane owns 2 channels and half a 16-row strip, the K-row window of ACTIVATED inputs rolls down the strip
// in registers (a window row is read, unpacked and activated once per strip).  FL = activation flags of X (1 BN, 2 ReLU,
// 4 dropout).  dw_fwd_kernel above: 169 us per TitaNet-L layer for 314 MB.
// ------------------------------------------------------------------------------------------
struct DwFwdSlabArgs {
  const bf16_t* X; BnAct act;
  const float* wdw;    // [C][KD]
  const float* bdw;    // [C]
  bf16_t* Q;           // [M][C], or null when only Q8 is wanted
  uint8_t* Q8;         // [M][C] e4m3 copy for the fp8 pointwise GEMM (TN_PREC_FP8; needs 4 channels per lane) or null
  int M, T, C, ntiles;
  // variable-length batches: the 256-row tiles that hold at least one valid frame (PGemmNtArgs::rowtiles) or null = all rows.
  // A padding-only tile is not walked at all: its rows keep what they held, which nobody reads (the pipelined GEMMs skip the
  // same tiles, every other consumer masks padding rows on load)
  const int* rowtiles; int n_rowtiles;
};
// MK: variable-length batch (a.act.rm.len), a compile-time flag: the fixed-length instantiation carries none of it
template <int KD, int FL, int CH, bool MK = false>
__global__ __launch_bounds__(512, 2) void dw_fwd_slab_kernel(DwFwdSlabArgs a) {
  constexpr int PADR = (KD - 1) / 2, ROWS = 64 + KD - 1, WPS = 4 / CH, RS = 8 * WPS;
  constexpr int TILE_B = ROWS * 512;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // sc, sh, bias, wd[KD] : [3 + KD][256] — staged through the SECOND tile buffer, which the first DMA does not touch and which is
  // only refilled behind the loop's first barrier (every thread has its constants in registers by then): 2 x 37 KB at K = 11
  float* cst = reinterpret_cast<float*>(smem + TILE_B);
  static_assert((3 + KD) * 256 * 4 <= TILE_B, "constants fit a tile buffer");
  const unsigned ring_lds = (unsigned)(uintptr_t)(tn_lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cl = (wave % WPS) * 64 * CH + lane * CH;             // first channel of the lane inside the slab
  const int strip = wave / WPS;
  const int nslab = a.C / 256;
  const int slab = blockIdx.x % nslab, first = blockIdx.x / nslab, stride = gridDim.x / nslab;
  const int cb = slab * 256;
  const uint32_t dkey = tn_act_key(a.act), dthr = a.act.drop_thr;
  const int* __restrict__ len = MK ? a.act.rm.len : nullptr;     // valid frames per utterance (uniform)
  if (tid < 256) {
    float s = 1.f, h = 0.f;
    if (FL & 1) bn_scale_shift(a.act, a.C, cb + tid, s, h);
    cst[tid] = s; cst[256 + tid] = h; cst[512 + tid] = a.bdw[cb + tid];
#pragma unroll
    for (int k = 0; k < KD; ++k) cst[(3 + k) * 256 + tid] = a.wdw[(size_t)(cb + tid) * KD + k];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // compiler-visible loads done before the first DMA
  // first output row of a 64-row tile (a listed 256-row tile = 4 of them)
  auto tile_row0 = [&](int tile) -> int { return (MK && a.rowtiles) ? tn_sload_i32(a.rowtiles, tile >> 2) * 256 + (tile & 3) * 64 : tile * 64; };
  auto dma_tile = [&](int tile, int buf) {
    const int raw0 = tile_row0(tile) - PADR;
#pragma unroll
    for (int i = 0; i < (ROWS / 2 + 7) / 8; ++i) {
      const int r = 2 * (wave + 8 * i);
      if (r < ROWS) {
        int gr = raw0 + r + (lane >> 5);
        gr = gr < 0 ? 0 : (gr >= a.M ? a.M - 1 : gr);            // rows outside the tensor: any valid row (never used)
        tn_dma16(a.X + (size_t)gr * a.C + cb + (lane & 31) * 8,
                 (unsigned)__builtin_amdgcn_readfirstlane((int)(ring_lds + buf * TILE_B + r * 512)));
      }
    }
  };
  if (first < a.ntiles) dma_tile(first, 0);
  __syncthreads();
  float sc[CH], sh[CH], bd[CH], wd[KD][CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    sc[i] = cst[cl + i]; sh[i] = cst[256 + cl + i]; bd[i] = cst[512 + cl + i];
#pragma unroll
    for (int k = 0; k < KD; ++k) wd[k][i] = cst[(3 + k) * 256 + cl + i];
  }
  int buf = 0;
  for (int tile = first; tile < a.ntiles; tile += stride, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this tile's DMA (and the previous tile's stores)
    __builtin_amdgcn_s_barrier();
    if (tile + stride < a.ntiles) dma_tile(tile + stride, buf ^ 1);
    const bf16_t* Xs = reinterpret_cast<const bf16_t*>(smem + buf * TILE_B);
    const int out0 = tile_row0(tile), raw0 = out0 - PADR;
    const int l0 = strip * RS;
    const int g_first = raw0 + l0, g_last = g_first + KD + RS - 2;
    // variable-length batches (a.act.rm.len): frames >= len[b] are padding — they read as zeros and are WRITTEN as zeros (the
    // pointwise GEMM then sees a plain operand whose padding rows give y = bias exactly, tn_pgemm.h pad_rows)
    const int tf = g_first >= 0 ? g_first % a.T : 0;
    const bool inside = g_first >= 0 && g_last < a.M && tf + KD + RS - 2 < a.T;              // window inside one utterance
    int Lf = a.T;
    if (len && inside) Lf = tn_sload_i32(len, g_first / a.T);
    const bool fast = inside && tf + KD + RS - 2 < Lf;                                       // wave-uniform
    auto row_ok = [&](int gr) -> bool {                                                      // wave-uniform argument
      if (gr < 0 || gr >= a.M) return false;
      if (!len) return true;
      const int b = gr / a.T;
      return gr - b * a.T < tn_sload_i32(len, b);
    };
    if (len && inside && tf + PADR >= Lf) {
      // every output row of the strip is padding
      float z[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) z[i] = 0.f;
#pragma unroll
      for (int o = 0; o < RS; ++o) {
        const int gr = out0 + l0 + o;
        if (a.Q) st_ch<CH>(a.Q + (size_t)gr * a.C + cb + cl, z);
        if (CH == 4 && a.Q8) *reinterpret_cast<uint32_t*>(a.Q8 + (size_t)gr * a.C + cb + cl) = 0u;
      }
    } else
    // one path: the window of activated rows rolls down the strip either way (a row is activated ONCE — with the dropout
    // hash per tap the boundary strips, 9 % of them at T = 300, cost more than all the others together); strips whose window
    // leaves the utterance of an output row only add a wave-uniform test per tap
    {
      float A[KD][CH];
      auto place = [&](int j, int slot) {
        const int gr = g_first + j;
        if (fast || row_ok(gr)) {
          ld_ch<CH>(Xs + (l0 + j) * 256 + cl, A[slot]);
          act_c<FL, CH>(A[slot], sc, sh, dkey, dthr, (uint32_t)gr, a.C, cb + cl);
        } else {
#pragma unroll
          for (int i = 0; i < CH; ++i) A[slot][i] = 0.f;
        }
      };
#pragma unroll
      for (int j = 0; j < KD - 1; ++j) place(j, j % KD);
#pragma unroll
      for (int o = 0; o < RS; ++o) {
        place(o + KD - 1, (o + KD - 1) % KD);
        const int gr = out0 + l0 + o;
        float q[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) q[i] = bd[i];
        if (fast) {
#pragma unroll
          for (int k = 0; k < KD; ++k)
#pragma unroll
            for (int i = 0; i < CH; ++i) q[i] = fmaf(wd[k][i], A[(o + k) % KD][i], q[i]);
        } else {
          const int t = gr % a.T;
#pragma unroll
          for (int k = 0; k < KD; ++k) {
            const int tt = t + k - PADR;
            if (tt >= 0 && tt < a.T) {
#pragma unroll
              for (int i = 0; i < CH; ++i) q[i] = fmaf(wd[k][i], A[(o + k) % KD][i], q[i]);
            }
          }
          if (len && !row_ok(gr)) {
#pragma unroll
            for (int i = 0; i < CH; ++i) q[i] = 0.f;
          }
        }
        if (fast || gr < a.M) {
          if (a.Q) st_ch<CH>(a.Q + (size_t)gr * a.C + cb + cl, q);      // (null: only the e4m3 copy is kept, fp8 weight gradient)
          if (CH == 4 && a.Q8) *reinterpret_cast<uint32_t*>(a.Q8 + (size_t)gr * a.C + cb + cl) = f2fp8x4(q[0], q[1], q[CH - 2], q[CH - 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}
template <int KD, int FL>
inline int launch_dw_fwd_slab_t(DwFwdSlabArgs a, int grid, hipStream_t st) {
  const size_t smem = (size_t)2 * (64 + KD - 1) * 512;
  // 4 channels per lane (2 measured 140 vs 103 us with dropout: one hash per 8 channels)
  auto kern = a.act.rm.len ? dw_fwd_slab_kernel<KD, FL, 4, true> : dw_fwd_slab_kernel<KD, FL, 4, false>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -4;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, a);
  return (int)hipGetLastError();
}
// -1000: no specialisation (caller runs dw_fwd_kernel)
inline int launch_dw_fwd_slab(DwFwdSlabArgs a, int KD, hipStream_t st) {
  if (a.C % 256 != 0) return -1000;
  if (!a.act.rm.len) { a.rowtiles = nullptr; a.n_rowtiles = 0; }
  a.ntiles = a.rowtiles ? a.n_rowtiles * 4 : (a.M + 63) / 64;
  if (a.ntiles <= 0) return 0;
  const int nslab = a.C / 256;
  // K = 7 (TitaNet-M): 80 KB of LDS and <= 114 VGPRs — TWO workgroups fit a CU, and one's tile-top wait (its DMA and the previous
  // tile's stores, vmcnt(0)) hides behind the other's arithmetic: 49.9 -> 41.3 us per layer at 76800 x 512 with 512 workgroups
  // (tools/dw_slab_harness, profiles/r05_dw_slab_wgs.txt).  K = 11 needs 140 - 152 VGPRs: one per CU (512 workgroups: 116 vs 113 us;
  // capped at 128 registers it spills 20 and runs 140 - 173 us)
  static const int forced_wgs = [] { const char* e = getenv("TN_DWF_WGS"); return e && atoi(e) > 0 ? atoi(e) : 0; }();      // (tuning switch)
  const int max_wgs = forced_wgs ? forced_wgs : (KD == 7 ? 512 : 256);
  int per = max_wgs / nslab;
  if (per < 1) per = 1;
  if (per > a.ntiles) per = a.ntiles;
  const int grid = per * nslab;
  const int fl = (a.act.mode != 0 ? 1 : 0) | (a.act.relu ? 2 : 0) | (a.act.drop_thr ? 4 : 0);
#define TN_DWFS(K)                                                             \
  case K:                                                                      \
    switch (fl) {                                                              \
      case 7: return launch_dw_fwd_slab_t<K, 7>(a, grid, st);                  \
      case 3: return launch_dw_fwd_slab_t<K, 3>(a, grid, st);                  \
      case 0: return launch_dw_fwd_slab_t<K, 0>(a, grid, st);                  \
      default: return -1000;                                                   \
    }
  switch (KD) {
    TN_DWFS(7)
    TN_DWFS(11)
    default: return -1000;
  }
#undef TN_DWFS
}

template <typename AT>
inline int launch_dw_fwd(const AT* X, const BnAct& act, const float* wdw, const float* bdw, AT* Q, int M, int T, int C, int KD,
                         hipStream_t st, uint8_t* Q8 = nullptr) {
  const int slabs = (C + 255) / 256, tiles = (M + 63) / 64;
  dim3 grid(std::min(tiles, std::max(1, 768 / slabs)), slabs);       // ~3 resident workgroups per CU
  const size_t smem = (size_t)(64 + KD - 1) * (256 + 8) * sizeof(AT) + 2 * 256 * sizeof(float);
  switch (KD) {
#define TN_DWF_CASE(K)                                                                                              \
  case K: {                                                                                                         \
    auto kern = dw_fwd_kernel<AT, K>;                                                                               \
    if (smem > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return -4; \
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, X, act, wdw, bdw, Q, M, T, C, Q8);                          \
  } break;
    TN_DWF_CASE(3) TN_DWF_CASE(5) TN_DWF_CASE(7) TN_DWF_CASE(9) TN_DWF_CASE(11) TN_DWF_CASE(13) TN_DWF_CASE(15)
#undef TN_DWF_CASE
    default: return -2;   // TN_E_UNSUPPORTED
  }
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Squeeze-and-Excitation: m = mean_T act(Y3); h = relu(W1 m); g = sigmoid(W2 h)
// (reference src/modules.py:173-189).  One workgroup per utterance — or, for small batches of long utterances (32 x 2000
// frames would occupy 32 of the 256 CUs), gridDim.y = P workgroups per utterance that store their partial column sums in
// `acc` ([B][P][C], plain stores: the result does not depend on the order the workgroups run in); a second launch with
// mode 2 (one workgroup per utterance) adds them in order and turns the sums into mean / h / g.
//   mode 0: everything in one workgroup;  1: partial sums of frames [p L/P, (p+1) L/P) -> acc[b][p];  2: acc[b][0..parts) -> m, h, g
//   mode 3 (round 6): 1 and 2 in ONE launch — the workgroup that arrives LAST at the utterance's counter (cnt[b], zero
//   between launches: the last arrival clears it) adds the partial sums in order and makes m / h / g: the result does not
//   depend on which workgroup that is.  The frames are split by the utterance's own length (L / P each), not T / P.
// ------------------------------------------------------------------------------------------
// FL >= 0: the activation flags as a compile-time constant (1 BatchNorm, 2 ReLU, 4 dropout) — the row loop of the generic form
// (FL = -1: run-time flags, a padding test with an integer division per row) is VALU-bound at hidden 512 / 1024
template <typename AT, int FL = -1>
__global__ __launch_bounds__(512) void se_squeeze_fc_kernel(const AT* __restrict__ Y, BnAct act, int T, int C, int Hr,
                                                            const float* __restrict__ W1, const float* __restrict__ W2,
                                                            float* __restrict__ m_out, float* __restrict__ h_out,
                                                            float* __restrict__ g_out, float* __restrict__ acc, int mode, int parts,
                                                            int* __restrict__ cnt = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_last;
  float* sc = reinterpret_cast<float*>(smem);
  float* sh = sc + C;
  float* mean = sh + C;
  float* hbuf = mean + C;          // [Hr]
  float* part = hbuf + ((Hr + 3) & ~3);   // [TG][C]
  const int tid = threadIdx.x, NT = blockDim.x, b = blockIdx.x;
  const int CV = C / 8, TG = NT / CV;
  act = tn_resolve_key(act);
  if (act.mode != 0)
    for (int c = tid; c < C; c += NT) bn_scale_shift(act, C, c, sc[c], sh[c]);
  __syncthreads();
  const int vc = tid % CV, tg = tid / CV;
  // frames of this workgroup; padding frames (>= len[b]) contribute nothing and are not read
  const int L = act.rm.len ? act.rm.len[b] : T;
  const int per = (L + (int)gridDim.y - 1) / (int)gridDim.y;
  const int t_lo = (int)blockIdx.y * per, t_hi = min(L, t_lo + per);
  if (tg < TG && mode != 2) {
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = 0.f;
    // 6 rows in flight per thread (one workgroup per utterance: with one load at a time this loop was an HBM round trip per
    // row, 116 us for the 157 MB of a TitaNet-L tensor).  Branch-free: rows past the end are CLAMPED to the last one and
    // added with weight zero — predicated loads compiled to one exec-masked block (with its own wait) per row, 2 TB/s
    constexpr int U = 6;
    float kr[8], hr[8];                          // the thread's BatchNorm constants in registers (LDS reads per element: bank conflicts)
#pragma unroll
    for (int i = 0; i < 8; ++i) { kr[i] = act.mode != 0 ? sc[vc * 8 + i] : 1.f; hr[i] = act.mode != 0 ? sh[vc * 8 + i] : 0.f; }
    for (int t0 = t_lo + tg; t0 < t_hi; t0 += TG * U) {
      float v[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = min(t0 + u * TG, t_hi - 1);
        load8(Y + ((size_t)b * T + t) * C + vc * 8, v[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u * TG;
        const uint32_t row = (uint32_t)b * T + min(t, t_hi - 1);
        if constexpr (FL >= 0) {
          // (every row visited is a valid frame: t < t_hi <= len[b])
          if (FL & 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[u][i] = v[u][i] * kr[i] + hr[i];
          }
          if (FL & 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[u][i] = fmaxf(v[u][i], 0.f);
          }
          if (FL & 4) tn_drop8(v[u], (row * (uint32_t)C + (uint32_t)(vc * 8)) >> 3, act.drop_key, act.drop_thr);      // (key resolved above)
        } else {
          act8(v[u], kr, hr, act, row, C, vc * 8);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] += (t < t_hi) ? v[u][i] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) part[tg * C + vc * 8 + i] = s[i];
  }
  __syncthreads();
  if (mode == 1) {
    for (int c = tid; c < C; c += NT) {
      float s = 0.f;
      for (int k = 0; k < TG; ++k) s += part[k * C + c];
      acc[((size_t)b * gridDim.y + blockIdx.y) * C + c] = s;
    }
    return;
  }
  if (mode == 3) {
    // The partial sums cross workgroups INSIDE a launch (the XCDs' L2s are not coherent with each other): device-scope
    // exchanges / loads, performed at the memory side, and a relaxed counter — a release / acquire pair at device scope is an
    // L2 write-back + invalidate per workgroup (measured: + 90 us per launch behind a GEMM's dirty output lines).  The
    // RETURNING exchange has been performed when its value is back, i.e. before the barrier that precedes the arrival.
    for (int c = tid; c < C; c += NT) {
      float s = 0.f;
      for (int k = 0; k < TG; ++k) s += part[k * C + c];
      const float old = __hip_atomic_exchange(acc + ((size_t)b * gridDim.y + blockIdx.y) * C + c, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" : : "v"(old));
    }
    __syncthreads();
    if (tid == 0) {
      s_last = __hip_atomic_fetch_add(cnt + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.y - 1;
      if (s_last) __hip_atomic_store(cnt + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
  }
  for (int c = tid; c < C; c += NT) {
    float s = 0.f;
    if (mode == 3)
      for (int k = 0; k < parts; ++k) s += __hip_atomic_load(acc + ((size_t)b * parts + k) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (mode == 2)
      for (int k = 0; k < parts; ++k) s += acc[((size_t)b * parts + k) * C + c];
    else
      for (int k = 0; k < TG; ++k) s += part[k * C + c];
    s *= 1.f / (float)max(L, 1);   // mean over the valid frames
    mean[c] = s;
    m_out[(size_t)b * C + c] = s;
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, NW = NT >> 6;
  for (int j = wave; j < Hr; j += NW) {
    // lane l: channels l, l + 64, ...
    float s = lane < C ? tn_dot_batched(W1 + (size_t)j * C + lane, 64, mean + lane, 64, (C - lane + 63) / 64) : 0.f;
    s = wave_sum(s);
    if (lane == 0) {
      s = fmaxf(s, 0.f);
      hbuf[j] = s;
      h_out[(size_t)b * Hr + j] = s;
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += NT) {
    const float s = tn_dot_batched(W2 + (size_t)c * Hr, 1, hbuf, 1, Hr);
    g_out[(size_t)b * C + c] = 1.f / (1.f + __expf(-s));
  }
}

// ------------------------------------------------------------------------------------------
// A = act(X) stored as a plain bf16 operand (wide models, the prolog's output): the first mega block's skip conv and its weight
// gradient then run on the pipelined GEMMs like every other block's (with the activation applied on load they were the
// generic GEMM and the generic weight gradient: 368 + 498 us per TitaNet-L step against 188 + ~125)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_store_kernel(const bf16_t* __restrict__ X, BnAct act, bf16_t* __restrict__ OUT, int M, int C) {
  extern __shared__ __attribute__((aligned(16))) float as_k[];      // sc, sh : [2][C]
  act = tn_resolve_key(act);
  for (int c = threadIdx.x; c < C; c += 256) bn_scale_shift(act, C, c, as_k[c], as_k[C + c]);
  __syncthreads();
  const int VC = C / 8;
  const size_t nvec = (size_t)M * VC;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    const int c0 = (int)(i % VC) * 8;
    float v[8];
    load8(X + i * 8, v);
    act8(v, as_k + c0, as_k + C + c0, act, (uint32_t)(i / VC), C, c0);
    store8(OUT + i * 8, v);
  }
}

// ------------------------------------------------------------------------------------------
// Mega-block tail: OUT = dropout(relu(BN(S) + g * act3(Y3)))   (reference src/models.py:467-472)
// ------------------------------------------------------------------------------------------
template <typename AT, bool MK = false>
__global__ __launch_bounds__(256) void combine_fwd_kernel(const AT* __restrict__ S, BnAct actS,
                                                          const AT* __restrict__ Y3, BnAct act3,
                                                          const float* __restrict__ gate, AT* __restrict__ OUT, int M,
                                                          int T, int C, int rows_per_block, uint32_t drop_thr,
                                                          uint32_t drop_key, float inv_keep, const uint32_t* key_add,
                                                          const int* __restrict__ rowtiles = nullptr) {
  // rowtiles (variable-length batches, rows_per_block divides 256): the 256-row tiles with valid frames; the grid covers only
  // those (padding-only tiles keep what they held: nobody reads them, tn_pgemm.h PGemmNtArgs::rowtiles)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* scS = reinterpret_cast<float*>(smem);
  float* shS = scS + C;
  float* sc3 = shS + C;
  float* sh3 = sc3 + C;
  const int tid = threadIdx.x, NT = blockDim.x;
  act3 = tn_resolve_key(act3);
  for (int c = tid; c < C; c += NT) {
    bn_scale_shift(actS, C, c, scS[c], shS[c]);
    bn_scale_shift(act3, C, c, sc3[c], sh3[c]);
  }
  __syncthreads();
  const int CV = C / 8;
  int r_begin = blockIdx.x * rows_per_block;
  if (MK && rowtiles) {
    const int per = 256 / rows_per_block;
    r_begin = rowtiles[blockIdx.x / per] * 256 + (blockIdx.x % per) * rows_per_block;
  }
  const int r_end = min(M, r_begin + rows_per_block);
  const uint32_t okey = key_add ? drop_key + *key_add : drop_key;
  if (NT % CV == 0) {
    // a thread keeps ONE 8-channel vector: its four BatchNorm constants live in registers (read from LDS per element they cost
    // 85 % bank-conflict cycles and a third of the wave time in LDS waits, profiles/r03_m10_sq_counters.json), no index division
    const int c0 = (tid % CV) * 8, rstep = NT / CV;
    float kS[8], hS[8], k3[8], h3[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { kS[q] = scS[c0 + q]; hS[q] = shS[c0 + q]; k3[q] = sc3[c0 + q]; h3[q] = sh3[c0 + q]; }
    // round 5: U rows in flight per thread, the utterance index carried along instead of divided out per row (the masked
    // variant divided three times per row and waited for len[b] each time: 58 us for the 45 k listed rows of configs[3] against 50
    // for 76.8 k unmasked ones), padding rows inside a listed tile are read like any other and stored as zeros
    BnAct a3 = act3;
    a3.rm.len = nullptr;                                   // (masked explicitly below)
    const int* __restrict__ len = MK ? act3.rm.len : nullptr;
    constexpr int U = 4;
    int row = r_begin + tid / CV;
    int b = row / T, rem = row - b * T;
    for (; row < r_end; row += U * rstep) {
      float s[U][8], y[U][8], g[U][8];
      int bb[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = row + u * rstep;
        const int rc = min(r, r_end - 1);                  // rows past the block: a clamped re-read, not stored
        bb[u] = r < r_end ? b : (r_end - 1) / T;
        ok[u] = !len || rem < len[bb[u]];
        load8(S + (size_t)rc * C + c0, s[u]);
        load8(Y3 + (size_t)rc * C + c0, y[u]);
        load8(gate + (size_t)bb[u] * C + c0, g[u]);
        rem += rstep;
        while (rem >= T) { rem -= T; ++b; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = row + u * rstep;
        if (r < r_end) {
          float o[8];
          act8(y[u], k3, h3, a3, (uint32_t)r, C, c0);
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = ok[u] ? fmaxf(s[u][q] * kS[q] + hS[q] + g[u][q] * y[u][q], 0.f) : 0.f;
          if (drop_thr) {
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] *= inv_keep;
            tn_drop8(o, ((uint32_t)r * (uint32_t)C + (uint32_t)c0) >> 3, okey, drop_thr);
          }
          store8(OUT + (size_t)r * C + c0, o);
        }
      }
    }
    return;
  }
  for (int i = tid; i < (r_end - r_begin) * CV; i += NT) {
    const int row = r_begin + i / CV, c0 = (i % CV) * 8;
    const int b = row / T;
    float s[8], y[8], g[8], o[8];
    load8(S + (size_t)row * C + c0, s);
    load8(Y3 + (size_t)row * C + c0, y);
    load8(gate + (size_t)b * C + c0, g);
    act8(y, sc3 + c0, sh3 + c0, act3, (uint32_t)row, C, c0);
    const bool pad = MK && !tn_row_valid(act3.rm, (uint32_t)row);              // padding rows are stored as zeros: the block
#pragma unroll                                                                 // output is a plain operand for its consumers
    for (int q = 0; q < 8; ++q) o[q] = pad ? 0.f : fmaxf(s[q] * scS[c0 + q] + shS[c0 + q] + g[q] * y[q], 0.f);
    if (drop_thr) {
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] *= inv_keep;
      tn_drop8(o, ((uint32_t)row * (uint32_t)C + (uint32_t)c0) >> 3, okey, drop_thr);
    }
    store8(OUT + (size_t)row * C + c0, o);
  }
}

// ------------------------------------------------------------------------------------------
// Attentive statistics pooling, second half (reference src/models.py:569-584): per (utterance,
// channel) online softmax over time of the energies, weighted mean and std of x = act(E).
// Lane-local over time (channels-last layout): no cross-lane reduction inside the loop.
//   pooled[b][c] = mean, pooled[b][D + c] = std; saves softmax max / 1/sum and q = sum alpha x^2.
// Also accumulates the BatchNorm1d(2D) batch statistics (reference src/models.py:506).
// ------------------------------------------------------------------------------------------
// CVB x TG = 256 threads: CVB vectors of 8 channels x TG time groups per workgroup; grid (B, ceil(D / (8 CVB))).  64 x 4 for
// large batches, 16 x 16 for small batches of long utterances (4x the workgroups, 4x shorter time loops)
template <typename AT, int CVB = 64, int TG = 4>
__global__ __launch_bounds__(256) void asp_pool_fwd_kernel(const AT* __restrict__ E, BnAct actE,
                                                           const AT* __restrict__ EN, int T, int D, float eps,
                                                           float* __restrict__ pooled, float* __restrict__ smax,
                                                           float* __restrict__ sinv, float* __restrict__ qout,
                                                           float* __restrict__ stats, float* __restrict__ partial = nullptr) {
  // gridDim.z = P > 1 (round 6, small batches of long utterances): the utterance's valid frames are split over P workgroups,
  // each stores its (max, sum, sum x, sum x^2) per channel in partial[b][part][4][D]; asp_pool_merge_kernel merges them with
  // the same rescaling the time groups of one workgroup are merged with below.
  static_assert(CVB * TG == 256, "256 threads");
  __shared__ float red[TG][4][CVB * 8];
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
  float m[8], l[8], s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { m[i] = -INFINITY; l[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f; }
  if (c0 < D) {
    const int Lb = actE.rm.len ? actE.rm.len[b] : T;     // softmax over the valid frames only
    const int per = (Lb + (int)gridDim.z - 1) / (int)gridDim.z;
    const int t_lo = (int)blockIdx.z * per, L = min(Lb, t_lo + per);      // this workgroup's frames [t_lo, L)
    // U rows in flight per thread (round 6): with one row per trip the loop was an HBM round trip per row — 123 trips for a
    // 1969-frame utterance at 16 time groups, 177 us on configs[3].  Rows past the end are clamped (loaded, not used); the
    // rows are consumed in the same order as before: bit-identical results.
    constexpr int U = 4;
    for (int t0 = t_lo + tg; t0 < L; t0 += TG * U) {
      float x[U][8], e[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t row = (uint32_t)b * T + min(t0 + u * TG, L - 1);
        load8(E + (size_t)row * D + c0, x[u]);
        load8(EN + (size_t)row * D + c0, e[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (t0 + u * TG < L) {
          act8(x[u], scs + vc * 8, shs + vc * 8, actE, (uint32_t)b * T + t0 + u * TG, D, c0);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float mn = fmaxf(m[i], e[u][i]);
            const float r = __expf(m[i] - mn);      // exp(-inf) = 0 on the first step
            const float p = __expf(e[u][i] - mn);
            l[i] = l[i] * r + p;
            s1[i] = s1[i] * r + p * x[u][i];
            s2[i] = s2[i] * r + p * x[u][i] * x[u][i];
            m[i] = mn;
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    red[tg][0][vc * 8 + i] = m[i];
    red[tg][1][vc * 8 + i] = l[i];
    red[tg][2][vc * 8 + i] = s1[i];
    red[tg][3][vc * 8 + i] = s2[i];
  }
  __syncthreads();
  for (int c = tid; c < CVB * 8; c += 256) {
    const int cg = cbase + c;
    if (cg >= D) continue;
    float M = -INFINITY;
    for (int k = 0; k < TG; ++k) M = fmaxf(M, red[k][0][c]);
    float L = 0.f, S1 = 0.f, S2 = 0.f;
    for (int k = 0; k < TG; ++k) {
      const float r = __expf(red[k][0][c] - M);
      L += red[k][1][c] * r; S1 += red[k][2][c] * r; S2 += red[k][3][c] * r;
    }
    if (gridDim.z > 1) {
      float* pp = partial + ((size_t)b * gridDim.z + blockIdx.z) * 4 * D + cg;
      pp[0] = M; pp[D] = L; pp[2 * D] = S1; pp[3 * D] = S2;
      continue;
    }
    const float inv = 1.f / L;
    const float mu = S1 * inv, q = S2 * inv;
    const float sd = sqrtf(fmaxf(q - mu * mu, eps));
    pooled[(size_t)b * 2 * D + cg] = mu;
    pooled[(size_t)b * 2 * D + D + cg] = sd;
    smax[(size_t)b * D + cg] = M;
    sinv[(size_t)b * D + cg] = inv;
    qout[(size_t)b * D + cg] = q;
    if (stats) {
      const int rep = b % TN_NREP;
      atomic_add_f32(&stats[(size_t)(rep * 2 + 0) * 2 * D + cg], mu);
      atomic_add_f32(&stats[(size_t)(rep * 2 + 1) * 2 * D + cg], mu * mu);
      atomic_add_f32(&stats[(size_t)(rep * 2 + 0) * 2 * D + D + cg], sd);
      atomic_add_f32(&stats[(size_t)(rep * 2 + 1) * 2 * D + D + cg], sd * sd);
    }
  }
}

// merges the P partial (max, sum, sum x, sum x^2) records of asp_pool_fwd_kernel<.., gridDim.z = P>; grid (B, ceil(D / 256))
__global__ __launch_bounds__(256) void asp_pool_merge_kernel(const float* __restrict__ partial, int P, int D, float eps,
                                                             float* __restrict__ pooled, float* __restrict__ smax,
                                                             float* __restrict__ sinv, float* __restrict__ qout,
                                                             float* __restrict__ stats) {
  const int b = blockIdx.x, cg = blockIdx.y * 256 + threadIdx.x;
  if (cg >= D) return;
  const float* pp = partial + (size_t)b * P * 4 * D + cg;
  float M = -INFINITY;
  for (int k = 0; k < P; ++k) M = fmaxf(M, pp[(size_t)k * 4 * D]);
  float L = 0.f, S1 = 0.f, S2 = 0.f;
  for (int k = 0; k < P; ++k) {
    const float* q = pp + (size_t)k * 4 * D;
    const float r = __expf(q[0] - M);          // (a part without frames: max = -inf, weight 0)
    L += q[D] * r; S1 += q[2 * D] * r; S2 += q[3 * D] * r;
  }
  const float inv = 1.f / L;
  const float mu = S1 * inv, q = S2 * inv;
  const float sd = sqrtf(fmaxf(q - mu * mu, eps));
  pooled[(size_t)b * 2 * D + cg] = mu;
  pooled[(size_t)b * 2 * D + D + cg] = sd;
  smax[(size_t)b * D + cg] = M;
  sinv[(size_t)b * D + cg] = inv;
  qout[(size_t)b * D + cg] = q;
  if (stats) {
    const int rep = b % TN_NREP;
    atomic_add_f32(&stats[(size_t)(rep * 2 + 0) * 2 * D + cg], mu);
    atomic_add_f32(&stats[(size_t)(rep * 2 + 1) * 2 * D + cg], mu * mu);
    atomic_add_f32(&stats[(size_t)(rep * 2 + 0) * 2 * D + D + cg], sd);
    atomic_add_f32(&stats[(size_t)(rep * 2 + 1) * 2 * D + D + cg], sd * sd);
  }
}

// ------------------------------------------------------------------------------------------
// Decoder tail, K % 1024 == 0 version (K = 2 * 1536 for every TitaNet size).  The first version (below, kept for other
// shapes) ran ONE 16-byte weight load per wave and loop trip (the trip count is a run-time value, so hipcc neither unrolls
// nor batches it): 16 outputs x 12 trips x an L2 round trip = 115 us for 0.3 GFLOP.  Here every wave owns TWO outputs at a
// time and a trip covers 1024 k: 8 independent weight loads in flight per lane (37 us); and the BatchNorm scale / shift of
// the pooled vector (16 replicated-sum loads per channel) comes from one small kernel instead of being recomputed by all
// 192 workgroups.
// ------------------------------------------------------------------------------------------
__global__ void bn_scale_shift_kernel(BnAct act, int K, float* __restrict__ scsh) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) {
    float sc, sh;
    bn_scale_shift(act, K, k, sc, sh);
    scsh[k] = sc; scsh[K + k] = sh;
  }
}
template <int NB, int ET>
__global__ __launch_bounds__(256) void tail_linear_fwd2_kernel(const float* __restrict__ pooled, const float* __restrict__ scsh, int B,
                                                               int K, int E, const float* __restrict__ W, const float* __restrict__ bias,
                                                               float* __restrict__ lin, float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* p = reinterpret_cast<float*>(smem);   // [NB][K] normalised pooled rows
  const int tid = threadIdx.x, b0 = blockIdx.x * NB, e0 = blockIdx.y * ET;
  for (int k = tid * 4; k < K; k += 1024) {
    const float4 sc = *reinterpret_cast<const float4*>(scsh + k), sh = *reinterpret_cast<const float4*>(scsh + K + k);
    float4 v[NB];
#pragma unroll
    for (int s = 0; s < NB; ++s) v[s] = (b0 + s < B) ? *reinterpret_cast<const float4*>(pooled + (size_t)(b0 + s) * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < NB; ++s) {
      const bool ok = b0 + s < B;
      *reinterpret_cast<float4*>(p + s * K + k) = ok ? make_float4(fmaf(v[s].x, sc.x, sh.x), fmaf(v[s].y, sc.y, sh.y), fmaf(v[s].z, sc.z, sh.z), fmaf(v[s].w, sc.w, sh.w))
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int e_end = min(E, e0 + ET);
  for (int e = e0 + 2 * wave; e < e_end; e += 8) {
    const bool two = e + 1 < e_end;
    const float* w0 = W + (size_t)e * K;
    const float* w1 = W + (size_t)(two ? e + 1 : e) * K;
    float s[2][NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) { s[0][q] = 0.f; s[1][q] = 0.f; }
    for (int kb = lane * 4; kb < K; kb += 1024) {
      float4 wv[2][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        wv[0][u] = *reinterpret_cast<const float4*>(w0 + kb + 256 * u);
        wv[1][u] = *reinterpret_cast<const float4*>(w1 + kb + 256 * u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const float4 pv = *reinterpret_cast<const float4*>(p + q * K + kb + 256 * u);
          s[0][q] += wv[0][u].x * pv.x + wv[0][u].y * pv.y + wv[0][u].z * pv.z + wv[0][u].w * pv.w;
          s[1][q] += wv[1][u].x * pv.x + wv[1][u].y * pv.y + wv[1][u].z * pv.z + wv[1][u].w * pv.w;
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < NB; ++q) s[h][q] = wave_sum(s[h][q]);
    if (lane == 0) {
      for (int h = 0; h < (two ? 2 : 1); ++h) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          if (b0 + q < B) {
            const float v = s[h][q] + bias[e + h];
            lin[(size_t)(b0 + q) * E + e + h] = v;
            t1 += v; t2 += v * v;
          }
        }
        if (stats) {
          const int rep = blockIdx.x % TN_NREP;
          atomic_add_f32(&stats[(size_t)(rep * 2 + 0) * E + e + h], t1);
          atomic_add_f32(&stats[(size_t)(rep * 2 + 1) * E + e + h], t2);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Decoder tail: lin = Linear(BN(pooled))  (reference src/models.py:506-511).  One workgroup per
// utterance; the K = 2D reduction is split over lanes with coalesced float4 weight reads.
// ------------------------------------------------------------------------------------------
// grid = (ceil(B/4), ceil(E/64)): 4 utterances per workgroup share every weight row read (the naive
// one-utterance-per-workgroup version re-read the whole 2.4 MB weight matrix 256 times).
__global__ __launch_bounds__(256) void tail_linear_fwd_kernel(const float* __restrict__ pooled, BnAct actP, int B, int K, int E,
                                                              const float* __restrict__ W, const float* __restrict__ bias,
                                                              float* __restrict__ lin, float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* p = reinterpret_cast<float*>(smem);   // [4][K] normalised pooled rows
  const int tid = threadIdx.x, b0 = blockIdx.x * 4, e0 = blockIdx.y * 64;
  for (int k = tid; k < K; k += 256) {
    float sc, sh;
    bn_scale_shift(actP, K, k, sc, sh);
#pragma unroll
    for (int s = 0; s < 4; ++s) p[s * K + k] = (b0 + s < B) ? pooled[(size_t)(b0 + s) * K + k] * sc + sh : 0.f;
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  for (int e = e0 + wave; e < min(E, e0 + 64); e += 4) {
    const float* w = W + (size_t)e * K;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane * 4; k < K; k += 256) {
      const float4 wv = *reinterpret_cast<const float4*>(w + k);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 pv = *reinterpret_cast<const float4*>(p + q * K + k);
        s[q] += wv.x * pv.x + wv.y * pv.y + wv.z * pv.z + wv.w * pv.w;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] = wave_sum(s[q]);
    if (lane == 0) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (b0 + q < B) {
          const float v = s[q] + bias[e];
          lin[(size_t)(b0 + q) * E + e] = v;
          t1 += v; t2 += v * v;
        }
      }
      if (stats) {
        const int rep = blockIdx.x % TN_NREP;
        atomic_add_f32(&stats[(size_t)(rep * 2 + 0) * E + e], t1);
        atomic_add_f32(&stats[(size_t)(rep * 2 + 1) * E + e], t2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Loss heads.  emb = BN(lin) (reference src/models.py:512-513); then
//   loss_type 0: inference, out = F.normalize(emb)                       (src/models.py:333)
//   loss_type 1: CELoss                                                  (src/losses.py:32-44)
//   loss_type 2: AngularMarginLoss (ArcFace/CosFace/SphereFace by m1,m2,m3) (src/losses.py:77-132)
// One workgroup per utterance.  Writes d(loss)/d(fc output) (already divided by B) for the
// backward pass, and for margin losses d(loss)/d(scale) when scale = ||x||.
// ------------------------------------------------------------------------------------------
struct HeadArgs {
  const float* lin;      // [B][E]
  BnAct actL;
  int B, E, NC;
  int loss_type;
  const float* W;        // [NC][E]
  const float* bias;     // [NC] (CE) or null
  const int64_t* targets;
  float scale; int has_scale; float m1, m2, m3, eps;
  float* emb;            // [B][E] pre-normalisation embeddings
  float* emb_norm;       // [B][E] normalised embeddings (workspace copy, used by backward)
  float* emb_user;       // [B][E] caller's output buffer or null
  int64_t* preds;        // [B]
  float* loss;           // scalar, pre-zeroed; accumulates mean
  float* dlogits;        // [B][NC]
  float* dscale;         // [B]
  float* logits;         // [B][NC] or null (CE logits / clamped cosines, for tests)
};

__global__ __launch_bounds__(256) void row_normalize_kernel(float* W, int R, int C) {
  // fc.weight.data = F.normalize(fc.weight.data, p=2, dim=1)  (reference src/losses.py:86) — in place
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = W[(size_t)r * C + c]; s += v * v; }
  s = wave_sum(s);
  const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
  for (int c = lane; c < C; c += 64) W[(size_t)r * C + c] *= inv;
}

__global__ __launch_bounds__(256) void head_fwd_kernel(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* x = reinterpret_cast<float*>(smem);   // [E] embedding (CE) or normalised embedding (margin)
  float* lg = x + a.E;                           // [NC]
  __shared__ float red[4];
  __shared__ float redv[4];
  __shared__ int redi[4];
  const int tid = threadIdx.x, b = blockIdx.x;
  float ss = 0.f;
  for (int e = tid; e < a.E; e += 256) {
    float sc, sh;
    bn_scale_shift(a.actL, a.E, e, sc, sh);
    const float v = a.lin[(size_t)b * a.E + e] * sc + sh;
    x[e] = v;
    a.emb[(size_t)b * a.E + e] = v;
    ss += v * v;
  }
  ss = block_sum_256(ss, red);
  const float norm = sqrtf(ss);
  if (a.loss_type == 2) {
    // normalized_inputs = inputs / ||inputs||  (no epsilon: src/losses.py:89-92)
    const float inv = 1.f / norm;
    for (int e = tid; e < a.E; e += 256) {
      const float v = x[e] * inv;
      x[e] = v;
      a.emb_norm[(size_t)b * a.E + e] = v;
      if (a.emb_user) a.emb_user[(size_t)b * a.E + e] = v;
    }
  } else {
    const float inv = 1.f / fmaxf(norm, 1e-12f);
    for (int e = tid; e < a.E; e += 256) {
      a.emb_norm[(size_t)b * a.E + e] = x[e] * inv;
      if (a.emb_user) a.emb_user[(size_t)b * a.E + e] = x[e] * inv;
    }
  }
  if (a.loss_type == 0) return;
  __syncthreads();
  int y = (int)a.targets[b];
  if (y < 0 || y >= a.NC) {
    // F.cross_entropy raises on an out-of-range target; a kernel cannot: poison the loss (the trainers' finite-loss guard,
    // reference src/learn.py:110-112, trips) and keep every access in range
    if (tid == 0) atomic_add_f32(a.loss, NAN);
    y = min(max(y, 0), a.NC - 1);
  }
  // logits / cosines
  float vmax = -INFINITY; int imax = 0x7fffffff;
  for (int c = tid; c < a.NC; c += 256) {
    const float* w = a.W + (size_t)c * a.E;
    float s = 0.f;
    if ((a.E & 3) == 0) {
      // 16-byte loads of the thread's own weight row (a quarter of the memory instructions of the scalar walk; the additions
      // keep their order e = 0, 1, 2, ..: bit-identical logits)
      for (int e = 0; e < a.E; e += 4) {
        const float4 wv = *reinterpret_cast<const float4*>(w + e), xv = *reinterpret_cast<const float4*>(x + e);
        s = fmaf(wv.x, xv.x, s); s = fmaf(wv.y, xv.y, s); s = fmaf(wv.z, xv.z, s); s = fmaf(wv.w, xv.w, s);
      }
    } else {
      for (int e = 0; e < a.E; ++e) s = fmaf(w[e], x[e], s);
    }
    if (a.loss_type == 1) s += a.bias[c];
    else s = fminf(fmaxf(s, -1.f), 1.f);
    lg[c] = s;
    if (a.logits) a.logits[(size_t)b * a.NC + c] = s;
    if (s > vmax) { vmax = s; imax = c; }
  }
  // block argmax (first index wins ties) and max
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(vmax, o, 64);
    const int oi = __shfl_xor(imax, o, 64);
    if (ov > vmax || (ov == vmax && oi < imax)) { vmax = ov; imax = oi; }
  }
  if ((tid & 63) == 0) { redv[tid >> 6] = vmax; redi[tid >> 6] = imax; }
  __syncthreads();
  vmax = redv[0]; imax = redi[0];
  for (int w = 1; w < 4; ++w)
    if (redv[w] > vmax || (redv[w] == vmax && redi[w] < imax)) { vmax = redv[w]; imax = redi[w]; }
  if (tid == 0) a.preds[b] = imax;
  const float invB = 1.f / (float)a.B;
  if (a.loss_type == 1) {
    float se = 0.f;
    for (int c = tid; c < a.NC; c += 256) se += __expf(lg[c] - vmax);
    se = block_sum_256(se, red);
    const float lse = vmax + logf(se);
    for (int c = tid; c < a.NC; c += 256) {
      const float p = __expf(lg[c] - lse);
      a.dlogits[(size_t)b * a.NC + c] = (p - (c == y ? 1.f : 0.f)) * invB;
    }
    if (tid == 0) atomic_add_f32(a.loss, (lse - lg[y]) * invB);
  } else {
    const float s = a.has_scale ? a.scale : norm;
    const float cy = lg[y];
    const float theta = acosf(cy);
    const float ang = a.m1 * theta + a.m2;
    const float num = s * (cosf(ang) - a.m3);
    float others = 0.f;
    for (int c = tid; c < a.NC; c += 256)
      if (c != y) others += expf(s * lg[c]);     // unstabilised exp, as the reference (src/losses.py:127)
    others = block_sum_256(others, red);
    const float den = expf(num) + others;
    const float dene = den + a.eps;
    const float dnum = -1.f + expf(num) / dene;
    float dsc = dnum * (cosf(ang) - a.m3);        // d loss / d scale (used only when scale = ||x||)
    float dsacc = 0.f;
    for (int c = tid; c < a.NC; c += 256) {
      float d;
      if (c == y) {
        const float sq = sqrtf(fmaxf(1.f - cy * cy, 0.f));
        d = dnum * s * a.m1 * sinf(ang) / sq;
      } else {
        const float p = expf(s * lg[c]) / dene;
        d = p * s;
        dsacc += p * lg[c];
      }
      a.dlogits[(size_t)b * a.NC + c] = d * invB;
    }
    dsacc = block_sum_256(dsacc, red);
    if (tid == 0) {
      a.dscale[b] = a.has_scale ? 0.f : (dsc + dsacc) * invB;
      atomic_add_f32(a.loss, -(num - logf(dene)) * invB);
    }
  }
}

// ------------------------------------------------------------------------------------------
// BatchNorm running statistics (train mode): running = (1-m) running + m batch, unbiased var
// (nn.BatchNorm1d; reference src/modules.py:128).  One launch for every BN layer of the model.
// ------------------------------------------------------------------------------------------
struct BnUpdateDesc {
  const float* stats;   // [TN_NREP][2][C]
  float* rmean;
  float* rvar;
  int C;
  int n;                // rows reduced
};

// eval mode: sums that make the batch-statistics formula reproduce the running statistics
__global__ void bn_eval_prepare_kernel(const BnUpdateDesc* descs, float* const* stats_out) {
  const BnUpdateDesc d = descs[blockIdx.y];
  float* st = stats_out[blockIdx.y];
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < d.C; c += gridDim.x * blockDim.x) {
    const float m = d.rmean[c], v = d.rvar[c], n = (float)d.n;
    st[c] = m * n;                    // replica 0 sum          (other replicas stay zero)
    st[d.C + c] = (v + m * m) * n;    // replica 0 sum of squares
  }
}

// n_rows_plan / n_rows_valid: layers reduced over all B*T rows of the plan use the valid-row count of a masked batch
__global__ void bn_running_update_kernel(const BnUpdateDesc* descs, float momentum, int64_t* nbt, int n_layers, int n_rows_plan,
                                         int n_rows_valid) {
  BnUpdateDesc d = descs[blockIdx.y];
  if (d.n == n_rows_plan) d.n = n_rows_valid;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < d.C; c += gridDim.x * blockDim.x) {
    float s = 0.f, q = 0.f;
    for (int r = 0; r < TN_NREP; ++r) { s += d.stats[(r * 2 + 0) * d.C + c]; q += d.stats[(r * 2 + 1) * d.C + c]; }
    const float inv = 1.f / (float)d.n;
    const float mean = s * inv;
    const float var = fmaxf(q * inv - mean * mean, 0.f);
    const float unb = var * ((float)d.n / (float)max(d.n - 1, 1));
    d.rmean[c] = (1.f - momentum) * d.rmean[c] + momentum * mean;
    d.rvar[c] = (1.f - momentum) * d.rvar[c] + momentum * unb;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) nbt[blockIdx.y] += 1;
}


// ------------------------------------------------------------------------------------------
// Decoder(simple_pool=True), reference src/models.py:497-502: AdaptiveAvgPool1d(1) over time of x = act(E)
// (the Linear(D, 2D) that follows is a plain f32 GEMM over B rows).  Lane-local over time like the attentive pool.
// ------------------------------------------------------------------------------------------
template <typename AT>
__global__ __launch_bounds__(256) void mean_pool_fwd_kernel(const AT* __restrict__ E, BnAct actE, int T, int D,
                                                            float* __restrict__ mu) {
  constexpr int CVB = 64, TG = 4;
  __shared__ float red[TG][CVB * 8];
  __shared__ float scs[CVB * 8], shs[CVB * 8];
  const int tid = threadIdx.x, b = blockIdx.x, cbase = blockIdx.y * CVB * 8;
  for (int c = tid; c < CVB * 8; c += 256) {
    float sc = 1.f, sh = 0.f;
    if (cbase + c < D) bn_scale_shift(actE, D, cbase + c, sc, sh);
    scs[c] = sc; shs[c] = sh;
  }
  __syncthreads();
  const int vc = tid % CVB, tg = tid / CVB, c0 = cbase + vc * 8;
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = 0.f;
  if (c0 < D) {
    for (int t = tg; t < T; t += TG) {
      const uint32_t row = (uint32_t)b * T + t;
      float x[8];
      load8(E + (size_t)row * D + c0, x);
      act8(x, scs + vc * 8, shs + vc * 8, actE, row, D, c0);
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += x[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[tg][vc * 8 + i] = s[i];
  __syncthreads();
  for (int c = tid; c < CVB * 8; c += 256) {
    if (cbase + c < D) mu[(size_t)b * D + cbase + c] = (red[0][c] + red[1][c] + red[2][c] + red[3][c]) / (float)T;
  }
}
