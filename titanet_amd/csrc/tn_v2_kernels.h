// titanet_amd — "v2" kernels: the three heavy kernels of a mega-block sub-block, specialised for the
// headline shape (hidden = 256 channels, bf16 activations; TitaNet-S, BASELINE.json configs[1]).
//
// What the generic v1 kernels lose on this HBM-bound shape is latency, not bandwidth: they run
// produce -> barrier -> MFMA -> barrier once per K chunk at 1-2 workgroups per CU.  Here instead:
//   * a row tile of the rows x 256 activation matrix is ONE contiguous 32 KB block of HBM (64 rows x
//     512 B): each thread prefetches the NEXT tile into registers (4 x 16 B per input stream) while
//     the current tile is transformed / multiplied / stored (global -> reg -> LDS double buffering:
//     the compiler counts vmcnt for us, loads stay in flight across the barriers);
//   * workgroups are persistent (grid = resident workgroups), the 256 x 256 weight matrix lives in
//     REGISTERS as MFMA B fragments for the whole kernel (each of the 8 waves owns 32 output columns
//     x all 256 K = 16 fragments = 64 VGPRs), so LDS only holds activations and the K loop runs
//     without any weight staging or barrier;
//   * per-channel reductions (BatchNorm sums, depthwise weight gradients) accumulate in registers
//     across all tiles of the workgroup and hit global atomics once per workgroup.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include "tn_common.h"
#include "tn_gemm.h"

#define V2_C 256            // channels (hidden width)
#define V2_R 64             // raw rows per tile
#define V2_AP (V2_C + 8)    // padded pitch of MFMA operand tiles (conflict-free ds_read_b128)
#define V2_NT 512
#define WG2_RK 32
#define WG2_PITCH 288      // 576-byte rows: the 4 rows of a transpose read land on distinct banks
#ifndef TN_NT_WGRAD_OPERANDS
// Infinity-Cache management by store policy (round 6).  The 256 MB memory-side cache holds about six hidden-width tensors of the
// benched batch, and the kernels of a mega block mostly read what the previous launch wrote.  Tensors that are written now and
// read only much later are stored NON-TEMPORAL so that they do not push the next kernels' operands out:
//   bit 0 (1)  the kept depthwise outputs Q of sub_fwd_v5 (read by the weight-gradient launch at the end of backward)
//   bit 1 (2)  the stored dS of dgrad_dw_v6 (same reader)
//   bit 3 (8)  the skip conv's output S of sub_fwd_v4 (read four launches and ~240 MB of writes later by the combine)
//   bit 4 (16) the stored dS of the skip conv's layer in dgrad_v2
//   bit 5 (32) the skip data gradient dgrad_v2 writes (added three launches and ~400 MB later by the first sub-block's pass)
//   bit 2 (4)  [off] non-temporal LOADS of Y in dgrad_dw_v6 (its last use): measured +0.03 ms
// Same-box round-robin of the step (profiles/r06_ab_nt_stores.txt): 8.344 ms with 0, 8.180 with 11, 8.142 with 27; 8.401 / 8.187 /
// 8.143 with 0 / 27 / 59 (-3.1 %).
#define TN_NT_WGRAD_OPERANDS 59
#endif
#ifndef V5_MD
#define V5_MD 0      // tuning only: fragment reads in flight of a hand-scheduled MFMA phase in sub_fwd_v5's consumers (0: hipcc's schedule)
#endif
#ifndef V2_DBG_SKIP
#define V2_DBG_SKIP 0   // tuning only: 1 skip MFMA, 2 skip stencil, 4 skip global stores, 8 skip act
#endif

// 16-byte vector <-> 8 floats with a uint4 register image (prefetch buffers)
__device__ __forceinline__ void unpack8(const uint4& a, float v[8]) {
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
  v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}

// act with per-thread register scale/shift (the thread's 8 channels never change)
__device__ __forceinline__ void act8_reg(float v[8], const float sc[8], const float sh[8], const BnAct& a, uint32_t row,
                                         int c0) {
  if (a.mode != 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = v[i] * sc[i] + sh[i];
  }
  if (a.relu) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
  }
  if (a.drop_thr) tn_drop8(v, (row * (uint32_t)V2_C + (uint32_t)c0) >> 3, tn_act_key(a), a.drop_thr);
}

// ------------------------------------------------------------------------------------------
// forward sub-block (reference src/modules.py:65-78 + BN statistics of :128):
//   Y = W_pw * dwconv_KD(act(X)) + b   (DW = true)      or      Y = W * act(X) + b   (DW = false)
// ------------------------------------------------------------------------------------------
struct SubFwdV2Args {
  const bf16_t* X;      // [M][256] raw input
  BnAct act;
  const float* wdw;     // [256][KD]
  const float* bdw;     // [256]
  const bf16_t* W;      // [256][256] bf16, row = output channel, K contiguous
  const float* bias;    // [256]
  bf16_t* Y;            // [M][256]
  float* stats;         // [TN_NREP][2][256] or null
  int M, T, ntiles;
  const uint4* Wswz;    // optional: W in MFMA-fragment order (swizzle256_kernel): the resident weight fragments then load with
                        // fully coalesced 1 KB reads instead of 32 rows x 32 B per instruction
  bf16_t* Q;            // [M][256] or null: the depthwise output (the pointwise GEMM's operand) is ALSO stored, so that the
                        // batched weight-gradient launch reads it instead of recomputing activation + stencil (sub_fwd_v5 only)
};

// activation with compile-time flags (FL bits: 1 = BatchNorm, 2 = ReLU, 4 = dropout): no uniform branches in the row loops
template <int FL>
__device__ __forceinline__ void act8_t(float v[8], const float sc[8], const float sh[8], uint32_t key, uint32_t thr, uint32_t row, int c0) {
  if (FL & 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], sc[i], sh[i]);
  }
  if (FL & 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
  }
  if (FL & 4) tn_drop8(v, (row * (uint32_t)V2_C + (uint32_t)c0) >> 3, key, thr);
}

// sub_fwd_v4: persistent MFMA kernel (W resident in registers) with the activation flags as template parameters, workgroup-uniform fast paths for
// interior tiles (no row-range tests) and single-utterance tiles (no tap-boundary tests: 79 % of the tiles at
// T = 300), and accumulators that start from the bias.
// MK (variable-length batch, a.act.rm.len): padding rows read as zeros.  Only for the activation-on-load variants of the skip
// conv (block 0 reads the raw prolog output); stored block outputs already hold zeros there.
template <int KD, bool DW, int FL, bool MK = false>
__global__ __launch_bounds__(V2_NT, 2) void sub_fwd_v4_kernel(SubFwdV2Args a) {
  static_assert(!(MK && DW), "the masked variant exists for the skip conv only");
  constexpr int PADR = DW ? (KD - 1) / 2 : 0;
  constexpr int OUTR = DW ? V2_R - (KD - 1) : V2_R;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Cs = reinterpret_cast<bf16_t*>(smem);     // [64][264] output staging tile; its first 32 KB double as
  bf16_t* Xa = Cs;                                  // [64][256] the activated input rows (dead once the stencil ran)
  bf16_t* As = Cs + V2_R * V2_AP;                   // [64][264] MFMA B operand (depthwise output)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vc = tid & 31, rq = tid >> 5;          // this thread's channel vector and row phase (rows rq + 16 q)
  const int c0 = vc * 8;
  const uint32_t dkey = tn_act_key(a.act), dthr = a.act.drop_thr;

  // ---- per-thread constants: BN scale/shift and depthwise taps of the thread's 8 channels.
  // Computed cooperatively (one channel per thread, 18 loads) and redistributed through LDS: having
  // every thread reduce the 8 statistic replicas of its own 8 channels costs 144 gather loads per
  // thread and dominated the kernel.
  float sc[8], sh[8];
  float wd[DW ? KD : 1][8], bd[8];
  {
    float* tmp = reinterpret_cast<float*>(As);      // scratch before the first tile: [2 + KD + 1][256]
    if (tid < V2_C) {
      float s = 1.f, h = 0.f;
      if (FL & 1) bn_scale_shift(a.act, V2_C, tid, s, h);
      tmp[tid] = s;
      tmp[V2_C + tid] = h;
      if (DW) {
        tmp[2 * V2_C + tid] = a.bdw[tid];
#pragma unroll
        for (int k = 0; k < KD; ++k) tmp[(3 + k) * V2_C + tid] = a.wdw[(size_t)tid * KD + k];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sc[i] = tmp[c0 + i];
      sh[i] = tmp[V2_C + c0 + i];
      if (DW) {
        bd[i] = tmp[2 * V2_C + c0 + i];
#pragma unroll
        for (int k = 0; k < KD; ++k) wd[k][i] = tmp[(3 + k) * V2_C + c0 + i];
      }
    }
    __syncthreads();
  }
  // ---- this wave's 32 output channels of W as MFMA A fragments, resident for the whole kernel.  The
  // activation rows are the B operand, so a lane ends up with 4 CONSECUTIVE channels of one row per
  // accumulator quad: the output tile is staged with 8-byte LDS writes (2-byte writes cost 4x more).
  const int half = lane >> 5;
  bf16x8_t wf[16];
  {
    const int co = wave * 32 + (lane & 31);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      wf[ks] = a.Wswz ? __builtin_bit_cast(bf16x8_t, a.Wswz[((size_t)wave * 16 + ks) * 64 + lane])
                      : *reinterpret_cast<const bf16x8_t*>(a.W + (size_t)co * V2_C + ks * 16 + half * 8);
  }
  float biasr[16];   // bias of this lane's 16 output channels: 32*wave + 8g + 4*half + j
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) biasr[4 * g + j] = a.bias[wave * 32 + 8 * g + 4 * half + j];
  float st_s[8], st_q[8];   // BN statistics of this thread's 8 channels (from the staged bf16 tile)
#pragma unroll
  for (int i = 0; i < 8; ++i) { st_s[i] = 0.f; st_q[i] = 0.f; }

  // unconditional buffer loads / stores, as in sub_fwd_v5 (rows outside the tensor: zeros / dropped)
  typedef __attribute__((ext_vector_type(4))) unsigned int v4_u32x4_t;
  const int tbytes = (int)((size_t)a.M * V2_C * sizeof(bf16_t));
  const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.X), 0, tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdY = __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, tbytes, 0x00020000);
  uint4 pf[4];
  auto prefetch = [&](int tile) {
    const int raw0 = tile * OUTR - PADR;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int gr = raw0 + rq + 16 * q;
      pf[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srdX, (gr * V2_C + c0) * (int)sizeof(bf16_t), 0, 0));
    }
  };
  int tile = blockIdx.x;
  if (tile < a.ntiles) prefetch(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    const int out0 = tile * OUTR, raw0 = out0 - PADR;
    __syncthreads();   // (1) previous tile's output staging (aliases Xa) has been stored
    // ---- registers -> LDS with the activation applied once per element.  Interior tiles (all 64 staged rows
    // exist: everything but the first / last tile) skip the per-row range tests.
    const bool interior = raw0 >= 0 && raw0 + V2_R <= a.M;                  // workgroup-uniform
    const bool one_utt = interior && (raw0 % a.T) + V2_R <= a.T;           // ... and inside ONE utterance
    if (MK) {
      const TileMask tm = tn_tile_mask(a.act.rm.len, a.T, a.M, raw0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = rq + 16 * q, gr = raw0 + r;
        float v[8];
        unpack8(pf[q], v);
        if (gr >= 0 && gr < a.M && tn_tile_valid(tm, gr)) act8_t<FL>(v, sc, sh, dkey, dthr, (uint32_t)gr, c0);
        else {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = 0.f;
        }
        store8(As + r * V2_AP + c0, v);
      }
    } else if (interior) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = rq + 16 * q;
        float v[8];
        unpack8(pf[q], v);
        act8_t<FL>(v, sc, sh, dkey, dthr, (uint32_t)(raw0 + r), c0);
        if (DW) store8(Xa + r * V2_C + c0, v);
        else store8(As + r * V2_AP + c0, v);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = rq + 16 * q, gr = raw0 + r;
        float v[8];
        unpack8(pf[q], v);
        if (gr >= 0 && gr < a.M) act8_t<FL>(v, sc, sh, dkey, dthr, (uint32_t)gr, c0);
        if (DW) store8(Xa + r * V2_C + c0, v);
        else store8(As + r * V2_AP + c0, v);
      }
    }
    prefetch(tile + gridDim.x);   // next tile's loads fly during the rest (past the last tile: rows beyond M, zeros)
    __syncthreads();   // (2)
    if (DW) {
      // ---- depthwise stencil over time -> MFMA operand tile.  Each thread owns 4 CONSECUTIVE output rows
      // (a sliding window over KD + 3 activated rows: every LDS row is read and unpacked once, not KD times).
      {
        const int o0 = rq * 4;
        float win[KD + 3][8];
#pragma unroll
        for (int j = 0; j < KD + 3; ++j) {
          if (o0 + j < V2_R) load8(Xa + (o0 + j) * V2_C + c0, win[j]);
          else {
#pragma unroll
            for (int i = 0; i < 8; ++i) win[j][i] = 0.f;
          }
        }
        if (one_utt) {
          // every tap of every output row of this tile is inside the utterance: no boundary tests
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(wd[0][i], win[q][i], bd[i]);
#pragma unroll
            for (int k = 1; k < KD; ++k)
#pragma unroll
              for (int i = 0; i < 8; ++i) acc[i] = fmaf(wd[k][i], win[q + k][i], acc[i]);
            store8(As + (o0 + q) * V2_AP + c0, acc);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int o = o0 + q;
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = bd[i];
            if (o < OUTR) {
              const int t = (out0 + o) % a.T;
#pragma unroll
              for (int k = 0; k < KD; ++k) {
                const int tt = t + k - PADR;
                if (tt >= 0 && tt < a.T) {
#pragma unroll
                  for (int i = 0; i < 8; ++i) acc[i] = fmaf(wd[k][i], win[q + k][i], acc[i]);
                }
              }
            }
            store8(As + o * V2_AP + c0, acc);
          }
        }
      }
      __syncthreads();   // (3)
    }
    // ---- pointwise GEMM: [64 x 256] x W^T, weights from registers
    f32x16_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = biasr[r]; acc1[r] = biasr[r]; }   // accumulate on top of the bias
    const bf16_t* brow = As + (lane & 31) * V2_AP + half * 8;
#pragma unroll
    for (int ks = 0; ks < ((0) ? 1 : 16); ++ks) {
      const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(brow + ks * 16);
      const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(brow + 32 * V2_AP + ks * 16);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b1, acc1, 0, 0, 0);
    }
    // ---- epilogue: lane = row (lane&31 [+32]), regs 4g..4g+3 = channels 32*wave + 8g + 4*half + 0..3
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = wave * 32 + 8 * g + 4 * half;
      uint2 w0, w1;
      w0.x = f2bf_pk(acc0[4 * g], acc0[4 * g + 1]);
      w0.y = f2bf_pk(acc0[4 * g + 2], acc0[4 * g + 3]);
      w1.x = f2bf_pk(acc1[4 * g], acc1[4 * g + 1]);
      w1.y = f2bf_pk(acc1[4 * g + 2], acc1[4 * g + 3]);
      *reinterpret_cast<uint2*>(Cs + (lane & 31) * V2_AP + co) = w0;
      *reinterpret_cast<uint2*>(Cs + (32 + (lane & 31)) * V2_AP + co) = w1;
    }
    __syncthreads();   // (4)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = rq + 16 * q, gr = out0 + o;
      const bool keep = o < OUTR && gr < a.M;
      const uint4 raw = *reinterpret_cast<const uint4*>(Cs + (keep ? o : 0) * V2_AP + c0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4_u32x4_t, raw), srdY, keep ? (gr * V2_C + c0) * (int)sizeof(bf16_t) : 0x7ffffff0, 0, (!DW && (TN_NT_WGRAD_OPERANDS & 8)) ? 2 : 0);
      if (keep) {
        float y[8];
        unpack8(raw, y);
#pragma unroll
        for (int i = 0; i < 8; ++i) { st_s[i] += y[i]; st_q[i] = fmaf(y[i], y[i], st_q[i]); }
      }
    }
  }
  if (a.stats) {
    // the 16 row phases hold the same channels: plain LDS stores + a short sum, then coalesced replicated atomics
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);      // [16][2][256]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      red[(rq * 2 + 0) * V2_C + c0 + i] = st_s[i];
      red[(rq * 2 + 1) * V2_C + c0 + i] = st_q[i];
    }
    __syncthreads();
    const int rep = blockIdx.x % TN_NREP;
    {
      const int which = tid >> 8, c = tid & 255;
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) v += red[(r * 2 + which) * V2_C + c];
      atomic_add_f32(&a.stats[(size_t)(rep * 2 + which) * V2_C + c], v);
    }
  }
}


template <int KD, bool DW, int FL>
inline int launch_sub_fwd_v4_t(SubFwdV2Args a, int grid, size_t smem, hipStream_t st) {
  auto kern = sub_fwd_v4_kernel<KD, DW, FL>;
  if constexpr (!DW && FL != 0) {
    if (a.act.rm.len) kern = sub_fwd_v4_kernel<KD, DW, FL, true>;
  } else if (a.act.rm.len && (DW || a.act.mode != 0)) return -1000;
  TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(V2_NT), smem, st, a);
  return (int)hipGetLastError();
}
template <int KD, bool DW>
inline int launch_sub_fwd_v4(SubFwdV2Args a, int resident_wgs, hipStream_t st) {
  if ((long)a.M * V2_C * 2 >= (1L << 31)) return -1000;      // 32-bit buffer offsets: the generic kernel takes such a batch
  constexpr int OUTR = DW ? V2_R - (KD - 1) : V2_R;
  a.ntiles = (a.M + OUTR - 1) / OUTR;
  const int grid = a.ntiles < resident_wgs ? a.ntiles : resident_wgs;
  const size_t smem = (size_t)(2 * V2_R * V2_AP) * sizeof(bf16_t);
  const int fl = (a.act.mode != 0 ? 1 : 0) | (a.act.relu ? 2 : 0) | (a.act.drop_thr ? 4 : 0);
  switch (fl) {
    case 0: return launch_sub_fwd_v4_t<KD, DW, 0>(a, grid, smem, st);   // block input (stored activated)
    case 3: return launch_sub_fwd_v4_t<KD, DW, 3>(a, grid, smem, st);   // BN + ReLU (eval, p = 0, prolog output)
    case 7: return launch_sub_fwd_v4_t<KD, DW, 7>(a, grid, smem, st);   // BN + ReLU + dropout
    default: return -1000;      // no specialisation for this activation (the caller runs the generic GEMM)
  }
}

// ------------------------------------------------------------------------------------------
// combine_fwd_v2: OUT = dropout(relu(BN(S) + g * act3(Y3))) for hidden = 256, bf16 (reference src/models.py:467-472).
// One workgroup per (utterance, row part); a thread owns a FIXED 8-channel vector, so the two BatchNorm
// scale/shift pairs and the SE gate live in registers (the generic kernel re-read them from LDS per element and
// spent ~60 instructions per vector on integer div/mod of a flat index); rows are 4-way unrolled so 8 loads fly.
// FL3 = activation flags of Y3 (1 BN, 2 ReLU, 4 dropout); DROP = dropout on the block output.
// ------------------------------------------------------------------------------------------
struct CombineFwdV2Args {
  const bf16_t* S; BnAct actS;
  const bf16_t* Y3; BnAct act3;
  const float* gate;     // [B][256]
  bf16_t* OUT;
  int T, parts;
  const int* len;        // valid frames per utterance or null: padding rows are written as zeros
  uint32_t drop_thr, drop_key;
  float inv_keep;
  const uint32_t* key_add;   // see BnAct::key_add
};
template <int FL3, bool DROP>
__global__ __launch_bounds__(256) void combine_fwd_v2_kernel(CombineFwdV2Args a) {
  __shared__ float cst[4 * V2_C];
  const int tid = threadIdx.x, vc = tid & 31, rq = tid >> 5, c0 = vc * 8;
  const int b = blockIdx.x / a.parts, part = blockIdx.x % a.parts;
  {
    float s, h;
    bn_scale_shift(a.actS, V2_C, tid, s, h);
    cst[tid] = s; cst[V2_C + tid] = h;
    s = 1.f; h = 0.f;
    if (FL3 & 1) bn_scale_shift(a.act3, V2_C, tid, s, h);
    cst[2 * V2_C + tid] = s; cst[3 * V2_C + tid] = h;
  }
  float g[8];
  {
    const float4 g0 = *reinterpret_cast<const float4*>(a.gate + (size_t)b * V2_C + c0);
    const float4 g1 = *reinterpret_cast<const float4*>(a.gate + (size_t)b * V2_C + c0 + 4);
    g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
  }
  __syncthreads();
  float scS[8], shS[8], sc3[8], sh3[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    scS[i] = cst[c0 + i]; shS[i] = cst[V2_C + c0 + i]; sc3[i] = cst[2 * V2_C + c0 + i]; sh3[i] = cst[3 * V2_C + c0 + i];
    if (DROP) { scS[i] *= a.inv_keep; shS[i] *= a.inv_keep; g[i] *= a.inv_keep; }   // relu(k x) = k relu(x), k > 0
  }
  const int per = (a.T + a.parts - 1) / a.parts;
  const int t0 = part * per, t_end = min(a.T, t0 + per);
  const int L = a.len ? a.len[b] : a.T;
  const int t1 = min(L, t_end);
  {
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int t = max(t0, L) + rq; t < t_end; t += 8) *reinterpret_cast<uint4*>(a.OUT + ((size_t)b * a.T + t) * V2_C + c0) = z;
  }
  const uint32_t dkey3 = tn_act_key(a.act3), dthr3 = a.act3.drop_thr;
  const uint32_t okey = a.key_add ? a.drop_key + *a.key_add : a.drop_key;
  constexpr int U = 4;
  for (int tb = t0 + rq; tb < t1; tb += 8 * U) {
    uint4 rs[U], ry[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = tb + 8 * u;
      if (t < t1) {
        const size_t o = ((size_t)b * a.T + t) * V2_C + c0;
        rs[u] = *reinterpret_cast<const uint4*>(a.S + o);
        ry[u] = *reinterpret_cast<const uint4*>(a.Y3 + o);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = tb + 8 * u;
      if (t < t1) {
        const uint32_t row = (uint32_t)b * a.T + t;
        float s[8], y[8], o[8];
        unpack8(rs[u], s);
        unpack8(ry[u], y);
        act8_t<FL3>(y, sc3, sh3, dkey3, dthr3, row, c0);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = fmaxf(fmaf(s[q], scS[q], fmaf(g[q], y[q], shS[q])), 0.f);
        if (DROP) tn_drop8(o, (row * (uint32_t)V2_C + (uint32_t)c0) >> 3, okey, a.drop_thr);
        store8(a.OUT + (size_t)row * V2_C + c0, o);
      }
    }
  }
}
inline int launch_combine_fwd_v2(const CombineFwdV2Args& a, int B, hipStream_t st) {
  const int fl3 = (a.act3.mode != 0 ? 1 : 0) | (a.act3.relu ? 2 : 0) | (a.act3.drop_thr ? 4 : 0);
  const dim3 grid(B * a.parts), blk(256);
  if (fl3 == 7 && a.drop_thr) hipLaunchKernelGGL((combine_fwd_v2_kernel<7, true>), grid, blk, 0, st, a);
  else if (fl3 == 3 && !a.drop_thr) hipLaunchKernelGGL((combine_fwd_v2_kernel<3, false>), grid, blk, 0, st, a);
  else return -1000;      // caller falls back to the generic kernel
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// se_squeeze_v2: m = mean_T act3(Y3); h = relu(W1 m); g = sigmoid(W2 h) for hidden = 256, bf16, Hr = 16
// (reference src/modules.py:173-189).  One workgroup (512 threads = 32 channel vectors x 16 row phases) per
// utterance; constants in registers, 4 rows in flight per thread, and the two tiny weight matrices are fetched
// BEFORE the streaming loop so that the serial mat-vec tail does not wait on HBM.
// ------------------------------------------------------------------------------------------
struct SeSqueezeV2Args {
  const bf16_t* Y; BnAct act;
  const float* W1;     // [16][256]
  const float* W2;     // [256][16]
  float* m_out; float* h_out; float* g_out;
  int T;
  const int* len;      // valid frames per utterance or null: the mean runs over them
};
template <int FL>
__global__ __launch_bounds__(512) void se_squeeze_v2_kernel(SeSqueezeV2Args a) {
  constexpr int HR = 16;
  __shared__ float cst[2 * V2_C];
  __shared__ float part[16][V2_C];
  __shared__ float mean[V2_C];
  __shared__ float hbuf[HR];
  const int tid = threadIdx.x, vc = tid & 31, tg = tid >> 5, c0 = vc * 8, b = blockIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // weights of the tail, in flight during the streaming loop
  float w1a[4], w1b[4];      // W1 rows wave and wave + 8, columns lane + 64 k
#pragma unroll
  for (int k = 0; k < 4; ++k) { w1a[k] = a.W1[(size_t)wave * V2_C + lane + 64 * k]; w1b[k] = a.W1[(size_t)(wave + 8) * V2_C + lane + 64 * k]; }
  float4 w2[4];              // W2 row tid (tid < 256)
  if (tid < V2_C) {
#pragma unroll
    for (int k = 0; k < 4; ++k) w2[k] = *reinterpret_cast<const float4*>(a.W2 + (size_t)tid * HR + 4 * k);
  }
  if (tid < V2_C) {
    float s = 1.f, h = 0.f;
    if (FL & 1) bn_scale_shift(a.act, V2_C, tid, s, h);
    cst[tid] = s; cst[V2_C + tid] = h;
  }
  __syncthreads();
  float sc[8], sh[8], acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { sc[i] = cst[c0 + i]; sh[i] = cst[V2_C + c0 + i]; acc[i] = 0.f; }
  const uint32_t dkey = tn_act_key(a.act), dthr = a.act.drop_thr;
  constexpr int U = 4;
  const int L = a.len ? a.len[b] : a.T;
  for (int tb = tg; tb < L; tb += 16 * U) {
    uint4 ry[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = tb + 16 * u;
      if (t < L) ry[u] = *reinterpret_cast<const uint4*>(a.Y + ((size_t)b * a.T + t) * V2_C + c0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = tb + 16 * u;
      if (t < L) {
        float v[8];
        unpack8(ry[u], v);
        act8_t<FL>(v, sc, sh, dkey, dthr, (uint32_t)b * a.T + t, c0);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += v[i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) part[tg][c0 + i] = acc[i];
  __syncthreads();
  if (tid < V2_C) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += part[k][tid];
    s *= 1.f / (float)max(L, 1);
    mean[tid] = s;
    a.m_out[(size_t)b * V2_C + tid] = s;
  }
  __syncthreads();
  {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s0 = fmaf(w1a[k], mean[lane + 64 * k], s0); s1 = fmaf(w1b[k], mean[lane + 64 * k], s1); }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (lane == 0) {
      s0 = fmaxf(s0, 0.f); s1 = fmaxf(s1, 0.f);
      hbuf[wave] = s0; hbuf[wave + 8] = s1;
      a.h_out[(size_t)b * HR + wave] = s0;
      a.h_out[(size_t)b * HR + wave + 8] = s1;
    }
  }
  __syncthreads();
  if (tid < V2_C) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s = fmaf(w2[k].x, hbuf[4 * k], s); s = fmaf(w2[k].y, hbuf[4 * k + 1], s);
      s = fmaf(w2[k].z, hbuf[4 * k + 2], s); s = fmaf(w2[k].w, hbuf[4 * k + 3], s);
    }
    a.g_out[(size_t)b * V2_C + tid] = 1.f / (1.f + __expf(-s));
  }
}
inline int launch_se_squeeze_v2(const SeSqueezeV2Args& a, int B, hipStream_t st) {
  const int fl = (a.act.mode != 0 ? 1 : 0) | (a.act.relu ? 2 : 0) | (a.act.drop_thr ? 4 : 0);
  if (fl == 7) hipLaunchKernelGGL((se_squeeze_v2_kernel<7>), dim3(B), dim3(512), 0, st, a);
  else if (fl == 3) hipLaunchKernelGGL((se_squeeze_v2_kernel<3>), dim3(B), dim3(512), 0, st, a);
  else return -1000;
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// se_combine_fwd_v3 (round 5): se_squeeze_v2 + combine_fwd_v2 in ONE launch that reads Y3 once.
// The SE gate of an utterance needs the mean over ALL its frames of act3(Y3) before the first output row can be formed, so the
// two-kernel form reads Y3 twice (17 x 39 MB per step).  An utterance of <= 320 frames x 256 channels is 160 KB of bf16: the
// 512 threads of its workgroup KEEP it in registers (<= 20 rows x 16 bytes per thread, the row / channel-vector assignment of
// se_squeeze_v2) across the two mat-vecs and apply the gate to the rows they hold; only the skip operand S is streamed in the
// second phase (first rows requested before the mat-vec tail, the rest one group of 4 rows ahead).  Same arithmetic in the same
// order as the two kernels it replaces: outputs, means, hidden layer and gates are bit-identical (tests/test_se_fused_gpu.py).
// One workgroup per utterance (B >= the CU count at the benched shape); T > 320: the two-kernel form.
// ------------------------------------------------------------------------------------------
#define SC3_MAXU 20
// the keep decisions of tn_drop8 as 8 bits (bit i = element i kept)
__device__ __forceinline__ uint32_t tn_drop8_bits(uint32_t idx8, uint32_t key, uint32_t thr) {
  const uint32_t x = tn_drop_shared(idx8, key);
  const uint32_t h0 = tn_drop_final(x, TN_DROP_C0), h1 = tn_drop_final(x, TN_DROP_C1);
  const uint32_t h2 = tn_drop_final(x, TN_DROP_C2), h3 = tn_drop_final(x, TN_DROP_C3);
  uint32_t m = 0;
  m |= ((h0 & 0xffffu) >= thr) ? 1u : 0u;  m |= ((h0 >> 16) >= thr) ? 2u : 0u;
  m |= ((h1 & 0xffffu) >= thr) ? 4u : 0u;  m |= ((h1 >> 16) >= thr) ? 8u : 0u;
  m |= ((h2 & 0xffffu) >= thr) ? 16u : 0u; m |= ((h2 >> 16) >= thr) ? 32u : 0u;
  m |= ((h3 & 0xffffu) >= thr) ? 64u : 0u; m |= ((h3 >> 16) >= thr) ? 128u : 0u;
  return m;
}
struct SeCombineV3Args {
  SeSqueezeV2Args se;        // Y = Y3, act = act3, W1, W2, m_out / h_out / g_out, T, len
  const bf16_t* S; BnAct actS;
  bf16_t* OUT;
  uint32_t drop_thr, drop_key;
  float inv_keep;
  const uint32_t* key_add;   // see BnAct::key_add
};
template <int FL3, bool DROP>
__global__ __launch_bounds__(512) void se_combine_fwd_v3_kernel(SeCombineV3Args aa) {
  constexpr int HR = 16, G = 4, NG = SC3_MAXU / G;
  const SeSqueezeV2Args& a = aa.se;
  __shared__ float cst[4 * V2_C];
  __shared__ float part[16][V2_C];
  __shared__ float mean[V2_C];
  __shared__ float gs[V2_C];
  __shared__ float hbuf[HR];
  const int tid = threadIdx.x, vc = tid & 31, tg = tid >> 5, c0 = vc * 8, b = blockIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int T = a.T;
  const int L = a.len ? a.len[b] : T;
  const size_t base = (size_t)b * T;
  // BatchNorm constants first: their loads must not queue behind the utterance's rows (vmcnt retires in order)
  float cs0 = 1.f, ch0 = 0.f, cs1 = 1.f, ch1 = 0.f;
  if (tid < V2_C) {
    if (FL3 & 1) bn_scale_shift(a.act, V2_C, tid, cs1, ch1);
    bn_scale_shift(aa.actS, V2_C, tid, cs0, ch0);
  }
  // the whole utterance: unconditional buffer loads through a descriptor of the utterance's VALID rows (frames past them read
  // as zeros and are masked out of the sums); one voffset per thread, the row step in the scalar offset
  typedef __attribute__((ext_vector_type(4))) unsigned int sc3_u32x4_t;
  const __amdgpu_buffer_rsrc_t srdY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.Y + base * V2_C), 0, L * V2_C * (int)sizeof(bf16_t), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdS = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(aa.S + base * V2_C), 0, L * V2_C * (int)sizeof(bf16_t), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc(aa.OUT + base * V2_C, 0, T * V2_C * (int)sizeof(bf16_t), 0x00020000);
  const int voff = (tg * V2_C + c0) * (int)sizeof(bf16_t);
  constexpr int USTEP = 16 * V2_C * (int)sizeof(bf16_t);
  // NO load may be out of range: a buffer load whose lanes are all past the descriptor returns its zeros at once, ahead of older
  // loads still in flight, and every counted wait behind it (the compiler's and the ones below) is then one short — rows past
  // the valid frames re-read the last valid row and are masked by `t < L`
  const int lastrow = max(L, 1) - 1;
  auto row_off = [&](int u) -> int { return (min(tg + 16 * u, lastrow) * V2_C + c0) * (int)sizeof(bf16_t); };
  uint4 ry[SC3_MAXU];
#pragma unroll
  for (int u = 0; u < SC3_MAXU; ++u) ry[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srdY, row_off(u), 0, (TN_NT_WGRAD_OPERANDS & 64) ? 2 : 0));
  // skip rows: three groups of 4 rows in flight (the first behind the utterance's rows, the second before the mat-vec tail)
  uint4 rs[3][G];
  auto load_s = [&](int g, uint4 (&dst)[G]) {
#pragma unroll
    for (int q = 0; q < G; ++q) dst[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srdS, row_off(g * G + q), 0, (TN_NT_WGRAD_OPERANDS & 64) ? 2 : 0));
  };
  load_s(0, rs[0]);
  if (tid < V2_C) { cst[tid] = cs0; cst[V2_C + tid] = ch0; cst[2 * V2_C + tid] = cs1; cst[3 * V2_C + tid] = ch1; }
  __syncthreads();
  float sc3[8], sh3[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { sc3[i] = cst[2 * V2_C + c0 + i]; sh3[i] = cst[3 * V2_C + c0 + i]; }
  const uint32_t dkey3 = tn_act_key(a.act), dthr3 = a.act.drop_thr;
  uint32_t km[SC3_MAXU / 4];
#pragma unroll
  for (int i = 0; i < SC3_MAXU / 4; ++i) km[i] = 0u;
  {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
    for (int u = 0; u < SC3_MAXU; ++u) {
      const int t = tg + 16 * u;
      float v[8];
      unpack8(ry[u], v);
      act8_t<FL3 & 3>(v, sc3, sh3, dkey3, dthr3, (uint32_t)b * T + t, c0);
      if (FL3 & 4) {
        // the keep bits of the row are kept for phase 2 (8 bits per row; left to itself hipcc stashes the 64-bit compare
        // results of all 160 elements in VGPR lanes for reuse, and the reused masks came back wrong in ~1 % of the elements)
        const uint32_t m = tn_drop8_bits((((uint32_t)b * T + t) * (uint32_t)V2_C + (uint32_t)c0) >> 3, dkey3, dthr3);
        km[u / 4] |= m << (8 * (u % 4));
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (m & (1u << i)) ? v[i] : 0.f;
      }
      const bool ok = t < L;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += ok ? v[i] : 0.f;
      __builtin_amdgcn_sched_barrier(0);      // (one row's arithmetic at a time, not all twenty interleaved: registers)
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) part[tg][c0 + i] = acc[i];
  }
  // weights of the tail (L2 hits: every workgroup reads the same 32 KB), in flight during the partial-sum exchange
  float w1a[4], w1b[4];      // W1 rows wave and wave + 8, columns lane + 64 k
#pragma unroll
  for (int k = 0; k < 4; ++k) { w1a[k] = a.W1[(size_t)wave * V2_C + lane + 64 * k]; w1b[k] = a.W1[(size_t)(wave + 8) * V2_C + lane + 64 * k]; }
  float4 w2[4];              // W2 row tid (tid < 256)
  if (tid < V2_C) {
#pragma unroll
    for (int k = 0; k < 4; ++k) w2[k] = *reinterpret_cast<const float4*>(a.W2 + (size_t)tid * HR + 4 * k);
  }
  load_s(1, rs[1]);
  __syncthreads();
  if (tid < V2_C) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += part[k][tid];
    s *= 1.f / (float)max(L, 1);
    mean[tid] = s;
    a.m_out[(size_t)b * V2_C + tid] = s;
  }
  __syncthreads();
  {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s0 = fmaf(w1a[k], mean[lane + 64 * k], s0); s1 = fmaf(w1b[k], mean[lane + 64 * k], s1); }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (lane == 0) {
      s0 = fmaxf(s0, 0.f); s1 = fmaxf(s1, 0.f);
      hbuf[wave] = s0; hbuf[wave + 8] = s1;
      a.h_out[(size_t)b * HR + wave] = s0;
      a.h_out[(size_t)b * HR + wave + 8] = s1;
    }
  }
  __syncthreads();
  if (tid < V2_C) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s = fmaf(w2[k].x, hbuf[4 * k], s); s = fmaf(w2[k].y, hbuf[4 * k + 1], s);
      s = fmaf(w2[k].z, hbuf[4 * k + 2], s); s = fmaf(w2[k].w, hbuf[4 * k + 3], s);
    }
    const float gv = 1.f / (1.f + __expf(-s));
    gs[tid] = gv;
    a.g_out[(size_t)b * V2_C + tid] = gv;
  }
  __syncthreads();
  // ---- phase 2: OUT = dropout(relu(BN(S) + g * act3(Y3))) on the rows this thread holds
  // (the rows stay PACKED across the tail: without this hipcc keeps phase 1's activated f32 values of all 20 rows for reuse —
  //  twice the registers, spills)
#pragma unroll
  for (int u = 0; u < SC3_MAXU; ++u) asm volatile("" : "+v"(ry[u].x), "+v"(ry[u].y), "+v"(ry[u].z), "+v"(ry[u].w));
#pragma unroll
  for (int i = 0; i < SC3_MAXU / 4; ++i) asm volatile("" : "+v"(km[i]));
  float scS[8], shS[8], g[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    scS[i] = cst[c0 + i]; shS[i] = cst[V2_C + c0 + i]; g[i] = gs[c0 + i];
    if (DROP) { scS[i] *= aa.inv_keep; shS[i] *= aa.inv_keep; g[i] *= aa.inv_keep; }   // relu(k x) = k relu(x), k > 0
  }
  const uint32_t okey = aa.key_add ? aa.drop_key + *aa.key_add : aa.drop_key;
#pragma unroll
  for (int gq = 0; gq < NG; ++gq) {
    if (gq + 2 < NG) load_s(gq + 2, rs[(gq + 2) % 3]);
    // hipcc counts its waits as if loads and stores retired in ONE order: with the previous groups' stores in flight its
    // vmcnt(newer loads + newer stores) is satisfied as soon as the STORES are acknowledged, this group's rows still on their
    // way (measured: wrong outputs from the third group on).  Loads do retire in order among themselves, so the wait that
    // holds is vmcnt(newer loads only) — conservative when stores are pending.
    if (gq + 2 < NG) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (gq + 1 < NG) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);      // (hipcc otherwise hoists every group's loads to the top: 80 more live registers, spills)
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const int u = gq * G + q, t = tg + 16 * u;
      const uint32_t row = (uint32_t)b * T + t;
      float s[8], y[8], o[8];
      unpack8(rs[gq % 3][q], s);
      unpack8(ry[u], y);
      act8_t<FL3 & 3>(y, sc3, sh3, dkey3, dthr3, row, c0);
      if (FL3 & 4) {
        const uint32_t m = km[u / 4] >> (8 * (u % 4));
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = (m & (1u << i)) ? y[i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = fmaxf(fmaf(s[i], scS[i], fmaf(g[i], y[i], shS[i])), 0.f);
      if (DROP) tn_drop8(o, (row * (uint32_t)V2_C + (uint32_t)c0) >> 3, okey, aa.drop_thr);
      const bool ok = t < L;                 // padding rows are written as zeros; rows >= T fall outside the descriptor
      sc3_u32x4_t w;
      w[0] = ok ? f2bf_pk(o[0], o[1]) : 0u; w[1] = ok ? f2bf_pk(o[2], o[3]) : 0u;
      w[2] = ok ? f2bf_pk(o[4], o[5]) : 0u; w[3] = ok ? f2bf_pk(o[6], o[7]) : 0u;
      // (the row step goes into the VECTOR offset, scalar offset 0: with an SGPR scalar offset hipcc's hazard recogniser assumes
      //  a 16-byte store has read its data registers at issue and lets a VALU write of the first one follow within 5
      //  instructions — on gfx950 the last lanes of the store then carried the NEW value: the first bf16 pair of ~0.03 % of
      //  the rows came out as the next load's address.  Found as zeros in elements 0, 1 of rows 48 + tg, 112 + tg, ..)
      __builtin_amdgcn_raw_buffer_store_b128(w, srdO, voff + u * USTEP, 0, 0);
      __builtin_amdgcn_sched_barrier(0);      // one row's arithmetic at a time: two interleaved rows do not fit the register file
    }
  }
}
// -1000: shape / flags outside the kernel (the caller runs se_squeeze_v2 + combine_fwd_v2)
inline int launch_se_combine_fwd_v3(const SeCombineV3Args& a, int B, hipStream_t st) {
  if (a.se.T > 16 * SC3_MAXU) return -1000;
  const int fl3 = (a.se.act.mode != 0 ? 1 : 0) | (a.se.act.relu ? 2 : 0) | (a.se.act.drop_thr ? 4 : 0);
  if (fl3 == 7 && a.drop_thr) hipLaunchKernelGGL((se_combine_fwd_v3_kernel<7, true>), dim3(B), dim3(512), 0, st, a);
  else if (fl3 == 3 && !a.drop_thr) hipLaunchKernelGGL((se_combine_fwd_v3_kernel<3, false>), dim3(B), dim3(512), 0, st, a);
  else return -1000;
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// sub_fwd_v5: PRODUCER / CONSUMER wave specialisation of the forward sub-block.
// With one 512-thread workgroup per CU every wave used to sit in the same barrier-separated phase
// (activation -> stencil -> MFMA -> epilogue -> store), so VALU, LDS and MFMA time ADDED up.  Here waves 0-3
// (one per SIMD) are producers: they turn the prefetched raw rows of tile i+1 into the MFMA operand tile
// (BatchNorm + ReLU + dropout on load, depthwise stencil), while waves 4-7 (the other wave of each SIMD) are
// consumers of tile i: pointwise GEMM with the whole 256 x 256 weight in THEIR registers (64 output channels
// = 128 VGPRs per wave), epilogue, coalesced store, BN statistics.  A SIMD therefore always has one VALU-bound
// and one MFMA-bound wave to pick from, the operand tile is double buffered, and only two workgroup barriers
// per tile remain.  The B-operand LDS reads halve (4 consumer waves instead of 8 re-read the tile).
// ------------------------------------------------------------------------------------------
// MK (variable-length batch, a.act.rm.len): the producers read padding rows as zeros and WRITE the depthwise output of padding
// rows as zeros (operand tile and kept copy), so the GEMM gives y == bias there: the host takes those rows out of the
// statistics again (stats_pad_fixup_kernel) and the weight gradients see a zero operand row.
template <int KD, bool DW, int FL, bool MK = false, int R = V2_R>
__global__ __launch_bounds__(V2_NT, 2) void sub_fwd_v5_kernel(SubFwdV2Args a) {
  constexpr int PADR = DW ? (KD - 1) / 2 : 0;
  constexpr int OUTR = DW ? R - (KD - 1) : R;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Cs = reinterpret_cast<bf16_t*>(smem);     // [64][264] output staging (consumers)
  bf16_t* As0 = Cs + R * V2_AP;                  // [2][64][264] MFMA B operand, double buffered
  bf16_t* Xa = As0 + 2 * R * V2_AP;              // [64][256] activated input rows (producers)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave < 4;
  const int ltid = tid & 255;                        // thread index inside the team
  const int vc = ltid & 31, rq = ltid >> 5, c0 = vc * 8;   // 32 channel vectors x 8 row phases
  const int half = lane >> 5, cw = wave & 3;         // consumer: 64 output channels cw*64 ..
  const uint32_t dkey = tn_act_key(a.act), dthr = a.act.drop_thr;
  const int stride = gridDim.x;
  const int first = blockIdx.x;
  if (first >= a.ntiles) return;

  // ---- per-thread constants through LDS (cooperative: one channel per thread)
  float sc[8], sh[8], wd[DW ? KD : 1][8], bd[8];
  {
    float* tmp = reinterpret_cast<float*>(As0);
    if (tid < V2_C) {
      float s = 1.f, h = 0.f;
      if (FL & 1) bn_scale_shift(a.act, V2_C, tid, s, h);
      tmp[tid] = s;
      tmp[V2_C + tid] = h;
      if (DW) {
        tmp[2 * V2_C + tid] = a.bdw[tid];
#pragma unroll
        for (int k = 0; k < KD; ++k) tmp[(3 + k) * V2_C + tid] = a.wdw[(size_t)tid * KD + k];
      }
    }
    __syncthreads();
    if (producer) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sc[i] = tmp[c0 + i];
        sh[i] = tmp[V2_C + c0 + i];
        if (DW) {
          bd[i] = tmp[2 * V2_C + c0 + i];
#pragma unroll
          for (int k = 0; k < KD; ++k) wd[k][i] = tmp[(3 + k) * V2_C + c0 + i];
        }
      }
    }
    __syncthreads();
  }
  // The two roles run in DISJOINT code regions (each with its own copy of the tile loop and the same number of
  // workgroup barriers), so that the register allocator sees max(producer, consumer) live values, not their sum.
  if (producer) {
    // ---- producer: raw rows of the next tile (8 rows per thread) -> activated rows -> stencil -> operand tile
    // unconditional BUFFER loads / stores (rows outside the tensor and tiles past the last one are out of the descriptor's
    // range: zeros / dropped): predicated accesses sit in exec-masked blocks, hipcc then waits with vmcnt(0) for the next
    // tile's rows — and with them for the kept depthwise output rows it has just stored
    typedef __attribute__((ext_vector_type(4))) unsigned int v5_u32x4_t;
    const int tbytes = (int)((size_t)a.M * V2_C * sizeof(bf16_t));
    const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.X), 0, tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srdQ = __builtin_amdgcn_make_buffer_rsrc(a.Q, 0, a.Q ? tbytes : 0, 0x00020000);
    constexpr int V5_OOB = 0x7ffffff0;
    auto store_q = [&](int row, bool keep, const float (&v)[8]) {
      v5_u32x4_t w;
      w[0] = f2bf_pk(v[0], v[1]); w[1] = f2bf_pk(v[2], v[3]); w[2] = f2bf_pk(v[4], v[5]); w[3] = f2bf_pk(v[6], v[7]);
      __builtin_amdgcn_raw_buffer_store_b128(w, srdQ, keep ? (row * V2_C + c0) * (int)sizeof(bf16_t) : V5_OOB, 0, (TN_NT_WGRAD_OPERANDS & 1) ? 2 : 0);
    };
    uint4 pf[R / 8];
    auto prefetch_q = [&](int tile, int q) {
      const int gr = tile * OUTR - PADR + rq + 8 * q;
      pf[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srdX, (tile < a.ntiles) ? (gr * V2_C + c0) * (int)sizeof(bf16_t) : V5_OOB, 0, (TN_NT_WGRAD_OPERANDS & 128) ? 2 : 0));
    };
    auto prefetch = [&](int tile) {
#pragma unroll
      for (int q = 0; q < R / 8; ++q) prefetch_q(tile, q);
    };
    TileMask tm = {0, 0, 0};
    auto produce_act = [&](int tile, bf16_t* As) {
      const int raw0 = tile * OUTR - PADR;
      const bool interior = raw0 >= 0 && raw0 + R <= a.M;
      if (MK) tm = tn_tile_mask(a.act.rm.len, a.T, a.M, raw0);         // (produce_stencil of the same tile reuses it)
#pragma unroll
      for (int q = 0; q < R / 8; ++q) {
        const int r = rq + 8 * q, gr = raw0 + r;
        float v[8];
        unpack8(pf[q], v);
        if (MK) {
          if (gr >= 0 && gr < a.M && tn_tile_valid(tm, gr)) act8_t<FL>(v, sc, sh, dkey, dthr, (uint32_t)gr, c0);
          else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = 0.f;
          }
        } else
        if (interior || (gr >= 0 && gr < a.M)) act8_t<FL>(v, sc, sh, dkey, dthr, (uint32_t)gr, c0);
        if (DW) store8(Xa + r * V2_C + c0, v);
        else store8(As + r * V2_AP + c0, v);
      }
      prefetch(tile + stride);     // (refilling each register right after its use measured 3 us slower here: the producers'
                                   //  LDS stores gate the consumers, and load issue in between delays them)
    };
    auto produce_stencil = [&](int tile, bf16_t* As) {
      const int out0 = tile * OUTR, raw0 = out0 - PADR;
      // (masked: the fast path also needs every row of the tile to be a valid frame)
      const bool one_utt = raw0 >= 0 && raw0 + R <= a.M && (raw0 % a.T) + R <= a.T && (!MK || raw0 + R <= tm.lim0);
#pragma unroll
      for (int grp = 0; grp < R / 32; ++grp) {
        const int o0 = grp * 32 + rq * 4;
        float win[KD + 3][8];
#pragma unroll
        for (int j = 0; j < KD + 3; ++j) {
          if (o0 + j < R) load8(Xa + (o0 + j) * V2_C + c0, win[j]);
          else {
#pragma unroll
            for (int i = 0; i < 8; ++i) win[j][i] = 0.f;
          }
        }
        if (one_utt) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(wd[0][i], win[q][i], bd[i]);
#pragma unroll
            for (int k = 1; k < KD; ++k)
#pragma unroll
              for (int i = 0; i < 8; ++i) acc[i] = fmaf(wd[k][i], win[q + k][i], acc[i]);
            store8(As + (o0 + q) * V2_AP + c0, acc);
            store_q(out0 + o0 + q, o0 + q < OUTR && out0 + o0 + q < a.M, acc);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int o = o0 + q;
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = bd[i];
            if (o < OUTR) {
              const int t = (out0 + o) % a.T;
#pragma unroll
              for (int k = 0; k < KD; ++k) {
                const int tt = t + k - PADR;
                if (tt >= 0 && tt < a.T) {
#pragma unroll
                  for (int i = 0; i < 8; ++i) acc[i] = fmaf(wd[k][i], win[q + k][i], acc[i]);
                }
              }
              if (MK && !tn_tile_valid(tm, out0 + o)) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = 0.f;
              }
            }
            store8(As + o * V2_AP + c0, acc);
            store_q(out0 + o, o < OUTR && out0 + o < a.M, acc);
          }
        }
      }
    };
    prefetch(first);
    produce_act(first, As0);
    __syncthreads();
    if (DW) produce_stencil(first, As0);
    __syncthreads();
    int buf = 0;
    for (int tile = first; tile < a.ntiles; tile += stride) {
      const int nxt = tile + stride;
      bf16_t* Anxt = As0 + (buf ^ 1) * R * V2_AP;
      if (nxt < a.ntiles) produce_act(nxt, Anxt);
      __syncthreads();
      if (DW && nxt < a.ntiles) produce_stencil(nxt, Anxt);
      __syncthreads();
      buf ^= 1;
    }
    if (a.stats) __syncthreads();
  } else {
    // ---- consumer: weights of 64 output channels as MFMA A fragments, bias, statistics
    bf16x8_t wf[2][16];
    float biasr[2][16];
    float st_s[8], st_q[8];
    typedef __attribute__((ext_vector_type(4))) unsigned int v5c_u32x4_t;
    const __amdgpu_buffer_rsrc_t srdYc = __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, (int)((size_t)a.M * V2_C * sizeof(bf16_t)), 0x00020000);
#pragma unroll
    for (int cbk = 0; cbk < 2; ++cbk) {
      const int co = cw * 64 + cbk * 32 + (lane & 31);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        wf[cbk][ks] = a.Wswz ? __builtin_bit_cast(bf16x8_t, a.Wswz[((size_t)(cw * 2 + cbk) * 16 + ks) * 64 + lane])
                             : *reinterpret_cast<const bf16x8_t*>(a.W + (size_t)co * V2_C + ks * 16 + half * 8);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) biasr[cbk][4 * g + j] = a.bias[cw * 64 + cbk * 32 + 8 * g + 4 * half + j];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { st_s[i] = 0.f; st_q[i] = 0.f; }
    __syncthreads();      // (pipeline fill: producers' activation)
    __syncthreads();      // (pipeline fill: producers' stencil)
    int buf = 0;
    for (int tile = first; tile < a.ntiles; tile += stride) {
      const bf16_t* As = As0 + buf * R * V2_AP;
      {
        f32x16_t acc[2][2];    // [channel block][row block]
#pragma unroll
        for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[cbk][0][r] = biasr[cbk][r]; acc[cbk][1][r] = biasr[cbk][r]; }
        const bf16_t* brow = As + (lane & 31) * V2_AP + half * 8;
#if V5_MD
        if constexpr (R == 32) {
          // hand-scheduled: V5_MD fragment reads in flight (tuning switch; this wave is the only MFMA wave of its SIMD)
          f32x16_t a2[2] = {acc[0][0], acc[1][0]};
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          tn_mfma_sched_lds<16, 1, 2, V5_MD, 32, 0>(&wf[0][0], brow, &a2[0]);
          acc[0][0] = a2[0]; acc[1][0] = a2[1];
        } else
#endif
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(brow + ks * 16);
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][ks], b0, acc[0][0], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][ks], b0, acc[1][0], 0, 0, 0);
          if (R == 64) {
            const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(brow + 32 * V2_AP + ks * 16);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][ks], b1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][ks], b1, acc[1][1], 0, 0, 0);
          }
        }
#pragma unroll
        for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int co = cw * 64 + cbk * 32 + 8 * g + 4 * half;
            uint2 w0, w1;
            w0.x = f2bf_pk(acc[cbk][0][4 * g], acc[cbk][0][4 * g + 1]); w0.y = f2bf_pk(acc[cbk][0][4 * g + 2], acc[cbk][0][4 * g + 3]);
            w1.x = f2bf_pk(acc[cbk][1][4 * g], acc[cbk][1][4 * g + 1]); w1.y = f2bf_pk(acc[cbk][1][4 * g + 2], acc[cbk][1][4 * g + 3]);
            *reinterpret_cast<uint2*>(Cs + (lane & 31) * V2_AP + co) = w0;
            if (R == 64) *reinterpret_cast<uint2*>(Cs + (32 + (lane & 31)) * V2_AP + co) = w1;
          }
      }
      __syncthreads();
      {
        const int out0 = tile * OUTR;
#pragma unroll
        for (int q = 0; q < R / 8; ++q) {
          const int o = rq + 8 * q, gr = out0 + o;
          const bool keep = o < OUTR && gr < a.M;
          const uint4 raw = *reinterpret_cast<const uint4*>(Cs + (keep ? o : 0) * V2_AP + c0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v5c_u32x4_t, raw), srdYc, keep ? (gr * V2_C + c0) * (int)sizeof(bf16_t) : 0x7ffffff0, 0, 0);
          if (keep) {
            float y[8];
            unpack8(raw, y);
#pragma unroll
            for (int i = 0; i < 8; ++i) { st_s[i] += y[i]; st_q[i] = fmaf(y[i], y[i], st_q[i]); }
          }
        }
      }
      __syncthreads();
      buf ^= 1;
    }
    if (a.stats) {
      float* red = reinterpret_cast<float*>(smem);      // [8][2][256]
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        red[(rq * 2 + 0) * V2_C + c0 + i] = st_s[i];
        red[(rq * 2 + 1) * V2_C + c0 + i] = st_q[i];
      }
      __syncthreads();
    }
  }
  if (a.stats) {
    const float* red = reinterpret_cast<const float*>(smem);
    const int rep = blockIdx.x % TN_NREP;
    const int which = tid >> 8, c = tid & 255;
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) v += red[(r * 2 + which) * V2_C + c];
    atomic_add_f32(&a.stats[(size_t)(rep * 2 + which) * V2_C + c], v);
  }
}

template <int KD, bool DW, int FL, int R>
inline int launch_sub_fwd_v5_t(SubFwdV2Args a, int grid, size_t smem, hipStream_t st) {
  auto kern = a.act.rm.len ? sub_fwd_v5_kernel<KD, DW, FL, true, R> : sub_fwd_v5_kernel<KD, DW, FL, false, R>;
  TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(V2_NT), smem, st, a);
  return (int)hipGetLastError();
}
// R: raw rows per tile (64, or 32: twice the tiles per workgroup — a shorter pipeline fill / drain per launch)
template <int KD, bool DW, int R = V2_R>
inline int launch_sub_fwd_v5(SubFwdV2Args a, int resident_wgs, hipStream_t st) {
  if ((long)a.M * V2_C * 2 >= (1L << 31)) return -1000;      // 32-bit buffer offsets: the generic kernel takes such a batch
  constexpr int OUTR = DW ? R - (KD - 1) : R;
  a.ntiles = (a.M + OUTR - 1) / OUTR;
  const int grid = a.ntiles < resident_wgs ? a.ntiles : resident_wgs;
  const size_t tiles = (size_t)(3 * R * V2_AP + R * V2_C) * sizeof(bf16_t), red = (size_t)8 * 2 * V2_C * sizeof(float);
  const size_t smem = tiles > red ? tiles : red;
  const int fl = (a.act.mode != 0 ? 1 : 0) | (a.act.relu ? 2 : 0) | (a.act.drop_thr ? 4 : 0);
  switch (fl) {
    case 0: return launch_sub_fwd_v5_t<KD, DW, 0, R>(a, grid, smem, st);
    case 3: return launch_sub_fwd_v5_t<KD, DW, 3, R>(a, grid, smem, st);
    case 7: return launch_sub_fwd_v5_t<KD, DW, 7, R>(a, grid, smem, st);
    default: return -1000;      // no specialisation for this activation (the caller runs the generic GEMM)
  }
}


// fragment-order copies of bf16 weight matrices [N][K] (N a multiple of 32, K of 16): one launch for all of them, after
// every parameter cast.  dst[((rb * K/16 + ks) * 64 + lane)] = 16 bytes of row rb*32 + (lane & 31), columns
// ks*16 + (lane >> 5)*8 .. +8  — i.e. exactly what lane `lane` of the wave owning row block rb feeds to MFMA k-step ks.
struct SwzDesc { const bf16_t* src; uint4* dst; int N, K; };
template <int DUMMY>
__global__ void swizzle256_kernel(const SwzDesc* __restrict__ tab) {
  const SwzDesc d = tab[blockIdx.y];
  const int KS = d.K / 16, total = d.N * d.K / 8;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int lane = i & 63, q = i >> 6, ks = q % KS, rb = q / KS;
    d.dst[i] = *reinterpret_cast<const uint4*>(d.src + (size_t)(rb * 32 + (lane & 31)) * d.K + ks * 16 + (lane >> 5) * 8);
  }
}
