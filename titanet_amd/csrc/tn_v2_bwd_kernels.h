// titanet_amd — "v2" backward kernels for the headline shape (hidden = 256, bf16); see tn_v2_kernels.h
// for the design notes.  Included by tn_bwd.hip only.
#pragma once
#include "tn_bwd_kernels.h"
#include "tn_v2_kernels.h"

// ==========================================================================================
// Batched pointwise weight gradients (all mega-block 1x1 convs of the model in ONE launch).
//
// d W[co][ci] = sum over rows r of dY[r][co] * Q[r][ci],
//     dY = BatchNorm-backward-on-load(dZ, Y)                      (P operand)
//     Q  = dwconv(act(Xprev))  (sub-blocks)  or  act(X)  (skip)   (recomputed, never stored)
//
// MI355X-first: the per-layer activation gradients dZ are KEPT (one buffer per layer — 2.7 GB at
// B=256, trivial against 288 GB of HBM) so the weight gradients are off the backward critical path
// and can be computed for all 68 layers at once.  The launch is perfectly load balanced: the
// (layer, 32-row chunk) work units are cut into one contiguous range per workgroup; a workgroup owns
// the FULL 256x256 output of its layer segment (8 waves x 128x64 accumulators), so operand tiles are
// produced once (no redundant recompute across output tiles) and only ~4 partial slabs per layer
// exist.  The contraction runs over LDS rows: bf16 fragments via ds_read_b64_tr_b16.
// ==========================================================================================
typedef __attribute__((ext_vector_type(4))) short v2_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short v2_s16x8_t;
// bf16 MFMA fragment (8 row-consecutive values of one column) out of a row-major LDS tile, via the
// gfx950 transpose read: lane (16-lane group g = lane>>4, i = lane&15) addresses 4 bf16 of row
// k0 + 8*(lane>>5) + (i>>2) [+4], columns cbase + 16*(g&1) + 4*(i&3) and receives column
// cbase + 16*(g&1) + i of rows +0..3 (semantics verified on hardware by tools/trprobe.hip).
__device__ __forceinline__ bf16x8_t wg_frag_v2(const bf16_t* tile, int k0, int cbase, int lane) {
  const int i = lane & 15;
  const bf16_t* p = tile + (k0 + 8 * (lane >> 5) + (i >> 2)) * WG2_PITCH + cbase + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
  const v2_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v2_s16x4_t*)(p));
  const v2_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v2_s16x4_t*)(p + 4 * WG2_PITCH));
  v2_s16x8_t v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, v);
}

struct WgradV2Desc {
  const bf16_t* dZ;
  const bf16_t* Y;       // null (with fstats null): P = dZ as stored (no BatchNorm between the gradient and the GEMM)
  const float* fstats;   // BN of Y: forward sums
  const float* bsums;    // BN of Y: backward sums (complete before this launch)
  const float* gamma;
  float inv_n, eps, batch;
  const bf16_t* X;       // raw input of the layer
  BnAct actX;
  const float* wdw;      // depthwise taps [256][KD] or null (plain 1x1 conv)
  const float* bdw;
  float* slabs;          // [max_parts][256*256] partial sums of this layer
  int drop_layer;        // dropout stream id of actX (key = tn_layer_key(seed, drop_layer)), set per step on device
  int ldp;               // row stride (elements) of dZ / Y: 256, or the full width when the layer is a 256-channel slab of a wider tensor
  int statC, chan0;      // channel count of the BN statistics arrays and this slab's first channel in them
  int ldq, q0;           // row stride (elements) of X and this unit's first channel in it (256 / 0 for the 256-wide models; wider
                         // models are cut into 256 x 256 output slabs: one unit per (P slab, Q slab) pair)
};


struct WgSeg {
  bf16_t* Pt; bf16_t* Qt; bf16_t* Xa; float* cst;
  int chunk0, nchunks, M, T;
};
// FL = activation flags of the layer input (1 BN, 2 ReLU, 4 dropout) or -1 = decide at run time
template <int FL>
__device__ __forceinline__ void wg2_act8(float v[8], const float* sc, const float* sh, const BnAct& a, uint32_t row, int c0) {
  if (FL < 0) { act8(v, sc, sh, a, row, V2_C, c0); return; }
  if (FL & 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], sc[i], sh[i]);
  }
  if (FL & 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
  }
  if (FL & 4) tn_drop8(v, (row * (uint32_t)V2_C + (uint32_t)c0) >> 3, tn_act_key(a), a.drop_thr);
}

template <int KD, bool DW, int FL>
__device__ __forceinline__ void wg2_chunks(const WgradV2Desc& d, const WgSeg& sg, f32x16_t (&acc)[4][2]) {
  constexpr int PADR = (KD - 1) / 2;
  bf16_t* Pt = sg.Pt; bf16_t* Qt = sg.Qt; bf16_t* Xa = sg.Xa; const float* cst = sg.cst;
  const int M = sg.M, T = sg.T, chunk0 = sg.chunk0, nchunks = sg.nchunks;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vc = tid & 31, rq = tid >> 5, c0 = vc * 8;
  const int wa = wave >> 2, wb = wave & 3;
  uint4 pz[2], py[2], px[3];
  // every prefetch register is refilled (for the next chunk) right after it was consumed, so that the memory pipe is
  // never empty: 5 - 7 loads per thread are in flight at any time, also while the tiles are transformed
  auto load_p = [&](int chunk, int q) {
    const int gr = chunk * WG2_RK + rq + 16 * q;
    if (gr < M) {
      pz[q] = *reinterpret_cast<const uint4*>(d.dZ + (size_t)gr * d.ldp + c0);
      py[q] = d.Y ? *reinterpret_cast<const uint4*>(d.Y + (size_t)gr * d.ldp + c0) : make_uint4(0, 0, 0, 0);
    } else {
      pz[q] = make_uint4(0, 0, 0, 0); py[q] = make_uint4(0, 0, 0, 0);
    }
  };
  auto load_x = [&](int chunk, int q) {
    const int i = rq + 16 * q;               // Xa row 0 .. 47 (need < 32 + KD - 1)
    const int gr = chunk * WG2_RK - PADR + i;
    if (i < WG2_RK + KD - 1 && gr >= 0 && gr < M) px[q] = *reinterpret_cast<const uint4*>(d.X + (size_t)gr * d.ldq + d.q0 + c0);
    else px[q] = make_uint4(0, 0, 0, 0);
  };
#pragma unroll
  for (int q = 0; q < 2; ++q) load_p(chunk0, q);
#pragma unroll
  for (int q = 0; q < 3; ++q) load_x(chunk0, q);
  for (int ch = 0; ch < nchunks; ++ch) {
    const int r0 = (chunk0 + ch) * WG2_RK;
    const bool interior = r0 - PADR >= 0 && r0 + WG2_RK + PADR <= M;               // workgroup-uniform
    const bool one_utt = interior && ((r0 - PADR) % T) + WG2_RK + 2 * PADR <= T;   // ... rows + halo in ONE utterance
    __syncthreads();   // previous MFMA done with Pt/Qt; constants visible
    // ---- P tile: BN backward on load  (out-of-range rows were loaded as zeros; their k2 term only matters
    //      for rows < M, so it is masked in the non-interior case)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = rq + 16 * q, gr = r0 + r;
      float z[8], y[8];
      unpack8(pz[q], z);
      unpack8(py[q], y);
      if (interior || gr < M) {
#pragma unroll
        for (int i = 0; i < 8; ++i) z[i] = cst[c0 + i] * z[i] + cst[V2_C + c0 + i] * y[i] + cst[2 * V2_C + c0 + i];
      }
      store8(Pt + r * WG2_PITCH + c0, z);
      if (ch + 1 < nchunks) load_p(chunk0 + ch + 1, q);
      __builtin_amdgcn_sched_barrier(0);   // bound the live temporaries (the 128 accumulators leave ~100 VGPRs)
    }
    // ---- activated input rows (with halo)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int i = rq + 16 * q, gr = r0 - PADR + i;
      if (i < WG2_RK + KD - 1) {
        if (!DW && FL == 0) {
          // stored operand used as it is (the kept depthwise outputs, an activated block input): a straight 16-byte copy
          // (rows outside the tensor were loaded as zeros)
          if (i >= PADR && i < PADR + WG2_RK) *reinterpret_cast<uint4*>(Qt + (i - PADR) * WG2_PITCH + c0) = px[q];
        } else {
          float v[8];
          unpack8(px[q], v);
          if (interior || (gr >= 0 && gr < M)) wg2_act8<FL>(v, cst + 3 * V2_C + c0, cst + 4 * V2_C + c0, d.actX, (uint32_t)gr, c0);
          if (DW) store8(Xa + i * V2_C + c0, v);
          else if (i >= PADR && i < PADR + WG2_RK) store8(Qt + (i - PADR) * WG2_PITCH + c0, v);   // plain 1x1: Q row r = input row r
        }
      }
      if (ch + 1 < nchunks) load_x(chunk0 + ch + 1, q);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (DW) {
      if (one_utt) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int r = rq + 16 * q;
          float o[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = cst[5 * V2_C + c0 + i];
#pragma unroll
          for (int k = 0; k < KD; ++k) {
            float v[8];
            load8(Xa + (r + k) * V2_C + c0, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = fmaf(cst[(6 + k) * V2_C + c0 + i], v[i], o[i]);
          }
          store8(Qt + r * WG2_PITCH + c0, o);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int r = rq + 16 * q;
          const int t = (r0 + r) % T;
          float o[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = cst[5 * V2_C + c0 + i];
#pragma unroll
          for (int k = 0; k < KD; ++k) {
            const int tt = t + k - PADR;
            if (tt >= 0 && tt < T) {
              float v[8];
              load8(Xa + (r + k) * V2_C + c0, v);
#pragma unroll
              for (int i = 0; i < 8; ++i) o[i] = fmaf(cst[(6 + k) * V2_C + c0 + i], v[i], o[i]);
            }
          }
          store8(Qt + r * WG2_PITCH + c0, o);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    }
    // ---- contraction over the 32 rows: 2 k-steps x (4 x 2) MFMA tiles per wave
#pragma unroll
    for (int ks = 0; ks < WG2_RK / 16; ++ks) {
      bf16x8_t af[4], bfr[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = wg_frag_v2(Pt, ks * 16, wa * 128 + i * 32, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = wg_frag_v2(Qt, ks * 16, wb * 64 + j * 32, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
}

// ALLOW_DW = false: every layer's second operand is a stored tensor (the forward kept the depthwise outputs): the
// depthwise-recompute variants of the chunk loop are not even instantiated — they cost the kernel 100 spilled VGPRs
// (the register allocation of a kernel is the worst case over all its branches).
template <int KD, bool ALLOW_DW>
__global__ __launch_bounds__(V2_NT, 2) void wgrad_batched_v2_kernel(const WgradV2Desc* __restrict__ descs, int n_layers,
                                                                     int M, int T, int chunks_per_layer,
                                                                     int units_per_wg, int* __restrict__ part_count,
                                                                     uint64_t seed, float inv_n_rows) {
  // inv_n_rows: 1 / (rows the BatchNorm statistics ran over) — the descriptors are written once per plan with 1 / M, a
  // variable-length batch has fewer valid rows every step
  constexpr int PADR = (KD - 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Pt = reinterpret_cast<bf16_t*>(smem);                 // [32][288]
  bf16_t* Qt = Pt + WG2_RK * WG2_PITCH;                          // [32][288]
  bf16_t* Xa = Qt + WG2_RK * WG2_PITCH;                          // [32 + KD - 1][256]
  float* cst = reinterpret_cast<float*>(Xa + (WG2_RK + KD - 1) * V2_C);   // k0,k1,k2,sc,sh,bd,wd[KD] : [6 + KD][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vc = tid & 31, rq = tid >> 5, c0 = vc * 8;
  const int wa = wave >> 2, wb = wave & 3;
  // (partition by units, not bytes: a unit whose P operand is a stored tensor reads two tensors per chunk instead of three,
  //  but a workgroup's time per chunk does not shrink with it — weighting the ranges by bytes made the launch 30 % slower,
  //  round 4 — so every chunk counts the same)
  const int total_units = n_layers * chunks_per_layer;
  int unit = blockIdx.x * units_per_wg;
  const int unit_end = min(total_units, unit + units_per_wg);

  while (unit < unit_end) {
    const int layer = unit / chunks_per_layer;
    const int chunk0 = unit % chunks_per_layer;
    const int nchunks = min(chunks_per_layer - chunk0, unit_end - unit);
    const WgradV2Desc& d = descs[layer];        // read field by field (scalar loads): a by-value copy of the 200-byte
                                                // descriptor lived in scratch (the kernel's "93 spilled VGPRs")
    const bool dw = ALLOW_DW && d.wdw != nullptr;
    __syncthreads();
    if (tid < V2_C) {
      BnBwd bb;
      bb.fstats = d.fstats; bb.bsums = d.bsums; bb.gamma = d.gamma; bb.inv_n = inv_n_rows; bb.eps = d.eps; bb.batch = d.batch;
      float k0 = 1.f, k1 = 0.f, k2 = 0.f, s, h;
      if (d.fstats) bn_bwd_coefs(bb, d.statC, d.chan0 + tid, k0, k1, k2);
      {
        BnAct ax;
        ax.stats = d.actX.stats; ax.gamma = d.actX.gamma; ax.beta = d.actX.beta; ax.inv_n = inv_n_rows; ax.eps = d.actX.eps; ax.mode = d.actX.mode;
        ax.drop_thr = d.actX.drop_thr; ax.inv_keep = d.actX.inv_keep;
        bn_scale_shift(ax, d.ldq, d.q0 + tid, s, h);
      }
      cst[tid] = k0; cst[V2_C + tid] = k1; cst[2 * V2_C + tid] = k2; cst[3 * V2_C + tid] = s; cst[4 * V2_C + tid] = h;
      cst[5 * V2_C + tid] = dw ? d.bdw[tid] : 0.f;
#pragma unroll
      for (int k = 0; k < KD; ++k) cst[(6 + k) * V2_C + tid] = dw ? d.wdw[(size_t)tid * KD + k] : 0.f;
    }
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // the chunk loop is specialised on (depthwise?, activation flags of the layer input): no runtime flag tests
    // and no boundary tests inside it for interior / single-utterance chunks
    {
      const int fl = (d.actX.mode != 0 ? 1 : 0) | (d.actX.relu ? 2 : 0) | (d.actX.drop_thr ? 4 : 0);
      WgSeg sg{Pt, Qt, Xa, cst, chunk0, nchunks, M, T};
      if (ALLOW_DW && dw) {
        if constexpr (ALLOW_DW) {
          switch (fl) {
            case 7: wg2_chunks<KD, true, 7>(d, sg, acc); break;
            case 3: wg2_chunks<KD, true, 3>(d, sg, acc); break;
            case 0: wg2_chunks<KD, true, 0>(d, sg, acc); break;
            default: wg2_chunks<KD, true, -1>(d, sg, acc); break;
          }
        }
      } else {
        // operands of this launch: stored tensors used as they are (kept depthwise outputs, activated block inputs, the
        // attention hidden layer) or BatchNorm + ReLU of a raw tensor (prolog output, epilog output); a dropout-hashing
        // variant is not instantiated (it cost the whole kernel its register allocation)
        if (fl == 0) wg2_chunks<KD, false, 0>(d, sg, acc);
        else wg2_chunks<KD, false, 3>(d, sg, acc);
      }
    }
    // ---- partial slab of this (layer, segment)
    int part = 0;
    if (tid == 0) part = atomicAdd(&part_count[layer], 1);
    part = __shfl(part, 0, 64);
    __syncthreads();
    // broadcast `part` from wave 0 to the other waves through LDS (cst is free now)
    if (tid == 0) reinterpret_cast<int*>(cst)[0] = part;
    __syncthreads();
    part = __builtin_amdgcn_readfirstlane(reinterpret_cast<int*>(cst)[0]);      // (uniform for the compiler too: read from LDS the slab
                                                                                //  descriptor was "divergent" and each of the 128 stores below sat in a waterfall loop)
    // buffer stores: ONE per-lane offset register + a scalar offset per accumulator register (128 precomputed 64-bit
    // addresses were hoisted out of the unit loop and lived in scratch: the kernel's "93 spilled VGPRs")
    float* slab = d.slabs + (size_t)part * V2_C * V2_C;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(slab, 0, V2_C * V2_C * (int)sizeof(float), 0x00020000);
    const int voff = ((wa * 128 + 4 * (lane >> 5)) * V2_C + wb * 64 + (lane & 31)) * (int)sizeof(float);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), srd, voff,
                                                ((i * 32 + (r & 3) + 8 * (r >> 2)) * V2_C + j * 32) * (int)sizeof(float), 0);
    unit += nchunks;
  }
}

// sum the partial slabs of every layer into its gradient tensor (fixed order per layer is not
// guaranteed across runs — parts are claimed by arrival — but each sum has <= ~6 terms)
struct WgradV2Out {
  const float* slabs;
  float* out;        // first element of this unit's 256 x 256 block of the weight gradient
  int ld;            // its row stride (256, or the layer's input width for slabs of a wider weight)
  int lim;           // 0 = the whole 256 x 256 unit, else rows | cols << 16 actually wanted (a 128-wide operand read as 256)
};
__global__ void wgrad_v2_reduce_kernel(const WgradV2Out* __restrict__ outs, const int* __restrict__ part_count) {
  const WgradV2Out o = outs[blockIdx.y];
  const int parts = part_count[blockIdx.y];
  const int nrow = o.lim ? (o.lim & 0xffff) : V2_C, ncol = o.lim ? (o.lim >> 16) : V2_C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V2_C * V2_C; i += gridDim.x * blockDim.x) {
    if ((i >> 8) >= nrow || (i & 255) >= ncol) continue;
    float s = 0.f;
    for (int k0 = 0; k0 < parts; k0 += 4) {        // the partial slabs of an element fetched together, not one round trip each
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (k0 + u < parts) ? o.slabs[(size_t)(k0 + u) * V2_C * V2_C + i] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) s += v[u];
    }
    o.out[(size_t)(i >> 8) * o.ld + (i & 255)] = s;
  }
}

// 8-byte vector <-> 4 floats (4 channels per lane: the stencil layout of the fused data-gradient kernel)
__device__ __forceinline__ void unpack4(const uint2& a, float v[4]) {
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
}

// depthwise gradient accumulators [TN_NREP][KD + 1][256] of the fused data-gradient kernels -> the gradient buffer
struct DwGradOut {
  const float* gacc;   // [TN_NREP][KD + 1][256]
  float* g_wdw;        // [256][KD]
  float* g_bdw;        // [256]
};
__global__ void dw_grad_finalize_kernel(const DwGradOut* __restrict__ outs, int KD) {
  const DwGradOut o = outs[blockIdx.x];
  for (int i = threadIdx.x; i < (KD + 1) * V2_C; i += blockDim.x) {
    const int k = i / V2_C, c = i % V2_C;
    float v = 0.f;
    for (int r = 0; r < TN_NREP; ++r) v += o.gacc[(size_t)(r * (KD + 1) + k) * V2_C + c];
    if (k < KD) o.g_wdw[(size_t)c * KD + k] = v;
    else o.g_bdw[c] = v;
  }
}

template <int FL>
__device__ __forceinline__ void act4_t(float v[4], const float sc[4], const float sh[4], uint32_t key, uint32_t thr, uint32_t row, int c0) {
  if (FL & 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = fmaf(v[i], sc[i], sh[i]);
  }
  if (FL & 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
  }
  if (FL & 4) tn_drop4(v, (row * (uint32_t)V2_C + (uint32_t)c0) >> 3, (uint32_t)(c0 >> 2) & 1u, key, thr);
}

struct CombineBwd1V2Args {
  const bf16_t* dOUT; const float* gate;
  const bf16_t* Y3; BnAct act3;
  const bf16_t* S; BnAct actS;
  bf16_t* dZ;
  float* dgate;        // [B][256] (atomic accumulate when parts > 1: pre-zeroed by the caller then)
  float* bsumsS;
  int T, parts;
  float inv_keep; uint32_t drop_thr, drop_key; const uint32_t* key_add;
};
template <int FL3, bool DROP>
__global__ __launch_bounds__(512) void combine_bwd1_v2_kernel(CombineBwd1V2Args a) {
  __shared__ float cst[6 * V2_C];        // sc3, sh3, scS, shS, meanS, rstdS
  __shared__ float part[16][3][V2_C];
  const int tid = threadIdx.x, vc = tid & 31, tg = tid >> 5, c0 = vc * 8;
  const int b = blockIdx.x / a.parts, prt = blockIdx.x % a.parts;
  const int per = (a.T + a.parts - 1) / a.parts;
  const int t0 = prt * per, t1 = min(a.T, t0 + per);
  if (tid < V2_C) {
    float s = 1.f, h = 0.f, ss, hs, ms, rs;
    if (FL3 & 1) bn_scale_shift(a.act3, V2_C, tid, s, h);
    bn_scale_shift(a.actS, V2_C, tid, ss, hs);
    bn_mean_rstd(a.actS, V2_C, tid, ms, rs);
    cst[tid] = s; cst[V2_C + tid] = h; cst[2 * V2_C + tid] = ss; cst[3 * V2_C + tid] = hs; cst[4 * V2_C + tid] = ms; cst[5 * V2_C + tid] = rs;
  }
  __syncthreads();
  float k3[8], h3[8], kS[8], hS[8], m8[8], r8[8], g8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    k3[i] = cst[c0 + i]; h3[i] = cst[V2_C + c0 + i]; kS[i] = cst[2 * V2_C + c0 + i]; hS[i] = cst[3 * V2_C + c0 + i];
    m8[i] = cst[4 * V2_C + c0 + i]; r8[i] = cst[5 * V2_C + c0 + i]; g8[i] = a.gate[(size_t)b * V2_C + c0 + i];
  }
  const uint32_t dkey3 = tn_act_key(a.act3), dthr3 = a.act3.drop_thr;
  const uint32_t okey = a.key_add ? a.drop_key + *a.key_add : a.drop_key;
  float dg[8], s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { dg[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f; }
  constexpr int U = 4;
  for (int tb = t0 + tg; tb < t1; tb += 16 * U) {
    uint4 rd[U], ry[U], rs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = tb + 16 * u;
      if (t < t1) {
        const size_t o = ((size_t)b * a.T + t) * V2_C + c0;
        rd[u] = *reinterpret_cast<const uint4*>(a.dOUT + o);
        ry[u] = *reinterpret_cast<const uint4*>(a.Y3 + o);
        rs[u] = *reinterpret_cast<const uint4*>(a.S + o);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = tb + 16 * u;
      if (t < t1) {
        const uint32_t row = (uint32_t)b * a.T + t;
        float d[8], y[8], sv[8], m[8];
        unpack8(rd[u], d);
        unpack8(ry[u], y);
        unpack8(rs[u], sv);
        act8_t<FL3>(y, k3, h3, dkey3, dthr3, row, c0);
#pragma unroll
        for (int i = 0; i < 8; ++i) m[i] = (fmaf(sv[i], kS[i], fmaf(g8[i], y[i], hS[i])) > 0.f) ? a.inv_keep : 0.f;
        if (DROP) tn_drop8(m, (row * (uint32_t)V2_C + (uint32_t)c0) >> 3, okey, a.drop_thr);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float z = d[i] * m[i];
          d[i] = z;
          dg[i] = fmaf(z, y[i], dg[i]);
          s1[i] += z;
          s2[i] = fmaf(z, (sv[i] - m8[i]) * r8[i], s2[i]);
        }
        store8(a.dZ + (size_t)row * V2_C + c0, d);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { part[tg][0][c0 + i] = dg[i]; part[tg][1][c0 + i] = s1[i]; part[tg][2][c0 + i] = s2[i]; }
  __syncthreads();
  const int rep = blockIdx.x % TN_NREP;
  for (int i = tid; i < 3 * V2_C; i += 512) {
    const int which = i / V2_C, c = i % V2_C;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += part[k][which][c];
    if (which == 0) {
      if (a.parts > 1) atomic_add_f32(&a.dgate[(size_t)b * V2_C + c], v);
      else a.dgate[(size_t)b * V2_C + c] = v;
    } else {
      atomic_add_f32(&a.bsumsS[(size_t)(rep * 2 + which - 1) * V2_C + c], v);
    }
  }
}
// -1000: no specialisation for this flag combination
inline int launch_combine_bwd1_v2(const CombineBwd1V2Args& a, int B, hipStream_t st) {
  const int fl3 = (a.act3.mode != 0 ? 1 : 0) | (a.act3.relu ? 2 : 0) | (a.act3.drop_thr ? 4 : 0);
  if (a.act3.rm.len) return -1000;
  const dim3 grid(B * a.parts), blk(512);
  if (fl3 == 7 && a.drop_thr) hipLaunchKernelGGL((combine_bwd1_v2_kernel<7, true>), grid, blk, 0, st, a);
  else if (fl3 == 3 && !a.drop_thr) hipLaunchKernelGGL((combine_bwd1_v2_kernel<3, false>), grid, blk, 0, st, a);
  else return -1000;
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// combine_bwd1_v3 (round 4): pass 1 of the mega-block tail backward that makes pass 2 unnecessary.
//
// Pass 2 (combine_bwd2_v2) existed to materialise  dY3bn[b,t,c] = (dZ[b,t,c] g[b,c] + dmean[b,c] / T) m3[b,t,c]  (m3 = d act3 /
// d bn3: ReLU sign, dropout keep, 1/(1-p)) and to take its BatchNorm-backward sums: 3 tensor passes per mega block.  But dY3bn
// is a function of dZ (written here), Y3 (read by every consumer anyway) and two numbers per (utterance, channel), and its
// sums over the rows are linear in four per-(utterance, channel) sums that THIS pass can take while it holds dZ and Y3:
//   B1 = sum_t m z,  B2 = sum_t m z y,  B3 = sum_t m,  B4 = sum_t m y      (m = [act3(y) > 0], y raw)
//   dgate = sc3 B2 + sh3 B1;   sum_t dY3bn = ga B1 + ub B3;   sum_t dY3bn y = ga B2 + ub B4     (ga = g/(1-p), ub = dmean/T/(1-p))
// So the workgroup of an utterance finishes the SE backward (reference src/modules.py:182-189) itself, publishes ga / ub
// ([B][2][256] floats) and adds the utterance's share to the BatchNorm-backward sums of sub-block 3; the sub-block's fused
// data-gradient kernel (dgrad_dw_v6<.., Z3>) rebuilds dY3bn on load from dZ, Y3, ga, ub.  One workgroup per utterance.
// ------------------------------------------------------------------------------------------
struct CombineBwd1V3Args {
  CombineBwd1V2Args a1;      // (dgate unused; parts = workgroups per utterance)
  const float* hid;          // [B][CH / 16] SE hidden layer (forward)
  const float* W1;           // [CH / 16][CH]
  const float* W2;           // [CH][CH / 16]
  float* dpre2; float* dpre1;
  float* gu;                 // [B][2][CH]: ga, ub
  float* bsums3;
  const int* len;            // valid frames per utterance (variable-length batch) or null: padding rows carry no gradient,
                             // the sums run over the valid rows, the SE mean ran over them
  float* bacc;               // parts > 1: [B][4][CH] accumulators of B1..B4 (zero on entry, left zero by the tail kernel)
};
// The per-utterance end of the pass: SE backward from B1..B4 of this thread's channels (B[q]: channel tid + 512 q), ga / ub,
// the utterance's share of the BatchNorm-backward sums of the last sub-block.  cst: sc3, sh3, .., mean3*rstd3 (6), rstd3 (7).
template <int FL3, int CH, bool PRE>
__device__ __forceinline__ void cb3_tail(const CombineBwd1V3Args& aa, int b, int L, const float* cst, float* p2, float* p1,
                                         float (&B)[(CH + 511) / 512][4], const float (*w2r)[4], const float* hidr, const float* w1r) {
  constexpr int HR = CH / 16, PT = (CH + 511) / 512;
  const CombineBwd1V2Args& a = aa.a1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rep = b % TN_NREP;
  float gg[PT];
#pragma unroll
  for (int q = 0; q < PT; ++q) {
    const int c = tid + 512 * q;
    gg[q] = 0.f;
    if (c < CH) {
      const float g = a.gate[(size_t)b * CH + c];
      gg[q] = g;
      const float dgate = fmaf(cst[c], B[q][1], cst[CH + c] * B[q][0]);
      const float d2 = dgate * g * (1.f - g);
      p2[c] = d2;
      aa.dpre2[(size_t)b * CH + c] = d2;
    }
  }
  __syncthreads();
  // p1[j] = relu'(hid[j]) * sum_c W2[c][j] p2[c]  (HR / 8 outputs per wave)
#pragma unroll
  for (int jj = 0; jj < HR / 8; ++jj) {
    const int j = wave + 8 * jj;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CH / 64; ++k) s = fmaf(PRE ? w2r[PRE ? jj : 0][PRE ? k : 0] : aa.W2[(size_t)(lane + 64 * k) * HR + j], p2[lane + 64 * k], s);
    s = wave_sum(s);
    if (lane == 0) {
      s = ((PRE ? hidr[PRE ? jj : 0] : aa.hid[(size_t)b * HR + j]) > 0.f) ? s : 0.f;
      p1[j] = s;
      aa.dpre1[(size_t)b * HR + j] = s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PT; ++q) {
    const int c = tid + 512 * q;
    if (c < CH) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < HR; ++j) s = fmaf(PRE ? w1r[PRE ? j : 0] : aa.W1[(size_t)j * CH + c], p1[j], s);
      const float on = (FL3 & 4) ? a.act3.inv_keep : 1.f;
      const float ga = gg[q] * on, ub = s / (float)max(L, 1) * on;
      aa.gu[((size_t)b * 2 + 0) * CH + c] = ga;
      aa.gu[((size_t)b * 2 + 1) * CH + c] = ub;
      const float sv = fmaf(ga, B[q][0], ub * B[q][2]), sy = fmaf(ga, B[q][1], ub * B[q][3]);
      atomic_add_f32(&aa.bsums3[(size_t)(rep * 2 + 0) * CH + c], sv);
      atomic_add_f32(&aa.bsums3[(size_t)(rep * 2 + 1) * CH + c], cst[7 * CH + c] * sy - cst[6 * CH + c] * sv);
    }
  }
}
// CH = hidden width (256: TitaNet-S; 512 / 1024: the wide models, whose last sub-block rebuilds the gradient in
// bn_bwd_apply_z3_kernel).  512 threads = (CH / 8 channel vectors) x TG row groups; dynamic LDS: 8 CH constants + [NS][6][CH] partial sums.
// a1.parts workgroups per utterance (small batches of long utterances): each takes a range of frames, B1..B4 meet in aa.bacc
// and combine_bwd1_v3_tail_kernel finishes the utterance; parts == 1: the workgroup does it itself.
template <int FL3, bool DROP, int CH>
__global__ __launch_bounds__(512) void combine_bwd1_v3_kernel(CombineBwd1V3Args aa) {
  constexpr int HR = CH / 16, CV = CH / 8, TG = 512 / CV, NS = CH == 256 ? 8 : TG, PT = (CH + 511) / 512;
  const CombineBwd1V2Args& a = aa.a1;
  extern __shared__ __attribute__((aligned(16))) float cb3_smem[];
  float* cst = cb3_smem;                 // sc3, sh3, scS, shS, meanS, rstdS, mean3*rstd3, rstd3 : [8][CH]
  float* part = cst + 8 * CH;            // [NS][6][CH]: B1..B4, skip sums
  float* p2 = part + NS * 6 * CH;        // [CH]
  float* p1 = p2 + CH;                   // [HR]
  const int tid = threadIdx.x, vc = tid % CV, tg = tid / CV, c0 = vc * 8;
  const int lane = tid & 63, wave = tid >> 6;
  const int parts = a.parts, b = blockIdx.x / parts, prt = blockIdx.x - b * parts;
  const int L = aa.len ? aa.len[b] : a.T;
  const int per = (a.T + parts - 1) / parts;
  const int t_lo = prt * per, t_end = min(a.T, t_lo + per), t_hi = min(L, t_end);
  for (int c = tid; c < CH; c += 512) {
    float s = 1.f, h = 0.f, ss, hs, ms, rs, m3 = 0.f, r3 = 1.f;
    if (FL3 & 1) { bn_scale_shift(a.act3, CH, c, s, h); bn_mean_rstd(a.act3, CH, c, m3, r3); }
    bn_scale_shift(a.actS, CH, c, ss, hs);
    bn_mean_rstd(a.actS, CH, c, ms, rs);
    cst[c] = s; cst[CH + c] = h; cst[2 * CH + c] = ss; cst[3 * CH + c] = hs; cst[4 * CH + c] = ms; cst[5 * CH + c] = rs;
    cst[6 * CH + c] = m3 * r3; cst[7 * CH + c] = r3;
  }
  __syncthreads();
  float k3[8], h3[8], kS[8], hS[8], m8[8], r8[8], g8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    k3[i] = cst[c0 + i]; h3[i] = cst[CH + c0 + i]; kS[i] = cst[2 * CH + c0 + i]; hS[i] = cst[3 * CH + c0 + i];
    m8[i] = cst[4 * CH + c0 + i]; r8[i] = cst[5 * CH + c0 + i]; g8[i] = a.gate[(size_t)b * CH + c0 + i];
  }
  const uint32_t dkey3 = tn_act_key(a.act3), dthr3 = a.act3.drop_thr;
  const uint32_t okey = a.key_add ? a.drop_key + *a.key_add : a.drop_key;
  // operands of the SE backward at the end (they do not depend on the row loop): requested now (hidden 256), so that the
  // serial tail of the workgroup — every workgroup reaches it at the same time — does not wait for dependent global round trips
  constexpr bool PRE = CH == 256;
  float w2r[PRE ? 2 : 1][4], hidr[2], w1r[PRE ? HR : 1];
  if (PRE) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = wave + 8 * jj;
#pragma unroll
      for (int k = 0; k < 4; ++k) w2r[PRE ? jj : 0][k] = aa.W2[(size_t)(lane + 64 * k) * HR + j];
      hidr[jj] = aa.hid[(size_t)b * HR + j];
    }
#pragma unroll
    for (int j = 0; j < (PRE ? HR : 1); ++j) w1r[j] = aa.W1[(size_t)j * CH + (tid & 255)];
  }
  float b1[8], b2[8], b3[8], b4[8], s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { b1[i] = 0.f; b2[i] = 0.f; b3[i] = 0.f; b4[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f; }
  constexpr int U = 4;
  for (int tb = t_lo + tg; tb < t_hi; tb += TG * U) {
    uint4 rd[U], ry[U], rs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = tb + TG * u;
      if (t < t_hi) {
        const size_t o = ((size_t)b * a.T + t) * CH + c0;
        rd[u] = *reinterpret_cast<const uint4*>(a.dOUT + o);
        ry[u] = *reinterpret_cast<const uint4*>(a.Y3 + o);
        rs[u] = *reinterpret_cast<const uint4*>(a.S + o);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = tb + TG * u;
      if (t < t_hi) {
        const uint32_t row = (uint32_t)b * a.T + t;
        float d[8], y[8], ya[8], sv[8], m[8];
        unpack8(rd[u], d);
        unpack8(ry[u], y);
        unpack8(rs[u], sv);
#pragma unroll
        for (int i = 0; i < 8; ++i) ya[i] = y[i];
        if (FL3 & 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) ya[i] = fmaf(ya[i], k3[i], h3[i]);
        }
        if (FL3 & 2) {
#pragma unroll
          for (int i = 0; i < 8; ++i) ya[i] = fmaxf(ya[i], 0.f);
        }
        if (FL3 & 4) tn_drop8(ya, (row * (uint32_t)CH + (uint32_t)c0) >> 3, dkey3, dthr3);
#pragma unroll
        for (int i = 0; i < 8; ++i) m[i] = (fmaf(sv[i], kS[i], fmaf(g8[i], ya[i], hS[i])) > 0.f) ? a.inv_keep : 0.f;
        if (DROP) tn_drop8(m, (row * (uint32_t)CH + (uint32_t)c0) >> 3, okey, a.drop_thr);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float z = d[i] * m[i];
          d[i] = z;
          s1[i] += z;
          s2[i] = fmaf(z, (sv[i] - m8[i]) * r8[i], s2[i]);
          const bool on3 = ya[i] > 0.f;              // ReLU passed and the element was kept (sc3 / sh3 carry 1/(1-p) > 0)
          const float zm = on3 ? z : 0.f, ym = on3 ? y[i] : 0.f;
          b1[i] += zm;
          b2[i] = fmaf(zm, y[i], b2[i]);
          b3[i] += on3 ? 1.f : 0.f;
          b4[i] += ym;
        }
        store8(a.dZ + (size_t)row * CH + c0, d);
      }
    }
  }
  // padding frames of this workgroup's range: no gradient (written, not read)
  if (aa.len) {
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    for (int t = max(t_lo, L) + tg; t < t_end; t += TG) *reinterpret_cast<uint4*>(a.dZ + ((size_t)b * a.T + t) * CH + c0) = z4;
  }
  // hidden 256: the two row groups of a wave (lanes l, l ^ 32) first; then the row groups through LDS
  if (CH == 256) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      b1[i] += __shfl_xor(b1[i], 32, 64); b2[i] += __shfl_xor(b2[i], 32, 64); b3[i] += __shfl_xor(b3[i], 32, 64);
      b4[i] += __shfl_xor(b4[i], 32, 64); s1[i] += __shfl_xor(s1[i], 32, 64); s2[i] += __shfl_xor(s2[i], 32, 64);
    }
  }
  if (CH != 256 || lane < 32) {
    float* mine = part + (size_t)(CH == 256 ? wave : tg) * 6 * CH + c0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      mine[i] = b1[i]; mine[CH + i] = b2[i]; mine[2 * CH + i] = b3[i];
      mine[3 * CH + i] = b4[i]; mine[4 * CH + i] = s1[i]; mine[5 * CH + i] = s2[i];
    }
  }
  __syncthreads();
  const int rep = b % TN_NREP;
  float B[PT][4];
#pragma unroll
  for (int q = 0; q < PT; ++q) {
    const int c = tid + 512 * q;
    B[q][0] = B[q][1] = B[q][2] = B[q][3] = 0.f;
    if (c < CH) {
      float v1 = 0.f, v2 = 0.f;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        const float* pk = part + (size_t)k * 6 * CH + c;
        B[q][0] += pk[0]; B[q][1] += pk[CH]; B[q][2] += pk[2 * CH]; B[q][3] += pk[3 * CH]; v1 += pk[4 * CH]; v2 += pk[5 * CH];
      }
      atomic_add_f32(&a.bsumsS[(size_t)(rep * 2 + 0) * CH + c], v1);
      atomic_add_f32(&a.bsumsS[(size_t)(rep * 2 + 1) * CH + c], v2);
      if (parts > 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) atomic_add_f32(&aa.bacc[((size_t)b * 4 + k) * CH + c], B[q][k]);
      }
    }
  }
  if (parts > 1) return;                 // workgroup-uniform: combine_bwd1_v3_tail_kernel finishes the utterance
  cb3_tail<FL3, CH, PRE>(aa, b, L, cst, p2, p1, B, w2r, hidr, w1r);
}
// the per-utterance end of a pass that ran as several workgroups per utterance (reads B1..B4 from aa.bacc and zeroes it again)
template <int FL3, int CH>
__global__ __launch_bounds__(512) void combine_bwd1_v3_tail_kernel(CombineBwd1V3Args aa) {
  constexpr int HR = CH / 16, PT = (CH + 511) / 512;
  const CombineBwd1V2Args& a = aa.a1;
  __shared__ float cst[8 * CH];          // only sc3, sh3 (0, 1) and mean3*rstd3, rstd3 (6, 7) are used
  __shared__ float p2[CH], p1[HR];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int L = aa.len ? aa.len[b] : a.T;
  float B[PT][4];
#pragma unroll
  for (int q = 0; q < PT; ++q) {
    const int c = tid + 512 * q;
    B[q][0] = B[q][1] = B[q][2] = B[q][3] = 0.f;
    if (c < CH) {
      float s = 1.f, h = 0.f, m3 = 0.f, r3 = 1.f;
      if (FL3 & 1) { bn_scale_shift(a.act3, CH, c, s, h); bn_mean_rstd(a.act3, CH, c, m3, r3); }
      cst[c] = s; cst[CH + c] = h; cst[6 * CH + c] = m3 * r3; cst[7 * CH + c] = r3;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float* pa = &aa.bacc[((size_t)b * 4 + k) * CH + c];
        B[q][k] = *pa;
        *pa = 0.f;
      }
    }
  }
  __syncthreads();
  float w2d[1][4] = {{0.f, 0.f, 0.f, 0.f}}, hd[2] = {0.f, 0.f}, w1d[1] = {0.f};
  cb3_tail<FL3, CH, false>(aa, b, L, cst, p2, p1, B, w2d, hd, w1d);
}
// -1000: no specialisation for this flag combination
template <int CH>
inline int launch_combine_bwd1_v3_c(const CombineBwd1V3Args& aa, int B, hipStream_t st) {
  const CombineBwd1V2Args& a = aa.a1;
  const int fl3 = (a.act3.mode != 0 ? 1 : 0) | (a.act3.relu ? 2 : 0) | (a.act3.drop_thr ? 4 : 0);
  if (a.parts < 1 || (a.parts > 1 && !aa.bacc)) return -1000;
  constexpr int CV = CH / 8, TG = 512 / CV, NS = CH == 256 ? 8 : TG;
  const size_t smem = (size_t)(8 * CH + NS * 6 * CH + CH + CH / 16) * sizeof(float);
  const dim3 grid(B * a.parts), blk(512);
  if (fl3 == 7 && a.drop_thr) {
    auto kern = combine_bwd1_v3_kernel<7, true, CH>;
    TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, grid, blk, smem, st, aa);
    if (a.parts > 1) hipLaunchKernelGGL((combine_bwd1_v3_tail_kernel<7, CH>), dim3(B), blk, 0, st, aa);
  } else if (fl3 == 3 && !a.drop_thr) {
    auto kern = combine_bwd1_v3_kernel<3, false, CH>;
    TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, grid, blk, smem, st, aa);
    if (a.parts > 1) hipLaunchKernelGGL((combine_bwd1_v3_tail_kernel<3, CH>), dim3(B), blk, 0, st, aa);
  } else return -1000;
  return (int)hipGetLastError();
}
inline int launch_combine_bwd1_v3(const CombineBwd1V3Args& aa, int B, int C, hipStream_t st) {
  switch (C) {
    case 256: return launch_combine_bwd1_v3_c<256>(aa, B, st);
    case 512: return launch_combine_bwd1_v3_c<512>(aa, B, st);
    case 1024: return launch_combine_bwd1_v3_c<1024>(aa, B, st);
    default: return -1000;
  }
}

// dS = BatchNorm-backward(dYbn, Y) with dYbn = (dZ ga[b] + ub[b]) [act3(Y) > 0] rebuilt on the fly from the tail's dZ
// (combine_bwd1_v3): the wide models' form of dgrad_dw_v6<.., Z3> — the streaming pass that makes the stored dS operand of
// the two pipelined GEMMs reads the tail's gradient directly, and the second tail pass (3 tensor passes per mega block) is gone.
// One workgroup = (utterance, 64-frame chunk); a thread keeps the constants of its 8 channels — k0 k1 k2, sc3 sh3 and the
// utterance's ga / ub — in registers and walks the chunk's rows (a flat grid-stride loop re-read ga / ub, 64 bytes of
// float32 per 16 bytes of gradient, for every vector: 79 us against 47 for the plain pass at hidden 512).
// OUT8 (fp8 plans): dS also as e4m3 bytes with one power-of-two scale per row + the exponent bytes (Fp8Rows, tn_common.h)
// NV (round 5): 8-channel vectors per lane.  NV = 2 at 1024 channels with fp8 outputs: ONE wave then holds a whole row (lane l:
// channels 8 l .. and 512 + 8 l ..), so the row maximum of the e4m3 row scale is a wave reduction — with one vector per lane a
// row is two waves and every U rows cost an LDS exchange between two workgroup barriers (200 us per launch against 116 for the
// bf16-only pass).
template <bool DROP3, int CH, bool OUT8, int NV = 1>
__global__ __launch_bounds__(256) void bn_bwd_apply_z3_kernel(const bf16_t* __restrict__ dZ, const bf16_t* __restrict__ Y, BnBwd bn, BnAct act3,
                                                              const float* __restrict__ gu, bf16_t* __restrict__ dS, int T, int chunk,
                                                              const int* __restrict__ len, Fp8Rows f8, Fp8Cols fc) {
  constexpr int VC = CH / 8 / NV, RG = 256 / VC;       // lanes per row, row groups per workgroup
  constexpr int VS = VC * 8;                            // channel distance between a lane's vectors
  constexpr bool XWAVE = OUT8 && VC > 64;               // a row spans two waves: row maxima through LDS
  __shared__ __attribute__((aligned(16))) float pg_k[5 * CH];      // k0, k1, k2, sc3, sh3
  __shared__ float wmax[4][4];
  __shared__ float cmax[OUT8 ? 256 * 8 * NV : 1];      // per-thread column maxima (fp8 weight gradient, Fp8Cols)
  for (int c = threadIdx.x; c < CH; c += 256) {
    bn_bwd_coefs(bn, CH, c, pg_k[c], pg_k[CH + c], pg_k[2 * CH + c]);
    bn_scale_shift(act3, CH, c, pg_k[3 * CH + c], pg_k[4 * CH + c]);
  }
  __syncthreads();
  const uint32_t dkey3 = tn_act_key(act3), dthr3 = act3.drop_thr;
  const int b = blockIdx.y, vc = threadIdx.x % VC, rg = threadIdx.x / VC, c0 = vc * 8;      // vector v of the lane: channels c0 + v VS ..
  float k0[NV][8], k1[NV][8], k2[NV][8], s3[NV][8], h3[NV][8], ga[NV][8], ub[NV][8];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + v * VS + u;
      k0[v][u] = pg_k[c]; k1[v][u] = pg_k[CH + c]; k2[v][u] = pg_k[2 * CH + c]; s3[v][u] = pg_k[3 * CH + c]; h3[v][u] = pg_k[4 * CH + c];
    }
    load8(gu + (size_t)b * 2 * CH + c0 + v * VS, ga[v]);
    load8(gu + (size_t)b * 2 * CH + CH + c0 + v * VS, ub[v]);
  }
  const bool cols = OUT8 && fc.q != nullptr;
  // column scales of the fp8 weight gradient: in LDS (16 more registers per lane would spill the NV = 2 form)
  __shared__ __attribute__((aligned(16))) float csc_l[OUT8 ? CH : 1];
  float cmx[NV][8];
  if (OUT8) {
    for (int c = threadIdx.x; c < CH; c += 256) {
      const float sc_ = cols ? tn_e4m3_col_scale(fc.amax_prev[c]) : 1.f;
      csc_l[c] = sc_;
      if (cols && blockIdx.x == 0 && blockIdx.y == 0) fc.cexp[c] = (uint8_t)(254u - ((__float_as_uint(sc_) >> 23) & 0xffu));      // E8M0 of 1 / scale
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int u = 0; u < 8; ++u) cmx[v][u] = 0.f;
  }
  const bool keep16 = !(OUT8 && cols && fc.skip_bf16);      // (both fp8 copies are the only ones read: no bf16 dS)
  const int t0 = blockIdx.x * chunk, t_end = min(T, t0 + chunk);
  const int t1 = len ? min(t_end, len[b]) : t_end;       // padding frames of a variable-length batch: dS = 0 (written below)
  if (len) {
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    for (int t = max(t0, t1) + rg; t < t_end; t += RG) {
      const size_t row = (size_t)b * T + t;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (keep16) *reinterpret_cast<uint4*>(dS + row * CH + c0 + v * VS) = z4;
        if (OUT8) {
          *reinterpret_cast<uint2*>(f8.q + row * CH + c0 + v * VS) = make_uint2(0u, 0u);
          if (cols) *reinterpret_cast<uint2*>(fc.q + row * CH + c0 + v * VS) = make_uint2(0u, 0u);
        }
      }
      if (OUT8 && c0 == 0) f8.rowexp[tn_rowexp_pos(row)] = (uint8_t)127;
    }
  }
  constexpr int U = NV == 1 ? 4 : 2;
  // (tb is uniform over the workgroup: with XWAVE the row maximum of the fp8 output crosses two waves)
  for (int tb = t0; tb < t1; tb += RG * U) {
    uint4 rz[U][NV], ry[U][NV];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int t = tb + rg + RG * q;
      if (t < t1) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const size_t o = ((size_t)b * T + t) * CH + c0 + v * VS;
          rz[q][v] = *reinterpret_cast<const uint4*>(dZ + o);
          ry[q][v] = *reinterpret_cast<const uint4*>(Y + o);
        }
      }
    }
    float zq[OUT8 ? U : 1][NV][8], mxq[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int t = tb + rg + RG * q;
      const bool live = t < t1;
      const uint32_t row = (uint32_t)b * T + t;
      float mx = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float z[8];
        if (live) {
          float y[8], m[8];
          unpack8(rz[q][v], z);
          unpack8(ry[q][v], y);
#pragma unroll
          for (int u = 0; u < 8; ++u) m[u] = (fmaf(y[u], s3[v][u], h3[v][u]) > 0.f) ? 1.f : 0.f;
          if (DROP3) tn_drop8(m, (row * (uint32_t)CH + (uint32_t)(c0 + v * VS)) >> 3, dkey3, dthr3);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float w = fmaf(z[u], ga[v][u], ub[v][u]) * m[u];
            z[u] = fmaf(k0[v][u], w, fmaf(k1[v][u], y[u], k2[v][u]));
          }
          if (keep16) store8(dS + (size_t)row * CH + c0 + v * VS, z);
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) z[u] = 0.f;
        }
        if (OUT8) {
#pragma unroll
          for (int u = 0; u < 8; ++u) { mx = fmaxf(mx, fabsf(z[u])); zq[q][v][u] = z[u]; }
        }
      }
      if (OUT8) mxq[q] = wave_max(mx);
    }
    if (OUT8) {
      if (XWAVE) {
        // row maxima of the U rows across the two waves of a row: ONE exchange for all of them
        const int w = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < U; ++q) wmax[w][q] = mxq[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < U; ++q) mxq[q] = fmaxf(mxq[q], wmax[w ^ 1][q]);
        __syncthreads();
      }
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int t = tb + rg + RG * q;
        if (t < t1) {
          const size_t row = (size_t)b * T + t;
          const float sc = tn_e4m3_row_scale(mxq[q]);
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            *reinterpret_cast<uint2*>(f8.q + row * CH + c0 + v * VS) = tn_e4m3_pack8(zq[q][v], 1.f / sc);
            if (cols) {
#pragma unroll
              for (int u = 0; u < 8; ++u) cmx[v][u] = fmaxf(cmx[v][u], fabsf(zq[q][v][u]));
              float cs8[8];
              *reinterpret_cast<float4*>(cs8) = *reinterpret_cast<const float4*>(csc_l + c0 + v * VS);
              *reinterpret_cast<float4*>(cs8 + 4) = *reinterpret_cast<const float4*>(csc_l + c0 + v * VS + 4);
              *reinterpret_cast<uint2*>(fc.q + row * CH + c0 + v * VS) = tn_e4m3_pack8_cols(zq[q][v], cs8);
            }
          }
          if (c0 == 0) f8.rowexp[tn_rowexp_pos(row)] = tn_e8m0_of_pow2(sc);
        }
      }
    }
  }
  if (OUT8) {
    if (cols) {      // workgroup-uniform
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int u = 0; u < 8; ++u) cmax[(threadIdx.x * NV + v) * 8 + u] = cmx[v][u];
      __syncthreads();
      // column c = vector v (c / VS) of lane (c % VS) / 8, held by the threads k VC + that lane, k < RG
      for (int c = threadIdx.x; c < CH; c += 256) {
        const int v = c / VS, ln = (c % VS) / 8;
        float m = 0.f;
        for (int k = 0; k < RG; ++k) m = fmaxf(m, cmax[((k * VC + ln) * NV + v) * 8 + (c & 7)]);
        if (m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(fc.amax_cur) + c, __float_as_uint(m));
      }
    }
  }
}
template <int CH>
inline int launch_bn_bwd_apply_z3_c(const bf16_t* dZ, const bf16_t* Y, const BnBwd& bn, const BnAct& act3, const float* gu, bf16_t* dS, int M,
                                    int T, hipStream_t st, Fp8Rows f8, Fp8Cols fc) {
  const int chunk = 64;
  const dim3 grid((T + chunk - 1) / chunk, M / T);
  const bool d3 = act3.drop_thr != 0;
#define TN_Z3(D, O) hipLaunchKernelGGL((bn_bwd_apply_z3_kernel<D, CH, O, ((O) && CH == 1024) ? 2 : 1>), grid, dim3(256), 0, st, dZ, Y, bn, act3, gu, dS, T, chunk, bn.rm.len, f8, fc)
  if (f8.q) { if (d3) TN_Z3(true, true); else TN_Z3(false, true); }
  else { if (d3) TN_Z3(true, false); else TN_Z3(false, false); }
#undef TN_Z3
  return (int)hipGetLastError();
}
inline int launch_bn_bwd_apply_z3(const bf16_t* dZ, const bf16_t* Y, const BnBwd& bn, const BnAct& act3, const float* gu, bf16_t* dS, int M,
                                  int C, int T, hipStream_t st, Fp8Rows f8 = Fp8Rows{nullptr, nullptr},
                                  Fp8Cols fc = Fp8Cols{nullptr, nullptr, nullptr, nullptr, 0}) {
  if (act3.mode == 0 || !act3.relu || M % T) return TN_E_UNSUPPORTED;
  if (!f8.q) fc.q = nullptr;
  if (C == 512) return launch_bn_bwd_apply_z3_c<512>(dZ, Y, bn, act3, gu, dS, M, T, st, f8, fc);
  if (C == 1024) return launch_bn_bwd_apply_z3_c<1024>(dZ, Y, bn, act3, gu, dS, M, T, st, f8, fc);
  return TN_E_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------
// combine_bwd2_v2: mega-block tail backward, pass 2, for hidden = 256 / bf16 / Hr = 16: SE backward (tiny
// mat-vecs, reference src/modules.py:182-189), then  dA3 = dZ * g + dmean / T ;  dY3bn = dA3 * d act3 / d bn
// -> stored + BN3 backward sums.  `parts` workgroups per utterance (each repeats the tiny SE backward and streams
// its share of the rows: 16 waves per CU overlap each other's serial prologue with streaming); per-thread
// constants in registers; the first rows are already in flight while the prologue runs.
// dgate comes from pass 1 in its own buffer (pass 2 of part 0 writes dpre2 / dpre1 for the SE weight gradients).
// ------------------------------------------------------------------------------------------
struct CombineBwd2V2Args {
  const bf16_t* dZ; const bf16_t* Y3; BnAct act3;
  const float* gate; const float* hid; const float* dgate;
  float* dpre2; float* dpre1;
  const float* W1;    // [16][256]
  const float* W2;    // [256][16]
  bf16_t* dYbn;
  float* bsums3;
  int T, parts;
  const int* len;     // valid frames per utterance or null: the SE mean ran over them; padding rows get a zero gradient
};
template <int FL3>
__global__ __launch_bounds__(512) void combine_bwd2_v2_kernel(CombineBwd2V2Args a) {
  constexpr int HR = 16;
  __shared__ float cst[4 * V2_C];       // sc3, sh3, mean3*rstd3, rstd3
  __shared__ float gS[V2_C], dmT[V2_C], p2[V2_C], p1[HR];
  __shared__ float part[16][2][V2_C];
  const int tid = threadIdx.x, vc = tid & 31, tg = tid >> 5, c0 = vc * 8;
  const int lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / a.parts, prt = blockIdx.x % a.parts;
  const int per = (a.T + a.parts - 1) / a.parts;
  const int L = a.len ? a.len[b] : a.T;
  const int t0 = prt * per, t_end = min(a.T, t0 + per), t1 = min(L, t_end);
  {
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int t = max(t0, L) + tg; t < t_end; t += 16) *reinterpret_cast<uint4*>(a.dYbn + ((size_t)b * a.T + t) * V2_C + c0) = z;
  }
  constexpr int U = 2;
  uint4 rd[U], ry[U];
  auto fetch = [&](int tb) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = tb + 16 * u;
      if (t < t1) {
        const size_t o = ((size_t)b * a.T + t) * V2_C + c0;
        rd[u] = *reinterpret_cast<const uint4*>(a.dZ + o);
        ry[u] = *reinterpret_cast<const uint4*>(a.Y3 + o);
      }
    }
  };
  fetch(t0 + tg);                        // in flight during the SE-backward prologue
  if (tid < V2_C) {
    float s = 1.f, h = 0.f, mean = 0.f, rstd = 1.f;
    if (FL3 & 1) { bn_scale_shift(a.act3, V2_C, tid, s, h); bn_mean_rstd(a.act3, V2_C, tid, mean, rstd); }
    cst[tid] = s; cst[V2_C + tid] = h; cst[2 * V2_C + tid] = mean * rstd; cst[3 * V2_C + tid] = rstd;
    const float g = a.gate[(size_t)b * V2_C + tid];
    gS[tid] = g;
    const float d2 = a.dgate[(size_t)b * V2_C + tid] * g * (1.f - g);
    p2[tid] = d2;
    if (prt == 0) a.dpre2[(size_t)b * V2_C + tid] = d2;
  }
  __syncthreads();
  // p1[j] = relu'(hid[j]) * sum_c W2[c][j] p2[c]  (2 outputs per wave)
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int j = wave + 8 * jj;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s = fmaf(a.W2[(size_t)(lane + 64 * k) * HR + j], p2[lane + 64 * k], s);
    s = wave_sum(s);
    if (lane == 0) {
      s = (a.hid[(size_t)b * HR + j] > 0.f) ? s : 0.f;
      p1[j] = s;
      if (prt == 0) a.dpre1[(size_t)b * HR + j] = s;
    }
  }
  __syncthreads();
  if (tid < V2_C) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < HR; ++j) s = fmaf(a.W1[(size_t)j * V2_C + tid], p1[j], s);
    dmT[tid] = s / (float)max(L, 1);
  }
  __syncthreads();
  float sc[8], sh[8], g[8], dm[8], s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = cst[c0 + i]; sh[i] = cst[V2_C + c0 + i]; g[i] = gS[c0 + i]; dm[i] = dmT[c0 + i]; s1[i] = 0.f; s2[i] = 0.f;
  }
  const float on = (FL3 & 4) ? a.act3.inv_keep : 1.f;
  const uint32_t dkey = tn_act_key(a.act3), dthr = a.act3.drop_thr;
  for (int tb = t0 + tg; tb < t1; tb += 16 * U) {
    uint4 cd[U], cy[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { cd[u] = rd[u]; cy[u] = ry[u]; }
    fetch(tb + 16 * U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = tb + 16 * u;
      if (t < t1) {
        const uint32_t row = (uint32_t)b * a.T + t;
        float d[8], y[8], m[8];
        unpack8(cd[u], d);
        unpack8(cy[u], y);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float z = (FL3 & 1) ? fmaf(y[i], sc[i], sh[i]) : y[i];
          m[i] = (!(FL3 & 2) || z > 0.f) ? on : 0.f;
        }
        if (FL3 & 4) tn_drop8(m, (row * (uint32_t)V2_C + (uint32_t)c0) >> 3, dkey, dthr);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float v = fmaf(d[i], g[i], dm[i]) * m[i];
          d[i] = v;
          s1[i] += v;
          s2[i] = fmaf(v, y[i], s2[i]);          // against the raw y; converted to xhat below (linear)
        }
        store8(a.dYbn + (size_t)row * V2_C + c0, d);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    part[tg][0][c0 + i] = s1[i];
    part[tg][1][c0 + i] = cst[3 * V2_C + c0 + i] * s2[i] - cst[2 * V2_C + c0 + i] * s1[i];
  }
  __syncthreads();
  {
    const int which = tid >> 8, c = tid & 255;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += part[k][which][c];
    atomic_add_f32(&a.bsums3[(size_t)((blockIdx.x % TN_NREP) * 2 + which) * V2_C + c], v);
  }
}
inline int launch_combine_bwd2_v2(const CombineBwd2V2Args& a, int B, hipStream_t st) {
  const int fl = (a.act3.mode != 0 ? 1 : 0) | (a.act3.relu ? 2 : 0) | (a.act3.drop_thr ? 4 : 0);
  const dim3 grid(B * a.parts), blk(512);
  if (fl == 7) hipLaunchKernelGGL((combine_bwd2_v2_kernel<7>), grid, blk, 0, st, a);
  else if (fl == 3) hipLaunchKernelGGL((combine_bwd2_v2_kernel<3>), grid, blk, 0, st, a);
  else return -1000;
  return (int)hipGetLastError();
}

// ==========================================================================================
// dw_bwd_slab: depthwise backward + activation backward for the WIDE models (hidden = 512 / 1024, K = 7 / 11 taps), one
// 256-channel slab of the C-wide tensors per workgroup.  The generic dw_bwd_kernel (64 x 64 tiles, fp32 LDS, a boundary
// test and two LDS reads per tap and output) ran at 1/6 of its memory time on TitaNet-L (617 us for 472 MB).  Here:
//   * raw dD and X rows of a 64-row tile (+ K - 1 halo rows) are moved by LDS-DMA into one of two LDS buffers while the
//     previous tile is processed (no staging registers: the tap windows below need them);
//   * a wave owns a strip of 8 output rows, a lane 4 channels; the K-row window of dD lives in REGISTERS and rolls down
//     the strip (slots = row index mod K, all compile-time): a window row is read from LDS and unpacked once per strip
//     instead of once per tap, and ONE window serves both the data gradient and the tap-weight gradient (see the fast
//     path); 2 K FMAs per output and channel remain, which is what bounds the kernel at K = 11;
//   * strips whose window crosses an utterance / batch boundary roll the same window with a wave-uniform test per tap.
// FL bits as in dw_bwd_v4: 1 BatchNorm on load of X, 2 ReLU, 4 dropout, 8 skip-path addend.
// ==========================================================================================
// wait until at most N younger vector-memory operations are outstanding, naming the registers of the asm loads this retires
// (no consumer is scheduled above the wait).  ONE statement with a compile-time count: a switch over run-time counts made
// hipcc merge the per-case "+v" values with v_mov copies placed BEFORE the wait — reads of registers whose loads were still
// in flight (non-finite gradients now and then at large batches).  The caller therefore waits for the SMALLEST number of DMA
// instructions a wave issues behind the loads, on every tile (the last tile of a workgroup issues a dummy set).
template <int N, int RS>
__device__ __forceinline__ void tn_wait_add(uint32_t (&r)[RS]) {
  static_assert(RS == 4 || RS == 8 || RS == 16, "register count of the addend rows");
  if constexpr (RS == 16)
    asm volatile("s_waitcnt vmcnt(%16)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]),
                   "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "n"(N));
  else if constexpr (RS == 8)
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "n"(N));
  else
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N));
}
struct DwBwdSlabArgs {
  const bf16_t* dD; const bf16_t* X; BnAct actX;
  const bf16_t* ADD;   // or null
  bf16_t* OUT;
  const float* wdw;    // [C][KD]
  float* g_wdw;        // [C][KD] (atomic accumulate, pre-zeroed)
  float* g_bdw;        // [C]
  float* bsumsX;       // [TN_NREP][2][C] or null
  // round 6: the tap-weight / bias gradient sums of a workgroup are STORED here ([gridDim.x][KD + 1][256], plain coalesced
  // stores) and added up by dw_part_reduce_kernel when the gradient bucket is finalised, instead of 2 K atomics per workgroup
  // on g_wdw (stride-K addresses, 256 workgroups per address): 14 of 48 us per TitaNet-M launch were those atomics.  null: atomics
  float* gpart;
  int M, T, C, ntiles;
  const int* rowtiles; int n_rowtiles;      // as DwFwdSlabArgs: the 256-row tiles with valid frames (variable-length batches) or null
};
// CH = channels per lane: 4 (one wave per strip of 8 output rows; K = 7: 82 -> 73 us per TitaNet-M layer, one dropout hash per 2 lanes
// instead of 4) or 2 (two waves per strip of 16 rows: half the window /
// weight / accumulator registers per lane, which is what K = 11 needs to stay out of scratch)
// MK: variable-length batch (a.actX.rm.len); a compile-time flag so that the fixed-length instantiation carries none of it
template <int KD, int FL, int CH, bool MK = false>
__global__ __launch_bounds__(512, 2) void dw_bwd_slab_kernel(DwBwdSlabArgs a) {
  constexpr int PADR = (KD - 1) / 2, ROWS = 64 + KD - 1, NT = 512;
  constexpr int TILE_B = ROWS * 512;                        // bytes of one stream's tile
  constexpr int WPS = 4 / CH, RS = 8 * WPS;                 // waves per strip, output rows per strip
  constexpr bool HAS_MASK = (FL & 7) != 0, HAS_ADD = (FL & 8) != 0;
  static_assert(KD % 2 == 1 && KD <= 15 && (CH == 2 || CH == 4), "odd tap counts up to 15; 2 or 4 channels per lane");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [2 buffers][dD | X][ROWS][256] bf16 raw, then mean*rstd, rstd : [2][256].  The start-up constants sc, sh, wd[KD]
  // ([2 + KD][256] floats) sit in buffer 1 until they are in registers (its first DMA is issued after that).
  float* fin = reinterpret_cast<float*>(smem + 4 * TILE_B);
  float* cst = reinterpret_cast<float*>(smem + 2 * TILE_B);
  const unsigned ring_lds = (unsigned)(uintptr_t)(tn_lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cl = (wave % WPS) * 64 * CH + lane * CH;           // first channel of the lane inside the slab
  const int strip = wave / WPS;
  const int nslab = a.C / V2_C;
  const int slab = blockIdx.x % nslab, first = blockIdx.x / nslab, stride = gridDim.x / nslab;
  const int cb = slab * V2_C;                                  // first channel of the slab
  const float mscale = (FL & 4) ? a.actX.inv_keep : 1.f;
  const uint32_t dkey = tn_act_key(a.actX), dthr = a.actX.drop_thr;
  const int* __restrict__ len = MK ? a.actX.rm.len : nullptr;  // valid frames per utterance (uniform)
  if (tid < V2_C) {
    float s = 1.f, h = 0.f, mean = 0.f, rstd = 1.f;
    if (FL & 1) { bn_scale_shift(a.actX, a.C, cb + tid, s, h); bn_mean_rstd(a.actX, a.C, cb + tid, mean, rstd); }
    cst[tid] = s; cst[V2_C + tid] = h; fin[tid] = mean * rstd; fin[V2_C + tid] = rstd;
#pragma unroll
    for (int k = 0; k < KD; ++k) cst[(2 + k) * V2_C + tid] = a.wdw[(size_t)(cb + tid) * KD + k];
  }
  // every compiler-visible load is complete before the first DMA: the waits below are plain vmcnt(0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // this wave's rows of a tile: 2 rows (1 KB) per instruction, rows 2 * (wave + 8 i)
  // (a wave issues NPIECE or NPIECE - 1 pieces; the wait for the addend rows counts NPIECE - 1 of them, see tn_wait_add)
  constexpr int NPIECE = (ROWS / 2 + 7) / 8, MINPIECE = (ROWS / 2) / 8;
  static_assert(MINPIECE == NPIECE - 1 || MINPIECE == NPIECE, "pieces per wave");
  auto tile_row0 = [&](int tile) -> int { return (MK && a.rowtiles) ? tn_sload_i32(a.rowtiles, tile >> 2) * 256 + (tile & 3) * 64 : tile * 64; };
  auto dma_tile = [&](int tile, int buf) {
    const int raw0 = tile_row0(tile) - PADR;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int r = 2 * (wave + 8 * i);
      if (r < ROWS) {
        int gr = raw0 + r + (lane >> 5);
        gr = gr < 0 ? 0 : (gr >= a.M ? a.M - 1 : gr);          // rows outside the tensor: any valid row (never used)
        const size_t o = (size_t)gr * a.C + cb + (lane & 31) * 8;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(ring_lds + buf * 2 * TILE_B + r * 512));
        tn_dma16(a.dD + o, dst);
        tn_dma16(a.X + o, dst + TILE_B);
      }
    }
  };
  if (first < a.ntiles) dma_tile(first, 0);
  __syncthreads();
  float sc[CH], sh[CH], wd[KD][CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    sc[i] = cst[cl + i]; sh[i] = cst[V2_C + cl + i];
#pragma unroll
    for (int k = 0; k < KD; ++k) wd[k][i] = cst[(2 + k) * V2_C + cl + i];
  }
  float gw[KD][CH], gb[CH], s1[CH], s2[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    gb[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f;
#pragma unroll
    for (int k = 0; k < KD; ++k) gw[k][i] = 0.f;
  }
  int buf = 0;
  for (int tile = first; tile < a.ntiles; tile += stride, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this tile's DMA (and the previous tile's stores)
    __builtin_amdgcn_s_barrier();                              // every wave's part landed; the other buffer is free again
    const int out0 = tile_row0(tile), raw0 = out0 - PADR;
    const int l0 = strip * RS;                                 // LDS row of the strip's first window row
    // skip-path addend rows of the strip.  A load hipcc counts would be waited for with vmcnt(its later loads), and the
    // DMA instructions issued below — which hipcc does not see — would then have to retire first (in-order counter): the
    // next tile's DMA serialised with this tile's arithmetic (+70 % on this variant).  So: asm loads issued BEFORE the DMA,
    // waited for with vmcnt(number of DMA instructions this wave issues after them), their registers named in the wait.
    constexpr int NADD = HAS_ADD ? RS * CH / 2 : 2;           // 32-bit words per lane
    uint32_t addr_[NADD];
    if constexpr (HAS_ADD) {
#pragma unroll
      for (int o = 0; o < RS; ++o) {
        int gr = out0 + l0 + o;
        gr = gr < a.M ? gr : a.M - 1;
        const bf16_t* src = a.ADD + (size_t)gr * a.C + cb + cl;
#pragma unroll
        for (int h = 0; h < CH / 2; ++h) asm volatile("global_load_dword %0, %1, off" : "=v"(addr_[o * (CH / 2) + h]) : "v"(src + 2 * h));
      }
    }
    const bool more = tile + stride < a.ntiles;
    if (more) dma_tile(tile + stride, buf ^ 1);
    else if (HAS_ADD) dma_tile(tile, buf ^ 1);                // last tile: the same count of DMA instructions (into the free buffer)
    float addv[HAS_ADD ? RS : 1][CH];
    if constexpr (HAS_ADD) {
      // at least 2 * MINPIECE DMA instructions were issued after the addend loads (a wave with one piece more also waits
      // for its oldest piece: issued right behind the addend rows, it lands with them)
      tn_wait_add<2 * MINPIECE, NADD>(addr_);
#pragma unroll
      for (int o = 0; o < RS; ++o)
#pragma unroll
        for (int h = 0; h < CH / 2; ++h) {
          addv[o][2 * h] = __uint_as_float(addr_[o * (CH / 2) + h] << 16);
          addv[o][2 * h + 1] = __uint_as_float(addr_[o * (CH / 2) + h] & 0xffff0000u);
        }
    }
    const bf16_t* Ds = reinterpret_cast<const bf16_t*>(smem + buf * 2 * TILE_B);
    const bf16_t* Xs = reinterpret_cast<const bf16_t*>(smem + buf * 2 * TILE_B + TILE_B);
    const int g_first = raw0 + l0, g_last = g_first + KD + RS - 2;   // window rows l0 .. l0 + KD + RS - 2
    // variable-length batches: dD is zero on padding rows (the BatchNorm-backward pass writes dS = 0 there); padding rows
    // of X read as zeros (no tap-weight gradient through them) and the data gradient is WRITTEN as zero there
    const int tf = g_first >= 0 ? g_first % a.T : 0;
    const bool inside = g_first >= 0 && g_last < a.M && tf + KD + RS - 2 < a.T;              // window inside one utterance
    int Lf = a.T;
    if (len && inside) Lf = tn_sload_i32(len, g_first / a.T);
    const bool fast = inside && tf + KD + RS - 2 < Lf;                                       // wave-uniform
    if (len && inside && tf + PADR >= Lf) {
      float z[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) z[i] = 0.f;
#pragma unroll
      for (int o = 0; o < RS; ++o) st_ch<CH>(a.OUT + (size_t)(out0 + l0 + o) * a.C + cb + cl, z);
    } else {
      // d w[k] = sum_r dD[r] A[r + k - pad] is summed here over the A rows of the strip (r' = r + k - pad): its dD operand is
      // then dD[r' - k + pad], the SAME row the data gradient of output row r' multiplies with w[k] — one window (of dD) serves
      // both sums and the activation is evaluated once per output row, not once per window row.
      // ONE body for both kinds of strip: strips whose window crosses an utterance / batch / length boundary roll the same
      // register window and only add a wave-uniform test per tap (a separate per-tap path reading LDS cost 4x a plain strip,
      // and with one barrier per tile every workgroup waited for its slowest strip: 189 -> 1xx us per TitaNet-L layer).
      auto body = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        float D[KD][CH];
#pragma unroll
        for (int j = 0; j < KD - 1; ++j) ld_ch<CH>(Ds + (l0 + j) * V2_C + cl, D[j % KD]);
        // frame of output row 0 of the strip and the valid frames of its utterance (boundary strips only)
        int t = 0, Lb = a.T;
        if constexpr (!FAST) {
          const int gr0 = out0 + l0;
          t = gr0 % a.T;
          if (len && gr0 < a.M) Lb = tn_sload_i32(len, gr0 / a.T);
        }
#pragma unroll
        for (int o = 0; o < RS; ++o) {
          ld_ch<CH>(Ds + (l0 + o + KD - 1) * V2_C + cl, D[(o + KD - 1) % KD]);
          const int gr = out0 + l0 + o;
          if (FAST || gr < a.M) {
            float y[CH], Ac[CH], dA[CH];
            ld_ch<CH>(Xs + (l0 + o + PADR) * V2_C + cl, y);
#pragma unroll
            for (int i = 0; i < CH; ++i) { Ac[i] = y[i]; dA[i] = 0.f; }
            act_c<(FL & 7), CH>(Ac, sc, sh, dkey, dthr, (uint32_t)gr, a.C, cb + cl);
            // padding frame (variable-length batch): no tap-weight gradient through it, its data gradient is written as zero
            // (dD of a padding row is zero where it was computed; row tiles that are padding only are skipped by the pipelined
            //  data-gradient GEMM and hold stale values: never read them as data)
            const bool pad = !FAST && t >= Lb;
            if (pad) {
#pragma unroll
              for (int i = 0; i < CH; ++i) Ac[i] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < KD; ++k) {
              const int sl = (o + KD - 1 - k) % KD;
              const int tb = t - k + PADR;             // frame of dD[gr - k + PADR]: inside this utterance (and a valid frame)?
              if (FAST || (tb >= 0 && tb < Lb)) {
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                  dA[i] = fmaf(wd[k][i], D[sl][i], dA[i]);
                  gw[k][i] = fmaf(Ac[i], D[sl][i], gw[k][i]);
                }
              }
            }
            if (!pad) {
#pragma unroll
              for (int i = 0; i < CH; ++i) gb[i] += D[(o + PADR) % KD][i];
            }
            if (HAS_ADD) {
#pragma unroll
              for (int i = 0; i < CH; ++i) dA[i] += addv[o][i];
            }
            if (pad) {
#pragma unroll
              for (int i = 0; i < CH; ++i) dA[i] = 0.f;
            }
            if (HAS_MASK) {
#pragma unroll
              for (int i = 0; i < CH; ++i) {
                const float m = (FL & 2) ? ((Ac[i] > 0.f) ? mscale : 0.f) : mscale;
                dA[i] *= m;
                s1[i] += dA[i];
                s2[i] = fmaf(dA[i], y[i], s2[i]);
              }
            }
            st_ch<CH>(a.OUT + (size_t)gr * a.C + cb + cl, dA);
            if constexpr (!FAST) {
              if (++t == a.T) {                        // the next output row opens the next utterance
                t = 0;
                Lb = (len && gr + 1 < a.M) ? tn_sload_i32(len, (gr + 1) / a.T) : a.T;
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);           // rows in order: bounds the live temporaries
        }
      };
      if (fast) body(std::true_type{});
      else body(std::false_type{});
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // s2 was accumulated against the RAW x:  sum dA * xhat = rstd * sum dA*x - mean*rstd * sum dA
#pragma unroll
  for (int i = 0; i < CH; ++i) s2[i] = fin[V2_C + cl + i] * s2[i] - fin[cl + i] * s1[i];
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);       // [strips][KD + 3][256] inside the tile buffers
  {
    float* mine = red + (size_t)strip * (KD + 3) * V2_C + cl;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
#pragma unroll
      for (int k = 0; k < KD; ++k) mine[k * V2_C + i] = gw[k][i];
      mine[KD * V2_C + i] = gb[i]; mine[(KD + 1) * V2_C + i] = s1[i]; mine[(KD + 2) * V2_C + i] = s2[i];
    }
  }
  __syncthreads();
  const int rep = (blockIdx.x / nslab) % TN_NREP;
  for (int i = tid; i < (KD + 3) * V2_C; i += NT) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 8 / WPS; ++w) v += red[(size_t)w * (KD + 3) * V2_C + i];
    const int k = i / V2_C, c = cb + i % V2_C;
    if (a.gpart && k <= KD) { a.gpart[(size_t)blockIdx.x * (KD + 1) * V2_C + i] = v; continue; }
    if (k < KD) atomic_add_f32(&a.g_wdw[(size_t)c * KD + k], v);
    else if (k == KD) atomic_add_f32(&a.g_bdw[c], v);
    else if (a.bsumsX && HAS_MASK) atomic_add_f32(&a.bsumsX[(size_t)(rep * 2 + (k - KD - 1)) * a.C + c], v);
  }
}
template <int KD, int FL>
inline int launch_dw_bwd_slab_t(DwBwdSlabArgs a, int grid, hipStream_t st) {
  constexpr int ROWS = 64 + KD - 1;
  const size_t tiles = (size_t)4 * ROWS * 512, red = (size_t)8 * (KD + 3) * V2_C * sizeof(float);
  const size_t smem = (tiles > red ? tiles : red) + (size_t)2 * V2_C * sizeof(float);
  auto kern = a.actX.rm.len ? dw_bwd_slab_kernel<KD, FL, (KD >= 11 ? 2 : 4), true> : dw_bwd_slab_kernel<KD, FL, (KD >= 11 ? 2 : 4), false>;
  TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, a);
  return (int)hipGetLastError();
}
// -1000: no specialisation for this (taps, flags) combination (caller runs the generic dw_bwd_kernel)
// workgroups per 256-channel slab of a dw_bwd_slab launch (the launcher's grid / nslab; dw_part_reduce_kernel's partial count)
inline int dw_bwd_slab_per(int M, int C, int n_rowtiles, int max_wgs) {
  const int ntiles = n_rowtiles > 0 ? n_rowtiles * 4 : (M + 63) / 64;
  int per = max_wgs / (C / V2_C);
  if (per < 1) per = 1;
  return per > ntiles ? ntiles : per;
}
// gradient sums of the depthwise taps / bias from the partial records of dw_bwd_slab_kernel (DwBwdSlabArgs::gpart):
// grid (layers, C / 256, KD + 1); workgroup blockIdx.x of the slab kernel worked on slab blockIdx.x % nslab.  Fixed order:
// the result does not depend on the order the workgroups ran in.
__global__ __launch_bounds__(256) void dw_part_reduce_kernel(const DwGradOut* __restrict__ outs, int KD, int nslab, int per) {
  const DwGradOut o = outs[blockIdx.x];
  const int slab = blockIdx.y, k = blockIdx.z, c = threadIdx.x;
  const float* src = o.gacc + ((size_t)slab * (KD + 1) + k) * V2_C + c;
  const size_t step = (size_t)nslab * (KD + 1) * V2_C;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int w = 0;
  for (; w + 8 <= per; w += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] += src[(size_t)(w + u) * step];
  }
  for (; w < per; ++w) v[0] += src[(size_t)w * step];
  const float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  const int cg = slab * V2_C + c;
  if (k < KD) o.g_wdw[(size_t)cg * KD + k] = s;
  else o.g_bdw[cg] = s;
}
template <int KD>
inline int launch_dw_bwd_slab(DwBwdSlabArgs a, int max_wgs, hipStream_t st) {
  if (a.C % V2_C != 0) return -1000;
  if (!a.actX.rm.len) { a.rowtiles = nullptr; a.n_rowtiles = 0; }
  a.ntiles = a.rowtiles ? a.n_rowtiles * 4 : (a.M + 63) / 64;
  if (a.ntiles <= 0) return 0;
  const int nslab = a.C / V2_C;
  int per = max_wgs / nslab;
  if (per < 1) per = 1;
  if (per > a.ntiles) per = a.ntiles;
  const int grid = per * nslab;
  const int fl = (a.actX.mode != 0 ? 1 : 0) | (a.actX.relu ? 2 : 0) | (a.actX.drop_thr ? 4 : 0) | (a.ADD ? 8 : 0);
  switch (fl) {
    case 7: return launch_dw_bwd_slab_t<KD, 7>(a, grid, st);
    case 11: return launch_dw_bwd_slab_t<KD, 11>(a, grid, st);
    case 8: return launch_dw_bwd_slab_t<KD, 8>(a, grid, st);
    default: return -1000;
  }
}

// ==========================================================================================
// Pointwise data gradient, lean version:  dD = BatchNorm-backward-on-load(dZ, Y) * W
// (the 1x1-conv dgrad of sub-blocks and skip connections).  R rows per tile; R = 32 keeps the kernel
// under 128 VGPRs so TWO workgroups (16 waves) share a CU and overlap each other's load / transform /
// MFMA / store phases; W^T stays resident in registers as MFMA A fragments.
// ==========================================================================================
struct DgradV2Args {
  const bf16_t* dZ;
  const bf16_t* Y;
  BnBwd bn;
  const bf16_t* Wt;    // [256 in-channels][256 out-channels]
  bf16_t* OUT;         // [M][256]
  int M, ntiles;
  const uint4* Wswz;   // optional: Wt in MFMA-fragment order [8 waves][16 k-steps][64 lanes] x 16 bytes
  bf16_t* dS_out;      // optional: the BatchNorm-backward'd gradient, stored [M][256] (operand of the pipelined weight gradient)
};

template <int R>
__global__ __launch_bounds__(V2_NT, (R == 32 ? 4 : 2)) void dgrad_v2_kernel(DgradV2Args a) {
  constexpr int NQ = R / 16;      // 16-byte vectors per thread per stream (512 threads cover 16 rows x 32 vectors)
  constexpr int NTILE = R / 32;   // 32-row MFMA N tiles
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Pt = reinterpret_cast<bf16_t*>(smem);      // [R][264] dY rows (MFMA B operand)
  bf16_t* Dt = Pt + R * V2_AP;                        // [R][264] dD rows
  float* cst = reinterpret_cast<float*>(Dt + R * V2_AP);   // k0, k1, k2 : [3][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vc = tid & 31, rq = tid >> 5, c0 = vc * 8, half = lane >> 5;
  if (tid < V2_C) {
    float k0, k1, k2;
    bn_bwd_coefs(a.bn, V2_C, tid, k0, k1, k2);
    cst[tid] = k0; cst[V2_C + tid] = k1; cst[2 * V2_C + tid] = k2;
  }
  bf16x8_t wf[16];
  {
    const int ci = wave * 32 + (lane & 31);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (a.Wswz) wf[ks] = __builtin_bit_cast(bf16x8_t, a.Wswz[((size_t)wave * 16 + ks) * 64 + lane]);   // fragment order: coalesced
      else wf[ks] = *reinterpret_cast<const bf16x8_t*>(a.Wt + (size_t)ci * V2_C + ks * 16 + half * 8);
    }
  }
  // unconditional buffer loads / stores, as in sub_fwd_v5 (rows beyond the tensor: zeros / dropped; an absent dS_out: every
  // store dropped): the compiler's vmcnt waits then stay counted instead of vmcnt(0) behind this tile's output stores
  typedef __attribute__((ext_vector_type(4))) unsigned int dg_u32x4_t;
  const int tbytes = (int)((size_t)a.M * V2_C * sizeof(bf16_t));
  const __amdgpu_buffer_rsrc_t srdZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.dZ), 0, tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.Y), 0, tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc(a.OUT, 0, tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdS = __builtin_amdgcn_make_buffer_rsrc(a.dS_out, 0, a.dS_out ? tbytes : 0, 0x00020000);
  uint4 pz[NQ], py[NQ];
  auto prefetch_q = [&](int tile, int q) {
    const int o = ((tile * R + rq + 16 * q) * V2_C + c0) * (int)sizeof(bf16_t);
    pz[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srdZ, o, 0, 0));
    py[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srdY, o, 0, 0));
  };
  int tile = blockIdx.x;
  if (tile < a.ntiles) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) prefetch_q(tile, q);
  }
  __syncthreads();
  float k0[8], k1[8], k2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { k0[i] = cst[c0 + i]; k1[i] = cst[V2_C + c0 + i]; k2[i] = cst[2 * V2_C + c0 + i]; }
  for (; tile < a.ntiles; tile += gridDim.x) {
    const int out0 = tile * R;
    __syncthreads();   // (1) previous tile's Dt rows have been stored
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int r = rq + 16 * q;
      float z[8], y[8];
      unpack8(pz[q], z);
      unpack8(py[q], y);
      if (out0 + r < a.M) {
#pragma unroll
        for (int i = 0; i < 8; ++i) z[i] = k0[i] * z[i] + k1[i] * y[i] + k2[i];
      }
      {
        dg_u32x4_t zs;
        zs[0] = f2bf_pk(z[0], z[1]); zs[1] = f2bf_pk(z[2], z[3]); zs[2] = f2bf_pk(z[4], z[5]); zs[3] = f2bf_pk(z[6], z[7]);
        __builtin_amdgcn_raw_buffer_store_b128(zs, srdS, ((out0 + r) * V2_C + c0) * (int)sizeof(bf16_t), 0, (TN_NT_WGRAD_OPERANDS & 16) ? 2 : 0);      // (rows >= M: out of range)
      }
      store8(Pt + r * V2_AP + c0, z);
    }
    // (refilling inside the loop above measured 2.5 us slower; past the last tile: rows beyond M, zeros)
#pragma unroll
    for (int q = 0; q < NQ; ++q) prefetch_q(tile + gridDim.x, q);
    __syncthreads();   // (2)
    f32x16_t acc[NTILE];
#pragma unroll
    for (int n = 0; n < NTILE; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const bf16_t* brow = Pt + (lane & 31) * V2_AP + half * 8;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
      for (int n = 0; n < NTILE; ++n) {
        const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(brow + n * 32 * V2_AP + ks * 16);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b, acc[n], 0, 0, 0);
      }
    }
#pragma unroll
    for (int n = 0; n < NTILE; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ci = wave * 32 + 8 * g + 4 * half;
        uint2 w;
        w.x = f2bf_pk(acc[n][4 * g], acc[n][4 * g + 1]);
        w.y = f2bf_pk(acc[n][4 * g + 2], acc[n][4 * g + 3]);
        *reinterpret_cast<uint2*>(Dt + (n * 32 + (lane & 31)) * V2_AP + ci) = w;
      }
    __syncthreads();   // (3)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int o = rq + 16 * q, gr = out0 + o;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dg_u32x4_t, *reinterpret_cast<const uint4*>(Dt + o * V2_AP + c0)), srdO,
                                             (gr * V2_C + c0) * (int)sizeof(bf16_t), 0, (TN_NT_WGRAD_OPERANDS & 32) ? 2 : 0);      // (rows >= M: out of range)
    }
  }
}

template <int R>
inline int launch_dgrad_v2(DgradV2Args a, int max_wgs, hipStream_t st) {
  if ((long)a.M * V2_C * 2 >= (1L << 31)) return -1000;      // 32-bit buffer offsets: the generic kernel takes such a batch
  a.ntiles = (a.M + R - 1) / R;
  const int grid = a.ntiles < max_wgs ? a.ntiles : max_wgs;
  const size_t smem = (size_t)2 * R * V2_AP * sizeof(bf16_t) + (size_t)3 * V2_C * sizeof(float);
  auto kern = dgrad_v2_kernel<R>;
  TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(V2_NT), smem, st, a);
  return (int)hipGetLastError();
}

// ==========================================================================================
// dgrad_dw_v6: pointwise data gradient + depthwise / activation backward of one sub-block in ONE pass
//   dD   = BatchNorm-backward-on-load(dZ, Y) * W                    (MFMA, W^T resident in registers)
//   dA   = sum_k w_dw[c][k] dD[r - k + pad]  (+ ADD)                  (transposed stencil over time, through LDS)
//   OUT  = dA * d act / d bn (X)  -> stored;  BN-backward sums of the layer that made X;  d w_dw, d b_dw
// A separate data-gradient kernel + depthwise-backward kernel (round 1) wrote dD (one rows x 256 tensor) to HBM and read it
// back: 6 passes per sub-block; this kernel moves 4 (dZ, Y, X in, OUT out).  Both predecessors ran at the rate of a plain
// streaming kernel, so the passes are the time.  Tiles are 32 GEMM rows that yield 30 output rows (the stencil needs dD[r-1] and dD[r+1]: consecutive tiles
// overlap by two rows, re-read from L2), small enough that the three input streams of the NEXT tile (48 KB per CU) stay in
// flight in registers (6 x 16 bytes per thread) without spilling next to the 64 weight VGPRs.
// FL bits: 1 = BatchNorm on load of X, 2 = ReLU, 4 = dropout, 8 = skip-path addend.
// ==========================================================================================
#define V6_R 32
#define V6_OUT 30
// position of channel c in a SPLIT row of 256 per-channel constants: the low four channels of every 8-channel vector, then the high four
__device__ __forceinline__ int v6_split(int c) { return ((c & 4) ? V2_C / 2 : 0) + (c >> 3) * 4 + (c & 3); }
struct DgradDwArgs {
  const bf16_t* dZ; const bf16_t* Y; BnBwd bn;       // gradient wrt the BatchNorm output of this sub-block, its raw output
  const uint4* Wswz;                                   // W^T in MFMA-fragment order (swizzle256_kernel)
  const bf16_t* X; BnAct actX;                         // raw input of the sub-block's depthwise conv + its activation
  const bf16_t* ADD;                                   // or null
  bf16_t* OUT;
  const float* wdw;
  float* gacc;                                         // [TN_NREP][KD + 1][256]
  float* bsumsX;                                       // or null
  int M, T, ntiles;
  // Z3 variants (the LAST sub-block of a mega block, round 4): dZ holds the gradient wrt the block's pre-activation sum (what
  // combine_bwd1_v3 wrote), and the gradient wrt this sub-block's BatchNorm output is rebuilt on load,
  //   dYbn = (dZ * ga[b][c] + ub[b][c]) * [act3(Y) > 0],
  // instead of being read from a tensor a second tail pass would have had to write (3 tensor passes per mega block less).
  const float* gu;                                     // [B][2][256]: ga, ub (combine_bwd1_v3), or null
  BnAct act3;                                          // activation of this sub-block's output (BatchNorm, ReLU, dropout)
  bf16_t* dS_out;                                      // the BatchNorm-backward'd gradient is also stored (rows x 256): the
                                                       // weight-gradient launch then reads it as a plain operand (any variant;
                                                       // or null)
};

// MK (variable-length batch, a.bn.rm.len): dS = 0 on padding rows (their dD then adds nothing to the valid rows next to them),
// padding rows of X read as zeros (no tap-weight gradient through them), the data gradient is written as zero there.
// P2 (round 4): two barriers per tile instead of three — the stencil of tile t and the transform of tile t + 1 share one phase
// (raw X rows double-buffered), the MFMAs of tile t + 1 the other.
// WGX (tuning harness only, tools/dgrad_dw_harness.hip): what the MATRIX-PIPE share of a fused pointwise weight gradient would
// add to a tile — 16 more MFMAs per wave (the 256 x 256 x 32-row product d W += dS^T Q of the tile, split over 8 waves) fed by
// transposing LDS fragment reads of the tile's rows, on two alternating accumulators (the real thing needs 128 accumulator
// registers per lane, which this kernel does not have: a LOWER bound of the in-kernel cost)
// MD (tuning harness only): fragment reads in flight in a HAND-SCHEDULED MFMA phase (tn_common.h: tn_mfma_chain_lds) instead of the
// compiler's read-wait-MFMA: bit-identical, 38.3 -> 36.5 - 37.7 us in the harness, nothing in the step (the other wave of the SIMD
// already covers the exposed LDS round trips): profiles/r05_dgrad_dw_md.txt.  0 = the compiler's own schedule
template <int FL, bool MK = false, bool Z3 = false, bool P2 = false, int WGX = 0, int MD = 0>
__global__ __launch_bounds__(V2_NT, 2) void dgrad_dw_v6_kernel(DgradDwArgs a) {
  constexpr int KD = 3, NT = V2_NT;
  constexpr bool HAS_MASK = (FL & 7) != 0, HAS_ADD = (FL & 8) != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Pt = reinterpret_cast<bf16_t*>(smem);          // [32][264] BN-backward'd dY rows (MFMA B operand)
  bf16_t* Dt = Pt + V6_R * V2_AP;                         // [32][264] dD rows
  bf16_t* Xs = Dt + V6_R * V2_AP;                         // [32][256] raw X rows (P2: two buffers)
  float* cst = reinterpret_cast<float*>(Xs + (P2 ? 2 : 1) * V6_R * V2_C);   // k0,k1,k2, sc,sh,mean*rstd,rstd, wd[3] : [10][256] (+ sc3, sh3 with Z3)
  float* gus = cst + 12 * V2_C;                               // Z3: ga, ub of the tile's two utterances [2][2][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int vc = tid & 31, rq = tid >> 5, c0 = vc * 8;   // transform layout: 8 channels x rows rq, rq + 16
  const int c4 = lane * 4;                               // stencil layout: 4 channels per lane, one wave per strip of 4 rows
  const int cs0 = vc * 4, cs1 = V2_C / 2 + vc * 4;       // this thread's two float4 of a SPLIT constant row (v6_split)
  const float mscale = (FL & 4) ? a.actX.inv_keep : 1.f;
  const uint32_t dkey = tn_act_key(a.actX), dthr = a.actX.drop_thr;
  if (tid < V2_C) {
    float k0, k1, k2, s = 1.f, h = 0.f, mean = 0.f, rstd = 1.f;
    bn_bwd_coefs(a.bn, V2_C, tid, k0, k1, k2);
    if (FL & 1) { bn_scale_shift(a.actX, V2_C, tid, s, h); bn_mean_rstd(a.actX, V2_C, tid, mean, rstd); }
    // the rows the transform reads per tile as two float4 per thread (k0, k1, k2, sc3, sh3, and gus below) are stored SPLIT:
    // channels 8 v .. 8 v + 3 of all 32 vectors first, then channels 8 v + 4 .. 8 v + 7 — each ds_read_b128 then has a 16-byte
    // lane stride (the natural order, a 32-byte stride, is a 2-way bank conflict on every one of them: round 5's counters had
    // the Z3 variant at 4x the plain variant's SQ_LDS_BANK_CONFLICT)
    const int ps = v6_split(tid);
    cst[ps] = k0; cst[V2_C + ps] = k1; cst[2 * V2_C + ps] = k2;
    cst[3 * V2_C + tid] = s; cst[4 * V2_C + tid] = h; cst[5 * V2_C + tid] = mean * rstd; cst[6 * V2_C + tid] = rstd;
#pragma unroll
    for (int k = 0; k < KD; ++k) cst[(7 + k) * V2_C + tid] = a.wdw[(size_t)tid * KD + k];
    if (Z3) {
      float s3 = 1.f, h3 = 0.f;
      bn_scale_shift(a.act3, V2_C, tid, s3, h3);
      cst[10 * V2_C + ps] = s3; cst[11 * V2_C + ps] = h3;
    }
  }
  const uint32_t dkey3 = Z3 ? tn_act_key(a.act3) : 0u, dthr3 = Z3 ? a.act3.drop_thr : 0u;
  // Z3: ga / ub of the (at most two) utterances a tile touches travel global -> register (one 8-byte load per thread, issued
  // with the tile prefetch) -> LDS (written behind the previous tile's stencil, read by this tile's transform)
  uint2 pg = make_uint2(0, 0);
  auto prefetch_g = [&](int tile) {
    const int g0t = tile * V6_OUT - 1;
    const int nbat = a.M / a.T;
    int bb = (g0t < 0 ? 0 : g0t) / a.T + (tid >> 8);
    bb = bb < nbat ? bb : nbat - 1;
    pg = *reinterpret_cast<const uint2*>(a.gu + (size_t)bb * 2 * V2_C + (tid & 255) * 2);
  };
  // (thread j < 256 of a half holds floats 2 j, 2 j + 1 of the utterance's [ga | ub] record: one row, two adjacent channels)
  auto publish_g = [&]() {
    const int i2 = (tid & 255) * 2;
    *reinterpret_cast<uint2*>(gus + (tid >> 8) * 2 * V2_C + (i2 & V2_C) + v6_split(i2 & (V2_C - 1))) = pg;
  };
  bf16x8_t wf[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) wf[ks] = __builtin_bit_cast(bf16x8_t, a.Wswz[((size_t)wave * 16 + ks) * 64 + lane]);
  uint4 pz[2], py[2], px[2];
  // each prefetch register is refilled for the next tile right after it was consumed: the memory pipe never runs empty
  auto prefetch_q = [&](int tile, int q) {
    const int gr = tile * V6_OUT - 1 + rq + 16 * q;
    const bool ok = gr >= 0 && gr < a.M;
    const size_t o = (size_t)gr * V2_C + c0;
    pz[q] = ok ? *reinterpret_cast<const uint4*>(a.dZ + o) : make_uint4(0, 0, 0, 0);
    if (TN_NT_WGRAD_OPERANDS & 4) {
      // (the raw output Y is read here for the LAST time in the step: streamed past the Infinity Cache)
      typedef __attribute__((ext_vector_type(4))) unsigned int v6_u4;
      const v6_u4 t = ok ? __builtin_nontemporal_load(reinterpret_cast<const v6_u4*>(a.Y + o)) : v6_u4{0u, 0u, 0u, 0u};
      py[q] = make_uint4(t[0], t[1], t[2], t[3]);
    } else
    py[q] = ok ? *reinterpret_cast<const uint4*>(a.Y + o) : make_uint4(0, 0, 0, 0);
    px[q] = ok ? *reinterpret_cast<const uint4*>(a.X + o) : make_uint4(0, 0, 0, 0);
  };
  int tile = blockIdx.x;
  if (Z3 && tile < a.ntiles) { prefetch_g(tile); publish_g(); }
  if (tile < a.ntiles) { prefetch_q(tile, 0); prefetch_q(tile, 1); }
  __syncthreads();
  float sc[4], sh[4], wd[KD][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sc[i] = cst[3 * V2_C + c4 + i]; sh[i] = cst[4 * V2_C + c4 + i];
#pragma unroll
    for (int k = 0; k < KD; ++k) wd[k][i] = cst[(7 + k) * V2_C + c4 + i];
  }
  float gw[KD][4], gb[4], s1[4], s2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    gb[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f;
#pragma unroll
    for (int k = 0; k < KD; ++k) gw[k][i] = 0.f;
  }
  // the three phases of a tile (Xb: the tile's raw X rows)
  auto transform = [&](int tile, bf16_t* Xb) {
    const int g0 = tile * V6_OUT - 1;          // global row of tile row 0
    TileMask tm = {0, 0, 0};
    if (MK) tm = tn_tile_mask(a.bn.rm.len, a.T, a.M, g0);
    const int e0 = Z3 ? ((g0 < 0 ? 0 : g0) / a.T + 1) * a.T : 0;     // first row of the tile's second utterance
    // ---- BN backward on load -> Pt;  raw X rows -> Xs
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = rq + 16 * q, gr = g0 + r;
      float z[8], y[8];
      unpack8(pz[q], z);
      unpack8(py[q], y);
      if (Z3) {
        // dYbn = (dZ ga + ub) [act3(y) > 0]: the gradient wrt this sub-block's BatchNorm output, rebuilt from the tail's dZ
        float m[8];
        {
          float c3[8], h3v[8];
          *reinterpret_cast<float4*>(c3) = *reinterpret_cast<const float4*>(cst + 10 * V2_C + cs0);
          *reinterpret_cast<float4*>(c3 + 4) = *reinterpret_cast<const float4*>(cst + 10 * V2_C + cs1);
          *reinterpret_cast<float4*>(h3v) = *reinterpret_cast<const float4*>(cst + 11 * V2_C + cs0);
          *reinterpret_cast<float4*>(h3v + 4) = *reinterpret_cast<const float4*>(cst + 11 * V2_C + cs1);
#pragma unroll
          for (int i = 0; i < 8; ++i) m[i] = (fmaf(y[i], c3[i], h3v[i]) > 0.f) ? 1.f : 0.f;
        }
        if (FL & 4) tn_drop8(m, ((uint32_t)gr * (uint32_t)V2_C + (uint32_t)c0) >> 3, dkey3, dthr3);
        const float* gsel = gus + (gr >= e0 ? 2 * V2_C : 0);
        float gav[8], ubv[8];
        *reinterpret_cast<float4*>(gav) = *reinterpret_cast<const float4*>(gsel + cs0);
        *reinterpret_cast<float4*>(gav + 4) = *reinterpret_cast<const float4*>(gsel + cs1);
        *reinterpret_cast<float4*>(ubv) = *reinterpret_cast<const float4*>(gsel + V2_C + cs0);
        *reinterpret_cast<float4*>(ubv + 4) = *reinterpret_cast<const float4*>(gsel + V2_C + cs1);
#pragma unroll
        for (int i = 0; i < 8; ++i) z[i] = fmaf(z[i], gav[i], ubv[i]) * m[i];
      }
      {
        // k0, k1, k2 from LDS (the registers go to the stencil window): unconditional 16-byte reads, then a select
        float k0v[8], k1v[8], k2v[8];
        *reinterpret_cast<float4*>(k0v) = *reinterpret_cast<const float4*>(cst + cs0);
        *reinterpret_cast<float4*>(k0v + 4) = *reinterpret_cast<const float4*>(cst + cs1);
        *reinterpret_cast<float4*>(k1v) = *reinterpret_cast<const float4*>(cst + V2_C + cs0);
        *reinterpret_cast<float4*>(k1v + 4) = *reinterpret_cast<const float4*>(cst + V2_C + cs1);
        *reinterpret_cast<float4*>(k2v) = *reinterpret_cast<const float4*>(cst + 2 * V2_C + cs0);
        *reinterpret_cast<float4*>(k2v + 4) = *reinterpret_cast<const float4*>(cst + 2 * V2_C + cs1);
        const bool ok = gr >= 0 && gr < a.M;
        const bool valid = !MK || tn_tile_valid(tm, gr);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float v = fmaf(k0v[i], z[i], fmaf(k1v[i], y[i], k2v[i]));
          z[i] = ok ? (valid ? v : 0.f) : (Z3 ? 0.f : z[i]);
        }
      }
      store8(Pt + r * V2_AP + c0, z);
      if (a.dS_out && r >= 1 && r <= V6_OUT && gr >= 0 && gr < a.M) {
        if (TN_NT_WGRAD_OPERANDS & 2) {
          typedef __attribute__((ext_vector_type(4))) unsigned int v6_s4;
          v6_s4 w4;
          w4[0] = f2bf_pk(z[0], z[1]); w4[1] = f2bf_pk(z[2], z[3]); w4[2] = f2bf_pk(z[4], z[5]); w4[3] = f2bf_pk(z[6], z[7]);
          __builtin_nontemporal_store(w4, reinterpret_cast<v6_s4*>(a.dS_out + (size_t)gr * V2_C + c0));
        } else store8(a.dS_out + (size_t)gr * V2_C + c0, z);
      }
      *reinterpret_cast<uint4*>(Xb + r * V2_C + c0) = px[q];
      if (tile + (int)gridDim.x < a.ntiles) prefetch_q(tile + gridDim.x, q);
    }
  };
  auto mfma = [&]() {
    // ---- dD = dY * W : 32 rows x this wave's 32 input channels
    {
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const bf16_t* brow = Pt + (lane & 31) * V2_AP + half * 8;
      if constexpr (MD > 0) {
        tn_mfma_chain_lds<16, MD, 32>(wf, brow, acc);
      } else {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], *reinterpret_cast<const bf16x8_t*>(brow + ks * 16), acc, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = f2bf_pk(acc[4 * g], acc[4 * g + 1]);
        w.y = f2bf_pk(acc[4 * g + 2], acc[4 * g + 3]);
        *reinterpret_cast<uint2*>(Dt + (lane & 31) * V2_AP + wave * 32 + 8 * g + 4 * half) = w;
      }
    }
  };
  f32x16_t wgx_acc[2];
  if constexpr (WGX != 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { wgx_acc[0][r] = 0.f; wgx_acc[1][r] = 0.f; }
  }
  auto wgx = [&](const bf16_t* Xb) {
    if constexpr (WGX != 0) {
      typedef __attribute__((ext_vector_type(4))) short s16x4_t;
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
      auto trfrag = [&](const bf16_t* base, int pitch_b, int r0, int cblk) {
        // rows r0 .. r0 + 15 x 32 channels of block cblk, transposed: lane (l & 31) -> channel, k = rows
        const char* q = reinterpret_cast<const char*>(base) + (size_t)(r0 + (lane >> 4) * 4) * 0 + (size_t)(r0 + ((lane & 15) >> 2)) * pitch_b + (cblk * 32 + (lane >> 4) * 8 + (lane & 3) * 2) * 2;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(q));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(q + 8 * pitch_b));
        s16x8_t v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(bf16x8_t, v);
      };
      // wave: output-channel blocks 2 (wave >> 1) .. + 1 of dS, input-channel blocks 4 (wave & 1) .. + 3 of the other operand
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        bf16x8_t af[2], bfr[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = trfrag(Pt, V2_AP * 2, kh * 16, 2 * (wave >> 1) + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = trfrag(Xb, V2_C * 2, kh * 16, 4 * (wave & 1) + j);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            wgx_acc[(i * 4 + j) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], wgx_acc[(i * 4 + j) & 1], 0, 0, 0);
      }
    }
  };
  auto stencil = [&](int tile, const bf16_t* Xb) {
    const int g0 = tile * V6_OUT - 1;
    TileMask tm = {0, 0, 0};
    if (MK) tm = tn_tile_mask(a.bn.rm.len, a.T, a.M, g0);
    // ---- transposed stencil + activation backward: wave = strip of 4 output rows (tile rows 1 + 4*wave ..), lane = 4 channels
    {
      // 30 output rows over 8 waves: strips of 4, 4, 4, 4, 4, 4, 3, 3 rows (a 2-row leftover strip would always take the
      // slow path and every wave would wait for it at the next barrier)
      const bool nr4 = wave < 6;
      const int NR = nr4 ? 4 : 3;
      const int i0 = nr4 ? 1 + 4 * wave : 25 + 3 * (wave - 6);     // first output row of the strip (tile row index)
      const int gfirst = g0 + i0 - 1, glast = g0 + i0 + NR;        // window rows i0-1 .. i0+NR
      // (masked: the strip lies inside one utterance, so its rows are valid frames iff the last one is)
      const bool fast = gfirst >= 0 && glast < a.M && (gfirst % a.T) + NR + 1 < a.T && (!MK || tn_tile_valid(tm, glast - 1));   // wave-uniform
      auto ldD = [&](int i, float* D) { unpack4(*reinterpret_cast<const uint2*>(Dt + i * V2_AP + c4), D); };
      auto ldX = [&](int i, float* Yr) { unpack4(*reinterpret_cast<const uint2*>(Xb + i * V2_C + c4), Yr); };
      uint2 addv[4];
      if (HAS_ADD) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int gr = g0 + i0 + q;
          addv[q] = (q < NR && gr >= 0 && gr < a.M) ? *reinterpret_cast<const uint2*>(a.ADD + (size_t)gr * V2_C + c4) : make_uint2(0, 0);
        }
      }
      // rolling 3-row window of dD (slots j % 3).  The tap-weight gradient d w[k] = sum_r dD[r] A[r + k - 1] is summed over
      // the A rows of the strip (r' = r + k - 1): its dD operand dD[r' - k + 1] is then a row of the SAME window the data
      // gradient of output row r' uses, and the activation is evaluated for the output rows only (4 per strip, not 6).
      // Boundary strips (utterance edges, ends of the batch, padding frames) roll the same window and add wave-uniform tests
      // per row and tap — every wave of the tile waits for the slowest strip at the next barrier.
      auto strip = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        float D[3][4];
        ldD(i0 - 1, D[0]);
        ldD(i0, D[1]);
        int t = 0;
        if constexpr (!FAST) t = (g0 + i0) % a.T;                 // frame of the strip's first output row (g0 + i0 >= 0)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (q == 3 && !nr4) break;
          const int P = q % 3, C_ = (q + 1) % 3, N = (q + 2) % 3;
          ldD(i0 + q + 1, D[N]);
          const int gr = g0 + i0 + q;
          if (FAST || gr < a.M) {
            float Yr[4], Ac[4], dA[4];
            ldX(i0 + q, Yr);
#pragma unroll
            for (int i = 0; i < 4; ++i) Ac[i] = Yr[i];
            act4_t<FL>(Ac, sc, sh, dkey, dthr, (uint32_t)gr, c4);
            const bool pad = !FAST && MK && !tn_tile_valid(tm, gr);
            if (pad) {
#pragma unroll
              for (int i = 0; i < 4; ++i) Ac[i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              gb[i] += D[C_][i];
              dA[i] = wd[1][i] * D[C_][i];
              gw[1][i] = fmaf(Ac[i], D[C_][i], gw[1][i]);
            }
            if (FAST || (t + 1 < a.T && gr + 1 < a.M)) {          // the next row belongs to the same utterance
#pragma unroll
              for (int i = 0; i < 4; ++i) { dA[i] = fmaf(wd[0][i], D[N][i], dA[i]); gw[0][i] = fmaf(Ac[i], D[N][i], gw[0][i]); }
            }
            if (FAST || t > 0) {
#pragma unroll
              for (int i = 0; i < 4; ++i) { dA[i] = fmaf(wd[2][i], D[P][i], dA[i]); gw[2][i] = fmaf(Ac[i], D[P][i], gw[2][i]); }
            }
            if (HAS_ADD) {
              float ad[4];
              unpack4(addv[q], ad);
#pragma unroll
              for (int i = 0; i < 4; ++i) dA[i] += ad[i];
            }
            if (pad) {
#pragma unroll
              for (int i = 0; i < 4; ++i) dA[i] = 0.f;
            }
            if (HAS_MASK) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float m = (FL & 2) ? ((Ac[i] > 0.f) ? mscale : 0.f) : mscale;
                dA[i] *= m;
                s1[i] += dA[i];
                s2[i] = fmaf(dA[i], Yr[i], s2[i]);
              }
            }
            uint2 ov;
            ov.x = f2bf_pk(dA[0], dA[1]);
            ov.y = f2bf_pk(dA[2], dA[3]);
            *reinterpret_cast<uint2*>(a.OUT + (size_t)gr * V2_C + c4) = ov;
            if constexpr (!FAST) {
              if (++t == a.T) t = 0;
            }
          }
          __builtin_amdgcn_sched_barrier(0);       // rows in order: bounds the live temporaries
        }
      };
      if (fast) strip(std::true_type{});
      else strip(std::false_type{});
    }
  };
  const int G = (int)gridDim.x;
  if (!P2) {
    for (; tile < a.ntiles; tile += G) {
      __syncthreads();   // (1) the previous tile's stencil is done with Dt / Xs, its MFMAs with Pt
      if (Z3 && tile + G < a.ntiles) prefetch_g(tile + G);      // (published behind this tile's stencil)
      transform(tile, Xs);
      __syncthreads();   // (2)
      mfma();
      wgx(Xs);
      __syncthreads();   // (3)
      stencil(tile, Xs);
      if (Z3 && tile + G < a.ntiles) publish_g();     // nobody reads gus between barrier (2) and the next barrier (1)
    }
  } else if (tile < a.ntiles) {
    // prologue: tile 0 up to its dD rows; then per tile: [transform(next) + stencil(this)] | barrier | [MFMAs(next)] | barrier
    __syncthreads();
    transform(tile, Xs);
    if (Z3 && tile + G < a.ntiles) prefetch_g(tile + G);
    __syncthreads();
    mfma();
    if (Z3 && tile + G < a.ntiles) publish_g();       // gus(tile) was read before the barrier above
    __syncthreads();
    int buf = 0;
    for (; tile < a.ntiles; tile += G) {
      const int next = tile + G;
      if (next < a.ntiles) {
        transform(next, Xs + (buf ^ 1) * V6_R * V2_C);       // Pt: the MFMAs of `tile` finished before the last barrier
        if (Z3 && next + G < a.ntiles) prefetch_g(next + G);
      }
      stencil(tile, Xs + buf * V6_R * V2_C);
      __syncthreads();
      if (next < a.ntiles) {
        mfma();                                               // Dt: the stencil of `tile` finished before the barrier
        if (Z3 && next + G < a.ntiles) publish_g();           // gus(next) was read in the phase above
      }
      __syncthreads();
      buf ^= 1;
    }
  }
  if constexpr (WGX != 0) {      // keep the probe's accumulators alive
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += wgx_acc[0][r] + wgx_acc[1][r];
    if (t == 12345.678f) gb[0] += t;
  }
  // s2 was accumulated against the RAW x:  sum dA * xhat = rstd * sum dA*x - mean*rstd * sum dA
#pragma unroll
  for (int i = 0; i < 4; ++i) s2[i] = cst[6 * V2_C + c4 + i] * s2[i] - cst[5 * V2_C + c4 + i] * s1[i];
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);       // [8 waves][KD + 3][256] (inside the tiles; cst stays intact)
  {
    float* mine = red + (size_t)wave * (KD + 3) * V2_C + c4;
#pragma unroll
    for (int k = 0; k < KD; ++k) *reinterpret_cast<float4*>(mine + k * V2_C) = make_float4(gw[k][0], gw[k][1], gw[k][2], gw[k][3]);
    *reinterpret_cast<float4*>(mine + KD * V2_C) = make_float4(gb[0], gb[1], gb[2], gb[3]);
    *reinterpret_cast<float4*>(mine + (KD + 1) * V2_C) = make_float4(s1[0], s1[1], s1[2], s1[3]);
    *reinterpret_cast<float4*>(mine + (KD + 2) * V2_C) = make_float4(s2[0], s2[1], s2[2], s2[3]);
  }
  __syncthreads();
  const int rep = blockIdx.x % TN_NREP;
  for (int i = tid; i < (KD + 3) * V2_C; i += NT) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += red[(size_t)w * (KD + 3) * V2_C + i];
    const int k = i / V2_C, c = i % V2_C;
    if (k <= KD) atomic_add_f32(&a.gacc[(size_t)(rep * (KD + 1) + k) * V2_C + c], v);
    else if (a.bsumsX && HAS_MASK) atomic_add_f32(&a.bsumsX[(size_t)(rep * 2 + (k - KD - 1)) * V2_C + c], v);
  }
}


template <int FL, bool P2>
inline int launch_dgrad_dw_v6_t(DgradDwArgs a, int grid, size_t smem, hipStream_t st) {
  auto kern = a.bn.rm.len ? dgrad_dw_v6_kernel<FL, true, false, P2> : dgrad_dw_v6_kernel<FL, false, false, P2>;
  if constexpr (FL == 7 || FL == 3) {
    if (a.gu) kern = a.bn.rm.len ? dgrad_dw_v6_kernel<FL, true, true, P2> : dgrad_dw_v6_kernel<FL, false, true, P2>;
  } else if (a.gu) return -1000;
  TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(V2_NT), smem, st, a);
  return (int)hipGetLastError();
}
// -1000: no specialisation for this flag combination.  p2: the two-barrier schedule (dgrad_dw_v6_kernel<.., P2>): bit-identical,
// measured 45.5 vs 40.7 - 46.8 us in the harness and +0.1 ms on the step (the merged phase is longer than the two it replaces:
// the waves that finish their stencil strip early now wait for the slowest transform as well) - kept as an option, off
inline int launch_dgrad_dw_v6(DgradDwArgs a, int max_wgs, hipStream_t st, bool p2 = false) {
  if (!a.Wswz) return -1000;
  a.ntiles = (a.M + V6_OUT - 1) / V6_OUT;
  const int grid = a.ntiles < max_wgs ? a.ntiles : max_wgs;
  const size_t tiles = (size_t)(2 * V6_R * V2_AP + (p2 ? 2 : 1) * V6_R * V2_C) * sizeof(bf16_t);
  const size_t red = (size_t)8 * 6 * V2_C * sizeof(float);
  const size_t smem = (tiles > red ? tiles : red) + (size_t)(a.gu ? 12 + 4 : 10) * V2_C * sizeof(float);
  const int fl = (a.actX.mode != 0 ? 1 : 0) | (a.actX.relu ? 2 : 0) | (a.actX.drop_thr ? 4 : 0) | (a.ADD ? 8 : 0);
  // (Z3: the output activation carries dropout exactly when the input activation does — one model-wide rate)
  if (a.gu && ((a.act3.drop_thr != 0) != (a.actX.drop_thr != 0) || a.act3.mode == 0 || !a.act3.relu)) return -1000;
  if (p2) {
    switch (fl) {
      case 7: return launch_dgrad_dw_v6_t<7, true>(a, grid, smem, st);
      case 3: return launch_dgrad_dw_v6_t<3, true>(a, grid, smem, st);
      case 8: return launch_dgrad_dw_v6_t<8, true>(a, grid, smem, st);
      case 11: return launch_dgrad_dw_v6_t<11, true>(a, grid, smem, st);
      default: return -1000;
    }
  }
  switch (fl) {
    case 7: return launch_dgrad_dw_v6_t<7, false>(a, grid, smem, st);
    case 3: return launch_dgrad_dw_v6_t<3, false>(a, grid, smem, st);
    case 8: return launch_dgrad_dw_v6_t<8, false>(a, grid, smem, st);
    case 11: return launch_dgrad_dw_v6_t<11, false>(a, grid, smem, st);
    default: return -1000;
  }
}
