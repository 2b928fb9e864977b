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
  const bf16_t* Y;
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
};


template <int KD>
__global__ __launch_bounds__(V2_NT, 2) void wgrad_batched_v2_kernel(const WgradV2Desc* __restrict__ descs, int n_layers,
                                                                     int M, int T, int chunks_per_layer,
                                                                     int units_per_wg, int* __restrict__ part_count,
                                                                     uint64_t seed) {
  constexpr int PADR = (KD - 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Pt = reinterpret_cast<bf16_t*>(smem);                 // [32][288]
  bf16_t* Qt = Pt + WG2_RK * WG2_PITCH;                          // [32][288]
  bf16_t* Xa = Qt + WG2_RK * WG2_PITCH;                          // [32 + KD - 1][256]
  float* cst = reinterpret_cast<float*>(Xa + (WG2_RK + KD - 1) * V2_C);   // k0,k1,k2,sc,sh,bd,wd[KD] : [6 + KD][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vc = tid & 31, rq = tid >> 5, c0 = vc * 8;
  const int wa = wave >> 2, wb = wave & 3;
  const int total_units = n_layers * chunks_per_layer;
  int unit = blockIdx.x * units_per_wg;
  const int unit_end = min(total_units, unit + units_per_wg);

  while (unit < unit_end) {
    const int layer = unit / chunks_per_layer;
    const int chunk0 = unit % chunks_per_layer;
    const int nchunks = min(chunks_per_layer - chunk0, unit_end - unit);
    WgradV2Desc d = descs[layer];
    if (d.actX.drop_thr) d.actX.drop_key = tn_layer_key(seed, (uint32_t)d.drop_layer);
    const bool dw = d.wdw != nullptr;
    __syncthreads();
    if (tid < V2_C) {
      BnBwd bb;
      bb.fstats = d.fstats; bb.bsums = d.bsums; bb.gamma = d.gamma; bb.inv_n = d.inv_n; bb.eps = d.eps; bb.batch = d.batch;
      float k0, k1, k2, s, h;
      bn_bwd_coefs(bb, V2_C, tid, k0, k1, k2);
      bn_scale_shift(d.actX, V2_C, tid, s, h);
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

    uint4 pz[2], py[2], px[3];
    auto prefetch = [&](int chunk) {
      const int r0 = chunk * WG2_RK;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int gr = r0 + rq + 16 * q;
        if (gr < M) {
          pz[q] = *reinterpret_cast<const uint4*>(d.dZ + (size_t)gr * V2_C + c0);
          py[q] = *reinterpret_cast<const uint4*>(d.Y + (size_t)gr * V2_C + c0);
        } else {
          pz[q] = make_uint4(0, 0, 0, 0); py[q] = make_uint4(0, 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int i = rq + 16 * q;               // Xa row 0 .. 47 (need < 32 + KD - 1)
        const int gr = r0 - PADR + i;
        if (i < WG2_RK + KD - 1 && gr >= 0 && gr < M) px[q] = *reinterpret_cast<const uint4*>(d.X + (size_t)gr * V2_C + c0);
        else px[q] = make_uint4(0, 0, 0, 0);
      }
    };
    prefetch(chunk0);
    for (int ch = 0; ch < nchunks; ++ch) {
      const int r0 = (chunk0 + ch) * WG2_RK;
      __syncthreads();   // previous MFMA done with Pt/Qt; constants visible
      // ---- P tile: BN backward on load
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r = rq + 16 * q, gr = r0 + r;
        float z[8], y[8];
        unpack8(pz[q], z);
        unpack8(py[q], y);
        if (gr < M) {
#pragma unroll
          for (int i = 0; i < 8; ++i) z[i] = cst[c0 + i] * z[i] + cst[V2_C + c0 + i] * y[i] + cst[2 * V2_C + c0 + i];
        }
        store8(Pt + r * WG2_PITCH + c0, z);
      }
      // ---- activated input rows (with halo)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int i = rq + 16 * q, gr = r0 - PADR + i;
        if (i < WG2_RK + KD - 1) {
          float v[8];
          unpack8(px[q], v);
          if (gr >= 0 && gr < M) act8(v, cst + 3 * V2_C + c0, cst + 4 * V2_C + c0, d.actX, (uint32_t)gr, V2_C, c0);
          if (dw) store8(Xa + i * V2_C + c0, v);
          else if (i >= PADR && i < PADR + WG2_RK) store8(Qt + (i - PADR) * WG2_PITCH + c0, v);   // plain 1x1: Q row r = input row r
        }
      }
      if (ch + 1 < nchunks) prefetch(chunk0 + ch + 1);
      __syncthreads();
      if (dw) {
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
    // ---- partial slab of this (layer, segment)
    int part = 0;
    if (tid == 0) part = atomicAdd(&part_count[layer], 1);
    part = __shfl(part, 0, 64);
    __syncthreads();
    // broadcast `part` from wave 0 to the other waves through LDS (cst is free now)
    if (tid == 0) reinterpret_cast<int*>(cst)[0] = part;
    __syncthreads();
    part = reinterpret_cast<int*>(cst)[0];
    float* slab = d.slabs + (size_t)part * V2_C * V2_C;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = wa * 128 + i * 32 + cd_row(r, lane);
          const int ci = wb * 64 + j * 32 + (lane & 31);
          slab[(size_t)co * V2_C + ci] = acc[i][j][r];
        }
    unit += nchunks;
  }
}

// sum the partial slabs of every layer into its gradient tensor (fixed order per layer is not
// guaranteed across runs — parts are claimed by arrival — but each sum has <= ~6 terms)
struct WgradV2Out {
  const float* slabs;
  float* out;
};
__global__ void wgrad_v2_reduce_kernel(const WgradV2Out* __restrict__ outs, const int* __restrict__ part_count) {
  const WgradV2Out o = outs[blockIdx.y];
  const int parts = part_count[blockIdx.y];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V2_C * V2_C; i += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < parts; ++k) s += o.slabs[(size_t)k * V2_C * V2_C + i];
    o.out[i] = s;
  }
}
