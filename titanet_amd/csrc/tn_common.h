// titanet_amd — common device helpers for the gfx950 (MI355X / CDNA4) kernels.
//
// Layout convention used by every kernel in this library ("rows x channels"):
//   an activation tensor of the reference shape [B, C, T] (reference src/models.py:386-404)
//   is stored as a row-major matrix [M = B*T rows][C channels], row = b*T + t, channels
//   contiguous.  Pointwise (1x1) convs are then NT GEMMs with both operands K-contiguous
//   (what the MFMA fragment loads want), depthwise convs are whole-row shifts, and every
//   per-channel reduction (BatchNorm statistics, SE mean, attentive softmax over time)
//   is a lane-local loop with fully coalesced 16-byte accesses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#define TN_NREP 8          // replicated atomic accumulators per statistic (spreads same-address atomics)
#define TN_WAVE 64

typedef unsigned short bf16_t;   // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ------------------------------------------------------------------------------------------
// bf16 <-> f32
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// f32 -> bf16, round-to-nearest-even in hardware: v_cvt_pk_bf16_f32 converts TWO values per
// instruction (the bit-twiddled software rounding costs ~5 VALU ops per value, and these kernels are
// VALU-bound on exactly this kind of per-element work).
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) {
  f32x2_t v;
  v[0] = lo; v[1] = hi;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

template <typename AT> struct Elem;
template <> struct Elem<float> {
  static constexpr int BK = 32;     // K elements staged per chunk
  static constexpr int PAD = 1;     // LDS row padding (elements)
  static constexpr int KM = 2;      // K per MFMA (v_mfma_f32_32x32x2_f32)
  __device__ static __forceinline__ float to_f(float v) { return v; }
  __device__ static __forceinline__ float from_f(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int BK = 64;
  static constexpr int PAD = 8;
  static constexpr int KM = 16;     // v_mfma_f32_32x32x16_bf16
  __device__ static __forceinline__ float to_f(bf16_t v) { return bf2f(v); }
  __device__ static __forceinline__ bf16_t from_f(float v) { return f2bf(v); }
};

// 8 consecutive elements <-> 8 floats (global or LDS; p must be 16-byte aligned for bf16,
// 16-byte aligned for float as two float4).
__device__ __forceinline__ void load8(const float* p, float v[8]) {
  float4 a = *reinterpret_cast<const float4*>(p);
  float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float v[8]) {
  uint4 a = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
  v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void store8(float* p, const float v[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float v[8]) {
  uint4 a;
  a.x = f2bf_pk(v[0], v[1]);
  a.y = f2bf_pk(v[2], v[3]);
  a.z = f2bf_pk(v[4], v[5]);
  a.w = f2bf_pk(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = a;
}
// LDS row writes: the float tiles use an odd row stride (bank-conflict-free ds_read_b32 for the
// f32 MFMA fragments), so they are written element-wise; bf16 rows are 16-byte aligned.
__device__ __forceinline__ void store8_lds(float* p, const float v[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = v[i];
}
__device__ __forceinline__ void store8_lds(bf16_t* p, const float v[8]) { store8(p, v); }
__device__ __forceinline__ void load8_lds(const float* p, float v[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = p[i];
}
__device__ __forceinline__ void load8_lds(const bf16_t* p, float v[8]) { load8(p, v); }

// ------------------------------------------------------------------------------------------
// counter-based dropout (restated in numpy by oracle/rng.py for the parity tests)
// ------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t tn_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t tn_layer_key(uint64_t seed, uint32_t layer) {
  uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
  return tn_mix32(lo ^ tn_mix32(hi + layer * 0x9E3779B9u + 1u));
}
// Keep decisions for the 8 consecutive elements 8*idx8 .. 8*idx8+7 (one shared mixing round per group
// of 8, then one multiply-xorshift finaliser per element pair: 16 bits per element, keep iff >= thr).
// The kernels are VALU-bound on exactly this arithmetic, hence the shared round.
#define TN_DROP_C0 0x846ca68bu
#define TN_DROP_C1 0x9e3779b1u
#define TN_DROP_C2 0x85ebca77u
#define TN_DROP_C3 0xc2b2ae3du
__host__ __device__ __forceinline__ uint32_t tn_drop_shared(uint32_t idx8, uint32_t key) {
  uint32_t x = idx8 + key;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
  return x;
}
__host__ __device__ __forceinline__ uint32_t tn_drop_final(uint32_t x, uint32_t c) {
  uint32_t h = x * c;
  return h ^ (h >> 16);    // low 16 bits -> even element, high 16 bits -> odd element of the pair
}
// zero the dropped ones among 8 consecutive channels (no rescale: 1/(1-p) is folded into BN scale/shift)
__device__ __forceinline__ void tn_drop8(float v[8], uint32_t idx8, uint32_t key, uint32_t thr) {
  const uint32_t x = tn_drop_shared(idx8, key);
  const uint32_t h0 = tn_drop_final(x, TN_DROP_C0), h1 = tn_drop_final(x, TN_DROP_C1);
  const uint32_t h2 = tn_drop_final(x, TN_DROP_C2), h3 = tn_drop_final(x, TN_DROP_C3);
  v[0] = ((h0 & 0xffffu) >= thr) ? v[0] : 0.f; v[1] = ((h0 >> 16) >= thr) ? v[1] : 0.f;
  v[2] = ((h1 & 0xffffu) >= thr) ? v[2] : 0.f; v[3] = ((h1 >> 16) >= thr) ? v[3] : 0.f;
  v[4] = ((h2 & 0xffffu) >= thr) ? v[4] : 0.f; v[5] = ((h2 >> 16) >= thr) ? v[5] : 0.f;
  v[6] = ((h3 & 0xffffu) >= thr) ? v[6] : 0.f; v[7] = ((h3 >> 16) >= thr) ? v[7] : 0.f;
}
// the 4 channels 4*half .. 4*half+3 of the group
__device__ __forceinline__ void tn_drop4(float v[4], uint32_t idx8, uint32_t half, uint32_t key, uint32_t thr) {
  const uint32_t x = tn_drop_shared(idx8, key);
  const uint32_t h0 = tn_drop_final(x, half ? TN_DROP_C2 : TN_DROP_C0), h1 = tn_drop_final(x, half ? TN_DROP_C3 : TN_DROP_C1);
  v[0] = ((h0 & 0xffffu) >= thr) ? v[0] : 0.f; v[1] = ((h0 >> 16) >= thr) ? v[1] : 0.f;
  v[2] = ((h1 & 0xffffu) >= thr) ? v[2] : 0.f; v[3] = ((h1 >> 16) >= thr) ? v[3] : 0.f;
}
// the channel pair `pair` (0..3) of the group
__device__ __forceinline__ void tn_drop2(float v[2], uint32_t idx8, uint32_t pair, uint32_t key, uint32_t thr) {
  const uint32_t x = tn_drop_shared(idx8, key);
  const uint32_t c = pair == 0 ? TN_DROP_C0 : (pair == 1 ? TN_DROP_C1 : (pair == 2 ? TN_DROP_C2 : TN_DROP_C3));
  const uint32_t h = tn_drop_final(x, c);
  v[0] = ((h & 0xffffu) >= thr) ? v[0] : 0.f; v[1] = ((h >> 16) >= thr) ? v[1] : 0.f;
}
__host__ __device__ __forceinline__ bool tn_keep_elem(uint32_t e, uint32_t key, uint32_t thr) {
  const uint32_t x = tn_drop_shared(e >> 3, key);
  const uint32_t cs[4] = {TN_DROP_C0, TN_DROP_C1, TN_DROP_C2, TN_DROP_C3};
  const uint32_t h = tn_drop_final(x, cs[(e >> 1) & 3u]);
  return ((e & 1u) ? (h >> 16) : (h & 0xffffu)) >= thr;
}

// ------------------------------------------------------------------------------------------
// "activation on load": how a consumer turns a stored RAW conv output into the tensor the
// reference would have materialised: BatchNorm (batch or running statistics) -> ReLU ->
// Dropout (reference src/modules.py:119-133).  Raw outputs are stored once; normalisation,
// activation and the dropout mask are recomputed by every consumer (never materialised).
// ------------------------------------------------------------------------------------------
// Variable-length batches (BASELINE.json configs[3]; the reference pads and then ignores the lengths,
// src/datasets.py:63-73, src/learn.py:88): utterance b has len[b] valid frames, rows b*T + t with t >= len[b] are padding.
// The mask lives in the ACTIVATION: every consumer sees act(.) = 0 on padded rows, which is exactly the zero padding the
// convolutions apply at the end of an un-padded utterance; reductions (BatchNorm sums, SE mean, attentive softmax) skip
// padded rows and divide by the valid count.  len == null: every row is valid (the reference's semantics, bit for bit).
struct RowMask {
  const int* len;   // [B] valid frames per utterance, or null
  int T;
};
__device__ __forceinline__ bool tn_row_valid(const RowMask& m, uint32_t row) {
  if (!m.len) return true;
  const uint32_t b = row / (uint32_t)m.T;
  return (int)(row - b * (uint32_t)m.T) < m.len[b];
}

struct BnAct {
  // [TN_NREP][2][C]: sum, sum of squares over the M rows.  Train mode: accumulated by the producing
  // kernel's epilogue.  Eval mode: bn_eval_prepare_kernel writes the sums that reproduce the running
  // statistics (sum = mean*n, sumsq = (var + mean^2)*n), so device code has ONE normalisation path
  // (a second, running-stats path in these kernels was miscompiled by hipcc 7.2: the channel index
  // register was left undefined on that branch).
  const float* stats;
  const float* gamma;   // [C]
  const float* beta;    // [C]
  float inv_n;          // 1 / rows
  float eps;
  int mode;             // 0 identity, 1 BatchNorm from `stats`
  int relu;             // apply max(.,0)
  uint32_t drop_thr;    // 0 = no dropout; else round(p * 65536)
  uint32_t drop_key;    // tn_layer_key(seed, layer)
  const uint32_t* key_add;   // device word added to drop_key (the plan's per-step word: 0 unless tn_plan_step_tick drives the
                             // step from device memory, which is what lets a whole training step replay as ONE hipGraph); or null
  float inv_keep;       // 1 / (1 - p)
  RowMask rm;           // padded rows read as 0 (len == null: no mask)
};

// per-channel (mean, rstd) from the replicated batch sums
__device__ __forceinline__ void bn_mean_rstd(const BnAct& a, int C, int c, float& mean, float& rstd) {
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int r = 0; r < TN_NREP; ++r) { s += a.stats[(r * 2 + 0) * C + c]; q += a.stats[(r * 2 + 1) * C + c]; }
  mean = s * a.inv_n;
  const float var = fmaxf(q * a.inv_n - mean * mean, 0.f);
  rstd = rsqrtf(var + a.eps);
}
// scale/shift so that bn(x) = x*scale + shift
__device__ __forceinline__ void bn_scale_shift(const BnAct& a, int C, int c, float& sc, float& sh) {
  if (a.mode == 0) { sc = 1.f; sh = 0.f; return; }
  float mean, rstd;
  bn_mean_rstd(a, C, c, mean, rstd);
  sc = a.gamma[c] * rstd;
  sh = a.beta[c] - mean * sc;
  // dropout survivors are scaled by 1/(1-p): relu commutes with a positive scale, so fold it in here
  if (a.drop_thr) { sc *= a.inv_keep; sh *= a.inv_keep; }
}

__device__ __forceinline__ uint32_t tn_act_key(const BnAct& a) { return a.key_add ? a.drop_key + *a.key_add : a.drop_key; }

// the dropout key with the plan's per-step word folded in ONCE: act8 / act8_grad_mask otherwise dereference key_add (a global
// load) for every 8-element vector of a streaming loop
__device__ __forceinline__ BnAct tn_resolve_key(BnAct a) {
  if (a.drop_thr && a.key_add) { a.drop_key += *a.key_add; a.key_add = nullptr; }
  return a;
}

// apply act to 8 consecutive channels of row `row` (element index = row*C + c0 + i)
__device__ __forceinline__ void act8(float v[8], const float* sc, const float* sh, const BnAct& a,
                                     uint32_t row, int C, int c0) {
  if (a.mode != 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = v[i] * sc[i] + sh[i];
  }
  if (a.relu) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
  }
  if (a.drop_thr) tn_drop8(v, (row * (uint32_t)C + (uint32_t)c0) >> 3, tn_act_key(a), a.drop_thr);
  if (a.rm.len && !tn_row_valid(a.rm, row)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
  }
}
// mask-only variant for the backward pass: given the raw value's post-BN sign and the keep
// bits, returns the multiplier d(act)/d(bn output) for each of the 8 channels.
__device__ __forceinline__ void act8_grad_mask(const float raw[8], float m[8], const float* sc, const float* sh,
                                               const BnAct& a, uint32_t row, int C, int c0) {
  const float on = a.drop_thr ? a.inv_keep : 1.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float z = (a.mode != 0) ? raw[i] * sc[i] + sh[i] : raw[i];     // (sc, sh carry the 1/(1-p) factor: sign unchanged)
    m[i] = (!a.relu || z > 0.f) ? on : 0.f;
  }
  if (a.drop_thr) tn_drop8(m, (row * (uint32_t)C + (uint32_t)c0) >> 3, tn_act_key(a), a.drop_thr);
  if (a.rm.len && !tn_row_valid(a.rm, row)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------
// wave / block reductions
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// sum over a 256-thread block (4 waves); red = 4 floats of LDS
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  // hardware global_atomic_add_f32 (built with -munsafe-fp-atomics), no return value
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// e4m3 round trip of 8 values that share one power-of-two scale (experiment: what an fp8 operand would carry)
__device__ __forceinline__ void tn_e4m3_roundtrip8(float v[8], float s, float inv_s) {
#pragma unroll
  for (int i = 0; i < 8; i += 4) {
    uint32_t w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i] * inv_s, v[i + 1] * inv_s, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i + 2] * inv_s, v[i + 3] * inv_s, w, true);
    v[i] = __builtin_amdgcn_cvt_f32_fp8(w, 0) * s; v[i + 1] = __builtin_amdgcn_cvt_f32_fp8(w, 1) * s;
    v[i + 2] = __builtin_amdgcn_cvt_f32_fp8(w, 2) * s; v[i + 3] = __builtin_amdgcn_cvt_f32_fp8(w, 3) * s;
  }
}
// the power-of-two scale that maps a row maximum onto the e4m3 range (448 = largest finite e4m3)
__device__ __forceinline__ float tn_e4m3_row_scale(float amax) {
  return amax > 0.f ? exp2f(ceilf(log2f(amax * (1.f / 448.f)))) : 1.f;
}
// ---- fp8 data gradient (TN_PREC_FP8 plans, tn_pgemm.h F8 + rowexp): the BatchNorm-backward passes also write dS as e4m3
// bytes, each row scaled by its own power of two 2^-e, and e as an E8M0 byte (127 + e) — the per-row block scale operand
// of v_mfma_scale_f32_32x32x64_f8f6f4, so the scale costs the GEMM nothing.  The exponent bytes are stored in the order the
// GEMM's lanes read them: the four 32-row fragments a lane (wave row wm, fragment row fi) multiplies in a 256-row tile are
// rows 128 h + 64 wm + 32 mt + fi — their bytes form ONE dword at index 32 wm + fi of the tile's 256 bytes.
__host__ __device__ __forceinline__ size_t tn_rowexp_pos(size_t row) {
  const size_t tile = row >> 8;
  const unsigned r = (unsigned)(row & 255);
  return tile * 256 + (size_t)((((r >> 6) & 1) * 32 + (r & 31)) * 4 + ((r >> 7) * 2 + ((r >> 5) & 1)));
}
// e4m3 bytes of 8 values times inv_s
__device__ __forceinline__ uint2 tn_e4m3_pack8(const float v[8], float inv_s) {
  uint2 w = make_uint2(0u, 0u);
  w.x = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv_s, v[1] * inv_s, w.x, false);
  w.x = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv_s, v[3] * inv_s, w.x, true);
  w.y = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv_s, v[5] * inv_s, w.y, false);
  w.y = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv_s, v[7] * inv_s, w.y, true);
  return w;
}
// the E8M0 byte of a power-of-two scale (its biased float exponent)
__device__ __forceinline__ uint8_t tn_e8m0_of_pow2(float sc) { return (uint8_t)((__float_as_uint(sc) >> 23) & 0xffu); }
struct Fp8Rows { uint8_t* q; uint8_t* rowexp; };      // [M][C] e4m3 bytes, [ceil(M / 256) * 256] exponent bytes (tn_rowexp_pos); or nulls
// fp8 weight gradient (round 5): dS as e4m3 bytes with ONE power-of-two scale per COLUMN (a contraction over the rows only lets a
// per-column scale factor out).  Delayed scaling: the scale of column c comes from the maximum of |dS[:, c]| of the PREVIOUS
// backward of this plan (amax_prev), aimed at 56 = 448 / 8 (three bits of headroom for growth from one step to the next; values
// beyond are clamped to +-448); the pass that writes the bytes also takes this step's maxima (amax_cur, zeroed by the caller,
// atomicMax on the float bits) and the E8M0 byte per column that undoes the scale (cexp: the A block-scale operand of
// pgemm_tn_f8_batched_kernel).  q == null: off.
struct Fp8Cols { uint8_t* q; const float* amax_prev; float* amax_cur; uint8_t* cexp; int skip_bf16; };      // skip_bf16: both fp8 copies are the only ones read — the bf16 dS is not stored
// power-of-two scale for a column whose previous maximum was amax: 2^floor(log2(56 / amax)), exponent within +-60; 1 without history
__device__ __forceinline__ float tn_e4m3_col_scale(float amax) {
  // (a non-finite maximum — an overflowed step leaves +inf in the record, NaNs never enter it — is no history either: 56 / inf = 0
  //  would clamp to 2^-60 and turn every byte of the column into zero on the next backward without any signal)
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
  const float r = 56.f / amax;
  int e = (int)((__float_as_uint(r) >> 23) & 0xffu) - 127;
  e = e < -60 ? -60 : (e > 60 ? 60 : e);
  return __uint_as_float((uint32_t)(e + 127) << 23);
}
// e4m3 bytes of 8 values times s, clamped to the finite range (an out-of-range input of the converter is a NaN byte)
__device__ __forceinline__ uint2 tn_e4m3_pack8_cols(const float v[8], const float s[8]) {
  float w[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) w[u] = __builtin_amdgcn_fmed3f(v[u] * s[u], -448.f, 448.f);
  uint2 o = make_uint2(0u, 0u);
  o.x = __builtin_amdgcn_cvt_pk_fp8_f32(w[0], w[1], o.x, false);
  o.x = __builtin_amdgcn_cvt_pk_fp8_f32(w[2], w[3], o.x, true);
  o.y = __builtin_amdgcn_cvt_pk_fp8_f32(w[4], w[5], o.y, false);
  o.y = __builtin_amdgcn_cvt_pk_fp8_f32(w[6], w[7], o.y, true);
  return o;
}

#define TN_CHECK_HIP(expr)                          \
  do {                                              \
    hipError_t _e = (expr);                         \
    if (_e != hipSuccess) return (int)_e;           \
  } while (0)

// sum_{i < n} w[i * ws] * x[i * xs]: the loads of 16 terms are issued together (a plain loop with a run-time trip count is
// an L2 round trip per term — the SE mat-vecs of the generic tail kernels spent 60-80 us per utterance batch that way)
__device__ __forceinline__ float tn_dot_batched(const float* __restrict__ w, int ws, const float* x, int xs, int n) {
  float s = 0.f;
  for (int i0 = 0; i0 < n; i0 += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = (i0 + u < n) ? w[(size_t)(i0 + u) * ws] : 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s = fmaf(v[u], (i0 + u < n) ? x[(i0 + u) * xs] : 0.f, s);
  }
  return s;
}

// ------------------------------------------------------------------------------------------
// helpers of the slab kernels of the wide models (dw_bwd_slab, dw_fwd_slab): LDS-DMA, 2 / 4 channels per lane
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) char tn_lds_char;
// wave-uniform 32-bit load through the scalar cache (lgkmcnt).  A compiler-visible VECTOR load inside these loops would be
// waited for with vmcnt(0) — behind the next tile's LDS-DMA, which hipcc does not see (in-order counter): serialised
__device__ __forceinline__ int tn_sload_i32(const int* base, int index) {
  int v;
  const int off = __builtin_amdgcn_readfirstlane(index) * 4;
  asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(base), "s"(off) : "memory");
  return v;
}
// POL: 0 default cache policy, 1 nt (streamed once), 2 sc1, 3 sc0 sc1
template <int POL = 0>
__device__ __forceinline__ void tn_dma16(const void* gptr, unsigned lds_addr) {
  unsigned keep;
  // hidden from hipcc's waitcnt bookkeeping (cdna_hip_programming.md: M0 written in the statement that reads it)
  if constexpr (POL == 1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(lds_addr) : "memory");
  else if constexpr (POL == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(lds_addr) : "memory");
  else if constexpr (POL == 3)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc0 sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(lds_addr) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(lds_addr) : "memory");
}
// ------------------------------------------------------------------------------------------
// Padding mask of a tile of <= T consecutive rows (variable-length batches): the tile touches at most two utterances, so
// three wave-uniform numbers decide every row:  valid(gr) = gr < e0 ? gr < lim0 : gr < lim1
//   e0 = first row of the second utterance, lim0 / lim1 = first padding row of the first / second one.
// ------------------------------------------------------------------------------------------
struct TileMask { int e0, lim0, lim1; };
__device__ __forceinline__ TileMask tn_tile_mask(const int* len, int T, int M, int row0) {
  const int r0 = row0 < 0 ? 0 : row0;
  const int b0 = r0 / T, nb = M / T;
  TileMask m;
  m.e0 = (b0 + 1) * T;
  m.lim0 = b0 * T + tn_sload_i32(len, b0 < nb ? b0 : nb - 1);
  m.lim1 = m.e0 + tn_sload_i32(len, b0 + 1 < nb ? b0 + 1 : nb - 1);
  return m;
}
__device__ __forceinline__ bool tn_tile_valid(const TileMask& m, int gr) { return gr < m.e0 ? gr < m.lim0 : gr < m.lim1; }

// Σy, Σy² of a BatchNorm input take the padding rows of a variable-length batch out again: the fast kernels compute those
// rows from all-zero operands, so y == bias there exactly (rounded to bf16 where the statistics are taken from the stored
// values), `pad` rows of them.  One workgroup; replica 0 carries the correction.
template <int DUMMY = 0>
__global__ void stats_pad_fixup_kernel(float* __restrict__ stats, const float* __restrict__ bias, float pad, int C, int round_bf16) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float b = bias ? bias[c] : 0.f;
    if (round_bf16) b = __uint_as_float((uint32_t)f2bf(b) << 16);
    atomicAdd(&stats[c], -pad * b);
    atomicAdd(&stats[C + c], -pad * b * b);
  }
}

template <int FL, int CH>
__device__ __forceinline__ void act_c(float (&v)[CH], const float (&sc)[CH], const float (&sh)[CH], uint32_t key, uint32_t thr, uint32_t row, int C, int c) {
  if (FL & 1) {
#pragma unroll
    for (int i = 0; i < CH; ++i) v[i] = fmaf(v[i], sc[i], sh[i]);
  }
  if (FL & 2) {
#pragma unroll
    for (int i = 0; i < CH; ++i) v[i] = fmaxf(v[i], 0.f);
  }
  if (FL & 4) {
    const uint32_t e = row * (uint32_t)C + (uint32_t)c;
    if (CH == 4) tn_drop4(v, e >> 3, (uint32_t)(c >> 2) & 1u, key, thr);
    else tn_drop2(v, e >> 3, (uint32_t)(c >> 1) & 3u, key, thr);
  }
}
// CH consecutive bf16 channels <-> floats
template <int CH>
__device__ __forceinline__ void ld_ch(const bf16_t* p, float (&v)[CH]) {
  if (CH == 4) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  } else {
    const uint32_t r = *reinterpret_cast<const uint32_t*>(p);
    v[0] = __uint_as_float(r << 16); v[1] = __uint_as_float(r & 0xffff0000u);
  }
}
template <int CH>
__device__ __forceinline__ void st_ch(bf16_t* p, const float (&v)[CH]) {
  if (CH == 4) {
    uint2 o;
    o.x = f2bf_pk(v[0], v[1]); o.y = f2bf_pk(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = o;
  } else {
    *reinterpret_cast<uint32_t*>(p) = f2bf_pk(v[0], v[1]);
  }
}

// ------------------------------------------------------------------------------------------
// A chain of NK MFMAs (32x32x16 bf16) whose A fragments live in registers and whose B fragments are 16-byte LDS reads at
// b_addr + ks * STEP_B, HAND-SCHEDULED: hipcc sinks every fragment read to just in front of its MFMA
// (ds_read, lgkmcnt(0), MFMA: an exposed LDS round trip, ~4 MFMA slots, per 32-cycle instruction — in every GEMM phase of the
// hidden-256 kernels, round 5), whatever the source order says.  Here the reads are inline asm, D of them in flight, and the
// waits are counted (LDS operations of a wave retire in order).  Precondition: no LDS operation of this wave is outstanding
// on entry (the callers come from a barrier, which hipcc precedes with lgkmcnt(0)); on exit all reads have been waited for.
// The wait statement names the fragment as "+v" and is followed by sched_barrier(0): hipcc otherwise hoists the register-only
// MFMA over an inline-asm wait (cdna_hip_programming.md, pitfall 18).
// ------------------------------------------------------------------------------------------
// General form: NK k-steps x NB row blocks of B fragments (LDS address b_addr + n * BLK_B + ks * STEP_B, read j = ks * NB + n),
// every fragment feeding NA MFMAs (channel blocks): acc[a * NB + n] += wf[a * NK + ks] x b(ks, n).  wf / acc point at register
// arrays (every index is a compile-time constant after inlining).
template <int J, int NK, int NB, int NA, int D, int STEP_B, int BLK_B>
__device__ __forceinline__ void tn_mfma_sched_step(const bf16x8_t* wf, unsigned b_addr, bf16x8_t (&bq)[D], f32x16_t* acc) {
  constexpr int NR = NK * NB;
  if constexpr (J < NR) {
    constexpr int W = (NR - 1 - J) < (D - 1) ? (NR - 1 - J) : (D - 1);
    constexpr int ks = J / NB, n = J % NB;
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bq[J % D]) : "n"(W));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < NA; ++a) acc[a * NB + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a * NK + ks], bq[J % D], acc[a * NB + n], 0, 0, 0);
    if constexpr (J + D < NR) {
      constexpr int j2 = J + D;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[J % D]) : "v"(b_addr), "n"((j2 % NB) * BLK_B + (j2 / NB) * STEP_B));
    }
    tn_mfma_sched_step<J + 1, NK, NB, NA, D, STEP_B, BLK_B>(wf, b_addr, bq, acc);
  }
}
template <int J, int NB, int D, int STEP_B, int BLK_B>
__device__ __forceinline__ void tn_mfma_sched_fill(unsigned b_addr, bf16x8_t (&bq)[D]) {
  if constexpr (J < D) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[J]) : "v"(b_addr), "n"((J % NB) * BLK_B + (J / NB) * STEP_B));
    tn_mfma_sched_fill<J + 1, NB, D, STEP_B, BLK_B>(b_addr, bq);
  }
}
template <int NK, int NB, int NA, int D, int STEP_B, int BLK_B>
__device__ __forceinline__ void tn_mfma_sched_lds(const bf16x8_t* wf, const void* b_lds, f32x16_t* acc) {
  static_assert(D <= NK * NB && (NB - 1) * BLK_B + (NK - 1) * STEP_B < 65536, "fragment offsets are 16-bit immediates");
  const unsigned b_addr = (unsigned)(uintptr_t)(const tn_lds_char*)b_lds;
  bf16x8_t bq[D];
  tn_mfma_sched_fill<0, NB, D, STEP_B, BLK_B>(b_addr, bq);
  tn_mfma_sched_step<0, NK, NB, NA, D, STEP_B, BLK_B>(wf, b_addr, bq, acc);
  __builtin_amdgcn_sched_barrier(0);
}
// one row block, one channel block (the data-gradient phase of dgrad_dw_v6)
template <int NK, int D, int STEP_B>
__device__ __forceinline__ void tn_mfma_chain_lds(const bf16x8_t (&wf)[NK], const void* b_lds, f32x16_t& acc) {
  tn_mfma_sched_lds<NK, 1, 1, D, STEP_B, 0>(&wf[0], b_lds, &acc);
}

// ------------------------------------------------------------------------------------------
// Zero fill of a large region on the stream.  hipMemsetAsync runs the runtime's generic fill kernel (measured ~95 us per
// call on the 50 - 100 MB gradient buffers of TitaNet-M / -L: 0.3 ms of every step); this is a plain full-chip 16-byte store
// loop (~20 us for 100 MB).  Small or unaligned regions go to hipMemsetAsync.
// ------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void tn_zero_kernel(uint4* __restrict__ p, size_t nvec) {
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) p[i] = z;
}
inline hipError_t tn_zero_async(void* ptr, size_t bytes, hipStream_t st) {
  if (bytes < ((size_t)1 << 20) || ((uintptr_t)ptr & 15)) return hipMemsetAsync(ptr, 0, bytes, st);
  const size_t nvec = bytes / 16, tail = bytes - nvec * 16;
  const size_t want = (nvec + 255) / 256;
  hipLaunchKernelGGL(tn_zero_kernel, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, st, reinterpret_cast<uint4*>(ptr), nvec);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && tail) e = hipMemsetAsync(reinterpret_cast<char*>(ptr) + nvec * 16, 0, tail, st);
  return e;
}
