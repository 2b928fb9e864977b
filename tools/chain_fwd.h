// titanet_amd — the utterance-resident forward chain of the mega blocks (round 6), headline shape only:
// hidden = 256, depthwise K = 3, bf16, training, fixed-length batch, frames <= 320, batch <= the CU count.
//
// What it replaces: per mega block the launches sub_fwd_v4 (skip conv) + 3 x sub_fwd_v5 + se_combine_fwd_v3 stream the block's
// activations through HBM 14 times (reference src/models.py:457-472, src/modules.py:65-78, :119-148, :173-189): every
// train-mode BatchNorm needs the statistics of the WHOLE batch before its output can be consumed, and a kernel boundary was
// the only grid-wide synchronisation.  Here ONE persistent launch walks all mega blocks with one 512-thread workgroup per
// utterance (batch 256 = the 256 CUs of an MI355X), keeps the utterance ON THE CU between the BatchNorm points and meets the
// other workgroups in an XCD-hierarchical grid barrier there.  Per block the HBM traffic is what backward needs and nothing
// else: raw S, Y1, Y2, Y3, the kept depthwise outputs Q1..Q3 and the block output are WRITTEN once (8 passes), S is read back
// once by the combine (L2 / Infinity-Cache warm) — 9 passes instead of 14.3, 0 launches instead of 5.
//
// Where the utterance lives: 300 x 256 bf16 = 154 KB does not fit LDS next to the MFMA operand tiles, but it is less than a
// third of the CU's 512 KB register file.  Waves 0-3 (one per SIMD: the PRODUCERS, as in sub_fwd_v5) own it: thread (vc, rq)
// holds the 8-channel vector vc of the CH_ROWS consecutive frames rq * RP .. of the utterance — CH_NREG rows as packed bf16 in
// VGPRs, the rest in a private LDS slot per thread (conflict-free 16-byte accesses) — and runs activation + depthwise stencil
// from there (a frame's neighbours are the thread's own registers; the two frames at the ends of a thread's range come from a
// 2-row halo exchange through LDS per layer).  Waves 4-7 (the CONSUMERS) hold the layer's 256 x 256 weight as MFMA A fragments
// (128 VGPRs) and multiply 32-frame operand tiles the producers stage in LDS; a GEMM tile is the SET of frames
// {RP * rq + 4 s + q}: any 32 frames make an MFMA column block.  The raw output tile goes through an LDS staging tile to HBM
// (consumers: coalesced 16-byte stores + BatchNorm sums, as sub_fwd_v5) and back into the producers' rows IN PLACE (the
// stencil of step s + 1 only needs frames the window registers already hold).
//
// Numerics: every element goes through the same arithmetic in the same order as the kernels this replaces (activation on load
// rounded to bf16 before the stencil, accumulators started from the bias, k-steps 0..15, statistics from the bf16-rounded
// outputs, the combine's fused multiply-adds), so Y1 / Q1 / S are bit-identical; later tensors differ through the summation
// order of the float atomics of the statistics and of the SE mean (8 row ranges of 40 frames instead of 16 strided phases) —
// the same run-to-run noise two launches of the old kernels show.
//
// Inter-workgroup protocol (cdna_hip_programming.md Guideline 16, MI355X_MICROARCH.md "barrier-xcd"): the only data crossing
// CUs are the BatchNorm sums — agent-scope float atomics (performed at the memory side), every issuing wave drains them
// (vmcnt(0)) before the workgroup arrives; arrival = one returning agent-scope atomic on the counter of the workgroup's shard
// (blockIdx & 7: dispatch puts those on one XCD, correctness does not depend on it), the last arrival of a shard bumps the top
// counter, the last shard publishes the epoch to the 8 generation words; everyone else polls its shard's word relaxed with
// s_sleep and takes ONE agent acquire after the match.  Every spin is bounded (CH_TIMEOUT_TICKS of the 100 MHz clock): on
// expiry the workgroup sets the error word and leaves, the others follow at their next barrier — a workgroup that is not
// resident (batch > free CUs) costs a late, flagged step, not a hung GPU.  All barrier words are zeroed by the host before
// every launch.
#pragma once
#include <stddef.h>

#include <utility>

#include "../titanet_amd/csrc/tn_v2_kernels.h"

// compile-time loop: f(ch_c<0>{}) .. f(ch_c<N - 1>{}) — the producers index their register-resident rows with constants
template <int V> using ch_c = std::integral_constant<int, V>;
template <int... I, typename F>
__device__ __forceinline__ void ch_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(ch_c<I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void ch_for(F&& f) { ch_for_impl(std::make_integer_sequence<int, N>{}, f); }

#ifndef CH_DBG
#define CH_DBG 0      // tuning only: 1 no MFMA chain, 2 no Y stores
#endif
#define CH_NT 512
#define CH_R 32            // frames per GEMM tile
#define CH_ROWS 40         // frames per producer thread (8 row ranges x 40 = 320)
#define CH_STEPS (CH_ROWS / 4)
#ifndef CH_NREG
#define CH_NREG 24         // ... of which in the producer's VGPRs; the others in the thread's LDS slot, where the consumer
                           // thread of the same index reaches them (SE sums and combine are shared between the two teams)
#endif
#define CH_MAXT (8 * CH_ROWS)
#define CH_MINT (8 * (CH_ROWS - 4) + 1)      // the row ranges are a fixed 40 frames: shorter utterances take the per-layer launches
#define CH_TIMEOUT_TICKS 200000000ull      // 2 s of the 100 MHz wall clock

// LDS carve (bytes)
#define CH_QB_OFF 0
#define CH_QB_BYTES (2 * CH_R * V2_AP * 2)
#define CH_RED_OFF (CH_QB_OFF + CH_QB_BYTES / 2)      // statistics exchange [8][2][256] floats / SE partial sums [16][256]: alias the SECOND operand buffer
#define CH_CS_OFF (CH_QB_OFF + CH_QB_BYTES)
#define CH_CS_BYTES (2 * CH_R * V2_AP * 2)            // output staging tiles (double buffered); the SE phase: mean[256], gate[256], hidden[16]
#define CH_HALO_OFF (CH_CS_OFF + CH_CS_BYTES)
#define CH_HALO_BYTES (8 * 2 * V2_C * 2)
#define CH_CST_OFF (CH_HALO_OFF + CH_HALO_BYTES)
#define CH_CST_BYTES (9 * V2_C * 4)      // rows: 0 sc, 1 sh, 2 scS, 3 shS, 4..6 depthwise taps, 7 depthwise bias, 8 pointwise bias
#define CH_FLAG_OFF (CH_CST_OFF + CH_CST_BYTES)
#define CH_HL_OFF (CH_FLAG_OFF + 16)
#define CH_HL_BYTES ((CH_ROWS - CH_NREG) * 256 * 16)
#define CH_SMEM (CH_HL_OFF + CH_HL_BYTES)
static_assert(CH_SMEM <= 160 * 1024, "LDS budget");
static_assert(16 * V2_C * 4 <= CH_QB_BYTES / 2, "SE partial sums fit the second operand buffer");

// grid barrier words (unsigned), 128 bytes apart
#define CH_BAR_CNT(x) ((x) * 32)
#define CH_BAR_TOP (8 * 32)
#define CH_BAR_GEN(x) (9 * 32 + (x) * 32)
#define CH_BAR_ERR (17 * 32)
#define CH_BAR_WORDS (18 * 32)

struct ChainLayer {      // one pointwise conv (+ the depthwise conv in front of it) and the BatchNorm behind it
  const uint4* Wswz;     // 256 x 256 bf16 in MFMA-fragment order (swizzle256_kernel)
  const float* bias;     // [256]
  bf16_t* Y;             // [B*T][256] raw output
  float* stats;          // [TN_NREP][2][256], zeroed by the host
  const float* gamma;
  const float* beta;
  const float* wdw;      // [256][3] or null (skip conv)
  const float* bdw;      // [256]
  bf16_t* Q;             // [B*T][256] kept depthwise output, or null
  uint32_t drop_key;     // tn_layer_key of the dropout behind this layer's BatchNorm (sub-blocks)
  uint32_t pad_;
};
struct ChainBlock {
  ChainLayer skip;
  ChainLayer sub[3];
  const float* se_w1;    // [16][256]
  const float* se_w2;    // [256][16]
  float* m_out; float* h_out; float* g_out;      // [B][256], [B][16], [B][256]
  bf16_t* OUT;           // [B*T][256] block output
  uint32_t out_key;      // dropout on the block output
  uint32_t pad_;
};
struct ChainArgs {
  const bf16_t* X0;      // input of block 0 [B*T][256]
  const float* x0_stats; const float* x0_gamma; const float* x0_beta;      // x0_mode 1: BatchNorm + ReLU on load (raw prolog output)
  int x0_mode;           // 0: stored activated
  const ChainBlock* blocks;
  int nblocks, B, T;
  float inv_n, eps, inv_keep;
  uint32_t drop_thr;     // 0: no dropout (the DROP template parameter must agree)
  const uint32_t* key_add;
  unsigned* bar;         // CH_BAR_WORDS words, zeroed before every launch
  unsigned long long* stamps;      // optional [nblocks][16] wall-clock stamps of workgroup 0 (harness), or null
};

typedef __attribute__((address_space(1))) unsigned ch_gu32;
#define CH_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// ONE lane.  false: timed out (error word set)
__device__ __forceinline__ bool ch_grid_barrier(unsigned* bar, unsigned epoch, int b, int B) {
  const int x = b & 7;
  const unsigned nx = (unsigned)((B - x + 7) / 8), nshards = (unsigned)(B < 8 ? B : 8);
  ch_gu32* cnt = (ch_gu32*)(bar + CH_BAR_CNT(x));
  ch_gu32* top = (ch_gu32*)(bar + CH_BAR_TOP);
  ch_gu32* gen = (ch_gu32*)(bar + CH_BAR_GEN(x));
  ch_gu32* err = (ch_gu32*)(bar + CH_BAR_ERR);
  const unsigned old = __hip_atomic_fetch_add(cnt, 1u, CH_RLX_AGENT);
  if (old == nx * epoch - 1u) {
    const unsigned t = __hip_atomic_fetch_add(top, 1u, CH_RLX_AGENT);
    if (t == nshards * epoch - 1u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) __hip_atomic_store((ch_gu32*)(bar + CH_BAR_GEN(i)), epoch, CH_RLX_AGENT);
    }
  }
  const unsigned long long t0 = wall_clock64();
  bool ok = true;
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load(gen, CH_RLX_AGENT) >= epoch) break;
    __builtin_amdgcn_s_sleep(2);
    if ((spins & 63u) == 63u) {
      if (__hip_atomic_load(err, CH_RLX_AGENT) != 0u) { ok = false; break; }
      if (wall_clock64() - t0 > CH_TIMEOUT_TICKS) { __hip_atomic_store(err, 0x80000000u | epoch, CH_RLX_AGENT); ok = false; break; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok;
}

__device__ __forceinline__ uint4 ch_pack8(const float v[8]) {
  uint4 w;
  w.x = f2bf_pk(v[0], v[1]); w.y = f2bf_pk(v[2], v[3]); w.z = f2bf_pk(v[4], v[5]); w.w = f2bf_pk(v[6], v[7]);
  return w;
}

template <typename P> using ch_gp = __attribute__((address_space(1))) P*;      // (an integer round trip loses the address space: flat loads)
template <typename P> __device__ __forceinline__ ch_gp<P> ch_g(P* p) { return (ch_gp<P>)p; }

// Table entries through the SCALAR cache.  The tables are written by the host before the launch; a compiler-visible load of
// them sits behind stores / atomics of this kernel and becomes a VECTOR load: its value is "divergent" (a waterfall loop around
// every buffer access built from it) and its wait is vmcnt(0) — behind every store in flight.
typedef __attribute__((ext_vector_type(4))) unsigned int ch_sq_t;
__device__ __forceinline__ ChainLayer ch_sload_layer(const ChainLayer* p) {
  static_assert(sizeof(ChainLayer) == 80, "five 16-byte scalar loads");
  ch_sq_t q0, q1, q2, q3, q4;
  const uint64_t base = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
  const uint64_t sb = ((uint64_t)hi << 32) | lo;
  asm volatile("s_load_dwordx4 %0, %5, 0x0\n\ts_load_dwordx4 %1, %5, 0x10\n\ts_load_dwordx4 %2, %5, 0x20\n\ts_load_dwordx4 %3, %5, 0x30\n\t"
               "s_load_dwordx4 %4, %5, 0x40\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(q0), "=&s"(q1), "=&s"(q2), "=&s"(q3), "=&s"(q4) : "s"(sb) : "memory");
  auto ptr = [](uint32_t l, uint32_t h) { return ((uint64_t)h << 32) | l; };
  ChainLayer L;
  L.Wswz = (const uint4*)ptr(q0[0], q0[1]); L.bias = (const float*)ptr(q0[2], q0[3]);
  L.Y = (bf16_t*)ptr(q1[0], q1[1]); L.stats = (float*)ptr(q1[2], q1[3]);
  L.gamma = (const float*)ptr(q2[0], q2[1]); L.beta = (const float*)ptr(q2[2], q2[3]);
  L.wdw = (const float*)ptr(q3[0], q3[1]); L.bdw = (const float*)ptr(q3[2], q3[3]);
  L.Q = (bf16_t*)ptr(q4[0], q4[1]); L.drop_key = q4[2]; L.pad_ = 0;
  return L;
}
struct ChainTail { const float* se_w1; const float* se_w2; float* m_out; float* h_out; float* g_out; bf16_t* OUT; uint32_t out_key; };
__device__ __forceinline__ ChainTail ch_sload_tail(const ChainBlock* p) {
  static_assert(sizeof(ChainBlock) == 4 * 80 + 56 && offsetof(ChainBlock, se_w1) == 320, "layout");
  ch_sq_t q0, q1, q2;
  typedef __attribute__((ext_vector_type(2))) unsigned int ch_sd_t;
  ch_sd_t q3;
  const uint64_t base = (uint64_t)p + 320;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
  const uint64_t sb = ((uint64_t)hi << 32) | lo;
  asm volatile("s_load_dwordx4 %0, %4, 0x0\n\ts_load_dwordx4 %1, %4, 0x10\n\ts_load_dwordx4 %2, %4, 0x20\n\ts_load_dwordx2 %3, %4, 0x30\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(q0), "=&s"(q1), "=&s"(q2), "=&s"(q3) : "s"(sb) : "memory");
  auto ptr = [](uint32_t l, uint32_t h) { return ((uint64_t)h << 32) | l; };
  ChainTail t;
  t.se_w1 = (const float*)ptr(q0[0], q0[1]); t.se_w2 = (const float*)ptr(q0[2], q0[3]);
  t.m_out = (float*)ptr(q1[0], q1[1]); t.h_out = (float*)ptr(q1[2], q1[3]);
  t.g_out = (float*)ptr(q2[0], q2[1]); t.OUT = (bf16_t*)ptr(q2[2], q2[3]);
  t.out_key = q3[0];
  return t;
}

// The consumer's MFMA chain of one 32-row tile (16 k-steps x 2 channel blocks, B fragments from LDS, 4 reads in flight; the
// schedule of tn_mfma_sched_lds) with a slice of independent VALU work behind every k-step: the statistics of the PREVIOUS tile
// ride in the shadow of the matrix pipe instead of running after it (the consumer is the only MFMA wave of its SIMD and its
// own VALU tail was serial: 1.24 us per tile alone, 0.85 of it the chain).  slice(ch_c<J>) must be register-only code.
// slice j of a tile's BatchNorm sums: row vector j / 4 (packed bf16, zeros for a frame past the utterance), channel pair j % 4.
// Inline asm: as plain arithmetic hipcc gathers all sixteen slices behind the chain.
struct ChStatSlice {
  uint4 (&raw4)[4];
  float (&st_s)[8];
  float (&st_q)[8];
  template <int J>
  __device__ __forceinline__ void operator()(ch_c<J>) {
    constexpr int q = J >> 2, pr = J & 3;
    const uint32_t w = pr == 0 ? raw4[q].x : (pr == 1 ? raw4[q].y : (pr == 2 ? raw4[q].z : raw4[q].w));
    float lo, hi;
    asm volatile("v_lshlrev_b32 %0, 16, %6\n\tv_and_b32 %1, 0xffff0000, %6\n\tv_add_f32 %2, %2, %0\n\tv_fmac_f32 %3, %0, %0\n\t"
                 "v_add_f32 %4, %4, %1\n\tv_fmac_f32 %5, %1, %1"
                 : "=&v"(lo), "=&v"(hi), "+v"(st_s[2 * pr]), "+v"(st_q[2 * pr]), "+v"(st_s[2 * pr + 1]), "+v"(st_q[2 * pr + 1]) : "v"(w));
  }
};
template <int J, typename F>
__device__ __forceinline__ void ch_mfma_step(const bf16x8_t* wf, unsigned b_addr, bf16x8_t (&bq)[4], f32x16_t* acc, F& slice) {
  if constexpr (J < 16) {
    constexpr int W = (15 - J) < 3 ? (15 - J) : 3;
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bq[J % 4]) : "n"(W));
    __builtin_amdgcn_sched_barrier(0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[J], bq[J % 4], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[16 + J], bq[J % 4], acc[1], 0, 0, 0);
    if constexpr (J + 4 < 16) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[J % 4]) : "v"(b_addr), "n"((J + 4) * 32));
    slice(ch_c<J>{});
    ch_mfma_step<J + 1>(wf, b_addr, bq, acc, slice);
  }
}
template <typename F>
__device__ __forceinline__ void ch_mfma_tile(const bf16x8_t* wf, const void* b_lds, f32x16_t* acc, F& slice) {
  const unsigned b_addr = (unsigned)(uintptr_t)(const tn_lds_char*)b_lds;
  bf16x8_t bq[4];
  asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(bq[0]) : "v"(b_addr));
  asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(bq[1]) : "v"(b_addr));
  asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(bq[2]) : "v"(b_addr));
  asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(bq[3]) : "v"(b_addr));
  ch_mfma_step<0>(wf, b_addr, bq, acc, slice);
  __builtin_amdgcn_sched_barrier(0);
}

// DROP: dropout behind the sub-block BatchNorms and on the block output (training with p > 0); STAMP: harness time stamps
template <bool DROP, bool STAMP>
__global__ __launch_bounds__(CH_NT, 2) void chain_fwd_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Qb = reinterpret_cast<bf16_t*>(smem + CH_QB_OFF);
  bf16_t* Cs = reinterpret_cast<bf16_t*>(smem + CH_CS_OFF);
  uint4* halo = reinterpret_cast<uint4*>(smem + CH_HALO_OFF);      // [8 ranges][2: first, last row][32 vectors], RAW rows
  float* cst = reinterpret_cast<float*>(smem + CH_CST_OFF);
  float* red = reinterpret_cast<float*>(smem + CH_RED_OFF);
  unsigned* flag = reinterpret_cast<unsigned*>(smem + CH_FLAG_OFF);
  uint4* hl = reinterpret_cast<uint4*>(smem + CH_HL_OFF);          // [CH_ROWS - CH_NREG][256 threads of a team]
  float* se_mean = reinterpret_cast<float*>(smem + CH_CS_OFF);     // SE phase (no pass is running): [256]
  float* se_gate = se_mean + V2_C;                                 // [256]
  float* se_hid = se_gate + V2_C;                                  // [16]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave < 4;
  const int ltid = tid & 255;
  const int vc = ltid & 31, rq = ltid >> 5, c0 = vc * 8;
  const int half = lane >> 5, cw = wave & 3;
  const int b = blockIdx.x, T = a.T, B = a.B;
  constexpr int RP = CH_ROWS;                    // frames per row range
  constexpr int NSTEP = CH_STEPS, NRS = CH_NREG / 4, NLS = NSTEP - NRS;      // steps: register rows, LDS rows
  const uint32_t kadd = (DROP && a.key_add) ? *a.key_add : 0u;
  const uint32_t dthr = a.drop_thr;
  const uint32_t rowbase = (uint32_t)b * (uint32_t)T;
  typedef __attribute__((ext_vector_type(4))) unsigned int ch_u32x4_t;
  const int ubytes = T * V2_C * (int)sizeof(bf16_t);
  constexpr int OOB = 0x7ffffff0;
  const int t_first = RP * rq;                   // first frame of this thread's row range
  uint4* myhl = hl + ltid;                       // the LDS-resident rows CH_NREG .. of range (vc, rq): slot (r - CH_NREG) * 256

  auto stamp = [&](int blk, int i) {
    if (STAMP) { if (b == 0 && tid == 0 && a.stamps) a.stamps[blk * 16 + i] = wall_clock64(); }
  };
  // harness only: per-iteration stamps of the passes of block 0 — thread 0 (a producer) and thread 256 (a consumer),
  // [pass 0..3][role][iteration 0..11][4: iteration start, before the barrier, after it, -]
  auto stamp2 = [&](int blk, int pass_id, int it, int k) {
    if (STAMP) {
      if (b == 0 && blk == 0 && (tid == 0 || tid == 256) && a.stamps) a.stamps[a.nblocks * 16 + (((pass_id * 2 + (tid >> 8)) * 12 + it) * 4 + k)] = wall_clock64();
    }
  };
  // BatchNorm scale / shift of channel c from the batch sums (bn_scale_shift's arithmetic)
  auto bn_consts = [&](const float* stats, const float* gamma, const float* beta, bool fold_keep, int c, float& sc, float& sh) {
    const ch_gp<const float> S = ch_g(stats);
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int r = 0; r < TN_NREP; ++r) { s += S[(r * 2 + 0) * V2_C + c]; q += S[(r * 2 + 1) * V2_C + c]; }
    const float mean = s * a.inv_n;
    const float var = fmaxf(q * a.inv_n - mean * mean, 0.f);
    const float rstd = rsqrtf(var + a.eps);
    sc = ch_g(gamma)[c] * rstd;
    sh = ch_g(beta)[c] - mean * sc;
    if (fold_keep) { sc *= a.inv_keep; sh *= a.inv_keep; }      // dropout survivors are scaled by 1 / (1 - p): folded in (bn_scale_shift)
  };
  // the grid barrier of epoch e.  false: leave
  auto grid_sync = [&](unsigned e) -> bool {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's statistics atomics (and stores) are performed
    __syncthreads();
    if (tid == 0) *flag = ch_grid_barrier(a.bar, e, b, B) ? 1u : 0u;
    __syncthreads();
    return *flag != 0u;
  };
  // statistics of a finished pass: the consumers' per-thread sums meet in `red`, then replicated atomics (as sub_fwd_v5)
  auto stats_atomics = [&](float* stats) {
    const int which = tid >> 8, c = tid & 255;
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) v += red[(r * 2 + which) * V2_C + c];
    __hip_atomic_fetch_add(ch_g(stats) + (size_t)((b % TN_NREP) * 2 + which) * V2_C + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // the combine of one row (se_combine_fwd_v3's arithmetic): y raw Y3 row, sv raw S row -> packed output row (zeros past the utterance)
  auto combine_row = [&](const uint4& yraw, const uint4& sraw, int t, const float (&sc3)[8], const float (&sh3)[8], const float (&scS)[8],
                         const float (&shS)[8], const float (&g)[8], uint32_t key3, uint32_t okey) -> uint4 {
    const uint32_t row = rowbase + (uint32_t)t;
    float sv[8], y[8], o[8];
    unpack8(sraw, sv);
    unpack8(yraw, y);
    act8_t<DROP ? 7 : 3>(y, sc3, sh3, key3, dthr, row, c0);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaxf(fmaf(sv[i], scS[i], fmaf(g[i], y[i], shS[i])), 0.f);
    if (DROP) tn_drop8(o, (row * (uint32_t)V2_C + (uint32_t)c0) >> 3, okey, dthr);
    return t < T ? ch_pack8(o) : make_uint4(0u, 0u, 0u, 0u);
  };
  // one row's share of the SE column sums
  auto sum_row = [&](const uint4& yraw, int t, const float (&sc3)[8], const float (&sh3)[8], uint32_t key3, float (&acc)[8]) {
    float v[8];
    unpack8(yraw, v);
    act8_t<DROP ? 7 : 3>(v, sc3, sh3, key3, dthr, rowbase + (uint32_t)t, c0);
    const bool ok = t < T;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += ok ? v[i] : 0.f;
  };

  if (producer) {
    // =====================================================================================================================
    // PRODUCERS: own the utterance.  The register-resident rows are walked by a RUN-TIME loop whose body always works on
    // held[0..3] and then rotates the 20 rows by 4 (40 register moves per step): a fully unrolled walk is ~150 KB of
    // straight-line code per mega block against a 64 KB instruction cache — measured: every phase ran at the speed of
    // instruction fetch.  After NRS rotations the rows are back in their natural order.
    // =====================================================================================================================
    uint4 held[CH_NREG];
    auto rotate4 = [&]() {
      const uint4 t0 = held[0], t1 = held[1], t2 = held[2], t3 = held[3];
#pragma unroll
      for (int i = 0; i + 4 < CH_NREG; ++i) held[i] = held[i + 4];
      held[CH_NREG - 4] = t0; held[CH_NREG - 3] = t1; held[CH_NREG - 2] = t2; held[CH_NREG - 1] = t3;
    };
    auto put_halo = [&]() {                      // first / last row of the range -> LDS (the neighbours read them behind a barrier)
      halo[(rq * 2 + 0) * 32 + vc] = held[0];
      halo[(rq * 2 + 1) * 32 + vc] = myhl[(CH_ROWS - CH_NREG - 1) * 256];
    };
    auto halo_left = [&]() -> uint4 { return halo[((rq > 0 ? rq - 1 : 0) * 2 + 1) * 32 + vc]; };      // (rq == 0: not a frame, masked by the caller)
    auto halo_right = [&]() -> uint4 { return halo[((rq < 7 ? rq + 1 : 7) * 2 + 0) * 32 + vc]; };
    // depthwise constants of a layer -> cst rows 4..7 (one channel per thread; read behind the next barrier)
    auto put_dw_consts = [&](const float* wdw, const float* bdw) {
#pragma unroll
      for (int k = 0; k < 3; ++k) cst[(4 + k) * V2_C + ltid] = ch_g(wdw)[(size_t)ltid * 3 + k];
      cst[7 * V2_C + ltid] = ch_g(bdw)[ltid];
    };

    // ---- prologue: the utterance's rows of the first block's input (never out of range: rows past the last frame re-read it)
    {
      const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.X0 + (size_t)rowbase * V2_C), 0, ubytes, 0x00020000);
      float s = 1.f, h = 0.f;
      if (a.x0_mode) bn_consts(a.x0_stats, a.x0_gamma, a.x0_beta, false, ltid, s, h);
      cst[ltid] = s; cst[V2_C + ltid] = h;
      if (a.nblocks > 0) { const ChainLayer L0 = ch_sload_layer(&a.blocks[0].sub[0]); put_dw_consts(L0.wdw, L0.bdw); }
      __syncthreads();                                                                 // [pro 1]
      float sc[8], sh[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { sc[i] = cst[c0 + i]; sh[i] = cst[V2_C + c0 + i]; }
      const bool bn = a.x0_mode != 0;
      auto prep = [&](const uint4& raw, int t) -> uint4 {
        float v[8];
        unpack8(raw, v);
        if (bn) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = fmaxf(fmaf(v[i], sc[i], sh[i]), 0.f);
        }
        const uint4 w = ch_pack8(v);
        return t < T ? w : make_uint4(0u, 0u, 0u, 0u);
      };
#pragma unroll 1
      for (int g = 0; g < NSTEP; ++g) {
        uint4 r4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          r4[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srdX, (min(t_first + 4 * g + q, T - 1) * V2_C + c0) * (int)sizeof(bf16_t), 0, 0));
        if (g < NRS) {
          // (the rotation puts group g at the top once all NRS groups are in)
          rotate4();
#pragma unroll
          for (int q = 0; q < 4; ++q) held[CH_NREG - 4 + q] = prep(r4[q], t_first + 4 * g + q);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) myhl[(4 * (g - NRS) + q) * 256] = prep(r4[q], t_first + 4 * g + q);
        }
      }
      put_halo();
      __syncthreads();                                                                 // [pro 2]
    }

    for (int blk = 0; blk < a.nblocks; ++blk) {
      const ChainBlock* Kp = a.blocks + blk;
      stamp(blk, 0);
      // ------------------------------------------------------------------------------------------------------------------
      // a pass = one pointwise conv over the utterance, NSTEP + 1 iterations: iteration `it` stages the operand tile of step
      // `it` (rows 4 it .. 4 it + 3 of every range) while the consumers multiply step it - 1, then takes the raw output of
      // step it - 1 back in place.  DW: BatchNorm + ReLU + dropout of each row on its way into the stencil window (ACT:
      // rows are raw outputs of the previous layer; each row once, rounded to bf16 as sub_fwd_v5's activation stage stores
      // it), depthwise stencil, kept output Q.  !DW: the skip conv (operand = the rows themselves, nothing comes back).
      // ------------------------------------------------------------------------------------------------------------------
      // (act is a RUN-TIME flag on purpose: as a template parameter the four rows of a step become one basic block, hipcc interleaves
      //  their dropout-hash chains and the 96 registers of resident rows no longer leave room: 157 VGPRs spilled, measured slower)
      auto pass = [&](const ChainLayer& L, bool act, uint32_t act_key, int pass_id, auto DWT) {
        constexpr bool DW = decltype(DWT)::value;
        float sc[8], sh[8], wd0[8], wd1[8], wd2[8], bd[8];
        float wa[8], wb[8];                       // frames r - 1, r of the running stencil window
        uint4 hr = make_uint4(0u, 0u, 0u, 0u);    // the right neighbour's first row (raw)
        const uint32_t key = act_key + kadd;
        // activation of a raw row at frame t (only valid frames count; everything else is the zero padding of the conv)
        auto act_row = [&](const uint4& raw, int t, bool inrange, float (&v)[8]) {
          unpack8(raw, v);
          if (act) {
            // (DROP: a row outside the utterance is dropped as a whole — a threshold no 16-bit hash reaches — instead of 8 selects)
            act8_t<DROP ? 7 : 3>(v, sc, sh, key, (DROP && !inrange) ? 0x10000u : dthr, rowbase + (uint32_t)t, c0);
            const uint4 w = ch_pack8(v);           // bf16 rounding of the activated value
            unpack8(w, v);
          }
          if ((!DROP || !act) && !inrange) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = 0.f;
          }
        };
        if constexpr (DW) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            sc[i] = cst[c0 + i]; sh[i] = cst[V2_C + c0 + i];
            wd0[i] = cst[4 * V2_C + c0 + i]; wd1[i] = cst[5 * V2_C + c0 + i]; wd2[i] = cst[6 * V2_C + c0 + i]; bd[i] = cst[7 * V2_C + c0 + i];
          }
          act_row(halo_left(), t_first - 1, rq > 0 && t_first - 1 < T, wa);
          act_row(held[0], t_first, t_first < T, wb);
          hr = halo_right();
          // (an opaque VALUE: left to itself hipcc turns `last ? halo : row` into a select of ADDRESSES and a flat load, whose wait
          //  is vmcnt(0) lgkmcnt(0) — behind every store in flight, once per step)
          asm volatile("" : "+v"(hr.x), "+v"(hr.y), "+v"(hr.z), "+v"(hr.w));
        }
        const __amdgpu_buffer_rsrc_t srdQ = __builtin_amdgcn_make_buffer_rsrc(DW ? L.Q + (size_t)rowbase * V2_C : (bf16_t*)nullptr, 0, (DW && L.Q) ? ubytes : 0, 0x00020000);
        // stencil + store of row `t` whose successor's raw row is nx (valid: inr)
        auto stencil_row = [&](const uint4& nx, bool inr, int t, bf16_t* dst) {
          float wc[8];
          act_row(nx, t + 1, inr && t + 1 < T, wc);
          float acc[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = fmaf(wd0[i], wa[i], bd[i]);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = fmaf(wd1[i], wb[i], acc[i]);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = fmaf(wd2[i], wc[i], acc[i]);
          const uint4 w = ch_pack8(acc);
          *reinterpret_cast<uint4*>(dst) = w;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ch_u32x4_t, w), srdQ, (t < T) ? (t * V2_C + c0) * (int)sizeof(bf16_t) : OOB, 0, 0);
#pragma unroll
          for (int i = 0; i < 8; ++i) { wa[i] = wb[i]; wb[i] = wc[i]; }
        };
        // ---- NSTEP + 2 iterations, ONE workgroup barrier each: iteration `it` stages the operand tile of step it (Qb[it & 1]) while the
        // consumers multiply step it - 1 into Cs[(it - 1) & 1] and store step it - 2 from Cs[it & 1], which is also where this
        // thread's raw output rows of step it - 2 come back from.
        // ---- the register rows
#pragma unroll 1
        for (int it = 0; it < NRS; ++it) {
          stamp2(blk, pass_id, it, 0);
          bf16_t* As = Qb + (it & 1) * CH_R * V2_AP + (4 * rq) * V2_AP + c0;
          if constexpr (DW) {
            uint4 n3 = myhl[0];                    // the successor of the last register row is the first LDS row
            asm volatile("" : "+v"(n3.x), "+v"(n3.y), "+v"(n3.z), "+v"(n3.w));
            const bool lastreg = it == NRS - 1;
            n3.x = lastreg ? n3.x : held[4].x; n3.y = lastreg ? n3.y : held[4].y; n3.z = lastreg ? n3.z : held[4].z; n3.w = lastreg ? n3.w : held[4].w;
            const int t0 = t_first + 4 * it;
            stencil_row(held[1], true, t0, As);
            stencil_row(held[2], true, t0 + 1, As + V2_AP);
            stencil_row(held[3], true, t0 + 2, As + 2 * V2_AP);
            stencil_row(n3, true, t0 + 3, As + 3 * V2_AP);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(As + q * V2_AP) = held[q];
          }
          rotate4();
          if constexpr (DW) {
            // the raw output of step it - 2 comes back in place: after it + 1 rotations its rows sit in held[NREG - 12 ..]
            if (it >= 2) {
              const bf16_t* Cb = Cs + (it & 1) * CH_R * V2_AP;
#pragma unroll
              for (int q = 0; q < 4; ++q) held[CH_NREG - 12 + q] = *reinterpret_cast<const uint4*>(Cb + (4 * rq + q) * V2_AP + c0);
            }
          }
          stamp2(blk, pass_id, it, 1);
          __syncthreads();                                                             // [I]
          stamp2(blk, pass_id, it, 2);
        }
        // ---- the LDS rows (+ the two iterations that only take the last outputs back)
#pragma unroll 1
        for (int it = NRS; it <= NSTEP + 1; ++it) {
          const int u = it - NRS;
          stamp2(blk, pass_id, it, 0);
          if (it < NSTEP) {
            bf16_t* As = Qb + (it & 1) * CH_R * V2_AP + (4 * rq) * V2_AP + c0;
            if constexpr (DW) {
              const int t0 = t_first + 4 * it;
              const bool lastlds = u == NLS - 1;
              uint4 n3 = myhl[(lastlds ? 4 * u + 3 : 4 * u + 4) * 256];      // (the last step's successor is the right neighbour's row)
              n3.x = lastlds ? hr.x : n3.x; n3.y = lastlds ? hr.y : n3.y; n3.z = lastlds ? hr.z : n3.z; n3.w = lastlds ? hr.w : n3.w;
              stencil_row(myhl[(4 * u + 1) * 256], true, t0, As);
              stencil_row(myhl[(4 * u + 2) * 256], true, t0 + 1, As + V2_AP);
              stencil_row(myhl[(4 * u + 3) * 256], true, t0 + 2, As + 2 * V2_AP);
              stencil_row(n3, lastlds ? rq < 7 : true, t0 + 3, As + 3 * V2_AP);
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(As + q * V2_AP) = myhl[(4 * u + q) * 256];
            }
          }
          if constexpr (DW) {
            // step it - 2: the last two register groups (the rows are back in their natural order), then the LDS rows
            const bf16_t* Cb = Cs + (it & 1) * CH_R * V2_AP;
            if (u == 0) {
#pragma unroll
              for (int q = 0; q < 4; ++q) held[CH_NREG - 8 + q] = *reinterpret_cast<const uint4*>(Cb + (4 * rq + q) * V2_AP + c0);
            } else if (u == 1) {
#pragma unroll
              for (int q = 0; q < 4; ++q) held[CH_NREG - 4 + q] = *reinterpret_cast<const uint4*>(Cb + (4 * rq + q) * V2_AP + c0);
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) myhl[(4 * (u - 2) + q) * 256] = *reinterpret_cast<const uint4*>(Cb + (4 * rq + q) * V2_AP + c0);
            }
          }
          stamp2(blk, pass_id, it, 1);
          __syncthreads();                                                             // [I]
          stamp2(blk, pass_id, it, 2);
        }
        if constexpr (DW) put_halo();            // RAW first / last rows for the next layer's window (visible behind [R1])
        __syncthreads();                                                               // [R1] the consumers' sums are in `red`
        stats_atomics(L.stats);
      };

      {
        const ChainLayer LK = ch_sload_layer(&Kp->skip);
        pass(LK, false, 0u, 0, std::false_type{});
      }
      stamp(blk, 1);
      // (a run-time loop: ONE copy of the pass in the instruction stream, not three)
      uint32_t prev_key = 0u;
#pragma unroll 1
      for (int j = 0; j < 3; ++j) {
        const ChainLayer L = ch_sload_layer(&Kp->sub[j]);
        pass(L, j > 0, prev_key, 1 + j, std::true_type{});
        prev_key = L.drop_key;
        stamp(blk, 2 + 2 * j);
        if (!grid_sync((unsigned)(blk * 3 + j + 1))) return;
        stamp(blk, 3 + 2 * j);
        if (j < 2) {
          // BatchNorm constants of this layer + the next layer's depthwise constants -> cst
          const ChainLayer Ln = ch_sload_layer(&Kp->sub[j + 1]);
          float s, h;
          bn_consts(L.stats, L.gamma, L.beta, DROP, ltid, s, h);
          cst[ltid] = s; cst[V2_C + ltid] = h;
          put_dw_consts(Ln.wdw, Ln.bdw);
          __syncthreads();                                                             // [C1]
        }
      }

      // ------------------------------------------------------------------------------------------------------------------
      // SE gate + residual combine (se_combine_fwd_v3's arithmetic): OUT = dropout(relu(BN(S) + g * act3(Y3)))
      // the register rows of every range here, the LDS-resident rows by the consumer thread of the same index
      // ------------------------------------------------------------------------------------------------------------------
      {
        const ChainLayer L3 = ch_sload_layer(&Kp->sub[2]);
        const ChainLayer LS = ch_sload_layer(&Kp->skip);
        const ChainTail TL = ch_sload_tail(Kp);
        const uint32_t okey = TL.out_key + kadd;
        float* part = red;                       // [16][256]
        {
          float s, h;
          bn_consts(L3.stats, L3.gamma, L3.beta, DROP, ltid, s, h);
          cst[ltid] = s; cst[V2_C + ltid] = h;
          bn_consts(LS.stats, LS.gamma, LS.beta, false, ltid, s, h);
          cst[2 * V2_C + ltid] = s; cst[3 * V2_C + ltid] = h;
        }
        // weights of the two mat-vecs (L2 hits), in flight across the column sums
        float w1a[4], w1b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { w1a[k] = ch_g(TL.se_w1)[(size_t)wave * V2_C + lane + 64 * k]; w1b[k] = ch_g(TL.se_w1)[(size_t)(wave + 8) * V2_C + lane + 64 * k]; }
        f32x4_t w2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w2[k] = *reinterpret_cast<ch_gp<const f32x4_t>>(ch_g(TL.se_w2) + (size_t)ltid * 16 + 4 * k);
        __syncthreads();                                                               // [S1]
        float sc3[8], sh3[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc3[i] = cst[c0 + i]; sh3[i] = cst[V2_C + c0 + i]; }
        const uint32_t key3 = L3.drop_key + kadd;
        {
          float acc[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll 1
          for (int g = 0; g < NRS; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sum_row(held[q], t_first + 4 * g + q, sc3, sh3, key3, acc);
            rotate4();
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) part[rq * V2_C + c0 + i] = acc[i];
        }
        __syncthreads();                                                               // [S2]
        {
          float s = 0.f;
#pragma unroll
          for (int k = 0; k < 16; ++k) s += part[k * V2_C + ltid];
          s *= 1.f / (float)T;
          se_mean[ltid] = s;
          ch_g(TL.m_out)[(size_t)b * V2_C + ltid] = s;
        }
        __syncthreads();                                                               // [S3]
        {
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) { s0 = fmaf(w1a[k], se_mean[lane + 64 * k], s0); s1 = fmaf(w1b[k], se_mean[lane + 64 * k], s1); }
          s0 = wave_sum(s0);
          s1 = wave_sum(s1);
          if (lane == 0) {
            s0 = fmaxf(s0, 0.f); s1 = fmaxf(s1, 0.f);
            se_hid[wave] = s0; se_hid[wave + 8] = s1;
            ch_g(TL.h_out)[(size_t)b * 16 + wave] = s0;
            ch_g(TL.h_out)[(size_t)b * 16 + wave + 8] = s1;
          }
        }
        __syncthreads();                                                               // [S4]
        {
          float s = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            s = fmaf(w2[k][0], se_hid[4 * k], s); s = fmaf(w2[k][1], se_hid[4 * k + 1], s);
            s = fmaf(w2[k][2], se_hid[4 * k + 2], s); s = fmaf(w2[k][3], se_hid[4 * k + 3], s);
          }
          const float gv = 1.f / (1.f + __expf(-s));
          se_gate[ltid] = gv;
          ch_g(TL.g_out)[(size_t)b * V2_C + ltid] = gv;
        }
        __syncthreads();                                                               // [S5]
        stamp(blk, 8);
        // ---- phase 2 on the register rows: 5 trips of 4 rows; the skip operand S streams in pairs of rows, two pairs ahead
        float scS[8], shS[8], g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          scS[i] = cst[2 * V2_C + c0 + i]; shS[i] = cst[3 * V2_C + c0 + i]; g[i] = se_gate[c0 + i];
          if (DROP) { scS[i] *= a.inv_keep; shS[i] *= a.inv_keep; g[i] *= a.inv_keep; }
        }
        const __amdgpu_buffer_rsrc_t srdS = __builtin_amdgcn_make_buffer_rsrc(LS.Y + (size_t)rowbase * V2_C, 0, ubytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc(TL.OUT + (size_t)rowbase * V2_C, 0, ubytes, 0x00020000);
        auto lds = [&](int t) -> uint4 {
          return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srdS, (min(t, T - 1) * V2_C + c0) * (int)sizeof(bf16_t), 0, 0));
        };
        auto sto = [&](const uint4& w, int t) {
          // (vector offset carries the row, scalar offset 0: the gfx950 store-data hazard of se_combine_fwd_v3)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ch_u32x4_t, w), srdO, t < T ? (t * V2_C + c0) * (int)sizeof(bf16_t) : OOB, 0, 0);
        };
        // (every store of the previous pass has long been acknowledged: the grid barrier drained them)
        uint4 sa0 = lds(t_first), sa1 = lds(t_first + 1), sb0 = lds(t_first + 2), sb1 = lds(t_first + 3);
#pragma unroll 1
        for (int gi = 0; gi < NRS; ++gi) {
          const int t0 = t_first + 4 * gi;
          // loads retire in order among themselves (not against the stores in between): the wait that holds is "at most the
          // NEWER LOADS outstanding" — conservative while stores are pending (se_combine_fwd_v3)
          asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          const uint4 o0 = combine_row(held[0], sa0, t0, sc3, sh3, scS, shS, g, key3, okey);
          const uint4 o1 = combine_row(held[1], sa1, t0 + 1, sc3, sh3, scS, shS, g, key3, okey);
          __builtin_amdgcn_sched_barrier(0);
          sa0 = lds(t0 + 4); sa1 = lds(t0 + 5);      // (past the last trip: re-reads of valid rows, never consumed)
          sto(o0, t0); sto(o1, t0 + 1);
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // 2 newer loads + 2 stores at most: the pair B has landed
          __builtin_amdgcn_sched_barrier(0);
          const uint4 o2 = combine_row(held[2], sb0, t0 + 2, sc3, sh3, scS, shS, g, key3, okey);
          const uint4 o3 = combine_row(held[3], sb1, t0 + 3, sc3, sh3, scS, shS, g, key3, okey);
          __builtin_amdgcn_sched_barrier(0);
          sb0 = lds(t0 + 6); sb1 = lds(t0 + 7);
          sto(o2, t0 + 2); sto(o3, t0 + 3);
          held[0] = o0; held[1] = o1; held[2] = o2; held[3] = o3;
          rotate4();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                               // [S6] the consumers' rows are in their slots
        put_halo();
        if (blk + 1 < a.nblocks) { const ChainLayer Ln = ch_sload_layer(&a.blocks[blk + 1].sub[0]); put_dw_consts(Ln.wdw, Ln.bdw); }
        __syncthreads();                                                               // [S7]
        stamp(blk, 9);
      }
    }
  } else {
    // =====================================================================================================================
    // CONSUMERS: the layer's weight in registers, MFMA, output tile -> HBM + statistics; their share of SE + combine
    // =====================================================================================================================
    bf16x8_t wf[2][16];
    auto load_weights = [&](const uint4* W) {
#pragma unroll
      for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) wf[cbk][ks] = __builtin_bit_cast(bf16x8_t, ch_g(W)[((size_t)(cw * 2 + cbk) * 16 + ks) * 64 + lane]);
    };
    if (a.nblocks > 0) { const ChainLayer L0 = ch_sload_layer(&a.blocks[0].skip); load_weights(L0.Wswz); }
    __syncthreads();                                                                   // [pro 1]
    __syncthreads();                                                                   // [pro 2]
    for (int blk = 0; blk < a.nblocks; ++blk) {
      const ChainBlock* Kp = a.blocks + blk;
      // next: the layer whose weights are fetched as soon as this pass's last tile is out (null: none)
      auto pass = [&](const ChainLayer& L, const ChainLayer* next, int pass_id) {
        const __amdgpu_buffer_rsrc_t srdY = __builtin_amdgcn_make_buffer_rsrc(L.Y + (size_t)rowbase * V2_C, 0, ubytes, 0x00020000);
        float st_s[8], st_q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { st_s[i] = 0.f; st_q[i] = 0.f; }
        cst[8 * V2_C + ltid] = ch_g(L.bias)[ltid];      // the accumulators start from the bias (read per tile: 32 registers saved)
        // the weights were requested behind stores of the previous pass: loads and stores do not retire in one order
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float* bsrc = cst + 8 * V2_C + cw * 64 + 4 * half;
#pragma unroll 1
        for (int it = 0; it <= NSTEP + 1; ++it) {
          stamp2(blk, pass_id, it, 0);
          // ---- step it - 2: staged output tile -> HBM (the stores go out first), statistics from the same registers below
          uint4 raw4[4];
          if (it >= 2) {
            const bf16_t* Cb = Cs + (it & 1) * CH_R * V2_AP;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int o = rq + 8 * q;                       // tile row; frame = RP * (o >> 2) + 4 s + (o & 3)
              const int t = RP * (o >> 2) + 4 * (it - 2) + (o & 3);
              raw4[q] = *reinterpret_cast<const uint4*>(Cb + o * V2_AP + c0);
#if !(CH_DBG & 2)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ch_u32x4_t, raw4[q]), srdY, t < T ? (t * V2_C + c0) * (int)sizeof(bf16_t) : OOB, 0, 0);
#endif
              // (a frame past the utterance adds zeros to the sums: 4 selects here instead of a branch inside the MFMA chain)
              if (!(t < T)) raw4[q] = make_uint4(0u, 0u, 0u, 0u);
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) raw4[q] = make_uint4(0u, 0u, 0u, 0u);
          }
          ChStatSlice stat_slice{raw4, st_s, st_q};
          // ---- step it - 1: the product
          if (it >= 1 && it <= NSTEP) {
            const int s = it - 1;
            const bf16_t* As = Qb + (s & 1) * CH_R * V2_AP;
            bf16_t* Cw = Cs + (s & 1) * CH_R * V2_AP;
            f32x16_t acc[2];
#pragma unroll
            for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(bsrc + cbk * 32 + 8 * g);
                acc[cbk][4 * g] = bv.x; acc[cbk][4 * g + 1] = bv.y; acc[cbk][4 * g + 2] = bv.z; acc[cbk][4 * g + 3] = bv.w;
              }
            const bf16_t* brow = As + (lane & 31) * V2_AP + half * 8;
            // hand-scheduled: 4 fragment reads in flight (hipcc's own schedule is read - wait - 2 MFMAs with ONE fragment
            // register, and this wave is the only MFMA wave of its SIMD: the LDS round trip was exposed 16 times per tile)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !(CH_DBG & 1)
            ch_mfma_tile(&wf[0][0], brow, &acc[0], stat_slice);
#else
            ch_for<16>(stat_slice);
#endif
#pragma unroll
            for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int co = cw * 64 + cbk * 32 + 8 * g + 4 * half;
                uint2 w0;
                w0.x = f2bf_pk(acc[cbk][4 * g], acc[cbk][4 * g + 1]); w0.y = f2bf_pk(acc[cbk][4 * g + 2], acc[cbk][4 * g + 3]);
                *reinterpret_cast<uint2*>(Cw + (lane & 31) * V2_AP + co) = w0;
              }
          } else {
            // (the last iteration has no product to hide behind)
            ch_for<16>(stat_slice);
          }
          stamp2(blk, pass_id, it, 1);
          __syncthreads();                                                             // [I]
          stamp2(blk, pass_id, it, 2);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          red[(rq * 2 + 0) * V2_C + c0 + i] = st_s[i];
          red[(rq * 2 + 1) * V2_C + c0 + i] = st_q[i];
        }
        if (next) { const ChainLayer Ln = ch_sload_layer(next); load_weights(Ln.Wswz); }
        __syncthreads();                                                               // [R1]
        stats_atomics(L.stats);
      };
      {
        const ChainLayer LK = ch_sload_layer(&Kp->skip);
        pass(LK, &Kp->sub[0], 0);
      }
#pragma unroll 1
      for (int j = 0; j < 3; ++j) {
        const ChainLayer L = ch_sload_layer(&Kp->sub[j]);
        pass(L, j < 2 ? &Kp->sub[j + 1] : (const ChainLayer*)nullptr, 1 + j);
        if (!grid_sync((unsigned)(blk * 3 + j + 1))) return;
        if (j < 2) __syncthreads();                                                    // [C1]
      }
      {
        const ChainLayer L3 = ch_sload_layer(&Kp->sub[2]);
        const ChainLayer LS = ch_sload_layer(&Kp->skip);
        const ChainTail TL = ch_sload_tail(Kp);
        const uint32_t okey = TL.out_key + kadd;
        float* part = red;
        float w1a[4], w1b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { w1a[k] = ch_g(TL.se_w1)[(size_t)wave * V2_C + lane + 64 * k]; w1b[k] = ch_g(TL.se_w1)[(size_t)(wave + 8) * V2_C + lane + 64 * k]; }
        __syncthreads();                                                               // [S1]
        float sc3[8], sh3[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc3[i] = cst[c0 + i]; sh3[i] = cst[V2_C + c0 + i]; }
        const uint32_t key3 = L3.drop_key + kadd;
        {
          float acc[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll 1
          for (int g = 0; g < NLS; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sum_row(myhl[(4 * g + q) * 256], t_first + CH_NREG + 4 * g + q, sc3, sh3, key3, acc);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) part[(8 + rq) * V2_C + c0 + i] = acc[i];
        }
        __syncthreads();                                                               // [S2]
        __syncthreads();                                                               // [S3]
        {
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) { s0 = fmaf(w1a[k], se_mean[lane + 64 * k], s0); s1 = fmaf(w1b[k], se_mean[lane + 64 * k], s1); }
          s0 = wave_sum(s0);
          s1 = wave_sum(s1);
          if (lane == 0) {
            s0 = fmaxf(s0, 0.f); s1 = fmaxf(s1, 0.f);
            se_hid[wave] = s0; se_hid[wave + 8] = s1;
            ch_g(TL.h_out)[(size_t)b * 16 + wave] = s0;
            ch_g(TL.h_out)[(size_t)b * 16 + wave + 8] = s1;
          }
        }
        __syncthreads();                                                               // [S4]
        __syncthreads();                                                               // [S5]
        float scS[8], shS[8], g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          scS[i] = cst[2 * V2_C + c0 + i]; shS[i] = cst[3 * V2_C + c0 + i]; g[i] = se_gate[c0 + i];
          if (DROP) { scS[i] *= a.inv_keep; shS[i] *= a.inv_keep; g[i] *= a.inv_keep; }
        }
        const __amdgpu_buffer_rsrc_t srdS = __builtin_amdgcn_make_buffer_rsrc(LS.Y + (size_t)rowbase * V2_C, 0, ubytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc(TL.OUT + (size_t)rowbase * V2_C, 0, ubytes, 0x00020000);
        auto lds = [&](int t) -> uint4 {
          return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srdS, (min(t, T - 1) * V2_C + c0) * (int)sizeof(bf16_t), 0, 0));
        };
        auto sto = [&](const uint4& w, int t) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ch_u32x4_t, w), srdO, t < T ? (t * V2_C + c0) * (int)sizeof(bf16_t) : OOB, 0, 0);
        };
        const int tl0 = t_first + CH_NREG;
        uint4 sa0 = lds(tl0), sa1 = lds(tl0 + 1), sb0 = lds(tl0 + 2), sb1 = lds(tl0 + 3);
#pragma unroll 1
        for (int gi = 0; gi < NLS; ++gi) {
          const int t0 = tl0 + 4 * gi;
          uint4* slot = myhl + (4 * gi) * 256;
          asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          const uint4 o0 = combine_row(slot[0], sa0, t0, sc3, sh3, scS, shS, g, key3, okey);
          const uint4 o1 = combine_row(slot[256], sa1, t0 + 1, sc3, sh3, scS, shS, g, key3, okey);
          __builtin_amdgcn_sched_barrier(0);
          sa0 = lds(t0 + 4); sa1 = lds(t0 + 5);
          sto(o0, t0); sto(o1, t0 + 1);
          slot[0] = o0; slot[256] = o1;
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          const uint4 o2 = combine_row(slot[512], sb0, t0 + 2, sc3, sh3, scS, shS, g, key3, okey);
          const uint4 o3 = combine_row(slot[768], sb1, t0 + 3, sc3, sh3, scS, shS, g, key3, okey);
          __builtin_amdgcn_sched_barrier(0);
          sb0 = lds(t0 + 6); sb1 = lds(t0 + 7);
          sto(o2, t0 + 2); sto(o3, t0 + 3);
          slot[512] = o2; slot[768] = o3;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (blk + 1 < a.nblocks) { const ChainLayer Ln = ch_sload_layer(&a.blocks[blk + 1].skip); load_weights(Ln.Wswz); }
        __syncthreads();                                                               // [S6]
        __syncthreads();                                                               // [S7]
      }
    }
  }
}

// -1000: shape / flags outside the kernel (the caller runs the per-layer launches)
inline int launch_chain_fwd(const ChainArgs& a, bool stamp, hipStream_t st) {
  if (a.T > CH_MAXT || a.T < CH_MINT || a.B < 1 || a.nblocks < 1) return -1000;
  if ((size_t)a.B * a.T * V2_C * 2 >= ((size_t)1 << 31)) return -1000;
  auto kern = a.drop_thr ? (stamp ? chain_fwd_kernel<true, true> : chain_fwd_kernel<true, false>)
                         : (stamp ? chain_fwd_kernel<false, true> : chain_fwd_kernel<false, false>);
  TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_SMEM));
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(CH_NT), CH_SMEM, st, a);
  return (int)hipGetLastError();
}
