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
#include <utility>

#include "tn_v2_kernels.h"

// compile-time loop: f(ch_c<0>{}) .. f(ch_c<N - 1>{}) — the producers index their register-resident rows with constants
template <int V> using ch_c = std::integral_constant<int, V>;
template <int... I, typename F>
__device__ __forceinline__ void ch_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(ch_c<I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void ch_for(F&& f) { ch_for_impl(std::make_integer_sequence<int, N>{}, f); }

#define CH_NT 512
#define CH_R 32            // frames per GEMM tile
#define CH_ROWS 40         // frames per producer thread (8 row ranges x 40 = 320)
#define CH_STEPS (CH_ROWS / 4)
#ifndef CH_NREG
#define CH_NREG 24         // ... of which in VGPRs (the others in the thread's LDS slot)
#endif
#define CH_MAXT (8 * CH_ROWS)
#define CH_TIMEOUT_TICKS 200000000ull      // 2 s of the 100 MHz wall clock

// LDS carve (bytes)
#define CH_QB_OFF 0
#define CH_QB_BYTES (2 * CH_R * V2_AP * 2)
#define CH_CS_OFF (CH_QB_OFF + CH_QB_BYTES)
#define CH_CS_BYTES (CH_R * V2_AP * 2)
#define CH_HALO_OFF (CH_CS_OFF + CH_CS_BYTES)
#define CH_HALO_BYTES (8 * 2 * V2_C * 2)
#define CH_CST_OFF (CH_HALO_OFF + CH_HALO_BYTES)
#define CH_CST_BYTES (8 * V2_C * 4)      // rows: 0 sc, 1 sh, 2 scS, 3 shS, 4..6 depthwise taps, 7 depthwise bias
#define CH_RED_OFF (CH_CST_OFF + CH_CST_BYTES)
#define CH_RED_BYTES (8 * 2 * V2_C * 4)  // statistics exchange [8][2][256]; the SE phase: part[8][256], mean[256], gate[256], hidden[16]
#define CH_FLAG_OFF (CH_RED_OFF + CH_RED_BYTES)
#define CH_HL_OFF (CH_FLAG_OFF + 16)
#define CH_HL_BYTES ((CH_ROWS - CH_NREG) * 256 * 16)
#define CH_SMEM (CH_HL_OFF + CH_HL_BYTES)

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

// DROP: dropout behind the sub-block BatchNorms and on the block output (training with p > 0); STAMP: harness time stamps
template <bool DROP, bool STAMP>
__global__ __launch_bounds__(CH_NT, 2) void chain_fwd_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Qb = reinterpret_cast<bf16_t*>(smem + CH_QB_OFF);
  bf16_t* Cs = reinterpret_cast<bf16_t*>(smem + CH_CS_OFF);
  uint4* halo = reinterpret_cast<uint4*>(smem + CH_HALO_OFF);      // [8 ranges][2: first, last row][32 vectors]
  float* cst = reinterpret_cast<float*>(smem + CH_CST_OFF);
  float* red = reinterpret_cast<float*>(smem + CH_RED_OFF);
  unsigned* flag = reinterpret_cast<unsigned*>(smem + CH_FLAG_OFF);
  uint4* hl = reinterpret_cast<uint4*>(smem + CH_HL_OFF);          // [CH_ROWS - CH_NREG][256 producer threads]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave < 4;
  const int ltid = tid & 255;
  const int vc = ltid & 31, rq = ltid >> 5, c0 = vc * 8;
  const int half = lane >> 5, cw = wave & 3;
  const int b = blockIdx.x, T = a.T, B = a.B;
  const int RP = (((T + 7) >> 3) + 3) & ~3;       // frames per row range (multiple of 4, <= CH_ROWS)
  const int nsteps = RP >> 2;
  const uint32_t kadd = (DROP && a.key_add) ? *a.key_add : 0u;
  const uint32_t dthr = a.drop_thr;
  const uint32_t rowbase = (uint32_t)b * (uint32_t)T;
  typedef __attribute__((ext_vector_type(4))) unsigned int ch_u32x4_t;
  const int ubytes = T * V2_C * (int)sizeof(bf16_t);
  constexpr int OOB = 0x7ffffff0;

  auto stamp = [&](int blk, int i) {
    if (STAMP) { if (b == 0 && tid == 0 && a.stamps) a.stamps[blk * 16 + i] = wall_clock64(); }
  };
  // BatchNorm scale / shift of channel c from the batch sums (bn_scale_shift's arithmetic)
  auto bn_consts = [&](const float* stats, const float* gamma, const float* beta, bool fold_keep, int c, float& sc, float& sh) {
    BnAct t;
    t.stats = stats; t.gamma = gamma; t.beta = beta; t.inv_n = a.inv_n; t.eps = a.eps; t.mode = 1;
    t.drop_thr = fold_keep ? 1u : 0u; t.inv_keep = a.inv_keep;
    bn_scale_shift(t, V2_C, c, sc, sh);
  };
  // the grid barrier of epoch e; every wave has drained its atomics.  false: leave
  auto grid_sync = [&](unsigned e) -> bool {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    atomic_add_f32(&stats[(size_t)((b % TN_NREP) * 2 + which) * V2_C + c], v);
  };

  if (producer) {
    // =====================================================================================================================
    // PRODUCERS: own the utterance
    // =====================================================================================================================
    uint4 held[CH_NREG];
    uint4* myhl = hl + ltid;
    auto get_row = [&](auto R) -> uint4 {
      constexpr int r = decltype(R)::value;
      if constexpr (r < CH_NREG) return held[r];
      else return myhl[(r - CH_NREG) * 256];
    };
    auto set_row = [&](auto R, const uint4& v) {
      constexpr int r = decltype(R)::value;
      if constexpr (r < CH_NREG) held[r] = v;
      else myhl[(r - CH_NREG) * 256] = v;
    };
    const int t_first = RP * rq;                 // first frame of this thread's range
    // the first frame again, opaque to the optimiser: without it every per-row quantity of the unrolled row loops (frame index,
    // dropout counter, validity mask: 40 rows x several) is loop-invariant in the block loop, gets hoisted and spilled
    auto tf = [&]() -> int { int v = t_first; asm volatile("" : "+v"(v)); return v; };
    auto put_halo = [&]() {                      // first / last row of the range -> LDS (the neighbours read them behind a barrier)
      halo[(rq * 2 + 0) * 32 + vc] = get_row(ch_c<0>{});
      ch_for<CH_STEPS>([&](auto S) {
        constexpr int s = decltype(S)::value;
        if (s == nsteps - 1) halo[(rq * 2 + 1) * 32 + vc] = get_row(ch_c<4 * s + 3>{});
      });
    };
    auto halo_left = [&]() -> uint4 { return rq > 0 ? halo[((rq - 1) * 2 + 1) * 32 + vc] : make_uint4(0u, 0u, 0u, 0u); };
    auto halo_right = [&]() -> uint4 { return rq < 7 ? halo[((rq + 1) * 2 + 0) * 32 + vc] : make_uint4(0u, 0u, 0u, 0u); };

    // ---- depthwise constants of layer L -> cst rows 4..7 (one channel per thread; read behind the next barrier)
    auto put_dw_consts = [&](const ChainLayer& L) {
#pragma unroll
      for (int k = 0; k < 3; ++k) cst[(4 + k) * V2_C + ltid] = L.wdw[(size_t)ltid * 3 + k];
      cst[7 * V2_C + ltid] = L.bdw[ltid];
    };

    // ---- prologue: the utterance's rows of the first block's input (never out of range: rows past the last frame re-read it)
    {
      const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.X0 + (size_t)rowbase * V2_C), 0, ubytes, 0x00020000);
      ch_for<CH_ROWS>([&](auto R) {
        constexpr int r = decltype(R)::value;
        set_row(R, __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srdX, (min(t_first + r, T - 1) * V2_C + c0) * (int)sizeof(bf16_t), 0, 0)));
        if constexpr (r % 8 == 7) __builtin_amdgcn_sched_barrier(0);      // (8 rows in flight at a time: all 40 at once spill)
      });
      float s = 1.f, h = 0.f;
      if (a.x0_mode) bn_consts(a.x0_stats, a.x0_gamma, a.x0_beta, false, ltid, s, h);
      cst[ltid] = s; cst[V2_C + ltid] = h;
      if (a.nblocks > 0) put_dw_consts(a.blocks[0].sub[0]);
      __syncthreads();                                                                 // [pro 1]
      float sc[8], sh[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { sc[i] = cst[c0 + i]; sh[i] = cst[V2_C + c0 + i]; }
      const bool bn = a.x0_mode != 0;
      ch_for<CH_ROWS>([&](auto R) {
        constexpr int r = decltype(R)::value;
        float v[8];
        unpack8(get_row(R), v);
        if (bn) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = fmaxf(fmaf(v[i], sc[i], sh[i]), 0.f);
        }
        const bool ok = r < RP && t_first + r < T;
        const uint4 w = ch_pack8(v);
        set_row(R, ok ? w : make_uint4(0u, 0u, 0u, 0u));
        __builtin_amdgcn_sched_barrier(0);
      });
      put_halo();
      __syncthreads();                                                                 // [pro 2]
    }

    for (int blk = 0; blk < a.nblocks; ++blk) {
      const ChainBlock& K = a.blocks[blk];
      stamp(blk, 0);
      // ------------------------------------------------------------------------------------------------------------------
      // a pass = one pointwise conv over the utterance.  DW: depthwise stencil in front, kept output Q, the raw output
      // replaces the thread's rows.  !DW: the skip conv (operand = the rows themselves, nothing comes back).
      // ------------------------------------------------------------------------------------------------------------------
      auto pass = [&](const ChainLayer& L, auto DWT) {
        constexpr bool DW = decltype(DWT)::value;
        float wd0[8], wd1[8], wd2[8], bd[8];
        float wa[8], wb[8];                       // frames r - 1, r of the running stencil window
        if constexpr (DW) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { wd0[i] = cst[4 * V2_C + c0 + i]; wd1[i] = cst[5 * V2_C + c0 + i]; wd2[i] = cst[6 * V2_C + c0 + i]; bd[i] = cst[7 * V2_C + c0 + i]; }
          unpack8(halo_left(), wa);
          unpack8(get_row(ch_c<0>{}), wb);
        }
        const int tbp = tf();
        const __amdgpu_buffer_rsrc_t srdQ = __builtin_amdgcn_make_buffer_rsrc(DW ? L.Q + (size_t)rowbase * V2_C : (bf16_t*)nullptr, 0, (DW && L.Q) ? ubytes : 0, 0x00020000);
        auto produce = [&](auto S) {
          constexpr int s = decltype(S)::value;
          bf16_t* As = Qb + (s & 1) * CH_R * V2_AP;
          ch_for<4>([&](auto QI) {
            constexpr int q = decltype(QI)::value, r = 4 * s + q;
            bf16_t* dst = As + (4 * rq + q) * V2_AP + c0;
            if constexpr (DW) {
              float wc[8];
              uint4 nx;
              if constexpr (r + 1 >= CH_ROWS) nx = halo_right();
              else if constexpr (q == 3) nx = (s == nsteps - 1) ? halo_right() : get_row(ch_c<(r + 1 < CH_ROWS ? r + 1 : 0)>{});
              else nx = get_row(ch_c<(r + 1 < CH_ROWS ? r + 1 : 0)>{});
              unpack8(nx, wc);
              float acc[8];
#pragma unroll
              for (int i = 0; i < 8; ++i) acc[i] = fmaf(wd0[i], wa[i], bd[i]);
#pragma unroll
              for (int i = 0; i < 8; ++i) acc[i] = fmaf(wd1[i], wb[i], acc[i]);
#pragma unroll
              for (int i = 0; i < 8; ++i) acc[i] = fmaf(wd2[i], wc[i], acc[i]);
              const uint4 w = ch_pack8(acc);
              *reinterpret_cast<uint4*>(dst) = w;
              const int t = tbp + r;
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ch_u32x4_t, w), srdQ, (t < T) ? (t * V2_C + c0) * (int)sizeof(bf16_t) : OOB, 0, 0);
#pragma unroll
              for (int i = 0; i < 8; ++i) { wa[i] = wb[i]; wb[i] = wc[i]; }
            } else {
              *reinterpret_cast<uint4*>(dst) = get_row(ch_c<r>{});
            }
          });
        };
        produce(ch_c<0>{});
        __syncthreads();                                                               // [P1]
        ch_for<CH_STEPS>([&](auto S) {
          constexpr int s = decltype(S)::value;
          if (s < nsteps) {
            if constexpr (s + 1 < CH_STEPS) { if (s + 1 < nsteps) produce(ch_c<s + 1>{}); }
            __syncthreads();                                                           // [A]
            if constexpr (DW) {
              // the raw output rows of this thread come back in place
              ch_for<4>([&](auto QI) {
                constexpr int q = decltype(QI)::value;
                set_row(ch_c<4 * s + q>{}, *reinterpret_cast<const uint4*>(Cs + (4 * rq + q) * V2_AP + c0));
              });
            }
            __syncthreads();                                                           // [B]
          }
        });
        __syncthreads();                                                               // [R1] the consumers' sums are in `red`
        stats_atomics(L.stats);
      };
      // BatchNorm + ReLU (+ dropout) of the raw rows in place, rounded to bf16 (what sub_fwd_v5's activation stage stores)
      auto act_rows = [&](const ChainLayer& L, const ChainLayer* next_dw) {
        {
          float s, h;
          bn_consts(L.stats, L.gamma, L.beta, DROP, ltid, s, h);
          cst[ltid] = s; cst[V2_C + ltid] = h;
          if (next_dw) put_dw_consts(*next_dw);
        }
        __syncthreads();                                                               // [C1]
        float sc[8], sh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc[i] = cst[c0 + i]; sh[i] = cst[V2_C + c0 + i]; }
        const uint32_t key = L.drop_key + kadd;
        const int tb = tf();
        ch_for<CH_ROWS>([&](auto R) {
          constexpr int r = decltype(R)::value;
          float v[8];
          unpack8(get_row(R), v);
          const uint32_t row = rowbase + (uint32_t)(tb + r);
          act8_t<DROP ? 7 : 3>(v, sc, sh, key, dthr, row, c0);
          const bool ok = r < RP && tb + r < T;
          const uint4 w = ch_pack8(v);
          set_row(R, ok ? w : make_uint4(0u, 0u, 0u, 0u));
          __builtin_amdgcn_sched_barrier(0);
        });
        put_halo();
        __syncthreads();                                                               // [C2]
      };

      pass(K.skip, std::false_type{});
      stamp(blk, 1);
      // (a run-time loop: ONE copy of the unrolled pass in the instruction stream, not three)
#pragma unroll 1
      for (int j = 0; j < 3; ++j) {
        pass(K.sub[j], std::true_type{});
        stamp(blk, 2 + 3 * j);
        if (!grid_sync((unsigned)(blk * 3 + j + 1))) return;
        stamp(blk, 3 + 3 * j);
        if (j < 2) { act_rows(K.sub[j], &K.sub[j + 1]); stamp(blk, 4 + 3 * j); }
      }
      // ------------------------------------------------------------------------------------------------------------------
      // SE gate + residual combine (se_combine_fwd_v3's arithmetic): OUT = dropout(relu(BN(S) + g * act3(Y3)))
      // ------------------------------------------------------------------------------------------------------------------
      {
        float* part = red;                       // [8][256]
        float* mean = red + 8 * V2_C;            // [256]
        float* gs = red + 9 * V2_C;              // [256]
        float* hbuf = red + 10 * V2_C;           // [16]
        {
          float s, h;
          bn_consts(K.sub[2].stats, K.sub[2].gamma, K.sub[2].beta, DROP, ltid, s, h);
          cst[ltid] = s; cst[V2_C + ltid] = h;
          bn_consts(K.skip.stats, K.skip.gamma, K.skip.beta, false, ltid, s, h);
          cst[2 * V2_C + ltid] = s; cst[3 * V2_C + ltid] = h;
          if (blk + 1 < a.nblocks) put_dw_consts(a.blocks[blk + 1].sub[0]);
        }
        // weights of the two mat-vecs (L2 hits), in flight across the column sums
        float w1a[4], w1b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { w1a[k] = K.se_w1[(size_t)wave * V2_C + lane + 64 * k]; w1b[k] = K.se_w1[(size_t)(wave + 8) * V2_C + lane + 64 * k]; }
        float4 w2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w2[k] = *reinterpret_cast<const float4*>(K.se_w2 + (size_t)ltid * 16 + 4 * k);
        __syncthreads();                                                               // [S1]
        float sc3[8], sh3[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc3[i] = cst[c0 + i]; sh3[i] = cst[V2_C + c0 + i]; }
        const uint32_t key3 = K.sub[2].drop_key + kadd;
        {
          float acc[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = 0.f;
          const int tb = tf();
          ch_for<CH_ROWS>([&](auto R) {
            constexpr int r = decltype(R)::value;
            float v[8];
            unpack8(get_row(R), v);
            const uint32_t row = rowbase + (uint32_t)(tb + r);
            act8_t<DROP ? 7 : 3>(v, sc3, sh3, key3, dthr, row, c0);
            const bool ok = r < RP && tb + r < T;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += ok ? v[i] : 0.f;
            __builtin_amdgcn_sched_barrier(0);
          });
#pragma unroll
          for (int i = 0; i < 8; ++i) part[rq * V2_C + c0 + i] = acc[i];
        }
        __syncthreads();                                                               // [S2]
        {
          float s = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) s += part[k * V2_C + ltid];
          s *= 1.f / (float)T;
          mean[ltid] = s;
          K.m_out[(size_t)b * V2_C + ltid] = s;
        }
        __syncthreads();                                                               // [S3]
        {
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) { s0 = fmaf(w1a[k], mean[lane + 64 * k], s0); s1 = fmaf(w1b[k], mean[lane + 64 * k], s1); }
          s0 = wave_sum(s0);
          s1 = wave_sum(s1);
          if (lane == 0) {
            s0 = fmaxf(s0, 0.f); s1 = fmaxf(s1, 0.f);
            hbuf[wave] = s0; hbuf[wave + 8] = s1;
            K.h_out[(size_t)b * 16 + wave] = s0;
            K.h_out[(size_t)b * 16 + wave + 8] = s1;
          }
        }
        __syncthreads();                                                               // [S4]
        {
          float s = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            s = fmaf(w2[k].x, hbuf[4 * k], s); s = fmaf(w2[k].y, hbuf[4 * k + 1], s);
            s = fmaf(w2[k].z, hbuf[4 * k + 2], s); s = fmaf(w2[k].w, hbuf[4 * k + 3], s);
          }
          const float gv = 1.f / (1.f + __expf(-s));
          gs[ltid] = gv;
          K.g_out[(size_t)b * V2_C + ltid] = gv;
        }
        __syncthreads();                                                               // [S5]
        stamp(blk, 10);
        // (the rows stay PACKED across the mat-vecs: hipcc otherwise keeps phase 1's activated f32 values of every row for reuse
        //  in phase 2 — 8 registers per row, spilled; se_combine_fwd_v3 has the same cut)
#pragma unroll
        for (int r = 0; r < CH_NREG; ++r) asm volatile("" : "+v"(held[r].x), "+v"(held[r].y), "+v"(held[r].z), "+v"(held[r].w));
        // ---- phase 2 on the rows this thread holds; the skip operand S streams in two groups of 4 rows ahead
        float scS[8], shS[8], g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          scS[i] = cst[2 * V2_C + c0 + i]; shS[i] = cst[3 * V2_C + c0 + i]; g[i] = gs[c0 + i];
          if (DROP) { scS[i] *= a.inv_keep; shS[i] *= a.inv_keep; g[i] *= a.inv_keep; }
        }
        const uint32_t okey = K.out_key + kadd;
        const __amdgpu_buffer_rsrc_t srdS = __builtin_amdgcn_make_buffer_rsrc(K.skip.Y + (size_t)rowbase * V2_C, 0, ubytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc(K.OUT + (size_t)rowbase * V2_C, 0, ubytes, 0x00020000);
        uint4 rs[2][4];
        const int tb2 = tf();
        auto load_s = [&](auto GI) {
          constexpr int gi = decltype(GI)::value;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            rs[gi & 1][q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srdS, (min(tb2 + 4 * gi + q, T - 1) * V2_C + c0) * (int)sizeof(bf16_t), 0, 0));
        };
        // (every store of the previous pass has long been acknowledged: the grid barrier drained them)
        load_s(ch_c<0>{});
        ch_for<CH_STEPS>([&](auto GI) {
          constexpr int gi = decltype(GI)::value;
          if (gi < nsteps) {
            // loads retire in order among themselves (not against the stores in between): the wait that holds is "at most the
            // NEWER LOADS outstanding" — conservative while stores are pending (se_combine_fwd_v3)
            if constexpr (gi + 1 < CH_STEPS) {
              if (gi + 1 < nsteps) { load_s(ch_c<gi + 1>{}); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
              else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            ch_for<4>([&](auto QI) {
              constexpr int q = decltype(QI)::value, r = 4 * gi + q;
              const int t = tb2 + r;
              const uint32_t row = rowbase + (uint32_t)t;
              float sv[8], y[8], o[8];
              unpack8(rs[gi & 1][q], sv);
              unpack8(get_row(ch_c<r>{}), y);
              act8_t<DROP ? 7 : 3>(y, sc3, sh3, key3, dthr, row, c0);
#pragma unroll
              for (int i = 0; i < 8; ++i) o[i] = fmaxf(fmaf(sv[i], scS[i], fmaf(g[i], y[i], shS[i])), 0.f);
              if (DROP) tn_drop8(o, (row * (uint32_t)V2_C + (uint32_t)c0) >> 3, okey, dthr);
              const bool ok = t < T;
              const uint4 w = ok ? ch_pack8(o) : make_uint4(0u, 0u, 0u, 0u);
              set_row(ch_c<r>{}, w);
              // (vector offset carries the row, scalar offset 0: the gfx950 store-data hazard of se_combine_fwd_v3)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ch_u32x4_t, w), srdO, ok ? (t * V2_C + c0) * (int)sizeof(bf16_t) : OOB, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
            });
          }
        });
        put_halo();
        __syncthreads();                                                               // [S6]
        stamp(blk, 11);
      }
    }
  } else {
    // =====================================================================================================================
    // CONSUMERS: the layer's weight in registers, MFMA, output tile -> HBM + statistics
    // =====================================================================================================================
    bf16x8_t wf[2][16];
    float biasr[2][16];
    auto load_weights = [&](const ChainLayer& L) {
#pragma unroll
      for (int cbk = 0; cbk < 2; ++cbk) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) wf[cbk][ks] = __builtin_bit_cast(bf16x8_t, L.Wswz[((size_t)(cw * 2 + cbk) * 16 + ks) * 64 + lane]);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int j = 0; j < 4; ++j) biasr[cbk][4 * g + j] = L.bias[cw * 64 + cbk * 32 + 8 * g + 4 * half + j];
      }
    };
    if (a.nblocks > 0) load_weights(a.blocks[0].skip);
    __syncthreads();                                                                   // [pro 1]
    __syncthreads();                                                                   // [pro 2]
    for (int blk = 0; blk < a.nblocks; ++blk) {
      const ChainBlock& K = a.blocks[blk];
      // next: the layer whose weights are fetched as soon as this pass's last MFMA has issued (null: none)
      auto pass = [&](const ChainLayer& L, const ChainLayer* next) {
        const __amdgpu_buffer_rsrc_t srdY = __builtin_amdgcn_make_buffer_rsrc(L.Y + (size_t)rowbase * V2_C, 0, ubytes, 0x00020000);
        float st_s[8], st_q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { st_s[i] = 0.f; st_q[i] = 0.f; }
        // the weights were requested behind stores of the previous pass: loads and stores do not retire in one order
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                               // [P1]
        for (int s = 0; s < nsteps; ++s) {
          const bf16_t* As = Qb + (s & 1) * CH_R * V2_AP;
          {
            f32x16_t acc[2];
#pragma unroll
            for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[cbk][r] = biasr[cbk][r];
            const bf16_t* brow = As + (lane & 31) * V2_AP + half * 8;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
              const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(brow + ks * 16);
              acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][ks], b0, acc[0], 0, 0, 0);
              acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][ks], b0, acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int co = cw * 64 + cbk * 32 + 8 * g + 4 * half;
                uint2 w0;
                w0.x = f2bf_pk(acc[cbk][4 * g], acc[cbk][4 * g + 1]); w0.y = f2bf_pk(acc[cbk][4 * g + 2], acc[cbk][4 * g + 3]);
                *reinterpret_cast<uint2*>(Cs + (lane & 31) * V2_AP + co) = w0;
              }
          }
          __syncthreads();                                                             // [A]
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int o = rq + 8 * q;                       // tile row; frame = RP * (o >> 2) + 4 s + (o & 3)
            const int t = RP * (o >> 2) + 4 * s + (o & 3);
            const bool keep = t < T;
            const uint4 raw = *reinterpret_cast<const uint4*>(Cs + o * V2_AP + c0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ch_u32x4_t, raw), srdY, keep ? (t * V2_C + c0) * (int)sizeof(bf16_t) : OOB, 0, 0);
            if (keep) {
              float y[8];
              unpack8(raw, y);
#pragma unroll
              for (int i = 0; i < 8; ++i) { st_s[i] += y[i]; st_q[i] = fmaf(y[i], y[i], st_q[i]); }
            }
          }
          __syncthreads();                                                             // [B]
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          red[(rq * 2 + 0) * V2_C + c0 + i] = st_s[i];
          red[(rq * 2 + 1) * V2_C + c0 + i] = st_q[i];
        }
        if (next) load_weights(*next);
        __syncthreads();                                                               // [R1]
        stats_atomics(L.stats);
      };
      auto act_sync = [&]() { __syncthreads(); __syncthreads(); };                     // [C1] [C2]
      pass(K.skip, &K.sub[0]);
#pragma unroll 1
      for (int j = 0; j < 3; ++j) {
        pass(K.sub[j], j < 2 ? &K.sub[j + 1] : (blk + 1 < a.nblocks ? &a.blocks[blk + 1].skip : (const ChainLayer*)nullptr));
        if (!grid_sync((unsigned)(blk * 3 + j + 1))) return;
        if (j < 2) act_sync();
      }
      {
        float* mean = red + 8 * V2_C;
        float* hbuf = red + 10 * V2_C;
        float w1a[4], w1b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { w1a[k] = K.se_w1[(size_t)wave * V2_C + lane + 64 * k]; w1b[k] = K.se_w1[(size_t)(wave + 8) * V2_C + lane + 64 * k]; }
        __syncthreads();                                                               // [S1]
        __syncthreads();                                                               // [S2]
        __syncthreads();                                                               // [S3]
        {
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) { s0 = fmaf(w1a[k], mean[lane + 64 * k], s0); s1 = fmaf(w1b[k], mean[lane + 64 * k], s1); }
          s0 = wave_sum(s0);
          s1 = wave_sum(s1);
          if (lane == 0) {
            s0 = fmaxf(s0, 0.f); s1 = fmaxf(s1, 0.f);
            hbuf[wave] = s0; hbuf[wave + 8] = s1;
            K.h_out[(size_t)b * 16 + wave] = s0;
            K.h_out[(size_t)b * 16 + wave + 8] = s1;
          }
        }
        __syncthreads();                                                               // [S4]
        __syncthreads();                                                               // [S5]
        __syncthreads();                                                               // [S6]
      }
    }
  }
}

// -1000: shape / flags outside the kernel (the caller runs the per-layer launches)
inline int launch_chain_fwd(const ChainArgs& a, bool stamp, hipStream_t st) {
  if (a.T > CH_MAXT || a.T < 8 || a.B < 1 || a.nblocks < 1) return -1000;
  if ((size_t)a.B * a.T * V2_C * 2 >= ((size_t)1 << 31)) return -1000;
  auto kern = a.drop_thr ? (stamp ? chain_fwd_kernel<true, true> : chain_fwd_kernel<true, false>)
                         : (stamp ? chain_fwd_kernel<false, true> : chain_fwd_kernel<false, false>);
  TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_SMEM));
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(CH_NT), CH_SMEM, st, a);
  return (int)hipGetLastError();
}
