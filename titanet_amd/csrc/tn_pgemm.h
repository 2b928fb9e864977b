// titanet_amd — pipelined MFMA GEMMs for the wide models (hidden 512 / 1024: the pointwise 1x1 convs of reference
// src/modules.py:60-80 are MFMA-bound there) and for every other product whose operands are STORED bf16
// matrices.
//
//   pgemm_nt_kernel   C[M x N] = A[M x K] * W[N x K]^T          (forward / data-gradient form: both operands K-contiguous)
//
// Design (MI355X_MICROARCH.md / cdna_hip_programming.md 5): 256 x 256 output tile per 512-thread workgroup, 8 waves as
// 2 (M) x 4 (N), 128 x 64 per wave = 4 x 2 v_mfma_f32_32x32x16_bf16 tiles (128 accumulator registers).  Operand tiles go
// HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) into a ring of K-step stages
// (4 x 32 KB, K = 32 each); the DMA instructions of a later K step are issued BETWEEN the MFMAs of the current one (their
// issue cost hides under the matrix pipe) and waited for with a counted vmcnt.  LDS images are lane-linear (a DMA
// instruction writes base + lane * 16), so the bank swizzle is applied to the per-lane SOURCE address and undone by the
// fragment reads: image [256 rows][4 chunks of 16 B], chunk ^= (row >> 2) & 3 (conflict-free ds_read_b128 fragments).
// (Measured alternatives: 2 stages of K = 64 with one barrier per step, fragments read behind the barrier: 0.73 PFLOP/s on
// 76800 x 1024 x 1024 against 0.84 for 4 x K = 32 in the same loop shape.)
// Workgroups are PERSISTENT (one per CU): the K steps of all tiles of a workgroup form one stream, so the first operand
// tiles of tile t + 1 are in flight while tile t runs its epilogue, and the tile order keeps the column tiles of one row
// tile on one XCD (they share the A rows through that XCD's L2).
// Epilogue without LDS and without barriers: the DMA places weight row 2 i + j of a wave's 64 output channels at LDS row
// 32 j + i, so lane i of the two MFMA column blocks owns the ADJACENT channels 2 i, 2 i + 1 of its rows: bias, BatchNorm
// statistics (lane-local sums, one shuffle, replicated global atomics) and a packed 4-byte store per row — a wave
// instruction writes two full 128-byte lines.
// EVERY vector-memory instruction of the kernel is issued from inline asm: hipcc's waitcnt insertion does not see them, so
// nothing drains the DMA ring behind our back (a compiler-visible load anywhere in the loop made it put s_waitcnt vmcnt(0)
// in front of the fragment reads and of every output row), and the vmcnt queue is counted by hand: it retires in order, so
// the output stores of a tile (buffer stores: rows >= M are dropped by the descriptor's bounds check, 64 per wave whatever
// the tile) drain under the next tile's K steps while the waits name "at most 63 outstanding".
#pragma once
#include "tn_gemm.h"

typedef __attribute__((ext_vector_type(4))) int pg_i32x4_t;

// XCD-contiguous order of the persistent workgroups: block b runs on XCD b % 8 (observed; used for speed only)
__device__ __forceinline__ int pg_virtual_id(int b, int G) { return (G % 8 == 0) ? (b % 8) * (G / 8) + b / 8 : b; }

template <int N>
__device__ __forceinline__ void pg_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// hidden vector-memory operations (see the header comment)
__device__ __forceinline__ void pg_store_dword(uint32_t v, unsigned voff, pg_i32x4_t srd, unsigned soff) {
  asm volatile("buffer_store_dword %0, %1, %2, %3 offen" ::"v"(v), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void pg_store_dword_nt(uint32_t v, unsigned voff, pg_i32x4_t srd, unsigned soff) {
  asm volatile("buffer_store_dword %0, %1, %2, %3 offen nt" ::"v"(v), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void pg_atomic_add(float* p, float v) { asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ pg_i32x4_t pg_make_srd(const void* base, unsigned bytes) {
  const uint64_t b = (uint64_t)base;
  pg_i32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xffffu));      // stride 0: raw buffer
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}

// LDS-DMA through a buffer descriptor (rows outside the tensor read as zeros: the bounds check)
__device__ __forceinline__ void pg_dma16_buf(unsigned voff, pg_i32x4_t srd, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(srd), "s"(lds_addr) : "memory");
}

// the same with the non-temporal policy (tuning switch RW_POL bit 0 of rwgemm_k512_v2)
__device__ __forceinline__ void pg_dma16_buf_nt(unsigned voff, pg_i32x4_t srd, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(srd), "s"(lds_addr) : "memory");
}

// what an absent bias / column scale reads in the deep-ring variants (N <= 3072; one copy per translation unit)
static __device__ const float pg_const_zeros[3072] = {};
// ... and what absent row exponents read: E8M0 127 = scale 1 in every byte
#define PG_X4 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu
#define PG_X16 PG_X4, PG_X4, PG_X4, PG_X4
static __device__ const uint32_t pg_const_unit_exp[64] = {PG_X16, PG_X16, PG_X16, PG_X16};
#undef PG_X16
#undef PG_X4

struct PGemmNtArgs {
  const bf16_t* A;   // [M][lda], K contiguous, used as stored
  int lda;
  // variable-length batches: the 256-row tiles that hold at least one valid frame (ascending tile indices, device memory) or
  // null = all of them.  Row tiles that are padding only are not computed at all — their output rows keep whatever they held:
  // every consumer of this path masks or zero-weights padding rows
  const int* rowtiles;
  int n_rowtiles;
  // F8 only: one E8M0 exponent byte per row of A (a power-of-two scale per row, stored in the lanes' order: tn_rowexp_pos in
  // tn_common.h) — the A block scale of the scaled MFMA; null = unit scales
  const uint8_t* rowexp;
};
struct PGemmEpiArgs {
  bf16_t* Y;               // [M][ldy]
  int ldy;
  const float* bias;       // [N] or null
  float* stats;            // [TN_NREP][2][N] column sums / sums of squares of y over the rows < M, or null
  const float* colscale;   // [N] or null: y = acc * colscale[n] + bias[n]
  float pad_rows;          // variable-length batches: that many rows < M of A are all-zero padding (y == bias there, exactly):
                           // their contribution is taken out of `stats` again (sums over the valid rows only)
  int nt_out;              // the output is stored non-temporal: nobody reads it within the next few launches (the skip conv's
                           // output, the skip data gradient: round 6, Infinity-Cache management by store policy)
};

// DBG (tuning only): 1 no MFMA, 2 no DMA after the prologue, 4 linear DMA source (wrong results), 8 no output stores
//
// Loop of one workgroup: K steps of 32 in a ring of 4 stages, ONE barrier per K step; behind it the fragments of the first
// k-step are read, then per k-step: fragment reads of the next one, one A and one B DMA instruction of the K step three
// ahead, 8 MFMAs.  Measured on 76800 x 1024 x 1024 (uniform random operands: the matrix pipe alone, fed from registers,
// sustains 1.65 - 1.9 PFLOP/s on such data and 2.4 on zeros — the chip clocks to its power budget —, tools/mfma_peak.hip):
//   this loop 0.84 PFLOP/s (191 us; the generic gemm_nt_kernel 0.58 - 0.60); without DMA and stores 1.27, without MFMA
//   and stores the L2 -> LDS stream alone takes 129 us (1.26 GB at 9.8 TB/s: 256 x 256 tiles at K = 1024 cannot be fed
//   faster), the output stores cost 20 - 40 us (they queue in front of the next tile's DMA);
//   2 stages of K = 64: 0.73;  fragments prefetched across the barrier + mid-step barrier: 0.75;  the two wave groups one
//   phase apart ("ping-pong", a barrier per 8 MFMAs): 0.76 - 0.79, with s_setprio around the MFMAs 0.75;  16-byte output
//   stores (quad transpose of the packed pairs, 16 instead of 64 store instructions per wave and tile): 4 % slower —
//   the store cost is the burst itself (every workgroup reaches its epilogue at the same time), not the instruction count.
//
// F8 = true: both operands are e4m3 BYTE matrices (pa.A [M][lda] bytes, g.W [N][K] bytes; TN_PREC_FP8 forward: the e4m3 copy
// of the depthwise output and the per-row-scaled weights, ea.colscale undoes the scales).  Same ring, same 64-byte tile rows
// (now 64 k per stage), same swizzle and the same two 16-byte fragment reads per operand row and stage — a lane's two pieces
// (k bytes [16 h, 16 h + 16) and [32 + 16 h, 48 + 16 h) of the 64-k block, the same for both operands) are ONE 32-byte
// operand of v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales): 8 MFMAs of 64 k per stage instead of 16 of 16 k, half the
// DMA bytes and half the matrix-pipe time per unit of K.
// HT: 128-row halves per output tile (2: 256 x 256 tiles; 1: 128 x 256 — twice the tiles, for shapes whose 256-row tile count
// leaves a fifth of the chip idle in the last round, e.g. 600 tiles at 76800 x 512: 200 workgroups x 3 against 240 x 5)
// NS: ring stages (4, or 5 = all 160 KB of LDS: one more K step in flight.  The operand stream of this kernel runs at
// (bytes in flight) / (latency): 3 x 32 KB per CU at ~2.5 us under load = the measured ~38 GB/s per CU; bias / column scales
// then live in registers, fetched per tile by inline-asm loads that retire behind the ring's own waits)
template <int DBG = 0, bool F8 = false, int HT = 2, int NS = 4>
__global__ __launch_bounds__(512, 2) void pgemm_nt_kernel(GemmShape g, PGemmNtArgs pa, PGemmEpiArgs ea, int tiles_n, int total_tiles) {
  constexpr int BK = F8 ? 64 : 32, NSTAGE = NS, AHEAD = NS - 1;
  constexpr int ROWB = 64;                   // bytes of a tile row
  constexpr int CPR = ROWB / 16;             // 16-byte chunks per row
  constexpr int RPI = 64 / CPR;              // rows per DMA instruction (1 KiB)
  constexpr int TM = 128 * HT;               // rows of the output tile
  constexpr int NQ = 256 / (8 * RPI);        // DMA instructions per wave and stage: weight tile
  constexpr int NQA = TM / (8 * RPI);        // ... A tile
  constexpr int TILE_A = TM * ROWB, TILE_B = TILE_A, STAGE_B = TILE_A + 256 * ROWB;     // (TILE_B: offset of the weight tile in a stage)
  constexpr int GRP = NQ + NQA;              // DMA instructions per wave and K step
  constexpr int STW = 32 * HT - 1;           // one less than the output stores a wave issues per tile
  constexpr int SWS = 2, SWM = CPR - 1;      // swizzle: chunk ^= (row >> SWS) & SWM
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(tn_lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int G = gridDim.x, v = pg_virtual_id(blockIdx.x, G);
  const int KT = g.K / BK;
  constexpr int ESZ = F8 ? 1 : 2;            // operand bytes per element
  const char* Wb = reinterpret_cast<const char*>(g.W);
  const char* Ab = reinterpret_cast<const char*>(pa.A);

  // ---- DMA side: instruction q of this wave fills LDS rows (q*8 + wave) * RPI + lane / CPR of both tiles
  const int dsub = lane / CPR;
  unsigned offA[NQ], offB[NQ];
  const int* __restrict__ rowtiles = pa.rowtiles;
  auto dma_setup = [&](int tile) {
    const int mi = tile / tiles_n, nt = tile - mi * tiles_n;
    const int mt = rowtiles ? tn_sload_i32(rowtiles, mi) : mi;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = (q * 8 + wave) * RPI + dsub;                    // LDS row of the tile
      const int chunk = (DBG & 4) ? (lane % CPR) : ((lane % CPR) ^ ((p >> SWS) & SWM));
      int ra = mt * TM + p;
      int rb = nt * 256 + (p & ~63) + 2 * (p & 31) + ((p >> 5) & 1);    // weight rows: 2 i + j at LDS row 32 j + i
      ra = ra < g.M ? ra : g.M - 1;            // rows outside the matrix: any valid row (their outputs are never stored)
      rb = rb < g.N ? rb : g.N - 1;
      if (q < NQA) offA[q] = ((unsigned)ra * (unsigned)pa.lda) * ESZ + chunk * 16;       // bytes
      offB[q] = ((unsigned)rb * (unsigned)g.K) * ESZ + chunk * 16;
    }
  };
  // DBG 16 (experiment): every workgroup walks K from its own starting step (rows of a K-major operand are a power-of-two
  // stride apart: workgroups marching through K in lockstep ask the same L2 channels for every row at the same time)
  const int krot = (DBG & 16) ? ((v / tiles_n) * 5) % KT : 0;      // (the column tiles of one row tile keep one phase: they share the A rows through L2)
  auto kmap = [&](int kt) { const int k2 = kt + krot; return k2 >= KT ? k2 - KT : k2; };
  constexpr int POLA = (DBG & 32) ? 1 : 0, POLB = (DBG & 64) ? 1 : 0;      // experiments: nt on the A / W stream
  auto dma_a = [&](int q, int kt, int stage) {
    if (q < NQA) tn_dma16<POLA>(Ab + (size_t)offA[q] + kmap(kt) * ROWB, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + stage * STAGE_B + (q * 8 + wave) * 1024)));
  };
  auto dma_b = [&](int q, int kt, int stage) {
    tn_dma16<POLB>(Wb + (size_t)offB[q] + kmap(kt) * ROWB, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + stage * STAGE_B + TILE_B + (q * 8 + wave) * 1024)));
  };
  // ---- MFMA side: fragment byte offsets inside a tile (row i of a 32-row block, k-step ks)
  const int fi = lane & 31, fh = lane >> 5;
  int foff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) foff[ks] = fi * ROWB + ((((ks << 1) + fh) ^ ((fi >> SWS) & SWM)) << 4);
  const int abase = wm * 64 * ROWB, bbase = TILE_B + wn * 64 * ROWB;

  f32x16_t acc[HT][2][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int h = 0; h < HT; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;
  };
  zero_acc();

  // ---- epilogue constants
  const pg_i32x4_t ysrd = pg_make_srd(ea.Y, (unsigned)((size_t)g.M * ea.ldy * 2));      // stores beyond row M - 1 are dropped
  const unsigned yvoff = (unsigned)(4 * fh * ea.ldy + 2 * fi) * 2;                        // this lane inside a (32-row, 64-column) block
  const unsigned row_b = (unsigned)ea.ldy * 2;
  // bias / column scales of all N columns sit in LDS behind the ring (the only compiler-visible loads of the kernel:
  // complete before the first DMA)
  float* cbias = reinterpret_cast<float*>(smem + NSTAGE * STAGE_B);
  if (NS == 4) {      // (deeper rings fill the LDS: their bias / scales travel in registers, fetch_bias)
    for (int i = tid; i < g.N; i += 512) {
      cbias[i] = ea.bias ? ea.bias[i] : 0.f;
      cbias[g.N + i] = ea.colscale ? ea.colscale[i] : 1.f;
    }
    __syncthreads();
  }
  // NS == 5: this lane's bias / scale pair of the NEXT tile to finish, requested one tile ahead (hidden loads: they join the
  // in-order vmcnt queue in front of a tile's K steps and have retired long before its epilogue reads them)
  f32x2_t bvn, cvn;
  auto fetch_bias = [&](int tile) {
    if (NS == 4) return;      // (NS == 4: bias / scales of all columns sit in LDS behind the ring)
    // branch-free (an absent bias / scale reads a constant array): a conditional load would leave the compiler free to
    // re-materialise the default into a register whose load is still in flight
    tile = tile < total_tiles ? tile : total_tiles - 1;
    const int mi = tile / tiles_n, nt = tile - mi * tiles_n;
    int nc = nt * 256 + wn * 64 + 2 * (lane & 31);
    nc = nc + 1 < g.N ? nc : 0;
    const float* pb = (ea.bias ? ea.bias : pg_const_zeros) + nc;
    const float* pc = (ea.colscale ? ea.colscale : pg_const_zeros) + nc;      // (absent: 0 + the 1 added in the epilogue)
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(bvn) : "v"(pb) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(cvn) : "v"(pc) : "memory");
  };

  // F8 with row scales: this lane's dword of exponent bytes (its four 32-row fragments of a 256-row tile), requested a whole
  // tile ahead by a hidden load (in-order vmcnt queue: retired long before the tile begins) and moved into xs_cur by a volatile
  // v_mov at the tile change
  // (branch-free like fetch_bias: absent exponents read a constant array of unit scales)
  uint32_t xs_cur, xs_next;
  constexpr bool has_xs = F8 && HT == 2;
  auto fetch_xs = [&](int tile) {
    if constexpr (has_xs) {
      tile = tile < total_tiles ? tile : total_tiles - 1;
      const int mi = tile / tiles_n;
      const int mt = rowtiles ? tn_sload_i32(rowtiles, mi) : mi;
      const uint8_t* px = pa.rowexp ? pa.rowexp + (size_t)mt * 256 + (size_t)(wm * 32 + (lane & 31)) * 4
                                    : reinterpret_cast<const uint8_t*>(pg_const_unit_exp) + (size_t)(wm * 32 + (lane & 31)) * 4;
      asm volatile("global_load_dword %0, %1, off" : "=v"(xs_next) : "v"(px) : "memory");
    }
  };
  auto take_xs = [&]() {
    if constexpr (has_xs) asm volatile("v_mov_b32 %0, %1" : "=v"(xs_cur) : "v"(xs_next));
  };
  int itile = v, ikt = 0;       // next K step to request
  int ctile = v, ckt = 0;       // K step being multiplied
  int cstage = 0, istage = 0;
  int fresh = AHEAD;            // K steps since this wave's last epilogue stores were issued (they sit in the vmcnt queue)
  // the stream never stops requesting: past the last real K step it re-requests step 0 of the last tile into stages nobody
  // reads any more (every wait below then has the same count, and no branch splits the MFMA stream)
  auto advance_issue = [&]() {
    if (itile < total_tiles && ++ikt == KT) { ikt = 0; itile += G; if (itile < total_tiles) dma_setup(itile); }
    istage = istage + 1 == NSTAGE ? 0 : istage + 1;
  };
  fetch_bias(ctile);
  fetch_xs(ctile);
  dma_setup(itile < total_tiles ? itile : total_tiles - 1);
#pragma unroll 1
  for (int d = 0; d < AHEAD; ++d) {                 // steps 0 .. AHEAD - 1 in flight
    const int kt = itile < total_tiles ? ikt : 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) { dma_a(q, kt, istage); dma_b(q, kt, istage); }
    advance_issue();
  }
  pg_wait<(AHEAD - 1) * GRP>();     // step 0 has landed (the later ones stay in flight)
  take_xs();
  fetch_xs(ctile + G);
  pg_barrier();
  while (ctile < total_tiles) {
    const int kt_req = itile < total_tiles ? ikt : 0;
    {
      const char* st = smem + cstage * STAGE_B;
      bf16x8_t af[2][2 * HT], bf[2][2];
      auto read_frags = [&](int ks, int buf) {
        const char* ap = st + abase + foff[ks];
        const char* bp = st + bbase + foff[ks];
        bf[buf][0] = *reinterpret_cast<const bf16x8_t*>(bp);
        bf[buf][1] = *reinterpret_cast<const bf16x8_t*>(bp + 32 * ROWB);
#pragma unroll
        for (int i = 0; i < 2 * HT; ++i) af[buf][i] = *reinterpret_cast<const bf16x8_t*>(ap + ((i >> 1) * 128 + (i & 1) * 32) * ROWB);
      };
      read_frags(0, 0);
      if constexpr (F8) {
        typedef __attribute__((ext_vector_type(8))) int i32x8_t;
        typedef __attribute__((ext_vector_type(4))) int i32x4_t;
        read_frags(1, 1);
        auto join = [](const bf16x8_t& lo, const bf16x8_t& hi) {
          const i32x4_t l = __builtin_bit_cast(i32x4_t, lo), h = __builtin_bit_cast(i32x4_t, hi);
          i32x8_t f;
          f[0] = l[0]; f[1] = l[1]; f[2] = l[2]; f[3] = l[3]; f[4] = h[0]; f[5] = h[1]; f[6] = h[2]; f[7] = h[3];
          return f;
        };
        constexpr int SC1 = 0x7f7f7f7f;          // E8M0 127: block scale 1
        const i32x8_t b0 = join(bf[0][0], bf[1][0]), b1 = join(bf[0][1], bf[1][1]);
#pragma unroll
        for (int i = 0; i < 2 * HT; ++i) {
          if (!(DBG & 2) && HT == 1) { dma_a(i, kt_req, istage); dma_b(i, kt_req, istage); }
          if (!(DBG & 2) && HT == 2 && (i & 1) == 0) { dma_a(i >> 1, kt_req, istage); dma_b(i >> 1, kt_req, istage); }     // of the K step three ahead
          const i32x8_t a8 = join(af[0][i], af[1][i]);
          if (DBG & 1) {
            asm volatile("" ::"v"(a8), "v"(b0), "v"(b1));
          } else {
            // A block scale: byte i of xs_cur = the exponent of this lane's row in fragment i (unit scales: 0x7f in every byte)
            const int xa = has_xs ? (int)xs_cur : SC1;
            auto mm = [&](auto sel) {
              constexpr int S = decltype(sel)::value;
              acc[i >> 1][i & 1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b0, acc[i >> 1][i & 1][0], 0, 0, S, xa, 0, SC1);
              acc[i >> 1][i & 1][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b1, acc[i >> 1][i & 1][1], 0, 0, S, xa, 0, SC1);
            };
            if (i == 0) mm(std::integral_constant<int, 0>{});
            else if (i == 1) mm(std::integral_constant<int, 1>{});
            else if (i == 2) mm(std::integral_constant<int, 2>{});
            else mm(std::integral_constant<int, 3>{});
          }
        }
      } else
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks == 0) read_frags(1, 1);
        if (!(DBG & 2)) { dma_a(ks, kt_req, istage); dma_b(ks, kt_req, istage); }     // of the K step three ahead
#pragma unroll
        for (int i = 0; i < 2 * HT; ++i) {
          if (DBG & 1) {
            asm volatile("" ::"v"(af[ks][i]), "v"(bf[ks][0]), "v"(bf[ks][1]));
          } else {
            acc[i >> 1][i & 1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bf[ks][0], acc[i >> 1][i & 1][0], 0, 0, 0);
            acc[i >> 1][i & 1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bf[ks][1], acc[i >> 1][i & 1][1], 0, 0, 0);
          }
        }
      }
    }
    advance_issue();
    cstage = cstage + 1 == NSTAGE ? 0 : cstage + 1;
    if (++ckt == KT) {
      // ---- epilogue of this tile, straight from the accumulators
      const int mi_ = ctile / tiles_n, nt_ = ctile - mi_ * tiles_n;
      const int mt_ = rowtiles ? tn_sload_i32(rowtiles, mi_) : mi_;
      const int ncol = nt_ * 256 + wn * 64 + 2 * fi;               // this lane's channel pair
      const bool cols_ok = nt_ * 256 + wn * 64 < g.N;                 // wave-uniform (N is a multiple of 64)
      const bool full_rows = mt_ * TM + TM <= g.M;
      float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
      if (cols_ok) {
        f32x2_t bv, cv;
        if (NS == 4) { bv = *reinterpret_cast<const f32x2_t*>(cbias + ncol); cv = *reinterpret_cast<const f32x2_t*>(cbias + g.N + ncol); }
        else { const float one = ea.colscale ? 0.f : 1.f; bv = bvn; cv[0] = cvn[0] + one; cv[1] = cvn[1] + one; }
#pragma unroll
        for (int h = 0; h < HT; ++h)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const int rblk = mt_ * TM + h * 128 + wm * 64 + mt * 32;     // first row of the 32-row block
            const unsigned sbase = (unsigned)rblk * row_b + (unsigned)(nt_ * 256 + wn * 64) * 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rr = (r & 3) + 8 * (r >> 2);
              float y0 = fmaf(acc[h][mt][0][r], cv[0], bv[0]), y1 = fmaf(acc[h][mt][1][r], cv[1], bv[1]);
              if (!(DBG & 8)) {
                if (ea.nt_out) pg_store_dword_nt(f2bf_pk(y0, y1), yvoff + (sbase + (unsigned)rr * row_b), ysrd, 0u);
                else pg_store_dword(f2bf_pk(y0, y1), yvoff + (sbase + (unsigned)rr * row_b), ysrd, 0u);   // (the bounds check covers the VGPR offset only)
              }
              if (!full_rows) {
                const bool ok = rblk + rr + 4 * fh < g.M;
                y0 = ok ? y0 : 0.f; y1 = ok ? y1 : 0.f;
              }
              s0 += y0; q0 = fmaf(y0, y0, q0);
              s1 += y1; q1 = fmaf(y1, y1, q1);
            }
          }
        if (ea.stats) {
          s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64);
          q0 += __shfl_xor(q0, 32, 64); q1 += __shfl_xor(q1, 32, 64);
          if (mi_ == 0 && wm == 0 && ea.pad_rows != 0.f) {       // once per column: one wave of the first row tile carries the correction
            s0 = fmaf(-ea.pad_rows, bv[0], s0); s1 = fmaf(-ea.pad_rows, bv[1], s1);
            q0 = fmaf(-ea.pad_rows * bv[0], bv[0], q0); q1 = fmaf(-ea.pad_rows * bv[1], bv[1], q1);
          }
          float* sp = ea.stats + (size_t)((blockIdx.x % TN_NREP) * 2 + fh) * g.N + ncol;     // lower half: sums, upper: squares
          pg_atomic_add(sp, fh ? q0 : s0);
          pg_atomic_add(sp + 1, fh ? q1 : s1);
        }
      }
      zero_acc();
      ckt = 0;
      ctile += G;
      fresh = (cols_ok && !(DBG & 8)) ? 0 : AHEAD;
      fetch_bias(ctile);
      take_xs();                  // the exponents of the tile that begins now (requested a tile ago)
      fetch_xs(ctile + G);
    }
    // this wave's part of the next stage has landed.  The vmcnt queue retires in order: the two groups requested after it
    // may stay in flight, and so may this tile's 64 output stores (+ statistics atomics) while they are YOUNGER than the
    // group waited for: "at most 63 outstanding" then retires the DMA groups that precede them
    if (fresh < AHEAD) pg_wait<STW>();
    else pg_wait<(AHEAD - 1) * GRP>();
    ++fresh;
    pg_barrier();           // ... everybody's part; and the stage refilled next is no longer read by anyone
  }
  pg_wait<0>();
}

// ==========================================================================================
// rwgemm_k512_kernel: C = A W^T for K = 512 (TitaNet-M: every pointwise conv of the forward and of the data gradient) with the
// WEIGHTS RESIDENT IN REGISTERS, the structure of the hidden-256 kernels (tn_v2_*): a workgroup owns one 256-column tile of
// the output for the whole launch — its W tile [256 out][512 k] is 256 KB = 128 VGPRs per lane as MFMA A fragments (wave w: out
// columns 32 w .. 32 w + 31) — and walks 64-row tiles of A: rows prefetched one tile ahead in registers (8 x 16 bytes per
// thread), staged in LDS as the MFMA B operand (padded pitch), 64 MFMAs per wave and tile, the result rows back through LDS
// for coalesced 16-byte stores.  pgemm_nt_kernel streams A AND W through the LDS ring for every tile: at K = 512 its operand
// stream (393 KB per 128 x 256 tile, two thirds of it weights, at ~38 GB/s per CU) bounds it at 64 us per 76800 x 512 x 512
// layer = 2.4 TB/s of its own bytes; here the only stream is A.
// Column statistics (BatchNorm batch sums) are taken from the STORED (bf16-rounded) values in the store phase: per-thread
// partial sums of a thread's 8 columns over all its rows and tiles, one LDS reduction + replicated atomics per workgroup.
// Row-tile lists / pad_rows as in pgemm_nt_kernel (a listed 256-row tile = 4 tiles here).
// ==========================================================================================
#define RW_K 512
#define RW_AP (RW_K + 8)       // pitch of the A tile rows in LDS (bf16 elements)
#define RW_DP (256 + 8)        // pitch of the result rows
#define RW_R 64
// EPI: bias + column statistics (forward); false: plain product (data gradient) — 32 more registers for fragment prefetch
template <bool EPI>
__global__ __launch_bounds__(512, 2) void rwgemm_k512_kernel(GemmShape g, PGemmNtArgs pa, PGemmEpiArgs ea, int tiles_n, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Pt = reinterpret_cast<bf16_t*>(smem);                 // [64][520] A rows (MFMA B operand)
  bf16_t* Dt = Pt + RW_R * RW_AP;                               // [64][264] result rows
  float* red = reinterpret_cast<float*>(smem);                  // [16][2][256] at the end (inside Pt)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int G = gridDim.x, v = pg_virtual_id(blockIdx.x, G);
  const int ct = v % tiles_n, first = v / tiles_n, stride = G / tiles_n;      // G is a multiple of tiles_n (launcher)
  const int col0 = ct * 256;
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);
  const int* __restrict__ rowtiles = pa.rowtiles;
  // ---- this wave's weight fragments: out column col0 + 32 wave + (lane & 31), k = 16 ks + 8 half .. + 8
  bf16x8_t wf[32];
  {
    const bf16_t* wr = W + (size_t)(col0 + wave * 32 + (lane & 31)) * RW_K + half * 8;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) wf[ks] = *reinterpret_cast<const bf16x8_t*>(wr + ks * 16);
  }
  // bias of this lane's 16 result columns (MFMA layout: column 8 g + 4 half + r of the wave's 32)
  float bv[EPI ? 16 : 1];
  if constexpr (EPI) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[4 * gq + r] = ea.bias ? ea.bias[col0 + wave * 32 + 8 * gq + 4 * half + r] : 0.f;
  }
  // ---- A rows: thread (rq = tid >> 6: rows rq + 8 q, vc = tid & 63: 16-byte vector of the 1 KB row)
  const int vc = tid & 63, rq = tid >> 6;
  auto tile_row0 = [&](int t) -> int { return rowtiles ? tn_sload_i32(rowtiles, t >> 2) * 256 + (t & 3) * RW_R : t * RW_R; };
  // BUFFER loads / stores (rows beyond M read as zeros / are dropped by the descriptor's bounds check), the loads issued from
  // inline asm and waited for with a COUNTED vmcnt: with predicated global loads / stores (each in its own exec-masked block)
  // — and even with plain buffer loads, because the loop header merges the entry path (8 loads) with the back edge (8 loads +
  // 4 stores) — hipcc waits with vmcnt(0) at the top of every tile, i.e. for the previous tile's output stores to be
  // acknowledged: 41 % of the wave time sat in that s_waitcnt.  Here the loads of the next tile are older than the stores
  // of this one (in-order queue), so "at most 4 outstanding" at the end of the tile retires exactly the loads.
  typedef __attribute__((ext_vector_type(4))) unsigned int rw_u32x4_t;
  const pg_i32x4_t srdA = pg_make_srd(pa.A, (unsigned)((size_t)g.M * pa.lda * sizeof(bf16_t)));
  const __amdgpu_buffer_rsrc_t srdY = __builtin_amdgcn_make_buffer_rsrc(ea.Y, 0, (int)((size_t)g.M * ea.ldy * sizeof(bf16_t)), 0x00020000);
  rw_u32x4_t pre[8];
  auto prefetch = [&](int t) {
    const int r0 = tile_row0(t);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned voff = (unsigned)(((r0 + rq + 8 * q) * pa.lda + vc * 8) * sizeof(bf16_t));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(pre[q]) : "v"(voff), "s"(srdA) : "memory");
    }
  };
#define RW_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(pre[0]), "+v"(pre[1]), "+v"(pre[2]), "+v"(pre[3]), "+v"(pre[4]), "+v"(pre[5]), "+v"(pre[6]), "+v"(pre[7]) : : "memory")
  // store phase: thread (sv = tid & 31: 8 columns, sr = tid >> 5: rows sr + 16 q)
  const int sv = tid & 31, sr = tid >> 5;
  float ssum[EPI ? 8 : 1], ssq[EPI ? 8 : 1];
  if constexpr (EPI) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { ssum[i] = 0.f; ssq[i] = 0.f; }
  }
  int tile = first;
  if (tile < ntiles) { prefetch(tile); RW_WAIT(0); }
  for (; tile < ntiles; tile += stride) {
    const int r0 = tile_row0(tile);
    // (no barrier here: every wave is past barrier (3) of the previous tile, so its MFMAs are done with Pt; and Dt is rewritten
    //  only behind barrier (2) below, which every wave reaches after its stores of the previous tile)
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<rw_u32x4_t*>(Pt + (rq + 8 * q) * RW_AP + vc * 8) = pre[q];
    prefetch(tile + stride < ntiles ? tile + stride : ntiles - 1);      // (unconditional: the wait below can then be counted)
    __syncthreads();   // (2)
    f32x16_t acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const bf16_t* brow = Pt + (lane & 31) * RW_AP + half * 8;
    // fragment reads PF k-steps ahead of their MFMAs (rotating registers): an LDS read issued right in front of its MFMA is a
    // ~100-cycle wait per 32-cycle instruction with two waves per SIMD
    constexpr int PF = EPI ? 2 : 8;
    bf16x8_t bq[PF][2];
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
      for (int n = 0; n < 2; ++n) bq[d][n] = *reinterpret_cast<const bf16x8_t*>(brow + n * 32 * RW_AP + d * 16);
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      bf16x8_t b0 = bq[ks % PF][0], b1 = bq[ks % PF][1];
      if (ks + PF < 32) {
#pragma unroll
        for (int n = 0; n < 2; ++n) bq[ks % PF][n] = *reinterpret_cast<const bf16x8_t*>(brow + n * 32 * RW_AP + (ks + PF) * 16);
      }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b1, acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        uint2 w;
        if constexpr (EPI) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[n][4 * gq + r] += bv[4 * gq + r];
        }
        w.x = f2bf_pk(acc[n][4 * gq], acc[n][4 * gq + 1]);
        w.y = f2bf_pk(acc[n][4 * gq + 2], acc[n][4 * gq + 3]);
        *reinterpret_cast<uint2*>(Dt + (n * 32 + (lane & 31)) * RW_DP + wave * 32 + 8 * gq + 4 * half) = w;
      }
    __syncthreads();   // (3)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = sr + 16 * q, gr = r0 + o;
      const rw_u32x4_t u = *reinterpret_cast<const rw_u32x4_t*>(Dt + o * RW_DP + sv * 8);
      __builtin_amdgcn_raw_buffer_store_b128(u, srdY, (int)((gr * ea.ldy + col0 + sv * 8) * sizeof(bf16_t)), 0, 0);
      if (EPI && ea.stats && gr < g.M) {
        {
          const uint32_t uw[4] = {u[0], u[1], u[2], u[3]};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float y0 = __uint_as_float(uw[i] << 16), y1 = __uint_as_float(uw[i] & 0xffff0000u);
            ssum[2 * i] += y0; ssq[2 * i] = fmaf(y0, y0, ssq[2 * i]);
            ssum[2 * i + 1] += y1; ssq[2 * i + 1] = fmaf(y1, y1, ssq[2 * i + 1]);
          }
        }
      }
    }
    RW_WAIT(4);        // the next tile's rows have landed (the 4 stores above, younger, may still be in flight)
  }
#undef RW_WAIT
  if (EPI && ea.stats) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[(sr * 2 + 0) * 256 + sv * 8 + i] = ssum[i]; red[(sr * 2 + 1) * 256 + sv * 8 + i] = ssq[i]; }
    __syncthreads();
    {
      const int which = tid >> 8, c = tid & 255;      // 512 threads: sums | sums of squares of the 256 columns
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += red[(k * 2 + which) * 256 + c];
      // variable-length batches: the all-zero padding rows inside the computed tiles gave y == bf16(bias) exactly — one
      // workgroup per column tile takes their contribution out again
      if (first == 0 && ea.pad_rows != 0.f && ea.bias) {
        const float b = bf2f((bf16_t)(f2bf_pk(ea.bias[col0 + c], 0.f) & 0xffffu));
        s = which ? fmaf(-ea.pad_rows * b, b, s) : fmaf(-ea.pad_rows, b, s);
      }
      atomic_add_f32(ea.stats + (size_t)((blockIdx.x % TN_NREP) * 2 + which) * g.N + col0 + c, s);
    }
  }
}

// ==========================================================================================
// rwgemm_k512_v2_kernel (round 5): the same product with the A rows streamed by LDS-DMA into a RING of 32-row stages.
// rwgemm_k512_kernel keeps ONE tile of register prefetch (64 KB per CU, requested in a burst at the top of a tile, consumed at
// the top of the next): its own counters said 41 % of the wave time sits in s_waitcnt and the matrix pipe is 30 % busy — the
// stream (2.9 - 3.1 TB/s of its own bytes), not the arithmetic, is the time.  Here the only per-lane state of the stream is the
// address: a DMA instruction moves one whole A row (64 lanes x 16 B = 1 KB = 512 bf16) into LDS at base + row * 1040 (the
// padded pitch makes the 16-byte fragment reads conflict-free without a swizzle: one instruction = one row), 4 instructions
// per wave and stage, 3 stages (96 KB per CU) always in flight behind a counted vmcnt.  Every vector-memory instruction of the
// loop is inline asm (see pgemm_nt_kernel): the in-order vmcnt queue of a wave holds, per iteration, 4 DMA instructions then 2
// output stores, so "at most 14 outstanding" at the top of iteration i + 1 retires exactly the rows of stage i + 1.
// Past its last tile a workgroup keeps requesting out-of-range rows (descriptor bounds check: zeros, no memory traffic): every
// wait has the same count and no branch splits the stream.
// ==========================================================================================
#ifndef RW_POL
#define RW_POL 0      // tuning only: 1 = nt on rwgemm_k512_v2's LDS-DMA loads, 2 = nt on its output stores
#endif
#ifndef RW_DBG
#define RW_DBG 0      // tuning only: 1 = rwgemm_k512_v2 runs 2 of its 32 k-steps (the stream without the arithmetic), 2 = no output stores
#endif
#define RW2_R 32
#define RW2_PITCH 1040       // bytes: 1 KB row + 16
#define RW2_NS 4
#define RW2_OP 80            // pitch (bytes) of a wave's private 32 x 32-channel output block
#ifdef RW_STAMPS
static __device__ unsigned long long rw_dbg[4];      // tuning harness: cycles in wait + barrier / MFMA loop / output, summed over waves
#endif
// Output without a workgroup barrier: a wave's 32 rows x 32 channels go through a PRIVATE 2.5 KB block of LDS (written in the
// accumulator layout, read back as 16-byte row pieces: 4 lanes cover the 64 contiguous bytes of a row), so the only barrier of
// a tile is the one that hands the ring stage over.  Two accumulators (even / odd k-steps): no MFMA waits for the one before it.
template <bool EPI>
__global__ __launch_bounds__(512, 2) void rwgemm_k512_v2_kernel(GemmShape g, PGemmNtArgs pa, PGemmEpiArgs ea, int tiles_n, int ntiles) {
  constexpr int STAGE_B = RW2_R * RW2_PITCH;      // 33280
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);                           // [16][2][256] at the end (inside the ring)
  const unsigned lds0 = (unsigned)(uintptr_t)(tn_lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* ob = smem + RW2_NS * STAGE_B + wave * (RW2_R * RW2_OP);         // this wave's output block
  const int G = gridDim.x, v = pg_virtual_id(blockIdx.x, G);
  const int ct = v % tiles_n, first = v / tiles_n, stride = G / tiles_n;      // G is a multiple of tiles_n (launcher)
  const int col0 = ct * 256;
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);
  const int* __restrict__ rowtiles = pa.rowtiles;
  bf16x8_t wf[32];
  {
    const bf16_t* wr = W + (size_t)(col0 + wave * 32 + (lane & 31)) * RW_K + half * 8;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) wf[ks] = *reinterpret_cast<const bf16x8_t*>(wr + ks * 16);
  }
  float bv[EPI ? 16 : 1];
  if constexpr (EPI) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[4 * gq + r] = ea.bias ? ea.bias[col0 + wave * 32 + 8 * gq + 4 * half + r] : 0.f;
  }
  const pg_i32x4_t srdA = pg_make_srd(pa.A, (unsigned)((size_t)g.M * pa.lda * sizeof(bf16_t)));
  const pg_i32x4_t srdY = pg_make_srd(ea.Y, (unsigned)((size_t)g.M * ea.ldy * sizeof(bf16_t)));
  // a listed 256-row tile = 8 tiles here; tiles past the end map to row M (out of the descriptor's range)
  auto tile_row0 = [&](int t) -> int {
    if (t >= ntiles) return g.M;
    return rowtiles ? tn_sload_i32(rowtiles, t >> 3) * 256 + (t & 7) * RW2_R : t * RW2_R;
  };
  // this wave's rows of a stage: wave, wave + 8, wave + 16, wave + 24
  auto dma_tile = [&](int t, int stage) {
    const int r0 = tile_row0(t);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = wave + 8 * q;
      const unsigned voff = (unsigned)(r0 + row) * (unsigned)(pa.lda * 2) + (unsigned)lane * 16u;
      if (RW_POL & 1) pg_dma16_buf_nt(voff, srdA, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + stage * STAGE_B + row * RW2_PITCH)));
      else pg_dma16_buf(voff, srdA, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + stage * STAGE_B + row * RW2_PITCH)));
    }
  };
  // output pieces of this lane: (row orow + 16 q, 16-byte piece opc of the wave's 64-byte row segment)
  const int orow = lane >> 2, opc = lane & 3;
  float ssum[EPI ? 8 : 1], ssq[EPI ? 8 : 1];
  if constexpr (EPI) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { ssum[i] = 0.f; ssq[i] = 0.f; }
  }
  // the compiler-visible loads above (weights, bias) must have retired before the hand-counted queue starts — and the
  // COMPILER must know it: it places its own vmcnt wait in front of the first use of a loaded register, which would be inside
  // the loop (a vmcnt(31 - ks) in front of every MFMA, i.e. a drain of the DMA ring and of the output stores per tile).  An
  // empty asm that reads every fragment here makes it wait for them here
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < 32; ++ks) asm volatile("" : "+v"(wf[ks]));
  if constexpr (EPI) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(bv[i]));
  }
#ifdef RW_STAMPS
  unsigned long long d_wait = 0, d_mfma = 0, d_out = 0;
#endif
  int tile = first, it = 0, stage = 0;
#pragma unroll 1
  for (int d = 0; d < RW2_NS - 1; ++d) dma_tile(tile + d * stride, d);
#pragma unroll 1
  for (; tile < ntiles; tile += stride, ++it) {
#ifdef RW_STAMPS
    const unsigned long long t0 = __builtin_readcyclecounter();
#endif
    // this wave's rows of the stage have landed (queue behind them: see the header comment)
    if (it == 0) pg_wait<8>();
    else if (it == 1) pg_wait<10>();
    else if (it == 2) pg_wait<12>();
    else pg_wait<14>();
    pg_barrier();      // everybody's rows have landed; every wave is past its MFMAs of the previous tile
    {
      const int ns = stage == 0 ? RW2_NS - 1 : stage - 1;      // the stage of the previous tile is free now
      dma_tile(tile + (RW2_NS - 1) * stride, ns);
    }
#ifdef RW_STAMPS
    const unsigned long long t1 = __builtin_readcyclecounter();
#endif
    const char* st_ = smem + stage * STAGE_B;
    f32x16_t acc, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }
    const char* brow = st_ + (lane & 31) * RW2_PITCH + half * 16;
    constexpr int PF = 8;
    bf16x8_t bq[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) bq[d] = *reinterpret_cast<const bf16x8_t*>(brow + d * 32);
#pragma unroll
    for (int ks = 0; ks < ((RW_DBG & 1) ? 2 : 32); ++ks) {
      const bf16x8_t b0 = bq[ks % PF];
      if (ks + PF < 32) bq[ks % PF] = *reinterpret_cast<const bf16x8_t*>(brow + (ks + PF) * 32);
      if (ks & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b0, acc1, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b0, acc, 0, 0, 0);
    }
#ifdef RW_STAMPS
    const unsigned long long t2 = __builtin_readcyclecounter();
#endif
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      uint2 w;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[4 * gq + r] += acc1[4 * gq + r] + (EPI ? bv[4 * gq + r] : 0.f);
      w.x = f2bf_pk(acc[4 * gq], acc[4 * gq + 1]);
      w.y = f2bf_pk(acc[4 * gq + 2], acc[4 * gq + 3]);
      *reinterpret_cast<uint2*>(ob + (lane & 31) * RW2_OP + (8 * gq + 4 * half) * 2) = w;
    }
    // (the block is private to the wave: the compiler's lgkmcnt wait orders the reads below behind the writes above)
    const int r0 = tile_row0(tile);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int o = orow + 16 * q, gr = r0 + o;
      typedef __attribute__((ext_vector_type(4))) unsigned int rw2_u32x4_t;
      const rw2_u32x4_t u = *reinterpret_cast<const rw2_u32x4_t*>(ob + o * RW2_OP + opc * 16);
      const unsigned voff = ((unsigned)gr * (unsigned)ea.ldy + (unsigned)(col0 + wave * 32 + opc * 8)) * 2u;
      // (s_nop: an inline-asm store gets no hazard slots from hipcc before a write of its data registers — found in the v3
      //  kernel below, where the next row's ds_read landed in them; here the two rows happen to get distinct registers)
      if ((RW_POL & 2) || ea.nt_out) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen nt\n\ts_nop 1" ::"v"(u), "v"(voff), "s"(srdY) : "memory");
      else if (!(RW_DBG & 2)) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(u), "v"(voff), "s"(srdY) : "memory");
      else asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(u), "v"(0x7ffffff0u), "s"(srdY) : "memory");
      if (EPI && ea.stats && gr < g.M) {
        const uint32_t uw[4] = {u[0], u[1], u[2], u[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float y0 = __uint_as_float(uw[i] << 16), y1 = __uint_as_float(uw[i] & 0xffff0000u);
          ssum[2 * i] += y0; ssq[2 * i] = fmaf(y0, y0, ssq[2 * i]);
          ssum[2 * i + 1] += y1; ssq[2 * i + 1] = fmaf(y1, y1, ssq[2 * i + 1]);
        }
      }
    }
    stage = stage + 1 == RW2_NS ? 0 : stage + 1;
#ifdef RW_STAMPS
    const unsigned long long t3 = __builtin_readcyclecounter();
    d_wait += t1 - t0; d_mfma += t2 - t1; d_out += t3 - t2;
#endif
  }
  pg_wait<0>();
#ifdef RW_STAMPS
  if (lane == 0) { atomicAdd(&rw_dbg[0], d_wait); atomicAdd(&rw_dbg[1], d_mfma); atomicAdd(&rw_dbg[2], d_out); atomicAdd(&rw_dbg[3], (unsigned long long)it); }
#endif
  if (EPI && ea.stats) {
    __syncthreads();
    {
      // columns of this lane: wave * 32 + opc * 8 + i; 16 lanes of the wave (orow) hold partial sums of the same columns
      const int sv = wave * 4 + opc, sr = orow;
#pragma unroll
      for (int i = 0; i < 8; ++i) { red[(sr * 2 + 0) * 256 + sv * 8 + i] = ssum[i]; red[(sr * 2 + 1) * 256 + sv * 8 + i] = ssq[i]; }
    }
    __syncthreads();
    {
      const int which = tid >> 8, c = tid & 255;      // 512 threads: sums | sums of squares of the 256 columns
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += red[(k * 2 + which) * 256 + c];
      if (first == 0 && ea.pad_rows != 0.f && ea.bias) {
        const float b = bf2f((bf16_t)(f2bf_pk(ea.bias[col0 + c], 0.f) & 0xffffu));
        s = which ? fmaf(-ea.pad_rows * b, b, s) : fmaf(-ea.pad_rows, b, s);
      }
      atomic_add_f32(ea.stats + (size_t)((blockIdx.x % TN_NREP) * 2 + which) * g.N + col0 + c, s);
    }
  }
}
// ==========================================================================================
// rwgemm_k512_v3_kernel (round 6): the ring of rwgemm_k512_v2 with 64 OUTPUT COLUMNS PER WAVE.  v2's eight waves own 32 columns
// each: every 16-byte B fragment read from LDS feeds ONE MFMA (1 KB of LDS per MFMA, 256 KB per 32-row tile: LDS reads and the
// matrix pipe are co-bound and did not overlap, 2.3 us per tile against 0.85).  Here a workgroup is FOUR waves, one per SIMD,
// each holding the 64 x 512 weight block of its columns — 256 registers per lane, which only fits at one wave per SIMD (512
// registers: hipcc keeps what does not fit the 256 architectural ones in AGPRs) — so a fragment feeds TWO MFMAs and a tile costs
// 128 KB of LDS reads.  The fragment reads are hand-scheduled (tn_mfma_sched_lds, 8 in flight: nothing else runs on the SIMD to
// cover an exposed LDS round trip).  DMA: 8 rows per wave and stage; output: a private 32 x 64-column block per wave.
// MEASURED (profiles/r06_rwgemm_v3.txt): 51.6 / 57.6 us against v2's 49.9 / 52.4 (plain / bias + statistics) at 76800 x 512 x 512 —
// halving the fragment reads buys nothing, the lone wave's serial epilogue costs 2 - 5 us: the kernel runs at ~3.1 TB/s of its own
// bytes whatever N is (r05 N sweep), i.e. it is the stream, not LDS or the matrix pipe.  Not the default (TN_RW_VARIANT=3 selects it).
// ==========================================================================================
#define RW3_OP 144           // pitch (bytes) of a wave's private 32 x 64-channel output block
template <bool EPI>
__global__ __launch_bounds__(256, 1) void rwgemm_k512_v3_kernel(GemmShape g, PGemmNtArgs pa, PGemmEpiArgs ea, int tiles_n, int ntiles) {
  constexpr int STAGE_B = RW2_R * RW2_PITCH;      // 33280
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);                           // [8][2][256] at the end (inside the ring)
  const unsigned lds0 = (unsigned)(uintptr_t)(tn_lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* ob = smem + RW2_NS * STAGE_B + wave * (RW2_R * RW3_OP);         // this wave's output block
  const int G = gridDim.x, v = pg_virtual_id(blockIdx.x, G);
  const int ct = v % tiles_n, first = v / tiles_n, stride = G / tiles_n;      // G is a multiple of tiles_n (launcher)
  const int col0 = ct * 256;
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);
  const int* __restrict__ rowtiles = pa.rowtiles;
  bf16x8_t wf[2 * 32];      // [column block][k-step]
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const bf16_t* wr = W + (size_t)(col0 + wave * 64 + cb * 32 + (lane & 31)) * RW_K + half * 8;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) wf[cb * 32 + ks] = *reinterpret_cast<const bf16x8_t*>(wr + ks * 16);
  }
  float bv[EPI ? 32 : 1];
  if constexpr (EPI) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[cb * 16 + 4 * gq + r] = ea.bias ? ea.bias[col0 + wave * 64 + cb * 32 + 8 * gq + 4 * half + r] : 0.f;
  }
  const pg_i32x4_t srdA = pg_make_srd(pa.A, (unsigned)((size_t)g.M * pa.lda * sizeof(bf16_t)));
  const pg_i32x4_t srdY = pg_make_srd(ea.Y, (unsigned)((size_t)g.M * ea.ldy * sizeof(bf16_t)));
  auto tile_row0 = [&](int t) -> int {
    if (t >= ntiles) return g.M;
    return rowtiles ? tn_sload_i32(rowtiles, t >> 3) * 256 + (t & 7) * RW2_R : t * RW2_R;
  };
  // this wave's rows of a stage: wave + 4 q
  auto dma_tile = [&](int t, int stage) {
    const int r0 = tile_row0(t);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = wave + 4 * q;
      const unsigned voff = (unsigned)(r0 + row) * (unsigned)(pa.lda * 2) + (unsigned)lane * 16u;
      pg_dma16_buf(voff, srdA, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + stage * STAGE_B + row * RW2_PITCH)));
    }
  };
  // output pieces of this lane: (row orow + 8 q, 16-byte piece opc of the wave's 128-byte row segment)
  const int orow = lane >> 3, opc = lane & 7;
  float ssum[EPI ? 8 : 1], ssq[EPI ? 8 : 1];
  if constexpr (EPI) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { ssum[i] = 0.f; ssq[i] = 0.f; }
  }
  // (see rwgemm_k512_v2: the compiler-visible loads retire here, and the compiler knows it)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < 64; ++ks) asm volatile("" : "+v"(wf[ks]));
  if constexpr (EPI) {
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(bv[i]));
  }
  int tile = first, it = 0, stage = 0;
#pragma unroll 1
  for (int d = 0; d < RW2_NS - 1; ++d) dma_tile(tile + d * stride, d);
#pragma unroll 1
  for (; tile < ntiles; tile += stride, ++it) {
    // this wave's rows of the stage have landed: per iteration the queue holds 8 DMA instructions then 4 output stores
    if (it == 0) pg_wait<16>();
    else if (it == 1) pg_wait<20>();
    else if (it == 2) pg_wait<24>();
    else pg_wait<28>();
    pg_barrier();      // everybody's rows have landed; every wave is past its MFMAs of the previous tile
    {
      const int ns = stage == 0 ? RW2_NS - 1 : stage - 1;      // the stage of the previous tile is free now
      dma_tile(tile + (RW2_NS - 1) * stride, ns);
    }
    const char* st_ = smem + stage * STAGE_B;
    f32x16_t acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = EPI ? bv[r] : 0.f; acc[1][r] = EPI ? bv[16 + r] : 0.f; }
    const char* brow = st_ + (lane & 31) * RW2_PITCH + half * 16;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tn_mfma_sched_lds<32, 1, 2, 8, 32, 0>(&wf[0], brow, &acc[0]);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        uint2 w;
        w.x = f2bf_pk(acc[cb][4 * gq], acc[cb][4 * gq + 1]);
        w.y = f2bf_pk(acc[cb][4 * gq + 2], acc[cb][4 * gq + 3]);
        *reinterpret_cast<uint2*>(ob + (lane & 31) * RW3_OP + (cb * 32 + 8 * gq + 4 * half) * 2) = w;
      }
    // (the block is private to the wave: the compiler's lgkmcnt wait orders the reads below behind the writes above)
    const int r0 = tile_row0(tile);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = orow + 8 * q, gr = r0 + o;
      typedef __attribute__((ext_vector_type(4))) unsigned int rw3_u32x4_t;
      const rw3_u32x4_t u = *reinterpret_cast<const rw3_u32x4_t*>(ob + o * RW3_OP + opc * 16);
      const unsigned voff = ((unsigned)gr * (unsigned)ea.ldy + (unsigned)(col0 + wave * 64 + opc * 8)) * 2u;
      // (s_nop: an inline-asm store gets no hazard slots from hipcc, and the next row's ds_read lands in the same data registers —
      //  without it the plain variant wrote 5 % wrong elements)
      asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(u), "v"(voff), "s"(srdY) : "memory");
      if (EPI && ea.stats && gr < g.M) {
        const uint32_t uw[4] = {u[0], u[1], u[2], u[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float y0 = __uint_as_float(uw[i] << 16), y1 = __uint_as_float(uw[i] & 0xffff0000u);
          ssum[2 * i] += y0; ssq[2 * i] = fmaf(y0, y0, ssq[2 * i]);
          ssum[2 * i + 1] += y1; ssq[2 * i + 1] = fmaf(y1, y1, ssq[2 * i + 1]);
        }
      }
    }
    stage = stage + 1 == RW2_NS ? 0 : stage + 1;
  }
  pg_wait<0>();
  if (EPI && ea.stats) {
    __syncthreads();
    {
      // columns of this lane: wave * 64 + opc * 8 + i; the 8 lanes groups (orow) hold partial sums of the same columns
      const int sv = wave * 8 + opc, sr = orow;
#pragma unroll
      for (int i = 0; i < 8; ++i) { red[(sr * 2 + 0) * 256 + sv * 8 + i] = ssum[i]; red[(sr * 2 + 1) * 256 + sv * 8 + i] = ssq[i]; }
    }
    __syncthreads();
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const int c = tid;      // 256 threads: one column each, sums then sums of squares
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += red[(k * 2 + which) * 256 + c];
      if (first == 0 && ea.pad_rows != 0.f && ea.bias) {
        const float b = bf2f((bf16_t)(f2bf_pk(ea.bias[col0 + c], 0.f) & 0xffffu));
        s = which ? fmaf(-ea.pad_rows * b, b, s) : fmaf(-ea.pad_rows, b, s);
      }
      atomic_add_f32(ea.stats + (size_t)((blockIdx.x % TN_NREP) * 2 + which) * g.N + col0 + c, s);
    }
  }
}
// ==========================================================================================
// rwgemm_k512_v4_kernel (round 6): v2's ring kernel as TWO INDEPENDENT 256-thread workgroups per CU on 128-column tiles.
// v2's cycle stamps: per wave and tile 1963 cycles in the MFMA loop (the pipe is full there: 2 waves x 32 MFMAs x 32 cycles) and
// 844 + 962 cycles waiting for the ring / in the output phase, during which the pipe idles — both waves of a SIMD are in the same
// phase of the same tile, the per-tile workgroup barrier keeps them there.  Two workgroups per CU share nothing but the CU: their
// phases drift apart and one's output phase runs under the other's MFMA loop.  Each wave still owns 32 columns (128 weight
// registers); ring of RW4_NS stages per workgroup (2 x 2 x 33 KB per CU); every A row is read by N / 128 workgroups, adjacent in
// launch order (one XCD: L2 hits).
// ==========================================================================================
#define RW4_NS 2
template <bool EPI>
__global__ __launch_bounds__(256, 2) void rwgemm_k512_v4_kernel(GemmShape g, PGemmNtArgs pa, PGemmEpiArgs ea, int tiles_n, int ntiles) {
  constexpr int STAGE_B = RW2_R * RW2_PITCH;      // 33280
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);                           // [16][2][128] at the end (inside the ring)
  const unsigned lds0 = (unsigned)(uintptr_t)(tn_lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* ob = smem + RW4_NS * STAGE_B + wave * (RW2_R * RW2_OP);         // this wave's output block
  const int G = gridDim.x, v = pg_virtual_id(blockIdx.x, G);
  const int ct = v % tiles_n, first = v / tiles_n, stride = G / tiles_n;      // G is a multiple of tiles_n (launcher)
  const int col0 = ct * 128;
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);
  const int* __restrict__ rowtiles = pa.rowtiles;
  bf16x8_t wf[32];
  {
    const bf16_t* wr = W + (size_t)(col0 + wave * 32 + (lane & 31)) * RW_K + half * 8;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) wf[ks] = *reinterpret_cast<const bf16x8_t*>(wr + ks * 16);
  }
  float bv[EPI ? 16 : 1];
  if constexpr (EPI) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[4 * gq + r] = ea.bias ? ea.bias[col0 + wave * 32 + 8 * gq + 4 * half + r] : 0.f;
  }
  const pg_i32x4_t srdA = pg_make_srd(pa.A, (unsigned)((size_t)g.M * pa.lda * sizeof(bf16_t)));
  const pg_i32x4_t srdY = pg_make_srd(ea.Y, (unsigned)((size_t)g.M * ea.ldy * sizeof(bf16_t)));
  auto tile_row0 = [&](int t) -> int {
    if (t >= ntiles) return g.M;
    return rowtiles ? tn_sload_i32(rowtiles, t >> 3) * 256 + (t & 7) * RW2_R : t * RW2_R;
  };
  // this wave's rows of a stage: wave + 4 q
  auto dma_tile = [&](int t, int stage) {
    const int r0 = tile_row0(t);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = wave + 4 * q;
      const unsigned voff = (unsigned)(r0 + row) * (unsigned)(pa.lda * 2) + (unsigned)lane * 16u;
      pg_dma16_buf(voff, srdA, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + stage * STAGE_B + row * RW2_PITCH)));
    }
  };
  const int orow = lane >> 2, opc = lane & 3;
  float ssum[EPI ? 8 : 1], ssq[EPI ? 8 : 1];
  if constexpr (EPI) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { ssum[i] = 0.f; ssq[i] = 0.f; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < 32; ++ks) asm volatile("" : "+v"(wf[ks]));
  if constexpr (EPI) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(bv[i]));
  }
  int tile = first, it = 0, stage = 0;
#pragma unroll 1
  for (int d = 0; d < RW4_NS - 1; ++d) dma_tile(tile + d * stride, d);
#pragma unroll 1
  for (; tile < ntiles; tile += stride, ++it) {
    // per iteration the queue holds 8 DMA instructions (issued at the top, for the next tile) then 2 output stores: what may
    // stay outstanding behind this tile's rows is the previous iteration's 2 stores
    if (it == 0) pg_wait<0>();
    else pg_wait<2>();
    pg_barrier();      // everybody's rows have landed; every wave is past its MFMAs of the previous tile
    dma_tile(tile + (RW4_NS - 1) * stride, stage ^ 1);
    const char* st_ = smem + stage * STAGE_B;
    f32x16_t acc, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }
    const char* brow = st_ + (lane & 31) * RW2_PITCH + half * 16;
    constexpr int PF = 8;
    bf16x8_t bq[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) bq[d] = *reinterpret_cast<const bf16x8_t*>(brow + d * 32);
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      const bf16x8_t b0 = bq[ks % PF];
      if (ks + PF < 32) bq[ks % PF] = *reinterpret_cast<const bf16x8_t*>(brow + (ks + PF) * 32);
      if (ks & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b0, acc1, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], b0, acc, 0, 0, 0);
    }
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      uint2 w;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[4 * gq + r] += acc1[4 * gq + r] + (EPI ? bv[4 * gq + r] : 0.f);
      w.x = f2bf_pk(acc[4 * gq], acc[4 * gq + 1]);
      w.y = f2bf_pk(acc[4 * gq + 2], acc[4 * gq + 3]);
      *reinterpret_cast<uint2*>(ob + (lane & 31) * RW2_OP + (8 * gq + 4 * half) * 2) = w;
    }
    const int r0 = tile_row0(tile);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int o = orow + 16 * q, gr = r0 + o;
      typedef __attribute__((ext_vector_type(4))) unsigned int rw4_u32x4_t;
      const rw4_u32x4_t u = *reinterpret_cast<const rw4_u32x4_t*>(ob + o * RW2_OP + opc * 16);
      const unsigned voff = ((unsigned)gr * (unsigned)ea.ldy + (unsigned)(col0 + wave * 32 + opc * 8)) * 2u;
      asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(u), "v"(voff), "s"(srdY) : "memory");
      if (EPI && ea.stats && gr < g.M) {
        const uint32_t uw[4] = {u[0], u[1], u[2], u[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float y0 = __uint_as_float(uw[i] << 16), y1 = __uint_as_float(uw[i] & 0xffff0000u);
          ssum[2 * i] += y0; ssq[2 * i] = fmaf(y0, y0, ssq[2 * i]);
          ssum[2 * i + 1] += y1; ssq[2 * i + 1] = fmaf(y1, y1, ssq[2 * i + 1]);
        }
      }
    }
    stage ^= 1;
  }
  pg_wait<0>();
  if (EPI && ea.stats) {
    __syncthreads();
    {
      const int sv = wave * 4 + opc, sr = orow;      // columns wave * 32 + opc * 8 + i of the 128; 16 row groups
#pragma unroll
      for (int i = 0; i < 8; ++i) { red[(sr * 2 + 0) * 128 + sv * 8 + i] = ssum[i]; red[(sr * 2 + 1) * 128 + sv * 8 + i] = ssq[i]; }
    }
    __syncthreads();
    {
      const int which = tid >> 7, c = tid & 127;      // 256 threads: sums | sums of squares of the 128 columns
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += red[(k * 2 + which) * 128 + c];
      if (first == 0 && ea.pad_rows != 0.f && ea.bias) {
        const float b = bf2f((bf16_t)(f2bf_pk(ea.bias[col0 + c], 0.f) & 0xffffu));
        s = which ? fmaf(-ea.pad_rows * b, b, s) : fmaf(-ea.pad_rows, b, s);
      }
      atomic_add_f32(ea.stats + (size_t)((blockIdx.x % TN_NREP) * 2 + which) * g.N + col0 + c, s);
    }
  }
}
// -1000: not this kernel's shape
// variant: 2 = the LDS-DMA ring kernel (falls through to 1 when its shape conditions fail), 1 = register prefetch
inline int launch_rwgemm_k512(const GemmShape& g, const PGemmNtArgs& pa, const PGemmEpiArgs& ea, hipStream_t st, int max_wgs, int variant = 2) {
  if (g.K != RW_K || g.N % 256 || g.N <= 0 || g.N > 1024 || pa.lda % 8 || ea.ldy % 8 || ea.colscale || pa.rowexp) return -1000;
  if ((size_t)g.M * pa.lda * 2 >= ((size_t)1 << 31) || (size_t)g.M * ea.ldy * 2 >= ((size_t)1 << 31)) return -1000;      // 32-bit buffer offsets
  const int tiles_n = g.N / 256;
  int grid = (max_wgs / (8 * tiles_n)) * 8 * tiles_n;   // multiple of 8 (XCD-contiguous order) and of the column tiles
  if (grid <= 0) return -1000;
  {
    static const int forced = [] { const char* e = getenv("TN_RW_VARIANT"); return e ? atoi(e) : 0; }();      // (debug switch)
    if (forced) variant = forced;
  }
  if (variant == 4) {
    // two independent 256-thread workgroups per CU on 128-column tiles (rwgemm_k512_v4_kernel)
    const int ntiles = pa.rowtiles ? pa.n_rowtiles * 8 : (g.M + RW2_R - 1) / RW2_R;
    const int tn4 = g.N / 128;
    int grid4 = ((2 * max_wgs) / (8 * tn4)) * 8 * tn4;
    if (ntiles <= 0) return 0;
    if (grid4 <= 0 || ntiles * tn4 < 2 * grid4 || pa.lda != RW_K) variant = 2;
    else {
      const size_t smem = (size_t)RW4_NS * RW2_R * RW2_PITCH + (size_t)4 * RW2_R * RW2_OP;
      auto kern = (ea.bias || ea.stats) ? rwgemm_k512_v4_kernel<true> : rwgemm_k512_v4_kernel<false>;
      TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      hipLaunchKernelGGL(kern, dim3(grid4), dim3(256), smem, st, g, pa, ea, tn4, ntiles);
      return (int)hipGetLastError();
    }
  }
  if (variant == 3) {
    // 64 columns per wave, one wave per SIMD (rwgemm_k512_v3_kernel); same shape conditions as variant 2
    const int ntiles = pa.rowtiles ? pa.n_rowtiles * 8 : (g.M + RW2_R - 1) / RW2_R;
    if (ntiles <= 0) return 0;
    if (ntiles * tiles_n < 2 * max_wgs || pa.lda != RW_K) variant = 2;
    else {
      const size_t smem = (size_t)RW2_NS * RW2_R * RW2_PITCH + (size_t)4 * RW2_R * RW3_OP;
      auto kern = (ea.bias || ea.stats) ? rwgemm_k512_v3_kernel<true> : rwgemm_k512_v3_kernel<false>;
      TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, g, pa, ea, tiles_n, ntiles);
      return (int)hipGetLastError();
    }
  }
  if (variant == 2) {
    // LDS-DMA ring of 32-row stages (rwgemm_k512_v2_kernel); the DMA moves whole 1 KB rows: lda == 512 only
    const int ntiles = pa.rowtiles ? pa.n_rowtiles * 8 : (g.M + RW2_R - 1) / RW2_R;
    if (ntiles <= 0) return 0;
    if (ntiles * tiles_n < 2 * max_wgs) return -1000;
    if (pa.lda != RW_K) variant = 1;      // a strided A: the register-prefetch kernel below takes any lda
  }
  if (variant == 2) {
    const int ntiles = pa.rowtiles ? pa.n_rowtiles * 8 : (g.M + RW2_R - 1) / RW2_R;
    const size_t smem = (size_t)RW2_NS * RW2_R * RW2_PITCH + (size_t)8 * RW2_R * RW2_OP;
    auto kern = (ea.bias || ea.stats) ? rwgemm_k512_v2_kernel<true> : rwgemm_k512_v2_kernel<false>;
    TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, g, pa, ea, tiles_n, ntiles);
    return (int)hipGetLastError();
  }
  const int ntiles = pa.rowtiles ? pa.n_rowtiles * 4 : (g.M + RW_R - 1) / RW_R;
  if (ntiles <= 0) return 0;
  if (ntiles * tiles_n < max_wgs) return -1000;      // small problems: the tiled kernel
  const size_t smem = (size_t)(RW_R * RW_AP + RW_R * RW_DP) * sizeof(bf16_t);
  auto kern = (ea.bias || ea.stats) ? rwgemm_k512_kernel<true> : rwgemm_k512_kernel<false>;
  TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, g, pa, ea, tiles_n, ntiles);
  return (int)hipGetLastError();
}

template <int DBG = 0, bool F8 = false>
inline int launch_pgemm_nt_t(const GemmShape& g, const PGemmNtArgs& pa, const PGemmEpiArgs& ea, hipStream_t st, int max_wgs, bool even_rounds = true) {
  if (g.K % (F8 ? 64 : 32) || g.K <= 0 || pa.lda % (F8 ? 16 : 8) || g.N % 64 || g.N > 3072 || ea.ldy % 2 || g.M <= 0) return TN_E_UNSUPPORTED;
  if (!F8 && (DBG & ~16) == 0) {      // K = 512: weights resident in registers (rwgemm_k512_kernel)
    const int rc = launch_rwgemm_k512(g, pa, ea, st, max_wgs);
    if (rc != -1000) return rc;
  }
  if ((long)g.M * pa.lda >= (1L << 32) || (long)g.N * g.K >= (1L << 32) || (long)g.M * ea.ldy * 2 >= (1L << 32)) return TN_E_UNSUPPORTED;
  const int tiles_n = (g.N + 255) / 256;
  // persistent workgroups with the same number of tiles each (1200 tiles: 240 x 5 beats 256 x 4.7, the last round of which
  // runs at 69 % occupancy: 201 vs 209 us), a multiple of 8 for the XCD-contiguous order
  auto plan_grid = [&](int total) {
    int grid = total < max_wgs ? total : max_wgs;
    if (grid >= 8 && even_rounds) {
      const int rounds = (total + grid - 1) / grid;
      grid = (((total + rounds - 1) / rounds) + 7) & ~7;
      if (grid > max_wgs) grid = max_wgs & ~7;
    }
    return grid;
  };
  const int total2 = (pa.rowtiles ? pa.n_rowtiles : (g.M + 255) / 256) * tiles_n;
  if (total2 <= 0) return 0;
  const int grid2 = plan_grid(total2);
  // 128-row tiles when the 256-row partition would leave more than an eighth of the launch's workgroup-rounds empty
  // (not with a row-tile list: those are 256-row tiles; not for the e4m3 form)
  bool half = false;
  int total = total2, grid = grid2;
  if (!F8 && !pa.rowtiles && even_rounds && total2 >= max_wgs) {
    const int rounds2 = (total2 + grid2 - 1) / grid2;
    const double use2 = (double)total2 / ((double)rounds2 * max_wgs);
    const int total1 = ((g.M + 127) / 128) * tiles_n, grid1 = plan_grid(total1), rounds1 = (total1 + grid1 - 1) / grid1;
    const double use1 = (double)total1 / ((double)rounds1 * max_wgs);
    if (use2 < 0.875 && use1 > use2 + 0.08) { half = true; total = total1; grid = grid1; }
  }
  if (half) {
    const size_t smem = (size_t)98304 + (size_t)2 * g.N * sizeof(float);
    auto kern = pgemm_nt_kernel<DBG, F8, 1, 4>;
    TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, g, pa, ea, tiles_n, total);
  } else if (g.N <= 3072) {
    // five 32 KB stages = the whole LDS of a CU: one more K step in flight (L/5: 21.8 -> 21.4 ms; neutral at K = 512)
    const size_t smem = (size_t)5 * 32768;
    auto kern = pgemm_nt_kernel<DBG, F8, 2, 5>;
    TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, g, pa, ea, tiles_n, total);
  } else {
    const size_t smem = (size_t)131072 + (size_t)2 * g.N * sizeof(float);
    auto kern = pgemm_nt_kernel<DBG, F8, 2, 4>;
    TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, g, pa, ea, tiles_n, total);
  }
  return (int)hipGetLastError();
}
inline int launch_pgemm_nt(const GemmShape& g, const PGemmNtArgs& pa, const PGemmEpiArgs& ea, hipStream_t st, int max_wgs = 256) {
  // K >= 1024: every row-tile group of workgroups starts its K walk at its own step (template bit 16, round 5: rows of a
  // K-major operand are 2 KB apart, and 256 workgroups marching through K in lockstep ask for the same 64-byte column of every
  // row at the same time — 210 -> 194 us at 76800 x 1024 x 1024, the operand stream alone 133 -> 128, profiles/r05_pgemm_harness_1024_b.txt;
  // at K = 512 the rotation loses, 52 -> 70 us).  Changes the summation order over K, not the arithmetic.
  static const bool krot_off = [] { const char* e = getenv("TN_KROT"); return e && atoi(e) == 0; }();      // (debug switch)
  if (g.K >= 1024 && !krot_off) return launch_pgemm_nt_t<16>(g, pa, ea, st, max_wgs);
  return launch_pgemm_nt_t<0>(g, pa, ea, st, max_wgs);
}
// e4m3 x e4m3 (pa.A and g.W are byte matrices, pa.lda in bytes = elements)
inline int launch_pgemm_nt_f8(const GemmShape& g, const PGemmNtArgs& pa, const PGemmEpiArgs& ea, hipStream_t st, int max_wgs = 256) {
  return launch_pgemm_nt_t<0, true>(g, pa, ea, st, max_wgs);
}

// ==========================================================================================
// pgemm_tn_kernel: weight-gradient form.  OUT[c][k] += sum over rows r of P[r][c] * Q[r][k]   (f32, global atomics)
//   P [rows][ldp], Q [rows][ldq]: row-major bf16 activations used as stored (the BatchNorm-backward'd gradient dS and the
//   kept GEMM operand of the layer); the contraction runs over the ROWS, so MFMA fragments are read with the transposing
//   LDS read (ds_read_b64_tr_b16) from row-major tiles [32 rows][256 channels] (16 KB per operand and K step of 32 rows).
//   LDS-DMA by BUFFER loads: rows beyond the tensor read as zeros (descriptor bounds check) and add nothing.
//   Swizzle (source side, undone by the reads): 16-byte chunk ^= (row & 3) << 2 — the 4 rows x 2 segments a transposing
//   read touches per 32 lanes then cover all 64 banks.
// One workgroup = one (256 x 256 output slab, row range) unit: units x splits workgroups, the splits of all slabs over the
// same rows adjacent in the XCD-contiguous order (they share the P / Q slabs through one L2).  Same ring / wait scheme as
// pgemm_nt_kernel.
// ==========================================================================================
typedef __attribute__((ext_vector_type(4))) short pg_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short pg_s16x8_t;
__device__ __forceinline__ bf16x8_t pg_tr_frag(const char* p) {
  const pg_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) pg_s16x4_t*)(p));
  const pg_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) pg_s16x4_t*)(p + 4 * 512));
  pg_s16x8_t v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, v);
}
struct PGemmTnArgs {
  const bf16_t* P; int ldp, np;     // [rows][ldp]; np = channels of P used (multiple of 256): output rows
  const bf16_t* Q; int ldq, nq;     // [rows][ldq]; nq = channels of Q used (multiple of 256): output columns
  int rows;
  float* out; int ldo;              // out[c][k] (+=), c < np, k < nq
  int splits, steps_per_split;      // row ranges: split s covers K steps [s * steps_per_split, ...)
  const int* rowtiles;              // as PGemmNtArgs::rowtiles: the contraction then runs over the 8 K steps of each listed
  int n_rowtiles;                   // 256-row tile only (P is zero on every padding row, so nothing is lost)
};

// one (256 x 256 output slab (tp, tq), K steps [s0, s0 + nsteps)) segment of a workgroup: prologue, ring loop, atomic flush
template <int NSTAGE>
__device__ __forceinline__ void pg_tn_segment(const PGemmTnArgs& a, int tp, int tq, int s0, int nsteps, char* smem) {
  constexpr int AHEAD = NSTAGE - 1, TILE_B = 16384, STAGE_B = 2 * TILE_B, GRP = 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(tn_lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave >> 2, wb = wave & 3;
  const int* __restrict__ rowtiles = a.rowtiles;

  const pg_i32x4_t psrd = pg_make_srd(a.P, (unsigned)((size_t)a.rows * a.ldp * 2));
  const pg_i32x4_t qsrd = pg_make_srd(a.Q, (unsigned)((size_t)a.rows * a.ldq * 2));
  // DMA: instruction q of this wave fills tile rows (q*8 + wave)*2 + (lane >> 5), 16-byte chunk lane & 31 of that row
  unsigned voffP[2], voffQ[2], baseP[2], baseQ[2];
  // first row of K step s (an index into the compacted list of steps when only the listed row tiles are contracted)
  auto step_row = [&](int s) { return rowtiles ? tn_sload_i32(rowtiles, s >> 3) * 256 + (s & 7) * 32 : s * 32; };
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = (q * 8 + wave) * 2 + (lane >> 5);
    const int chunk = (lane & 31) ^ ((r & 3) << 2);
    baseP[q] = (unsigned)(((size_t)r * a.ldp + tp * 256 + chunk * 8) * 2);
    baseQ[q] = (unsigned)(((size_t)r * a.ldq + tq * 256 + chunk * 8) * 2);
  }
  {
    const unsigned row0 = (unsigned)step_row(s0);
#pragma unroll
    for (int q = 0; q < 2; ++q) { voffP[q] = baseP[q] + row0 * (unsigned)a.ldp * 2u; voffQ[q] = baseQ[q] + row0 * (unsigned)a.ldq * 2u; }
  }
  const unsigned stepP = 32u * (unsigned)a.ldp * 2u, stepQ = 32u * (unsigned)a.ldq * 2u;
  int istage = 0, sreq = s0;          // K step the next DMA group requests
  auto dma_p = [&](int q) { pg_dma16_buf(voffP[q], psrd, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + istage * STAGE_B + (q * 8 + wave) * 1024))); };
  auto dma_q = [&](int q) { pg_dma16_buf(voffQ[q], qsrd, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + istage * STAGE_B + TILE_B + (q * 8 + wave) * 1024))); };
  auto advance_issue = [&]() {
    ++sreq;
    if (rowtiles && (sreq & 7) == 0) {
      // next listed row tile (past the end of the list: beyond the tensor -> zeros by the bounds check)
      const unsigned row0 = (sreq >> 3) < a.n_rowtiles ? (unsigned)step_row(sreq) : (unsigned)a.rows + 256u;
#pragma unroll
      for (int q = 0; q < 2; ++q) { voffP[q] = baseP[q] + row0 * (unsigned)a.ldp * 2u; voffQ[q] = baseQ[q] + row0 * (unsigned)a.ldq * 2u; }
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) { voffP[q] += stepP; voffQ[q] += stepQ; }      // past the last row: zeros (bounds check)
    }
    istage = istage + 1 == NSTAGE ? 0 : istage + 1;
  };
  // fragments
  const int i16 = lane & 15, g1 = (lane >> 4) & 1, half = lane >> 5, rw = (i16 >> 2) & 3;
  const int lbase = half * 4096 + (i16 >> 2) * 512 + (2 * g1 + ((i16 & 3) >> 1)) * 16 + (i16 & 1) * 8;
  int aoff[4], boff[2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) aoff[mi] = lbase + wa * 256 + ((mi ^ rw) << 6);
#pragma unroll
  for (int nj = 0; nj < 2; ++nj) boff[nj] = TILE_B + lbase + (wb >> 1) * 256 + (((((wb & 1) << 1) | nj) ^ rw) << 6);

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll 1
  for (int d = 0; d < AHEAD; ++d) {
    dma_p(0); dma_q(0); dma_p(1); dma_q(1);
    advance_issue();
  }
  pg_wait<(AHEAD - 1) * GRP>();
  pg_barrier();
  int cstage = 0;
  for (int s = 0; s < nsteps; ++s) {
    const char* st = smem + cstage * STAGE_B;
    bf16x8_t af[2][4], bf[2][2];
    auto read_frags = [&](int ks, int buf) {
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) bf[buf][nj] = pg_tr_frag(st + boff[nj] + ks * 8192);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) af[buf][mi] = pg_tr_frag(st + aoff[mi] + ks * 8192);
    };
    read_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 0) read_frags(1, 1);
      dma_p(ks); dma_q(ks);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][mi], bf[ks][0], acc[mi][0], 0, 0, 0);
        acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][mi], bf[ks][1], acc[mi][1], 0, 0, 0);
      }
    }
    advance_issue();
    cstage = cstage + 1 == NSTAGE ? 0 : cstage + 1;
    pg_wait<(AHEAD - 1) * GRP>();
    pg_barrier();
  }
  pg_wait<0>();
  float* out = a.out + (size_t)(tp * 256 + wa * 128) * a.ldo + tq * 256 + wb * 64 + (lane & 31);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        atomic_add_f32(out + (size_t)(mi * 32 + cd_row(r, lane)) * a.ldo + nj * 32, acc[mi][nj][r]);
}

template <int DUMMY>
__global__ __launch_bounds__(512, 2) void pgemm_tn_kernel(PGemmTnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int G = gridDim.x, v = pg_virtual_id(blockIdx.x, G);
  const int tiles_q = a.nq / 256, units = (a.np / 256) * tiles_q;
  const int unit = v % units, split = v / units;
  const int tp = unit / tiles_q, tq = unit - tp * tiles_q;
  const int nsteps_all = a.rowtiles ? a.n_rowtiles * 8 : (a.rows + 31) / 32;
  const int s0 = split * a.steps_per_split;
  int nsteps = nsteps_all - s0;
  nsteps = nsteps < a.steps_per_split ? nsteps : a.steps_per_split;
  if (nsteps <= 0) return;                                         // workgroup-uniform
  pg_tn_segment<4>(a, tp, tq, s0, nsteps, smem);
}

// ==========================================================================================
// pgemm_tn_batched_kernel (round 4): the weight gradients of MANY layers of one width in ONE launch at the end of backward
// (per gradient bucket).  Layer by layer, 256 workgroups each end with 65536 f32 atomics — 12.6 - 16.7 M per layer, which at
// hidden 512 takes as long as the contraction itself (the atomics resolve memory-side at ~0.2 G/us).  The BatchNorm-backward'd
// gradients dS and the GEMM operands Q of every layer are kept anyway (one buffer per layer), so the contraction can wait:
// here a GROUP of U = (H / 256)^2 workgroups (adjacent on one XCD: they share the P / Q slabs through its L2) walks a
// contiguous range of the (layer, K step) space — unit u of the group accumulates slab u of whatever layer the range is in,
// and flushes once per layer it touches: ~2 flushes per workgroup and STEP instead of one per layer.
// ==========================================================================================
struct PGemmTnDesc {
  const bf16_t* P; const bf16_t* Q; float* out;
  int ldp, ldq, ldo, tiles_q;       // tiles_p * tiles_q == U for every descriptor of a launch
};
template <int DUMMY>
__global__ __launch_bounds__(512, 2) void pgemm_tn_batched_kernel(const PGemmTnDesc* __restrict__ descs, int n_descs, int rows, int U,
                                                                   int steps_per_group, const int* rowtiles, int n_rowtiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int G = gridDim.x, v = pg_virtual_id(blockIdx.x, G);
  const int group = v / U, unit = v - group * U;
  const int nsteps = rowtiles ? n_rowtiles * 8 : (rows + 31) / 32;
  const long total = (long)n_descs * nsteps;
  long pos = (long)group * steps_per_group;
  long end = pos + steps_per_group;
  end = end < total ? end : total;
  while (pos < end) {
    const int d = (int)(pos / nsteps), s0 = (int)(pos - (long)d * nsteps);
    const long left = end - pos;
    const int n = (int)(left < (long)(nsteps - s0) ? left : (long)(nsteps - s0));
    PGemmTnArgs a;
    a.P = descs[d].P; a.Q = descs[d].Q; a.out = descs[d].out;
    a.ldp = descs[d].ldp; a.ldq = descs[d].ldq; a.ldo = descs[d].ldo;
    a.rows = rows; a.rowtiles = rowtiles; a.n_rowtiles = n_rowtiles;
    a.np = 0; a.nq = 0; a.splits = 0; a.steps_per_split = 0;
    const int tq_n = descs[d].tiles_q;
    pg_barrier();                       // the previous segment's last fragment reads are done before its ring is refilled
    pg_tn_segment<5>(a, unit / tq_n, unit % tq_n, s0, n, smem);      // five 32 KB stages: the whole LDS
    pos += n;
  }
}
// ==========================================================================================
// fp8 weight gradient (round 5): OUT[o][c] += sum over rows r of  (P8[r][o] 2^-e[o]) * Q8[r][c]
//   P8: the BatchNorm-backward'd gradient dS as e4m3 bytes scaled by ONE power of two per COLUMN o (the contraction runs over
//       the rows, so only a per-column scale factors out of it; bn_bwd_apply writes it with the previous step's column maxima),
//   Q8: the layer's kept depthwise output as e4m3 bytes (unit scale: what the fp8 forward GEMM already reads),
//   cexp: one E8M0 byte per column of P8 = the scale 2^-e[o] to undo, passed to v_mfma_scale_f32_32x32x64_f8f6f4 as the A
//       block scale (a lane's output row IS a column of P8, so the per-column scale costs nothing: both 32-k blocks of a lane
//       carry the same byte).
// K steps of 64 rows: tiles [64 rows][256 bytes] (16 KB per operand and stage, the ring of pg_tn_segment), read with the
// byte-transposing LDS read.  ds_read_b64_tr_b8 (probed, tools/tr8_probe.hip, profiles/r05_tr8_probe.txt): in every group of 16
// lanes, lane l receives byte (l & 7) of the 8-byte pieces addressed by lanes 2 j + (l >> 3), j = 0..7 — so with lane m of a
// group pointing at row (m >> 1), columns 8 (m & 1) .. + 7, lane l gets column l of rows 0..7: 16 columns x 8 rows per group,
// four instructions (rows 8 t ..) for the 32 rows of a lane's k-half, the same row order for both operands (which is all a
// dot product needs).  Swizzle (source side of the DMA, undone by the reads): 32-byte chunk ^= row & 7 — the 8 rows x 32 bytes a
// half-wave touches per instruction then cover all 64 banks once.
// ==========================================================================================
typedef __attribute__((ext_vector_type(8))) int pg_i32x8_t;
typedef __attribute__((ext_vector_type(2))) int pg_i32x2_t;
__device__ __forceinline__ pg_i32x8_t pg_tr8_frag(const char* p) {
  pg_i32x8_t f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const pg_i32x2_t v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) pg_i32x2_t*)(p + t * 2048));
    f[2 * t] = v[0]; f[2 * t + 1] = v[1];
  }
  return f;
}
struct PGemmTnF8Desc {
  const uint8_t* P; const uint8_t* Q; float* out; const uint8_t* cexp;      // cexp[ldp]: E8M0 byte per column of P
  int ldp, ldq, ldo, tiles_q;       // row strides in bytes (= elements); tiles_p * tiles_q == U for every descriptor of a launch
};
template <int NSTAGE>
__device__ __forceinline__ void pg_tn_segment_f8(const PGemmTnF8Desc& a, int rows, const int* __restrict__ rowtiles, int n_rowtiles,
                                                 int tp, int tq, int s0, int nsteps, char* smem) {
  constexpr int AHEAD = NSTAGE - 1, TILE_B = 16384, STAGE_B = 2 * TILE_B, GRP = 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(tn_lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave >> 2, wb = wave & 3;
  const pg_i32x4_t psrd = pg_make_srd(a.P, (unsigned)((size_t)rows * a.ldp));
  const pg_i32x4_t qsrd = pg_make_srd(a.Q, (unsigned)((size_t)rows * a.ldq));
  // DMA: instruction q of this wave fills tile rows (q*8 + wave)*4 + (lane >> 4), 16-byte chunk lane & 15 of that row
  unsigned voffP[2], voffQ[2], baseP[2], baseQ[2];
  auto step_row = [&](int s) { return rowtiles ? tn_sload_i32(rowtiles, s >> 2) * 256 + (s & 3) * 64 : s * 64; };
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = (q * 8 + wave) * 4 + (lane >> 4);
    const int chunk = (lane & 15) ^ ((r & 7) << 1);
    baseP[q] = (unsigned)((size_t)r * a.ldp + tp * 256 + chunk * 16);
    baseQ[q] = (unsigned)((size_t)r * a.ldq + tq * 256 + chunk * 16);
  }
  {
    const unsigned row0 = (unsigned)step_row(s0);
#pragma unroll
    for (int q = 0; q < 2; ++q) { voffP[q] = baseP[q] + row0 * (unsigned)a.ldp; voffQ[q] = baseQ[q] + row0 * (unsigned)a.ldq; }
  }
  const unsigned stepP = 64u * (unsigned)a.ldp, stepQ = 64u * (unsigned)a.ldq;
  int istage = 0, sreq = s0;
  auto dma_p = [&](int q) { pg_dma16_buf(voffP[q], psrd, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + istage * STAGE_B + (q * 8 + wave) * 1024))); };
  auto dma_q = [&](int q) { pg_dma16_buf(voffQ[q], qsrd, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + istage * STAGE_B + TILE_B + (q * 8 + wave) * 1024))); };
  auto advance_issue = [&]() {
    ++sreq;
    if (rowtiles && (sreq & 3) == 0) {
      const unsigned row0 = (sreq >> 2) < n_rowtiles ? (unsigned)step_row(sreq) : (unsigned)rows + 256u;
#pragma unroll
      for (int q = 0; q < 2; ++q) { voffP[q] = baseP[q] + row0 * (unsigned)a.ldp; voffQ[q] = baseQ[q] + row0 * (unsigned)a.ldq; }
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) { voffP[q] += stepP; voffQ[q] += stepQ; }      // past the last row: zeros (bounds check)
    }
    istage = istage + 1 == NSTAGE ? 0 : istage + 1;
  };
  // fragments: lane L -> group g = L >> 4 (k-half g >> 1, column half g & 1), m = L & 15: piece (row 32 (g >> 1) + (m >> 1) [+ 8 t],
  // columns 16 (g & 1) + 8 (m & 1) .. + 7) of the 32-column block; rw = m >> 1 = row & 7 drives the swizzle
  const int m16 = lane & 15, g1 = (lane >> 4) & 1, half = lane >> 5, rw = m16 >> 1;
  const int lbase = (half * 32 + rw) * 256 + 16 * g1 + 8 * (m16 & 1);
  int aoff[4], boff[2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) aoff[mi] = lbase + ((((wa * 4 + mi)) ^ rw) << 5);
#pragma unroll
  for (int nj = 0; nj < 2; ++nj) boff[nj] = TILE_B + lbase + ((((wb * 2 + nj)) ^ rw) << 5);
  // A block scales: byte mi of the dword = the E8M0 byte of P column tp * 256 + wa * 128 + mi * 32 + (lane & 31)
  int xa = 0;
  {
    const uint8_t* ce = a.cexp + tp * 256 + wa * 128 + (lane & 31);
    xa = (int)ce[0] | ((int)ce[32] << 8) | ((int)ce[64] << 16) | ((int)ce[96] << 24);
  }
  constexpr int SC1 = 0x7f7f7f7f;
  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the exponent bytes above: compiler-visible loads, before the counted queue)
  asm volatile("" : "+v"(xa));
#pragma unroll 1
  for (int d = 0; d < AHEAD; ++d) {
    dma_p(0); dma_q(0); dma_p(1); dma_q(1);
    advance_issue();
  }
  pg_wait<(AHEAD - 1) * GRP>();
  pg_barrier();
  int cstage = 0;
  for (int s = 0; s < nsteps; ++s) {
    const char* st = smem + cstage * STAGE_B;
    pg_i32x8_t bfr[2];
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) bfr[nj] = pg_tr8_frag(st + boff[nj]);
    dma_p(0); dma_q(0); dma_p(1); dma_q(1);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const pg_i32x8_t af = pg_tr8_frag(st + aoff[mi]);
      auto mm = [&](auto sel) {
        constexpr int S = decltype(sel)::value;
        acc[mi][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af, bfr[0], acc[mi][0], 0, 0, S, xa, 0, SC1);
        acc[mi][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af, bfr[1], acc[mi][1], 0, 0, S, xa, 0, SC1);
      };
      if (mi == 0) mm(std::integral_constant<int, 0>{});
      else if (mi == 1) mm(std::integral_constant<int, 1>{});
      else if (mi == 2) mm(std::integral_constant<int, 2>{});
      else mm(std::integral_constant<int, 3>{});
    }
    advance_issue();
    cstage = cstage + 1 == NSTAGE ? 0 : cstage + 1;
    pg_wait<(AHEAD - 1) * GRP>();
    pg_barrier();
  }
  pg_wait<0>();
  float* out = a.out + (size_t)(tp * 256 + wa * 128) * a.ldo + tq * 256 + wb * 64 + (lane & 31);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        atomic_add_f32(out + (size_t)(mi * 32 + cd_row(r, lane)) * a.ldo + nj * 32, acc[mi][nj][r]);
}
template <int DUMMY>
__global__ __launch_bounds__(512, 2) void pgemm_tn_f8_batched_kernel(const PGemmTnF8Desc* __restrict__ descs, int n_descs, int rows, int U,
                                                                      int steps_per_group, const int* rowtiles, int n_rowtiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int G = gridDim.x, v = pg_virtual_id(blockIdx.x, G);
  const int group = v / U, unit = v - group * U;
  const int nsteps = rowtiles ? n_rowtiles * 4 : (rows + 63) / 64;
  const long total = (long)n_descs * nsteps;
  long pos = (long)group * steps_per_group;
  long end = pos + steps_per_group;
  end = end < total ? end : total;
  while (pos < end) {
    const int d = (int)(pos / nsteps), s0 = (int)(pos - (long)d * nsteps);
    const long left = end - pos;
    const int n = (int)(left < (long)(nsteps - s0) ? left : (long)(nsteps - s0));
    PGemmTnF8Desc a = descs[d];
    pg_barrier();                       // the previous segment's last fragment reads are done before its ring is refilled
    pg_tn_segment_f8<5>(a, rows, rowtiles, n_rowtiles, unit / a.tiles_q, unit % a.tiles_q, s0, n, smem);
    pos += n;
  }
}
inline int launch_pgemm_tn_f8_batched(const PGemmTnF8Desc* descs_dev, int n_descs, int rows, int U, const int* rowtiles, int n_rowtiles, hipStream_t st,
                                      int max_wgs = 256, int ld_max = 1024) {
  if (n_descs <= 0) return 0;
  if (U <= 0 || U > max_wgs || rows <= 0 || ld_max % 16) return TN_E_UNSUPPORTED;
  if ((long)(rows + 512) * ld_max >= (1L << 32)) return TN_E_UNSUPPORTED;
  const int nsteps = rowtiles ? n_rowtiles * 4 : (rows + 63) / 64;
  if (nsteps <= 0) return 0;
  const long total = (long)n_descs * nsteps;
  int groups = max_wgs / U;
  if ((long)groups > total) groups = (int)total;
  const int spg = (int)((total + groups - 1) / groups);
  groups = (int)((total + spg - 1) / spg);
  const int grid = groups * U;
  auto kern = pgemm_tn_f8_batched_kernel<0>;
  TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 163840, st, descs_dev, n_descs, rows, U, spg, rowtiles, n_rowtiles);
  return (int)hipGetLastError();
}

// ld_max: the widest row stride (elements) among the descriptors' operands — they live in device memory, so the 32-bit
// buffer-offset limit of the kernel is checked against what the caller says it put there
inline int launch_pgemm_tn_batched(const PGemmTnDesc* descs_dev, int n_descs, int rows, int U, const int* rowtiles, int n_rowtiles, hipStream_t st,
                                   int max_wgs = 256, int ld_max = 1024) {
  if (n_descs <= 0) return 0;
  if (U <= 0 || U > max_wgs || rows <= 0 || ld_max % 8) return TN_E_UNSUPPORTED;
  if ((long)(rows + 512) * ld_max * 2 >= (1L << 32)) return TN_E_UNSUPPORTED;
  const int nsteps = rowtiles ? n_rowtiles * 8 : (rows + 31) / 32;
  if (nsteps <= 0) return 0;
  const long total = (long)n_descs * nsteps;
  int groups = max_wgs / U;
  if ((long)groups > total) groups = (int)total;
  const int spg = (int)((total + groups - 1) / groups);
  groups = (int)((total + spg - 1) / spg);
  int grid = groups * U;
  // (pg_virtual_id keeps a group on one XCD only for grids that are multiples of 8; U is 4 or 16 and groups a power-of-two
  //  fraction of 256 in practice)
  auto kern = pgemm_tn_batched_kernel<0>;
  TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 163840, st, descs_dev, n_descs, rows, U, spg, rowtiles, n_rowtiles);
  return (int)hipGetLastError();
}

inline int launch_pgemm_tn(PGemmTnArgs a, hipStream_t st, int max_wgs = 256) {
  if (a.np % 256 || a.nq % 256 || a.np <= 0 || a.nq <= 0 || a.ldp % 8 || a.ldq % 8 || a.rows <= 0) return TN_E_UNSUPPORTED;
  if ((long)(a.rows + 512) * a.ldp * 2 >= (1L << 32) || (long)(a.rows + 512) * a.ldq * 2 >= (1L << 32)) return TN_E_UNSUPPORTED;
  const int units = (a.np / 256) * (a.nq / 256);
  if (units > max_wgs) return TN_E_UNSUPPORTED;
  const int nsteps = a.rowtiles ? a.n_rowtiles * 8 : (a.rows + 31) / 32;
  if (nsteps <= 0) return 0;
  // 4 slabs (512 x 512 weights): every workgroup ends with 65536 atomics, and 64 row splits make that 16.7 M per layer — as
  // long as the K loop itself.  48 splits (192 workgroups): 75.7 vs 85.9 us at 76800 rows (tools/pgemm_tn_harness)
  if (units == 4 && max_wgs == 256) max_wgs = 192;
  int splits = max_wgs / units;
  if (splits > nsteps) splits = nsteps;
  a.steps_per_split = (nsteps + splits - 1) / splits;
  a.splits = splits;
  const int grid = units * splits;
  auto kern = pgemm_tn_kernel<0>;
  TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 131072, st, a);
  return (int)hipGetLastError();
}

