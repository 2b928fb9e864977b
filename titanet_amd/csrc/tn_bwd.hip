// titanet_amd — backward orchestration: loss.backward() through the whole network as one call.
#include <string.h>

#include <algorithm>
#include <vector>

#include "tn_bwd_kernels.h"
#include "tn_internal.h"
#include "tn_pgemm.h"
#include "tn_v2_bwd_kernels.h"
#include "tn_v2_wide_kernels.h"
#ifdef TN_DEBUG_NAN
#include <stdio.h>
__global__ void dbg_count_bad_kernel(const unsigned short* p, size_t n, int is_f32, unsigned* out) {
  unsigned c = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (is_f32) { const unsigned u = reinterpret_cast<const unsigned*>(p)[i]; c += ((u >> 23) & 0xff) == 0xff; }
    else c += ((p[i] >> 7) & 0xff) == 0xff;
  }
  if (c) atomicAdd(out, c);
}
static void dbg_check(const char* name, const void* ptr, size_t n, int is_f32, hipStream_t st) {
  static unsigned* d = nullptr;
  if (!d) (void)hipMalloc(&d, 4);
  (void)hipMemsetAsync(d, 0, 4, st);
  hipLaunchKernelGGL(dbg_count_bad_kernel, dim3(512), dim3(256), 0, st, (const unsigned short*)ptr, n, is_f32, d);
  unsigned h = 0;
  (void)hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, st);
  (void)hipStreamSynchronize(st);
  fprintf(stderr, "[nan] %-28s %u non-finite of %zu (%s)\n", name, h, n, hipGetErrorString(hipGetLastError()));
}
#define DBG(name, off, n) dbg_check(name, ws + (off), (size_t)(n), 0, st)
#define DBGF(name, ptr, n) dbg_check(name, ptr, (size_t)(n), 1, st)
#else
#define DBG(name, off, n)
#define DBGF(name, ptr, n)
#endif

namespace {

BnBwd make_bnbwd(const tn_plan* p, const BnRef& bn, int rows, int training) {
  BnBwd b;
  memset(&b, 0, sizeof(b));
  b.fstats = (const float*)(p->ws + p->stats[bn.id]);
  b.bsums = (const float*)(p->ws + p->bsums[bn.id]);
  b.gamma = p->params + bn.gamma;
  b.inv_n = 1.f / (float)rows;
  b.eps = 1e-5f;
  b.batch = training ? 1.f : 0.f;
  if (p->masked && rows == p->M) {
    b.rm = plan_row_mask(p);
    if (training) b.inv_n = 1.f / (float)std::max(p->n_valid, 1);
  }
  return b;
}

template <typename AT, typename Prod, typename Epi>
int gemm_any(const GemmShape& g, const typename Prod::Args& pa, const typename Epi::Args& ea, hipStream_t st) {
  if (g.N > 128) return launch_gemm<AT, 2, 4, Prod, Epi>(g, pa, ea, 0, st);
  return launch_gemm<AT, 2, 2, Prod, Epi>(g, pa, ea, 0, st);
}

template <typename AT>
int launch_dw_bwd(const DwBwdArgs& a, int KD, hipStream_t st) {
  DwBwdArgs args = a;
  const int row_tiles = (a.M + 63) / 64;
  // ~512 workgroups: each one pays a serial reduction epilogue, so more tiles per workgroup for wide models
  args.tiles_per_wg = std::max(1, std::min(16, row_tiles * ((a.C + 63) / 64) / 512));
  dim3 grid((row_tiles + args.tiles_per_wg - 1) / args.tiles_per_wg, (a.C + 63) / 64);
  switch (KD) {
#define TN_DW_CASE(K) \
  case K: hipLaunchKernelGGL((dw_bwd_kernel<AT, K>), grid, dim3(256), 0, st, args); break;
    TN_DW_CASE(1) TN_DW_CASE(3) TN_DW_CASE(5) TN_DW_CASE(7) TN_DW_CASE(9) TN_DW_CASE(11) TN_DW_CASE(13) TN_DW_CASE(15)
#undef TN_DW_CASE
    default: return TN_E_UNSUPPORTED;
  }
  return (int)hipGetLastError();
}

template <typename AT>
int backward_impl(tn_plan* p, float gs, const float* gs_dev, const float* g_emb, float* grad_input, hipStream_t st) {
  const tn_model* m = p->model;
  const tn_config& c = m->cfg;
  const int M = p->M, H = c.hidden, D = c.enc_out, A = c.attn_hidden, Hr = c.hidden / c.se_reduction, T = p->T, B = p->B;
  const int training = p->last_training;
  const uint64_t seed = p->last_seed;
  const float pd = c.dropout;
  char* ws = p->ws;
  float* params = p->params;
  float* grads = p->grads;
  float* slabs = (float*)(ws + p->slabs);
  const int nsub = c.n_sub_blocks;
  auto bsum = [&](const BnRef& bn) -> float* { return (float*)(ws + p->bsums[bn.id]); };
  auto wt = [&](const WcRef& r) -> const void* { return ws + r.wt; };

  TN_CHECK_HIP(tn_zero_async(grads, (size_t)m->n_params * sizeof(float), st));
  TN_CHECK_HIP(tn_zero_async(ws + p->bzero_begin, p->bzero_bytes, st));
  const int use_v2 = (p->masked && T < 64) ? 0 : p->use_v2;     // variable-length batches: as the forward
  if (p->masked && c.simple_pool) return TN_E_UNSUPPORTED;
  auto identity_rows = [&]() { BnAct a = identity_act(); a.rm = plan_row_mask(p); return a; };
  // wide models (hidden 512 / 1024, bf16, train; variable-length batches: dS = 0 on padding rows, so the padding rows of the
  // other operand never matter): every 1x1 conv's backward = one in-place BatchNorm-backward
  // pass (dS) + the two pipelined LDS-DMA GEMMs of tn_pgemm.h on stored operands (data gradient dS * W, weight gradient
  // dS^T * Q straight into the gradient buffer), layer by layer
  const bool pipe = sizeof(AT) == 2 && !use_v2 && p->wide_wgrad && training && H % 256 == 0 && D % 256 == 0;
  const bool batched_wgrad = sizeof(AT) == 2 && use_v2 && training && p->wg2_layers > 0;
  // ds_ready (fused tail, the last sub-block of a mega block): dz already holds dS (bn_bwd_apply_z3_kernel rebuilt the layer's
  // incoming gradient from the tail's dZ before the skip path's in-place pass overwrote that)
  auto pipe_layer = [&](size_t dz, size_t y, const BnRef& bn, int Cout, const WcRef& wc, int Cin, size_t dx_out, const void* q, bool q_plain,
                        const BnAct& qact, int64_t wgrad_off, bool ds_ready = false, bool defer_tn = false,
                        Fp8Rows f8rows = Fp8Rows{nullptr, nullptr}, size_t w8t = 0, size_t w8ts = 0,
                        Fp8Cols fcols = Fp8Cols{nullptr, nullptr, nullptr, nullptr, 0}, bool nt_out = false) -> int {
    DBG("pipe dZ in", dz, (size_t)M * Cout); DBG("pipe Y", y, (size_t)M * Cout);
    DBGF("pipe bsums", ws + p->bsums[bn.id], TN_NREP * 2 * Cout); DBGF("pipe fstats", ws + p->stats[bn.id], TN_NREP * 2 * Cout);
    int rc = 0;
    // f8rows (fp8 plans, hidden x hidden layers): the pass also writes dS as e4m3 rows + row exponents, the data gradient below
    // reads those
    const bool f8 = f8rows.q != nullptr;
    const bool listed = p->masked && p->n_rowtiles > 0;
    const int* rowtiles = listed ? (const int*)(ws + p->rowtiles) : nullptr;
    if (!ds_ready) rc = launch_bn_bwd_apply((bf16_t*)(ws + dz), (const bf16_t*)(ws + y), make_bnbwd(p, bn, M, training), M, Cout, st,
                                            p->fp8_bwd_emu && Cout == H, f8rows, rowtiles, listed ? p->n_rowtiles : 0, fcols);
    if (rc) return rc;
    DBG("pipe dS", dz, (size_t)M * Cout);
    if (f8) {
      // dX = (dS8 2^e) * (W^T8 s[ci]): e4m3 x e4m3 on v_mfma_scale_f32_32x32x64_f8f6f4, row exponents as its A block scale,
      // the weight scales in the epilogue
      GemmShape g8{M, Cin, Cout, ws + w8t};
      PGemmNtArgs pa8{(const bf16_t*)f8rows.q, Cout, rowtiles, listed ? p->n_rowtiles : 0, f8rows.rowexp};
      PGemmEpiArgs pe8{(bf16_t*)(ws + dx_out), Cin, nullptr, nullptr, (const float*)(ws + w8ts), 0.f, nt_out ? 1 : 0};
      rc = launch_pgemm_nt_f8(g8, pa8, pe8, st);
    } else {
      GemmShape g{M, Cin, Cout, ws + wc.wt};
      PGemmNtArgs pa{(const bf16_t*)(ws + dz), Cout, rowtiles, listed ? p->n_rowtiles : 0};
      PGemmEpiArgs pe{(bf16_t*)(ws + dx_out), Cin, nullptr, nullptr, nullptr, 0.f, nt_out ? 1 : 0};
      rc = launch_pgemm_nt(g, pa, pe, st);
    }
    if (rc) return rc;
    DBG("pipe dX", dx_out, (size_t)M * Cin);
    if (q_plain && defer_tn) return 0;      // its weight gradient rides in the bucket's batched launch (finalize_bucket)
    if (q_plain) {
      PGemmTnArgs ta{(const bf16_t*)(ws + dz), Cout, Cout, (const bf16_t*)q, Cin, Cin, M, grads + wgrad_off, Cin, 0, 0, rowtiles, listed ? p->n_rowtiles : 0};
      ProfScope ps(p, TN_PROF_BWD_WGRAD, st);
      return launch_pgemm_tn(ta, st);
    }
    ProdPlain::Args pp{ws + dz, Cout, identity_act()};
    ProdPlain::Args qa{q, Cin, qact};
    return launch_wgrad<AT, ProdPlain, ProdPlain>(M, Cout, Cin, pp, qa, 0, slabs, p->slab_bytes, grads + wgrad_off, st);
  };
  // (a row mask on an otherwise plain operand is moot here: its partner dS is zero on the padding rows)
  auto is_plain = [](const BnAct& a) { return a.mode == 0 && !a.relu && !a.drop_thr; };
  // wide models: the pointwise weight gradients of the mega blocks in ONE pipelined launch per gradient bucket
  // (pgemm_tn_batched_kernel: ~2 atomic flushes per workgroup and step instead of one per layer)
  const bool tn_batched = pipe && p->tn_table != 0 && !p->v2_tn;
  // fp8 weight gradient (round 5): the sub-block layers' contraction on the f8f6f4 MFMA from the per-column-scaled e4m3 dS the
  // BatchNorm-backward passes write beside their other outputs, and the kept e4m3 depthwise outputs.  Delayed scaling: a plan's
  // first backward has no column maxima yet — it runs the bf16 contraction and only RECORDS them
  const bool f8_wgrad = tn_batched && p->fp8_wgrad && p->tn_f8_table != 0 && training;
  const bool f8_wgrad_now = f8_wgrad && p->fp8_hist_valid;
  // the forward decided from the same flag whether to store the bf16 depthwise outputs: a change in between (a rebind, anything
  // that drops the history) would send the bf16 contraction over tensors this forward never wrote
  if (f8_wgrad && !f8_wgrad_now && p->fwd_q16_skipped) return TN_E_STATE;
  auto fcols_of = [&](const BlockWs& bw_, int j) -> Fp8Cols {
    if (!f8_wgrad || bw_.dS8c.empty()) return Fp8Cols{nullptr, nullptr, nullptr, nullptr, 0};
    // (skip_bf16: with the maxima on record the data gradient reads the row-scaled copy and the weight gradient the column-scaled
    //  one — the in-place pass then stores no bf16 dS at all: 2 of the 4 bytes it writes per element)
    return Fp8Cols{(uint8_t*)(ws + bw_.dS8c[j]), (const float*)(ws + bw_.amax_prev[j]), (float*)(ws + bw_.amax_cur[j]), (uint8_t*)(ws + bw_.cexp[j]),
                   f8_wgrad_now ? 1 : 0};
  };
  // table layout (plan_upload_bwd_tables): blocks from the last down, per block the skip conv (blocks > 0), then the
  // sub-blocks from the last down
  // (the first block's skip conv joins when its input — the activated prolog output — is kept as a stored operand, p->a0)
  auto tn_entries = [&](int blk) { return nsub + ((blk > 0 || p->a0) ? 1 : 0); };
  auto tn_offset = [&](int blk) { int o = 0; for (int k = c.n_mega_blocks - 1; k > blk; --k) o += tn_entries(k); return o; };
  const bool v2_bwd = sizeof(AT) == 2 && use_v2;
  // wide models: tap / bias gradients of the depthwise convs from stored partial sums (dw_part_reduce_kernel, per bucket)
  const bool dw_part = sizeof(AT) == 2 && !use_v2 && p->dw_part != 0 && training && p->wide_dw_bwd && H % V2_C == 0;
  const bool ov = p->overlap && v2_bwd && p->side != nullptr;      // independent launches on the plan's side stream (tn_internal.h)
  // round 4: the mega-block tail backward in ONE pass (combine_bwd1_v3 finishes the SE backward per utterance; the last
  // sub-block's fused data-gradient kernel rebuilds its incoming gradient on load and stores the BatchNorm-backward'd dS for
  // the weight-gradient launch): fixed-length training batches of the headline shape
  // (variable-length batches included: padding rows carry no gradient, sums over the valid rows; small batches of long
  //  utterances: several workgroups per utterance + a finishing kernel)
  const bool tail_ok = nsub >= 2 && p->se_gu != 0 && (p->tail_parts == 1 || p->se_bacc != 0);
  const bool fuse_tail = v2_bwd && batched_wgrad && H == V2_C && Hr == 16 && c.kernel == 3 && tail_ok;
  // ... and of the wide models on the pipelined path: the rebuild sits in the streaming pass that makes the stored dS operand
  const bool fuse_tail_wide = pipe && (H == 512 || H == 1024) && Hr * 16 == H && tail_ok &&
                              !p->fp8_bwd_emu;      // (the e4m3 experiment rounds dS in the plain pass)
  // headline shape: the sub-block pointwise weight gradients of a bucket as one pipelined TN contraction over the dS that
  // dgrad_dw_v6 stores and the kept depthwise outputs (pgemm_tn_batched: 5.6 TB/s against 3.9 for the rebuild-on-load units of
  // wgrad_batched_v2, which keeps the skip convs, the epilog and the pooling units: third descriptor table)
  const bool v2_tn = fuse_tail && p->v2_tn && p->tn_table != 0;
  const int nb = c.n_mega_blocks;
  const int per_blk = nsub + 1;
  int rc_fin = 0;
  // ---- everything that turns accumulated sums / kept tensors into the final gradients of ONE bucket, then the bucket's
  // event: with grad_groups > 1 the data-parallel trainer all-reduces bucket k while the backward of earlier blocks runs
  auto finalize_bucket = [&](int k) {
    const tn_plan::GradBucket& bk = p->buckets[k];
    const bool has_blocks = bk.blk_hi >= bk.blk_lo;
    if (bk.prolog) {
      const int cur_ = p->prolog_cur;
      ProdDy::Args pa{ws + p->dA[cur_], ws + p->Y0, H, make_bnbwd(p, m->prolog_bn, M, training)};
      int rc;
      if (p->prolog_taps) {
        ProdTaps::Args qa{ws + p->x0, c.n_mels, c.prolog_kernel, T};
        rc = launch_wgrad<AT, ProdDy, ProdTaps>(M, H, c.n_mels * c.prolog_kernel, pa, qa, 0, slabs, p->slab_bytes,
                                                (float*)(ws + p->prolog_gtmp), st);
        hipLaunchKernelGGL(prolog_wgrad_untap_kernel, dim3(64), dim3(256), 0, st, (const float*)(ws + p->prolog_gtmp), H, c.n_mels,
                           c.prolog_kernel, grads + m->prolog_w);
      } else {
        ProdIm2col::Args qa{p->last_input, c.n_mels, c.prolog_kernel, T, plan_row_mask(p).len};
        rc = launch_wgrad<AT, ProdDy, ProdIm2col>(M, H, c.n_mels * c.prolog_kernel, pa, qa, 0, slabs, p->slab_bytes,
                                                  grads + m->prolog_w, st);
      }
      if (rc) { rc_fin = rc; return; }
    }
    if (dw_part && has_blocks)
      hipLaunchKernelGGL(dw_part_reduce_kernel, dim3((bk.blk_hi - bk.blk_lo + 1) * nsub, H / V2_C, c.kernel + 1), dim3(256), 0, st,
                         (const DwGradOut*)(ws + p->dw_table) + (size_t)bk.blk_lo * nsub, c.kernel, H / V2_C,
                         dw_bwd_slab_per(M, H, (p->masked && p->skip_pad_tiles) ? p->n_rowtiles : 0, 256));
    if (v2_bwd && has_blocks)
      hipLaunchKernelGGL(dw_grad_finalize_kernel, dim3((bk.blk_hi - bk.blk_lo + 1) * nsub), dim3(256), 0, st,
                         (const DwGradOut*)(ws + p->dw_table) + (size_t)bk.blk_lo * nsub, c.kernel);
    if (tn_batched && has_blocks && f8_wgrad_now) {
      // the skip convs (bf16 operands) and the sub-block layers (e4m3 operands) as two launches; tables in backward order
      const bool listed = p->masked && p->n_rowtiles > 0;
      const int* rt = listed ? (const int*)(ws + p->rowtiles) : nullptr;
      auto has_skip = [&](int blk) { return (blk > 0 || p->a0) ? 1 : 0; };
      int sfirst = 0, scount = 0;
      for (int k2 = nb - 1; k2 > bk.blk_hi; --k2) sfirst += has_skip(k2);
      for (int k2 = bk.blk_hi; k2 >= bk.blk_lo; --k2) scount += has_skip(k2);
      ProfScope ps(p, TN_PROF_BWD_WGRAD, st);
      int rc = launch_pgemm_tn_batched((const PGemmTnDesc*)(ws + p->tn_skip_table) + sfirst, scount, M, (H / 256) * (H / 256), rt,
                                       listed ? p->n_rowtiles : 0, st, 256, H);
      if (rc) { rc_fin = rc; return; }
      rc = launch_pgemm_tn_f8_batched((const PGemmTnF8Desc*)(ws + p->tn_f8_table) + (size_t)(nb - 1 - bk.blk_hi) * nsub,
                                      (bk.blk_hi - bk.blk_lo + 1) * nsub, M, (H / 256) * (H / 256), rt, listed ? p->n_rowtiles : 0, st, 256, H);
      if (rc) { rc_fin = rc; return; }
    } else
    if (tn_batched && has_blocks) {
      const int first = tn_offset(bk.blk_hi), count = tn_offset(bk.blk_lo) + tn_entries(bk.blk_lo) - first;
      const bool listed = p->masked && p->n_rowtiles > 0;
      ProfScope ps(p, TN_PROF_BWD_WGRAD, st);
      const int rc = launch_pgemm_tn_batched((const PGemmTnDesc*)(ws + p->tn_table) + first, count, M, (H / 256) * (H / 256),
                                             listed ? (const int*)(ws + p->rowtiles) : nullptr, listed ? p->n_rowtiles : 0, st, 256, H);
      if (rc) { rc_fin = rc; return; }
    }
    if (v2_tn && has_blocks) {
      // table: blocks from the last down, per block the skip conv (blocks > 0), then the sub-blocks from the last down
      auto ent = [&](int blk) { return nsub + (blk > 0 ? 1 : 0); };
      int first = 0, count = 0;
      for (int k2 = nb - 1; k2 > bk.blk_hi; --k2) first += ent(k2);
      for (int k2 = bk.blk_hi; k2 >= bk.blk_lo; --k2) count += ent(k2);
      ProfScope ps(p, TN_PROF_BWD_WGRAD, st);
      const int rc = launch_pgemm_tn_batched((const PGemmTnDesc*)(ws + p->tn_table) + first, count, M, 1, nullptr, 0, st, 256, H);
      if (rc) { rc_fin = rc; return; }
    }
    if (batched_wgrad) {
      const int upb = per_blk * p->wg2_upl;      // weight-gradient units per mega block
      int first = has_blocks ? bk.blk_lo * upb : nb * upb;
      int count = (has_blocks ? (bk.blk_hi - bk.blk_lo + 1) * upb : 0) + (bk.tail ? p->wg2_epi_slabs + p->wg2_asp_units : 0);
      if (v2_tn) {      // compact table: [block 0's skip conv, epilog / pooling units]
        const bool b0 = has_blocks && bk.blk_lo == 0;
        first = b0 ? 0 : 1;
        count = (b0 ? 1 : 0) + (bk.tail ? p->wg2_epi_slabs + p->wg2_asp_units : 0);
      }
      // variable-length batches: the first unit (block 0's skip conv: its other operand is the ACTIVATED prolog output, not
      // zero on padding rows) is done by the generic masked kernel instead (see the skip connection below)
      if (p->masked && has_blocks && bk.blk_lo == 0) { first += p->wg2_upl; count -= p->wg2_upl; }
      if (has_blocks && bk.tail && bk.blk_hi != nb - 1) { rc_fin = TN_E_STATE; return; }   // ranges must be contiguous
      if (count > 0) {
        const int chunks = (M + 31) / 32;
        const long total = (long)count * chunks;
        const int upw = (int)((total + p->wg2_grid - 1) / p->wg2_grid);
        const size_t smem = (size_t)(2 * WG2_RK * WG2_PITCH + (WG2_RK + 2) * V2_C) * sizeof(bf16_t) + (size_t)(6 + 3) * V2_C * sizeof(float);
        auto kern = wgrad_batched_v2_kernel<3, false>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
          rc_fin = TN_E_STATE; return;
        }
        {
          ProfScope ps(p, TN_PROF_BWD_WGRAD, st);
          hipLaunchKernelGGL(kern, dim3(p->wg2_grid), dim3(V2_NT), smem, st,
                             (const WgradV2Desc*)(ws + p->wg2_desc) + (v2_tn ? 2 * p->wg2_layers : fuse_tail ? p->wg2_layers : 0) + first, count, M, T,
                             chunks, upw, (int*)(ws + p->wg2_count) + first, seed, 1.f / (float)(p->masked ? std::max(p->n_valid, 1) : M));
        }
        hipLaunchKernelGGL(wgrad_v2_reduce_kernel, dim3(32, count), dim3(256), 0, st, (const WgradV2Out*)(ws + (v2_tn ? p->wg2_out3 : p->wg2_out)) + first,
                           (const int*)(ws + p->wg2_count) + first);
      }
    }
    // gradients that are functions of the accumulated sums: BatchNorm affine + conv biases, SE weights
    const BnGradDesc* bt = (const BnGradDesc*)(ws + (training ? p->bwd_table : p->bwd_table_eval));
    auto bn_range = [&](int lo, int hi) {      // BN ids [lo, hi)
      if (hi > lo) hipLaunchKernelGGL(bn_param_grad_kernel, dim3(2, hi - lo), dim3(256), 0, st, bt + lo);
    };
    if (bk.prolog && has_blocks && bk.tail) bn_range(0, m->n_bn);
    else {
      if (bk.prolog) bn_range(0, 1);
      if (has_blocks) bn_range(1 + per_blk * bk.blk_lo, 1 + per_blk * (bk.blk_hi + 1));
      if (bk.tail) bn_range(1 + per_blk * nb, m->n_bn);
    }
    if (has_blocks)
      hipLaunchKernelGGL(se_wgrad_kernel, dim3((2 * H * Hr + 255) / 256, bk.blk_hi - bk.blk_lo + 1), dim3(256), 0, st,
                         (const SeGradDesc*)(ws + p->se_table) + bk.blk_lo, B, H, Hr);
    if (k < (int)p->bucket_events.size()) (void)hipEventRecord(p->bucket_events[k], st);
  };
  const bool grouped = p->buckets.size() > 1;

  // ================= loss head -> d emb =================
  {
    HeadBwdArgs ha;
    memset(&ha, 0, sizeof(ha));
    ha.B = B; ha.E = c.emb; ha.NC = c.n_classes;
    ha.loss_type = p->last_has_loss ? c.loss_type : TN_LOSS_NONE;
    ha.dlogits = (const float*)(ws + p->dlogits);
    ha.dscale = (const float*)(ws + p->dscale);
    ha.emb = (const float*)(ws + p->emb);
    ha.emb_norm = (const float*)(ws + p->emb_norm);
    ha.W = m->fc_w >= 0 ? params + m->fc_w : nullptr;
    ha.gs = gs; ha.gs_dev = gs_dev; ha.g_embnorm = g_emb;
    ha.g_W = m->fc_w >= 0 ? grads + m->fc_w : nullptr;
    ha.g_bias = m->fc_b >= 0 ? grads + m->fc_b : nullptr;
    ha.demb = (float*)(ws + p->demb);
    if (ha.loss_type != TN_LOSS_NONE) {
      const int n = c.n_classes * c.emb;
      hipLaunchKernelGGL(head_bwd_w_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ha);
    }
    hipLaunchKernelGGL(head_bwd_x_kernel, dim3(B), dim3(256), (size_t)c.emb * sizeof(float), st, ha);
  }
  // ================= decoder tail =================
  {
    BnAct actL = make_act(p, m->lin_bn, B, training, 0, 0.f, seed, 0);
    BnAct actP = c.simple_pool ? identity_act() : make_act(p, m->pool_bn, B, training, 0, 0.f, seed, 0);
    const float* demb = (const float*)(ws + p->demb);
    float* dlin = (float*)(ws + p->dlin);
    float* dpool = (float*)(ws + p->dpooled);
    const float* lin = (const float*)(ws + p->lin);
    const float* pooled = (const float*)(ws + p->pooled);
    const int K2 = 2 * D;
    hipLaunchKernelGGL(rows_bn_bwd_sums_kernel, dim3((c.emb + 255) / 256, (B + 15) / 16), dim3(256), 0, st, demb, lin, actL, B, c.emb, bsum(m->lin_bn));
    hipLaunchKernelGGL(rows_bn_bwd_apply_kernel, dim3((B * c.emb + 255) / 256), dim3(256), 0, st, demb, lin,
                       make_bnbwd(p, m->lin_bn, B, training), B, c.emb, dlin);
    {
      // (B * 8 + 2048 floats of dynamic LDS: past the 64 KB default from B = 1793 on)
      const size_t dw_smem = (size_t)(B * 8 + 4 * 8 * 64) * sizeof(float);
      if (dw_smem > (size_t)160 * 1024) return TN_E_UNSUPPORTED;
      if (dw_smem > (size_t)48 * 1024)
        TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tail_bwd_dw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dw_smem));
    }
    hipLaunchKernelGGL(tail_bwd_dw_kernel, dim3((K2 + 63) / 64, (c.emb + 7) / 8), dim3(256), (size_t)(B * 8 + 4 * 8 * 64) * sizeof(float), st, (const float*)dlin, pooled, actP, B, K2,
                       c.emb, grads + m->lin_w);
    // d pbn -> (in place) d pooled
    hipLaunchKernelGGL(tail_bwd_dp_kernel, dim3((K2 + 255) / 256, (B + 3) / 4), dim3(256), 0, st, (const float*)dlin, params + m->lin_w, B, K2,
                       c.emb, dpool);
    if (!c.simple_pool) {
      hipLaunchKernelGGL(rows_bn_bwd_sums_kernel, dim3((K2 + 255) / 256, (B + 15) / 16), dim3(256), 0, st, (const float*)dpool, pooled, actP, B, K2,
                         bsum(m->pool_bn));
      hipLaunchKernelGGL(rows_bn_bwd_apply_kernel, dim3((B * K2 + 255) / 256), dim3(256), 0, st, (const float*)dpool, pooled,
                         make_bnbwd(p, m->pool_bn, B, training), B, K2, dpool);
    }
  }
  // ================= attentive statistics pooling =================
  BnAct acte = make_act(p, m->epi_bn, M, training, 1, 0.f, seed, 0);
  const bool attn_v2 = sizeof(AT) == 2 && !p->generic && D % 256 == 0 && A == 128 && (size_t)M * D * 2 < ((size_t)1 << 31) && ((use_v2 && H == 256) || (H >= 512 && !use_v2 && p->wide_dw_bwd));
  if (c.simple_pool) {
    // ================= simple pool: Linear(D, 2D) over B rows, then the mean over time =================
    const float* dpool = (const float*)(ws + p->dpooled);
    const float* mu = (const float*)(ws + p->mu);
    {
      ProdPlain::Args pa{dpool, 2 * D, identity_act()};
      ProdPlain::Args qa{mu, D, identity_act()};
      int rc = launch_wgrad<float, ProdPlain, ProdPlain>(B, 2 * D, D, pa, qa, 0, slabs, p->slab_bytes, grads + m->pool2_w, st);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(rows_colsum_kernel, dim3((2 * D + 255) / 256), dim3(256), 0, st, dpool, B, 2 * D, grads + m->pool2_b);
    hipLaunchKernelGGL(rows_matmul_nn_kernel, dim3((D + 255) / 256, (B + 7) / 8), dim3(256), 0, st, dpool, params + m->pool2_w, B, 2 * D,
                       D, (float*)(ws + p->dmu));
    hipLaunchKernelGGL(mean_pool_bwd_kernel<AT>, dim3(B, (D + 511) / 512), dim3(256), 0, st, (const float*)(ws + p->dmu),
                       (const AT*)(ws + p->E), acte, T, D, (AT*)(ws + p->dEbn), bsum(m->epi_bn));
  } else
  {
    if (attn_v2 && p->asp_fused) {
      // energies recomputed from the tanh outputs (the forward stored none): d e -> dE, direct d x -> dEbn
      AspV2Args fa;
      memset(&fa, 0, sizeof(fa));
      fa.HID = (const bf16_t*)(ws + p->HID); fa.W = (const bf16_t*)(ws + p->wwout.w);
      fa.bias = params + m->asp_bout;
      fa.E = (const bf16_t*)(ws + p->E); fa.actE = acte;
      fa.pooled = (float*)(ws + p->pooled); fa.smax = (float*)(ws + p->smax); fa.sinv = (float*)(ws + p->sinv); fa.qv = (float*)(ws + p->qv);
      fa.dpooled = (const float*)(ws + p->dpooled); fa.dEN = (bf16_t*)(ws + p->dE); fa.DXD = (bf16_t*)(ws + p->dEbn);
      fa.g_bout = grads + m->asp_bout; fa.B = B; fa.T = T; fa.D = D; fa.eps = 1e-6f;
      int rc = launch_asp_v2<1>(fa, st);
      if (rc) return rc;
    } else
    if (p->tail_parts > 1)
      hipLaunchKernelGGL((asp_bwd_de_kernel<AT, 16, 16>), dim3(B, (D + 127) / 128, p->tail_parts), dim3(256), 0, st, (const AT*)(ws + p->E), acte,
                         (const AT*)(ws + p->EN), T, D, 1e-6f, (const float*)(ws + p->pooled), (const float*)(ws + p->qv),
                         (const float*)(ws + p->smax), (const float*)(ws + p->sinv), (const float*)(ws + p->dpooled),
                         (AT*)(ws + p->dE), (AT*)(ws + p->dEbn), grads + m->asp_bout);
    else
      hipLaunchKernelGGL(asp_bwd_de_kernel<AT>, dim3(B, (D + 511) / 512), dim3(256), 0, st, (const AT*)(ws + p->E), acte,
                         (const AT*)(ws + p->EN), T, D, 1e-6f, (const float*)(ws + p->pooled), (const float*)(ws + p->qv),
                         (const float*)(ws + p->smax), (const float*)(ws + p->sinv), (const float*)(ws + p->dpooled),
                         (AT*)(ws + p->dE), (AT*)(ws + p->dEbn), grads + m->asp_bout);
    // d W_out[c][a] = sum_r dEN[r][c] * hid[r][a]      (units of the batched weight-gradient launch when that runs)
    const bool asp_batched = batched_wgrad && p->wg2_asp_units > 0;
    if (!asp_batched) {
      ProdPlain::Args pa{ws + p->dE, D, identity_act()};
      ProdPlain::Args qa{ws + p->HID, A, identity_act()};
      int rc = launch_wgrad<AT, ProdPlain, ProdPlain>(M, D, A, pa, qa, 0, slabs, p->slab_bytes, grads + m->asp_wout, st);
      if (rc) return rc;
    }
    // d hid_pre = (dEN * W_out) .* (1 - hid^2)
    {
      int rc;
      if (attn_v2) {
        WideInArgs wa;
        memset(&wa, 0, sizeof(wa));
        wa.A = (const bf16_t*)(ws + p->dE); wa.W = (const bf16_t*)wt(p->wwout); wa.H = (const bf16_t*)(ws + p->HID);
        wa.colsum = grads + m->asp_bin; wa.Y = (bf16_t*)(ws + p->dHP); wa.M = M; wa.KW = D;
        rc = launch_wide_in_v2<1>(wa, 256, st);
      } else {
        GemmShape g{M, A, D, wt(p->wwout)};
        ProdPlain::Args pa{ws + p->dE, D, identity_act()};
        EpiTanhBwd::Args ea{ws + p->dHP, A, ws + p->HID, grads + m->asp_bin};
        rc = gemm_any<AT, ProdPlain, EpiTanhBwd>(g, pa, ea, st);
      }
      if (rc) return rc;
    }
    // d W_in[a][c] = sum_r dHP[r][a] * x[r][c],  x = act(E)
    if (!asp_batched) {
      ProdPlain::Args pa{ws + p->dHP, A, identity_act()};
      ProdPlain::Args qa{ws + p->E, D, acte};
      int rc = launch_wgrad<AT, ProdPlain, ProdPlain>(M, A, D, pa, qa, 0, slabs, p->slab_bytes, grads + m->asp_win, st);
      if (rc) return rc;
    }
    // d x = dHP * W_in + direct term; through the epilog relu -> dEbn (+ BN backward sums)
    {
      int rc;
      if (attn_v2) {
        WideOutArgs wa;
        memset(&wa, 0, sizeof(wa));
        wa.X = (const bf16_t*)(ws + p->dHP); wa.W = (const bf16_t*)wt(p->wwin); wa.Y = (bf16_t*)(ws + p->dEbn);
        wa.Wswz = p->wwin.swt ? (const uint4*)(ws + p->wwin.swt) : nullptr;
        wa.RAW = (const bf16_t*)(ws + p->E); wa.actR = acte; wa.bsums = bsum(m->epi_bn); wa.M = M; wa.N = D;
        rc = launch_wide_out_v2<128, 2>(wa, 256, st);
      } else {
        GemmShape g{M, D, A, wt(p->wwin)};
        ProdPlain::Args pa{ws + p->dHP, A, identity_act()};
        EpiAddMaskStore::Args ea{ws + p->dEbn, D, ws + p->E, acte, bsum(m->epi_bn)};
        rc = gemm_any<AT, ProdPlain, EpiAddMaskStore>(g, pa, ea, st);
      }
      if (rc) return rc;
    }
  }
  // ================= epilog 1x1 conv =================
  const void* x_last = nb > 0 ? (const void*)(ws + p->blk[nb - 1].OUT) : (const void*)(ws + p->Y0);
  BnAct act0 = make_act(p, m->prolog_bn, M, training, 1, 0.f, seed, 0);
  BnAct act_last = nb > 0 ? identity_rows() : act0;
  int cur = 0;   // dA[cur] holds the gradient wrt the current block output
  if (pipe && nb > 0) {
    int rc = pipe_layer(p->dEbn, p->E, m->epi_bn, D, p->wepi, H, p->dA[cur], x_last, is_plain(act_last), act_last, m->epi_w);
    if (rc) return rc;
  } else
  {
    ProdDy::Args pa{ws + p->dEbn, ws + p->E, D, make_bnbwd(p, m->epi_bn, M, training)};
    {
      ProdPlain::Args qa{x_last, H, act_last};
      if (!(batched_wgrad && p->wg2_epi_slabs > 0)) {
        int rc = launch_wgrad<AT, ProdDy, ProdPlain>(M, D, H, pa, qa, 0, slabs, p->slab_bytes, grads + m->epi_w, st);
        if (rc) return rc;
      }
    }
    int rc;
    // (the pipelined generic GEMM, BatchNorm backward on load, does this K = 1536 product in 183 us; a weights-in-registers
    //  kernel that re-streamed its weight slab per 64 rows took 207 us and was removed)
    {
      GemmShape g{M, H, D, wt(p->wepi)};
      EpiStoreArgs ea{ws + p->dA[cur], H, nullptr, nullptr};
      rc = gemm_any<AT, ProdDy, EpiStore>(g, pa, ea, st);
    }
    if (rc) return rc;
  }
  if (grouped) { finalize_bucket(0); if (rc_fin) return rc_fin; }
  // ================= mega blocks, last to first =================
  int next_bucket = grouped ? 1 : 0;
  for (int i = nb - 1; i >= 0; --i) {
    const MegaBlockRef& mb = m->blocks[i];
    BlockWs& bw = p->blk[i];
    const void* xin = i > 0 ? (const void*)(ws + p->blk[i - 1].OUT) : (const void*)(ws + p->Y0);
    BnAct actx = i > 0 ? identity_rows() : act0;
    BnAct act3 = make_act(p, mb.sub[nsub - 1].bn, M, training, 1, pd, seed, i * (nsub + 1) + nsub - 1);
    BnAct acts = make_act(p, mb.bnskip, M, training, 0, 0.f, seed, 0);
    const float inv_keep = (training && pd > 0.f) ? 1.f / (1.f - pd) : 1.f;
    const int CV = H / 8, TG = 512 / CV;
    {
      size_t smem = (size_t)(7 * H + TG * 3 * H) * sizeof(float);
      auto k1 = combine_bwd1_kernel<AT>;
      if (smem > 64 * 1024) TN_CHECK_HIP(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      uint32_t othr = 0, okey = 0;
      if (training && pd > 0.f) {
        othr = (uint32_t)lrintf(pd * 65536.f);
        okey = tn_layer_key(seed, (uint32_t)(i * (nsub + 1) + nsub));      // the block output's dropout stream (forward: combine)
      }
      int rc1 = -1000;
      float* dgate_out = (float*)(ws + ((v2_bwd && Hr == 16) ? bw.dgate : bw.dpre2));
      if (fuse_tail || fuse_tail_wide) {
        CombineBwd1V3Args c3;
        memset(&c3, 0, sizeof(c3));
        CombineBwd1V2Args& c1 = c3.a1;
        c1.dOUT = (const bf16_t*)(ws + p->dA[cur]); c1.gate = (const float*)(ws + bw.g); c1.Y3 = (const bf16_t*)(ws + bw.Y[nsub - 1]);
        c1.act3 = act3; c1.S = (const bf16_t*)(ws + bw.S); c1.actS = acts; c1.dZ = (bf16_t*)(ws + bw.dZk);
        c1.bsumsS = bsum(mb.bnskip); c1.T = T; c1.parts = p->tail_parts; c1.inv_keep = inv_keep; c1.drop_thr = othr; c1.drop_key = okey;
        c1.key_add = (const uint32_t*)(ws + p->step_state) + 2;
        c3.len = plan_row_mask(p).len; c3.bacc = p->se_bacc ? (float*)(ws + p->se_bacc) : nullptr;
        c3.hid = (const float*)(ws + bw.h); c3.W1 = params + mb.se_w1; c3.W2 = params + mb.se_w2;
        c3.dpre2 = (float*)(ws + bw.dpre2); c3.dpre1 = (float*)(ws + bw.dpre1); c3.gu = (float*)(ws + p->se_gu);
        c3.bsums3 = bsum(mb.sub[nsub - 1].bn);
        int rc = launch_combine_bwd1_v3(c3, B, H, st);
        if (rc) return rc == -1000 ? TN_E_UNSUPPORTED : rc;
        if (fuse_tail_wide) {
          // dS of the last sub-block NOW: the skip connection's BatchNorm-backward pass below works in place on dZk
          rc = launch_bn_bwd_apply_z3((const bf16_t*)(ws + bw.dZk), (const bf16_t*)(ws + bw.Y[nsub - 1]),
                                      make_bnbwd(p, mb.sub[nsub - 1].bn, M, training), act3, (const float*)(ws + p->se_gu),
                                      (bf16_t*)(ws + bw.dY[nsub - 1]), M, H, T, st,
                                      (p->fp8_bwd && !bw.w8t.empty()) ? Fp8Rows{(uint8_t*)(ws + p->ds8), (uint8_t*)(ws + p->dsexp)}
                                                                     : Fp8Rows{nullptr, nullptr},
                                      fcols_of(bw, nsub - 1));
          if (rc) return rc;
        }
      } else {
      if (sizeof(AT) == 2 && v2_bwd && H == V2_C) {
        CombineBwd1V2Args c1;
        memset(&c1, 0, sizeof(c1));
        c1.dOUT = (const bf16_t*)(ws + p->dA[cur]); c1.gate = (const float*)(ws + bw.g); c1.Y3 = (const bf16_t*)(ws + bw.Y[nsub - 1]);
        c1.act3 = act3; c1.S = (const bf16_t*)(ws + bw.S); c1.actS = acts; c1.dZ = (bf16_t*)(ws + bw.dZk); c1.dgate = dgate_out;
        c1.bsumsS = bsum(mb.bnskip); c1.T = T; c1.parts = 1; c1.inv_keep = inv_keep; c1.drop_thr = othr; c1.drop_key = okey;
        c1.key_add = (const uint32_t*)(ws + p->step_state) + 2;
        rc1 = launch_combine_bwd1_v2(c1, B, st);
        if (rc1 > 0) return rc1;
      }
      // generic kernels: tail_parts workgroups per utterance for small batches of long utterances (dgate accumulated in
      // the pre-zeroed dgate_acc, pass 2 reads it and writes dpre2)
      const int parts = p->tail_parts;
      float* dgate_acc = parts > 1 ? (float*)(ws + p->dgate_acc) + (size_t)i * B * H : nullptr;
      if (rc1 == -1000)
      hipLaunchKernelGGL(k1, dim3(B, parts), dim3(512), smem, st, (const AT*)(ws + p->dA[cur]), (const float*)(ws + bw.g),
                         (const AT*)(ws + bw.Y[nsub - 1]), act3, (const AT*)(ws + bw.S), acts, T, H, inv_keep, othr, okey,
                         (const uint32_t*)(ws + p->step_state) + 2, (AT*)(ws + bw.dZk),
                         dgate_acc ? dgate_acc : (float*)(ws + ((v2_bwd && Hr == 16) ? bw.dgate : bw.dpre2)), bsum(mb.bnskip));
      int rc2 = -1000;
      const bool split = rc1 == -1000 && dgate_acc;        // the generic pass 1 ran in parts: so does pass 2
      if (v2_bwd && Hr == 16 && !split) {
        // pass 1 left dgate in bw.dpre2; copy-free hand-over: pass 2 (v2) reads it from bw.dgate, so move the pointer roles:
        // pass 1 wrote to bw.dgate (see above), pass 2 writes bw.dpre2
        CombineBwd2V2Args ca;
        memset(&ca, 0, sizeof(ca));
        ca.dZ = (const bf16_t*)(ws + bw.dZk); ca.Y3 = (const bf16_t*)(ws + bw.Y[nsub - 1]); ca.act3 = act3;
        ca.gate = (const float*)(ws + bw.g); ca.hid = (const float*)(ws + bw.h); ca.dgate = (const float*)(ws + bw.dgate);
        ca.dpre2 = (float*)(ws + bw.dpre2); ca.dpre1 = (float*)(ws + bw.dpre1);
        ca.W1 = params + mb.se_w1; ca.W2 = params + mb.se_w2; ca.dYbn = (bf16_t*)(ws + bw.dY[nsub - 1]);
        ca.bsums3 = bsum(mb.sub[nsub - 1].bn); ca.T = T; ca.parts = 2; ca.len = plan_row_mask(p).len;
        rc2 = launch_combine_bwd2_v2(ca, B, st);
        if (rc2 > 0) return rc2;
      }
      if (rc2 == -1000) {
        if (v2_bwd && Hr == 16 && !split)   // pass 1 wrote dgate to bw.dgate; the generic pass 2 works in place on bw.dpre2
          TN_CHECK_HIP(hipMemcpyAsync(ws + bw.dpre2, ws + bw.dgate, (size_t)B * H * sizeof(float), hipMemcpyDeviceToDevice, st));
        smem = (size_t)(7 * H + ((Hr + 3) & ~3) + TG * 2 * H) * sizeof(float);
        auto k2 = combine_bwd2_kernel<AT>;
        if (smem > 64 * 1024) TN_CHECK_HIP(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(k2, dim3(B, split ? parts : 1), dim3(512), smem, st, (const AT*)(ws + bw.dZk), (const AT*)(ws + bw.Y[nsub - 1]), act3,
                           (const float*)(ws + bw.g), (const float*)(ws + bw.h), split ? (const float*)dgate_acc : (const float*)(ws + bw.dpre2),
                           (float*)(ws + bw.dpre2), (float*)(ws + bw.dpre1),
                           params + mb.se_w1, params + mb.se_w2, T, H, Hr, (AT*)(ws + bw.dY[nsub - 1]), bsum(mb.sub[nsub - 1].bn));
      }
      }
    }
    DBG("block dOUT", p->dA[cur], (size_t)M * H); DBG("combine dZk", bw.dZk, (size_t)M * H); DBG("combine dY3", bw.dY[nsub - 1], (size_t)M * H);
    // ---- skip connection: d S = BN-backward(dZ); dXs = dS * W_skip; d W_skip = dS^T X
    bool skip_on_side = false;
    if (pipe) {
      // (fp8 plans: its own e4m3 dS buffer — the last sub-block's, written by the one-pass tail above, is still pending)
      const bool f8s = p->fp8_bwd && bw.w8t_skip != 0;
      const bool a0 = i == 0 && p->a0 != 0;        // the skip conv read the stored activated prolog output (forward_impl)
      const void* xs = a0 ? (const void*)(ws + p->a0) : xin;
      const BnAct axs = a0 ? identity_rows() : actx;
      int rc = pipe_layer(bw.dZk, bw.S, mb.bnskip, H, bw.wskip, H, p->dXs, xs, is_plain(axs), axs, mb.wskip, false, tn_batched && (i > 0 || a0),
                          f8s ? Fp8Rows{(uint8_t*)(ws + p->ds8s), (uint8_t*)(ws + p->dsexps)} : Fp8Rows{nullptr, nullptr},
                          f8s ? bw.w8t_skip : 0, f8s ? bw.w8ts_skip : 0, Fp8Cols{nullptr, nullptr, nullptr, nullptr, 0}, p->nt_skip);
      if (rc) return rc;
    } else
    {
      ProdDy::Args pa{ws + bw.dZk, ws + bw.S, H, make_bnbwd(p, mb.bnskip, M, training)};
      ProdPlain::Args qa{xin, H, actx};
      int rc = 0;
      if (!batched_wgrad || (p->masked && i == 0))
        rc = launch_wgrad<AT, ProdDy, ProdPlain>(M, H, H, pa, qa, 0, slabs, p->slab_bytes, grads + mb.wskip, st);
      if (rc) return rc;
      if (v2_bwd) {
        DgradV2Args va;
        memset(&va, 0, sizeof(va));
        va.dZ = (const bf16_t*)(ws + bw.dZk); va.Y = (const bf16_t*)(ws + bw.S); va.bn = pa.bn;
        va.Wt = (const bf16_t*)(ws + bw.wskip.wt); va.OUT = (bf16_t*)(ws + p->dXs); va.M = M;
        va.Wswz = bw.wskip.swt ? (const uint4*)(ws + bw.wskip.swt) : nullptr;
        va.dS_out = (v2_tn && bw.dS_skip) ? (bf16_t*)(ws + bw.dS_skip) : nullptr;      // operand of the pipelined weight-gradient launch
        // side stream: the skip data gradient depends on the tail pass above only, and only the FIRST sub-block's launch
        // (its ADD operand) needs it — it runs beside the data-gradient launches of the other sub-blocks.  (p->dXs is shared by
        // all blocks: the fork event of block i - 1 sits behind block i's last reader on `st`)
        hipStream_t ss = st;
        if (ov && nsub >= 2) {
          TN_CHECK_HIP(hipEventRecord(p->ov_events[4 * i + 2], st));
          TN_CHECK_HIP(hipStreamWaitEvent(p->side, p->ov_events[4 * i + 2], 0));
          ss = p->side;
        }
        {
          ProfScope ps(p, TN_PROF_BWD_DGRAD, ss);
          rc = launch_dgrad_v2<64>(va, 256, ss);
        }
        if (ss != st) { TN_CHECK_HIP(hipEventRecord(p->ov_events[4 * i + 3], p->side)); skip_on_side = true; }
      } else {
        GemmShape g{M, H, H, wt(bw.wskip)};
        EpiStoreArgs ea{ws + p->dXs, H, nullptr, nullptr};
        rc = gemm_any<AT, ProdDy, EpiStore>(g, pa, ea, st);
      }
      if (rc) return rc;
    }
    // ---- sub-blocks, last to first
    for (int j = nsub - 1; j >= 0; --j) {
      const SubBlockRef& sb = mb.sub[j];
      const void* sin = j > 0 ? (const void*)(ws + bw.Y[j - 1]) : xin;
      BnAct asin = j > 0 ? make_act(p, mb.sub[j - 1].bn, M, training, 1, pd, seed, i * (nsub + 1) + j - 1) : actx;
      ProdDy::Args pa{ws + bw.dY[j], ws + bw.Y[j], H, make_bnbwd(p, sb.bn, M, training)};
      if (!batched_wgrad && !pipe) {
        int rc;
        if (p->save_q && training) {
          // the forward kept the depthwise output (the pointwise GEMM's operand): plain operand, no activation / stencil recompute
          ProdPlain::Args qa{ws + bw.Q[j], H, identity_act()};
          rc = launch_wgrad<AT, ProdDy, ProdPlain>(M, H, H, pa, qa, 0, slabs, p->slab_bytes, grads + sb.wpw, st, p, TN_PROF_BWD_WGRAD);
        } else {
          ProdDw::Args qa{sin, H, asin, params + sb.wdw, params + sb.bdw, c.kernel, T, nullptr};
          rc = launch_wgrad<AT, ProdDy, ProdDw>(M, H, H, pa, qa, c.kernel, slabs, p->slab_bytes, grads + sb.wpw, st, p,
                                                TN_PROF_BWD_WGRAD);
        }
        if (rc) return rc;
      }
      if (v2_bwd) {
        // (A + B) in one pass: dD never leaves the CU (dgrad_dw_v6)
        DgradDwArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.dZ = (const bf16_t*)(ws + bw.dY[j]); fa.Y = (const bf16_t*)(ws + bw.Y[j]); fa.bn = pa.bn;
        fa.Wswz = bw.wpw[j].swt ? (const uint4*)(ws + bw.wpw[j].swt) : nullptr;
        fa.X = (const bf16_t*)sin; fa.actX = asin; fa.wdw = params + sb.wdw; fa.M = M; fa.T = T;
        fa.gacc = (float*)(ws + p->dw_gacc) + (size_t)(i * nsub + j) * TN_NREP * (c.kernel + 1) * H;
        if (fuse_tail && j == nsub - 1) {
          // the incoming gradient is rebuilt from the tail's dZ; dS goes where dY[j] would have been (the weight-gradient unit
          // of this layer reads it as a plain operand: second descriptor table)
          fa.dZ = (const bf16_t*)(ws + bw.dZk); fa.gu = (const float*)(ws + p->se_gu); fa.act3 = act3;
          fa.dS_out = (bf16_t*)(ws + bw.dY[j]);
        } else if (v2_tn) {
          fa.dS_out = (bf16_t*)(ws + bw.dS[j]);      // kept for the pipelined weight-gradient launch (finalize_bucket)
        }
        if (j > 0) {
          fa.ADD = nullptr; fa.OUT = (bf16_t*)(ws + bw.dY[j - 1]); fa.bsumsX = bsum(mb.sub[j - 1].bn);
        } else {
          fa.ADD = (const bf16_t*)(ws + p->dXs); fa.OUT = (bf16_t*)(ws + p->dA[cur ^ 1]);
          fa.bsumsX = (i == 0) ? bsum(m->prolog_bn) : nullptr;
          if (skip_on_side) { TN_CHECK_HIP(hipStreamWaitEvent(st, p->ov_events[4 * i + 3], 0)); skip_on_side = false; }   // join: dXs
        }
        int rc;
        {
          ProfScope ps(p, TN_PROF_BWD_DW, st);
          rc = launch_dgrad_dw_v6(fa, 256, st);
        }
        if (rc) return rc == -1000 ? TN_E_UNSUPPORTED : rc;
        continue;
      }
      if (pipe) {
        const bool z3 = fuse_tail_wide && j == nsub - 1;
        const bool f8 = p->fp8_bwd && !bw.w8t.empty();
        int rc = pipe_layer(bw.dY[j], bw.Y[j], sb.bn, H, bw.wpw[j], H, p->dD, ws + bw.Q[j], true, identity_act(), sb.wpw,
                            z3, tn_batched, f8 ? Fp8Rows{(uint8_t*)(ws + p->ds8), (uint8_t*)(ws + p->dsexp)} : Fp8Rows{nullptr, nullptr},
                            f8 ? bw.w8t[j] : 0, f8 ? bw.w8ts[j] : 0, f8 ? fcols_of(bw, j) : Fp8Cols{nullptr, nullptr, nullptr, nullptr, 0}, (p->rw_nt & 2) != 0);
        if (rc) return rc;
      } else {
        GemmShape g{M, H, H, wt(bw.wpw[j])};
        EpiStoreArgs ea{ws + p->dD, H, nullptr, nullptr};
        int rc;
        {
          ProfScope ps(p, TN_PROF_BWD_DGRAD, st);
          rc = gemm_any<AT, ProdDy, EpiStore>(g, pa, ea, st);
        }
        if (rc) return rc;
      }
      DwBwdArgs da;
      memset(&da, 0, sizeof(da));
      da.dD = ws + p->dD;
      da.XRAW = sin;
      da.actX = asin;
      da.wdw = params + sb.wdw;
      da.g_wdw = grads + sb.wdw;
      da.g_bdw = grads + sb.bdw;
      da.M = M; da.T = T; da.C = H;
      if (j > 0) {
        da.ADD = nullptr;
        da.OUT = ws + bw.dY[j - 1];
        da.bsumsX = bsum(mb.sub[j - 1].bn);
      } else {
        da.ADD = ws + p->dXs;
        da.OUT = ws + p->dA[cur ^ 1];
        da.bsumsX = (i == 0) ? bsum(m->prolog_bn) : nullptr;
      }
      int rc = -1000;
      if (sizeof(AT) == 2 && H % V2_C == 0 && (c.kernel == 7 || c.kernel == 11) && p->wide_dw_bwd) {
        // wide models: streaming slab kernel with the tap windows in registers (tn_v2_bwd_kernels.h)
        DwBwdSlabArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.dD = (const bf16_t*)da.dD; sa.X = (const bf16_t*)da.XRAW; sa.actX = da.actX; sa.ADD = (const bf16_t*)da.ADD;
        sa.OUT = (bf16_t*)da.OUT; sa.wdw = da.wdw; sa.g_wdw = da.g_wdw; sa.g_bdw = da.g_bdw; sa.bsumsX = da.bsumsX;
        sa.M = M; sa.T = T; sa.C = H;
        if (p->masked && p->skip_pad_tiles && p->n_rowtiles > 0) { sa.rowtiles = (const int*)(ws + p->rowtiles); sa.n_rowtiles = p->n_rowtiles; }
        if (dw_part) sa.gpart = (float*)(ws + p->dw_part + (size_t)(i * nsub + j) * p->dw_part_stride);
        ProfScope ps(p, TN_PROF_BWD_DW, st);
        rc = c.kernel == 7 ? launch_dw_bwd_slab<7>(sa, 256, st) : launch_dw_bwd_slab<11>(sa, 256, st);
        if (rc > 0) return rc;
        if (rc == -1000 && dw_part) return TN_E_STATE;      // (the bucket's reduction would overwrite the generic kernel's sums)
      }
      if (rc == -1000) {
        ProfScope ps(p, TN_PROF_BWD_DW, st);
        rc = launch_dw_bwd<AT>(da, c.kernel, st);
      }
      if (rc) return rc;
      DBG("dw_bwd OUT", (const char*)da.OUT - ws, (size_t)M * H);
    }
    cur ^= 1;
    if (grouped && next_bucket < (int)p->buckets.size() && p->buckets[next_bucket].blk_lo == i && !p->buckets[next_bucket].prolog) {
      finalize_bucket(next_bucket++);
      if (rc_fin) return rc_fin;
    }
  }
  // ================= prolog conv =================
  {
    if (nb == 0) {
      // no mega blocks: dA[cur] holds d act0(Y0); push it through the prolog relu/BN mask
      return TN_E_UNSUPPORTED;
    }
    p->prolog_cur = cur;
    if (grad_input && !p->last_input && !p->prolog_taps) return TN_E_STATE;
    if (grad_input) {
      ProdDy::Args pa{ws + p->dA[cur], ws + p->Y0, H, make_bnbwd(p, m->prolog_bn, M, training)};
      const int n = B * c.n_mels * T;
      hipLaunchKernelGGL(prolog_input_grad_kernel<AT>, dim3((n + 255) / 256), dim3(256), 0, st, (const AT*)(ws + p->dA[cur]),
                         (const AT*)(ws + p->Y0), pa.bn, params + m->prolog_w, B, c.n_mels, T, H, c.prolog_kernel, grad_input);
    }
  }
  // ================= the last bucket (the only one without grouping): prolog weight gradient, the deferred pointwise
  // weight gradients (v2: all of the bucket's layers in one balanced launch), sums -> BatchNorm / bias / SE gradients
  finalize_bucket((int)p->buckets.size() - 1);
  if (rc_fin) return rc_fin;
  if (f8_wgrad && nb > 0 && !p->blk[0].amax_cur.empty()) {
    // this step's column maxima become the next step's scales (one contiguous run per side, plan layout)
    TN_CHECK_HIP(hipMemcpyAsync(ws + p->blk[0].amax_prev[0], ws + p->blk[0].amax_cur[0], (size_t)nb * nsub * H * sizeof(float), hipMemcpyDeviceToDevice, st));
    p->fp8_hist_valid = true;
  }
  return (int)hipGetLastError();
}

}  // namespace

// descriptor tables for the two "all layers at once" kernels; uploaded at bind time
int plan_upload_bwd_tables(tn_plan* p, hipStream_t st) {
  const tn_model* m = p->model;
  if (!p->grads) return 0;
  std::vector<BnGradDesc> bd(m->n_bn);
  std::vector<int64_t> bias_of(m->n_bn, -1);
  bias_of[m->prolog_bn.id] = m->prolog_b;
  for (auto& mb : m->blocks) {
    for (auto& sb : mb.sub) bias_of[sb.bn.id] = sb.bpw;
    bias_of[mb.bnskip.id] = mb.bskip;
  }
  bias_of[m->epi_bn.id] = m->epi_b;
  bias_of[m->lin_bn.id] = m->lin_b;
  for (int i = 0; i < m->n_bn; ++i) {
    const BnRef& r = m->all_bn[i];
    const int rows = (i == m->pool_bn.id || i == m->lin_bn.id) ? p->B : p->M;
    bd[i].bn = make_bnbwd(p, r, rows, 1);   // mode patched per call (see below): tables hold train mode
    bd[i].g_gamma = p->grads + r.gamma;
    bd[i].g_beta = p->grads + r.beta;
    bd[i].g_bias = bias_of[i] >= 0 ? p->grads + bias_of[i] : nullptr;
    bd[i].C = r.C;
    bd[i].n = rows;
  }
  if (sizeof(BnGradDesc) * bd.size() > p->bwd_table_bytes) return TN_E_STATE;
  TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->bwd_table, bd.data(), bd.size() * sizeof(BnGradDesc), hipMemcpyHostToDevice, st));
  // eval-mode copy (mode 2) right behind it
  for (auto& d : bd) d.bn.batch = 0.f;
  TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->bwd_table_eval, bd.data(), bd.size() * sizeof(BnGradDesc), hipMemcpyHostToDevice, st));
  std::vector<SeGradDesc> sd(m->blocks.size());
  for (size_t i = 0; i < m->blocks.size(); ++i) {
    sd[i].dpre2 = (const float*)(p->ws + p->blk[i].dpre2);
    sd[i].hid = (const float*)(p->ws + p->blk[i].h);
    sd[i].dpre1 = (const float*)(p->ws + p->blk[i].dpre1);
    sd[i].mean = (const float*)(p->ws + p->blk[i].m);
    sd[i].g_w1 = p->grads + m->blocks[i].se_w1;
    sd[i].g_w2 = p->grads + m->blocks[i].se_w2;
  }
  if (!sd.empty())
    TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->se_table, sd.data(), sd.size() * sizeof(SeGradDesc), hipMemcpyHostToDevice, st));
  if (p->tn_table) {
    const tn_config& c = m->cfg;
    const int nsub = c.n_sub_blocks, H = c.hidden;
    std::vector<PGemmTnDesc> td;
    if (p->v2_tn) {
      // headline-shape plans: the sub-block layers only, P = the dS that dgrad_dw_v6 stored (the last sub-block's sits where its
      // incoming gradient would have been: one-pass tail), Q = the kept depthwise output
      for (int i = c.n_mega_blocks - 1; i >= 0; --i) {
        const BlockWs& bw = p->blk[i];
        // the skip conv of blocks > 0: P = the dS stored by dgrad_v2, Q = the previous block's output (stored activated)
        if (i > 0) td.push_back(PGemmTnDesc{(const bf16_t*)(p->ws + bw.dS_skip), (const bf16_t*)(p->ws + p->blk[i - 1].OUT),
                                            p->grads + m->blocks[i].wskip, H, H, H, 1});
        for (int j = nsub - 1; j >= 0; --j)
          td.push_back(PGemmTnDesc{(const bf16_t*)(p->ws + (j == nsub - 1 ? bw.dY[j] : bw.dS[j])), (const bf16_t*)(p->ws + bw.Q[j]),
                                   p->grads + m->blocks[i].sub[j].wpw, H, H, H, 1});
      }
    } else
    for (int i = c.n_mega_blocks - 1; i >= 0; --i) {
      const BlockWs& bw = p->blk[i];
      if (i > 0 || p->a0)
        td.push_back(PGemmTnDesc{(const bf16_t*)(p->ws + bw.dZk), (const bf16_t*)(p->ws + (i > 0 ? p->blk[i - 1].OUT : p->a0)),
                                 p->grads + m->blocks[i].wskip, H, H, H, H / 256});
      for (int j = nsub - 1; j >= 0; --j)
        td.push_back(PGemmTnDesc{(const bf16_t*)(p->ws + bw.dY[j]), (const bf16_t*)(p->ws + bw.Q[j]), p->grads + m->blocks[i].sub[j].wpw,
                                 H, H, H, H / 256});
    }
    if (sizeof(PGemmTnDesc) > 64) return TN_E_STATE;
    if (!td.empty())
      TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->tn_table, td.data(), td.size() * sizeof(PGemmTnDesc), hipMemcpyHostToDevice, st));
    std::vector<PGemmTnDesc> tsk;
    std::vector<PGemmTnF8Desc> tf8;
    if (p->fp8_wgrad && p->tn_f8_table && !p->v2_tn) {
      // fp8 weight gradient: the skip convs alone (bf16 operands) and the sub-block layers (e4m3 operands), backward order
      if (sizeof(PGemmTnF8Desc) > 64) return TN_E_STATE;
      for (int i = c.n_mega_blocks - 1; i >= 0; --i) {
        const BlockWs& bw = p->blk[i];
        if (i > 0 || p->a0)
          tsk.push_back(PGemmTnDesc{(const bf16_t*)(p->ws + bw.dZk), (const bf16_t*)(p->ws + (i > 0 ? p->blk[i - 1].OUT : p->a0)),
                                    p->grads + m->blocks[i].wskip, H, H, H, H / 256});
        for (int j = nsub - 1; j >= 0; --j)
          tf8.push_back(PGemmTnF8Desc{(const uint8_t*)(p->ws + bw.dS8c[j]), (const uint8_t*)(p->ws + bw.Q8[j]), p->grads + m->blocks[i].sub[j].wpw,
                                      (const uint8_t*)(p->ws + bw.cexp[j]), H, H, H, H / 256});
      }
      if (!tsk.empty()) TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->tn_skip_table, tsk.data(), tsk.size() * sizeof(PGemmTnDesc), hipMemcpyHostToDevice, st));
      if (!tf8.empty()) TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->tn_f8_table, tf8.data(), tf8.size() * sizeof(PGemmTnF8Desc), hipMemcpyHostToDevice, st));
    }
    TN_CHECK_HIP(hipStreamSynchronize(st));
  }
  if (p->use_v2 && p->wg2_layers > 0) {
    const tn_config& c = m->cfg;
    const int nsub = c.n_sub_blocks, M = p->M, H = c.hidden, hs = H / 256;
    std::vector<WgradV2Desc> wd;
    std::vector<WgradV2Out> wo;
    const size_t slab_stride = (size_t)p->wg2_maxparts * 256 * 256;
    // one unit per 256 x 256 slab of d W[out][in]: P = BN-backward(dZ, Y)[:, out slab], Q = the layer's GEMM operand[:, in slab]
    auto add = [&](size_t dz, size_t y, const BnRef& bn, int p_width, const void* x, const BnAct& ax, int q_width, int drop_layer,
                   int64_t wdw, int64_t bdw, int64_t wout, int po, int qo) {
      WgradV2Desc d;
      memset(&d, 0, sizeof(d));
      d.dZ = (const bf16_t*)(p->ws + dz) + (size_t)po * 256;
      d.Y = (const bf16_t*)(p->ws + y) + (size_t)po * 256;
      d.fstats = (const float*)(p->ws + p->stats[bn.id]);
      d.bsums = (const float*)(p->ws + p->bsums[bn.id]);
      d.gamma = p->params + bn.gamma;          // indexed with chan0 + channel, like the statistics
      d.inv_n = 1.f / (float)M; d.eps = 1e-5f; d.batch = 1.f;
      d.X = (const bf16_t*)x;
      d.actX = ax;
      d.drop_layer = drop_layer;
      d.wdw = wdw >= 0 ? p->params + wdw : nullptr;
      d.bdw = bdw >= 0 ? p->params + bdw : nullptr;
      d.ldp = p_width; d.statC = p_width; d.chan0 = po * 256;
      d.ldq = q_width; d.q0 = qo * 256;
      d.slabs = (float*)(p->ws + p->wg2_slabs) + wd.size() * slab_stride;
      WgradV2Out o{d.slabs, p->grads + wout + (int64_t)po * 256 * q_width + (int64_t)qo * 256, q_width, 0};
      wd.push_back(d);
      wo.push_back(o);
    };
    const float pd = c.dropout;
    for (int i = 0; i < c.n_mega_blocks; ++i) {
      const MegaBlockRef& mb = m->blocks[i];
      const BlockWs& bw = p->blk[i];
      const void* xin = i > 0 ? (const void*)(p->ws + p->blk[i - 1].OUT) : (const void*)(p->ws + p->Y0);
      BnAct actx = i > 0 ? identity_act() : make_act(p, m->prolog_bn, M, 1, 1, 0.f, 0, 0);
      for (int po = 0; po < hs; ++po)
        for (int qo = 0; qo < hs; ++qo) add(bw.dZk, bw.S, mb.bnskip, H, xin, actx, H, 0, -1, -1, mb.wskip, po, qo);
      for (int j = 0; j < nsub; ++j) {
        const void* sin = j > 0 ? (const void*)(p->ws + bw.Y[j - 1]) : xin;
        BnAct asin = j > 0 ? make_act(p, mb.sub[j - 1].bn, M, 1, 1, pd, 0, 0) : actx;
        for (int po = 0; po < hs; ++po)
          for (int qo = 0; qo < hs; ++qo) {
            // the forward kept the depthwise output: a plain operand, no activation / stencil recompute
            add(bw.dY[j], bw.Y[j], mb.sub[j].bn, H, (const void*)(p->ws + bw.Q[j]), identity_act(), H, 0, -1, -1, mb.sub[j].wpw, po, qo);
          }
      }
    }
    // epilog conv: d W[D][H] in 256 x 256 slabs = BN-backward(dEbn, E)[:, out slab]^T * x_last[:, in slab]
    // (x_last = last block output, stored activated)
    if (p->wg2_epi_slabs > 0)
      for (int po = 0; po < c.enc_out / 256; ++po)
        for (int qo = 0; qo < hs; ++qo)
          add(p->dEbn, p->E, m->epi_bn, c.enc_out, (const void*)(p->ws + p->blk[c.n_mega_blocks - 1].OUT), identity_act(), H, 0, -1, -1,
              m->epi_w, po, qo);
    if (p->wg2_asp_units > 0) {
      const int D = c.enc_out, A = c.attn_hidden;
      auto add_plain = [&](const bf16_t* P, int ldp, const bf16_t* Q, int ldq, int q0, const BnAct& aq, float* out, int ld, int lim) {
        WgradV2Desc d;
        memset(&d, 0, sizeof(d));
        d.dZ = P; d.ldp = ldp; d.statC = 256;
        d.inv_n = 1.f / (float)M; d.eps = 1e-5f; d.batch = 1.f;
        d.X = Q; d.actX = aq; d.ldq = ldq; d.q0 = q0;
        d.slabs = (float*)(p->ws + p->wg2_slabs) + wd.size() * slab_stride;
        wd.push_back(d);
        wo.push_back(WgradV2Out{d.slabs, out, ld, lim});
      };
      const BnAct acte = make_act(p, m->epi_bn, M, 1, 1, 0.f, 0, 0);
      // d W_out[c][a] = sum_r dEN[r][c] hid[r][a]: P = 256-channel slab of dEN, Q = hid (128 wide, read as 256)
      for (int s = 0; s < D / 256; ++s)
        add_plain((const bf16_t*)(p->ws + p->dE) + s * 256, D, (const bf16_t*)(p->ws + p->HID), A, 0, identity_act(),
                  p->grads + m->asp_wout + (int64_t)s * 256 * A, A, 256 | (A << 16));
      // d W_in[a][c] = sum_r dHP[r][a] x[r][c], x = relu(BN(E)): P = dHP (128 wide, read as 256), Q = slab of E
      for (int s = 0; s < D / 256; ++s)
        add_plain((const bf16_t*)(p->ws + p->dHP), A, (const bf16_t*)(p->ws + p->E), D, s * 256, acte,
                  p->grads + m->asp_win + (int64_t)s * 256, D, A | (256 << 16));
    }
    if ((int)wd.size() != p->wg2_layers) return TN_E_STATE;
    for (const auto& d : wd)
      if (d.actX.drop_thr || d.wdw) return TN_E_STATE;      // the launch has no dropout-hashing / depthwise-recompute variant
    if (sizeof(WgradV2Desc) > 256 || sizeof(WgradV2Out) > 32) return TN_E_STATE;
    TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->wg2_desc, wd.data(), wd.size() * sizeof(WgradV2Desc), hipMemcpyHostToDevice, st));
    // second table: the fused-tail flow of backward_impl — the last sub-block's gradient buffer then holds the
    // BatchNorm-backward'd dS itself (stored by dgrad_dw_v6<.., Z3>): a plain P operand
    std::vector<WgradV2Desc> wd2;      // (source of an asynchronous copy: lives until the synchronise that ends this block)
    if (hs == 1 && nsub >= 2) {
      wd2 = wd;
      for (int i = 0; i < c.n_mega_blocks; ++i) {
        WgradV2Desc& d = wd2[(size_t)i * (nsub + 1) + 1 + (nsub - 1)];
        d.Y = nullptr; d.fstats = nullptr; d.bsums = nullptr;
      }
      TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->wg2_desc + (size_t)p->wg2_layers * sizeof(WgradV2Desc), wd2.data(),
                                  wd2.size() * sizeof(WgradV2Desc), hipMemcpyHostToDevice, st));
    }
    TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->wg2_out, wo.data(), wo.size() * sizeof(WgradV2Out), hipMemcpyHostToDevice, st));
    if (p->v2_tn) {
      // third table: what stays in wgrad_batched_v2 when the mega blocks' pointwise layers go to the pipelined TN launch — the
      // first block's skip conv (its operand is the prolog output through its activation), then the epilog / pooling units
      std::vector<WgradV2Desc> wd3;
      std::vector<WgradV2Out> wo3;
      wd3.push_back(wd[0]); wo3.push_back(wo[0]);
      for (size_t u = (size_t)c.n_mega_blocks * (nsub + 1); u < wd.size(); ++u) { wd3.push_back(wd[u]); wo3.push_back(wo[u]); }
      // their own slab ranges (plan_layout_tail): unit 0 may be cut over the whole grid when it is alone in its bucket's launch
      for (size_t u = 0; u < wd3.size(); ++u) {
        const size_t first = u == 0 ? 0 : (size_t)p->wg2_parts3_u0 + (u - 1) * (size_t)p->wg2_parts3_tail;
        wd3[u].slabs = (float*)(p->ws + p->wg2_slabs) + first * 256 * 256;
        wo3[u].slabs = wd3[u].slabs;
      }
      TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->wg2_desc + (size_t)2 * p->wg2_layers * sizeof(WgradV2Desc), wd3.data(),
                                  wd3.size() * sizeof(WgradV2Desc), hipMemcpyHostToDevice, st));
      TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->wg2_out3, wo3.data(), wo3.size() * sizeof(WgradV2Out), hipMemcpyHostToDevice, st));
      TN_CHECK_HIP(hipStreamSynchronize(st));      // (wd3 / wo3 die with this scope)
    }
    std::vector<DwGradOut> dg;
    for (int i = 0; i < c.n_mega_blocks; ++i)
      for (int j = 0; j < nsub; ++j) {
        DwGradOut o;
        o.gacc = (const float*)(p->ws + p->dw_gacc) + (size_t)(i * nsub + j) * TN_NREP * (c.kernel + 1) * c.hidden;
        o.g_wdw = p->grads + m->blocks[i].sub[j].wdw;
        o.g_bdw = p->grads + m->blocks[i].sub[j].bdw;
        dg.push_back(o);
      }
    if (sizeof(DwGradOut) > 32) return TN_E_STATE;
    TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->dw_table, dg.data(), dg.size() * sizeof(DwGradOut), hipMemcpyHostToDevice, st));
    TN_CHECK_HIP(hipStreamSynchronize(st));      // wd, wd2, wo, dg are pageable sources of the copies above
  }
  if (p->dw_part && !p->use_v2) {
    // wide models: where dw_part_reduce_kernel finds a layer's partial records and puts its tap / bias gradients
    const tn_config& c = m->cfg;
    std::vector<DwGradOut> dg;
    for (int i = 0; i < c.n_mega_blocks; ++i)
      for (int j = 0; j < c.n_sub_blocks; ++j) {
        DwGradOut o;
        o.gacc = (const float*)(p->ws + p->dw_part + (size_t)(i * c.n_sub_blocks + j) * p->dw_part_stride);
        o.g_wdw = p->grads + m->blocks[i].sub[j].wdw;
        o.g_bdw = p->grads + m->blocks[i].sub[j].bdw;
        dg.push_back(o);
      }
    if (sizeof(DwGradOut) > 32) return TN_E_STATE;
    TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->dw_table, dg.data(), dg.size() * sizeof(DwGradOut), hipMemcpyHostToDevice, st));
    TN_CHECK_HIP(hipStreamSynchronize(st));
  }
  TN_CHECK_HIP(hipStreamSynchronize(st));
  return 0;
}

// MetricLearningLoss.forward's autograd (reference src/losses.py:32-44, :77-132): backward of tn_head_forward from its `save`
extern "C" int tn_head_backward(int32_t loss_type, int32_t batch, int32_t emb, int32_t n_classes, const float* fc_weight,
                                const float* save, float grad_scale, const float* grad_loss_dev, const float* grad_normalized,
                                float* grad_inputs, float* grad_weight, float* grad_bias, void* stream) {
  if (!fc_weight || !save || !grad_inputs || !grad_weight) return TN_E_BADARG;
  if (batch <= 0 || emb <= 0 || n_classes <= 0) return TN_E_BADARG;
  if (loss_type != TN_LOSS_CE && loss_type != TN_LOSS_MARGIN) return TN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  HeadBwdArgs ha;
  memset(&ha, 0, sizeof(ha));
  ha.B = batch; ha.E = emb; ha.NC = n_classes; ha.loss_type = loss_type;
  ha.dlogits = save;
  ha.dscale = save + (size_t)batch * n_classes;
  ha.emb = ha.dscale + batch;
  ha.emb_norm = ha.emb + (size_t)batch * emb;
  ha.W = fc_weight; ha.gs = grad_scale; ha.gs_dev = grad_loss_dev; ha.g_embnorm = grad_normalized;
  ha.g_W = grad_weight; ha.g_bias = grad_bias; ha.demb = grad_inputs;
  hipLaunchKernelGGL(head_bwd_w_kernel, dim3((n_classes * emb + 255) / 256), dim3(256), 0, st, ha);
  hipLaunchKernelGGL(head_bwd_x_kernel, dim3(batch), dim3(256), (size_t)emb * sizeof(float), st, ha);
  return (int)hipGetLastError();
}

int plan_backward(tn_plan* p, float grad_scale, const float* grad_scale_dev, const float* grad_emb, float* grad_input,
                  hipStream_t st) {
  if (p->prec == TN_PREC_BF16) return backward_impl<bf16_t>(p, grad_scale, grad_scale_dev, grad_emb, grad_input, st);
  return backward_impl<float>(p, grad_scale, grad_scale_dev, grad_emb, grad_input, st);
}
