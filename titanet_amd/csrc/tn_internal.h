// titanet_amd — host-side model layout and execution plan (internal).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/titanet_amd.h"
#include "tn_common.h"

struct TensorInfo {
  std::string name;
  int kind;
  int64_t offset, numel;
  int ndim;
  int64_t shape[4];
};

struct BnRef {
  int id = -1;      // index into the BN tables (statistics, nbt)
  int C = 0;
  int64_t gamma = 0, beta = 0;    // offsets into params
  int64_t rmean = 0, rvar = 0;    // offsets into bnbuf
};

struct SubBlockRef {
  int64_t wdw, bdw, wpw, bpw;
  BnRef bn;
};

struct MegaBlockRef {
  std::vector<SubBlockRef> sub;
  int64_t se_w1, se_w2;
  int64_t wskip, bskip;
  BnRef bnskip;
};

struct tn_model {
  tn_config cfg;
  std::vector<TensorInfo> tensors;
  int64_t n_params = 0, n_buffers = 0;
  int n_bn = 0;
  int64_t prolog_w, prolog_b;
  BnRef prolog_bn;
  std::vector<MegaBlockRef> blocks;
  int64_t epi_w, epi_b;
  BnRef epi_bn;
  int64_t asp_win = -1, asp_bin = -1, asp_wout = -1, asp_bout = -1;
  int64_t pool2_w = -1, pool2_b = -1;   // Decoder(simple_pool=True): Linear(D, 2D) after the mean over time
  BnRef pool_bn;
  int64_t lin_w, lin_b;
  BnRef lin_bn;
  int64_t fc_w = -1, fc_b = -1;
  std::vector<BnRef> all_bn;   // by id
};

// compute-precision weight copies (offsets in BYTES into the workspace)
struct WcRef {
  size_t w = 0;    // [N][K]
  size_t wt = 0;   // [K][N]
  size_t sw = 0;   // 256 x 256 bf16 only: w in MFMA-fragment order [8 row blocks][16 k-steps][64 lanes] x 16 bytes, or 0
  size_t swt = 0;  // ... and wt in that order
};

struct BlockWs {
  std::vector<size_t> Y;   // raw sub-block outputs
  std::vector<size_t> Q;   // saved depthwise outputs (operand of the pointwise GEMM) for the batched weight gradients, or empty
  size_t S, OUT;
  size_t m, h, g;          // SE: mean [B][C], hidden [B][Hr], gate [B][C]  (float)
  std::vector<WcRef> wpw;
  size_t dS_skip = 0;            // ... and of the skip connection's conv as stored by dgrad_v2 (blocks > 0)
  std::vector<size_t> dS;        // headline-shape plans: the BatchNorm-backward'd gradient of sub-block j as stored by dgrad_dw_v6 (0: none)
  std::vector<size_t> w8, w8s;   // TN_PREC_FP8: e4m3 pointwise weights [H][H] and their per-row scales [H] (float)
  std::vector<size_t> w8t, w8ts; // fp8 data gradient: e4m3 rows of W^T [ci][co] and their per-input-channel scales [H]
  size_t w8t_skip = 0, w8ts_skip = 0;   // ... of the skip connection's 1x1 conv
  // fp8 weight gradient (round 5, tn_plan::fp8_wgrad): per sub-block layer the kept e4m3 depthwise output, the e4m3 copy of dS
  // scaled per COLUMN, the E8M0 byte per column the contraction undoes, and the column maxima of |dS| (previous step: persistent;
  // this step: inside the region zeroed at the start of every backward)
  std::vector<size_t> Q8, dS8c, cexp, amax_prev, amax_cur;
  WcRef wskip;
  // backward (float): SE pre-activation grads
  size_t dpre2, dpre1, dgate;
  // per-layer activation gradients kept for the batched weight-gradient launch (v2)
  std::vector<size_t> dY;   // d loss / d BN-output of sub-block j
  size_t dZk;               // d loss / d BN-output of the skip connection
};

struct tn_plan {
  const tn_model* model;
  int B, T, M, prec;
  int use_v2 = 0;           // specialised hidden=256 bf16 kernels (tn_v2_kernels.h)
  bool generic = false;     // TN_GENERIC=1: generic kernel templates everywhere (debug / cross-check switch)
  size_t esz;               // activation element size
  size_t ws_bytes = 0;
  size_t bound_bytes = 0;       // size of the workspace handed to tn_plan_bind
  size_t ws_fixed_bytes = 0;    // end of the part of the layout that does not depend on grad_groups
  // bound buffers
  float* params = nullptr;
  float* grads = nullptr;
  float* bnbuf = nullptr;
  int64_t* nbt = nullptr;
  char* ws = nullptr;
  bool bound = false;
  // workspace layout (byte offsets)
  size_t zero_begin, zero_bytes;        // region cleared at the start of every forward
  bool fp8 = false;                     // TN_PREC_FP8: forward pointwise GEMMs of the sub-blocks on the fp8 matrix cores
  bool fp8_bwd_emu = false;             // experiment (TN_FP8_BWD_EMU=1 at plan creation, fp8 plans): every BatchNorm-backward'd
                                        // gradient dS of the pipelined path is rounded through e4m3 (one power-of-two scale per
                                        // row) before the data- and weight-gradient GEMMs read it: the accuracy an fp8 backward
                                        // would have, measured before its kernels exist
  bool v2_tn = false;                   // headline-shape plans: sub-block pointwise weight gradients as ONE pipelined TN contraction
  size_t wg2_out3 = 0;                  // ... and the compact unit tables of what stays in wgrad_batched_v2 (skip convs, epilog, ASP)
  size_t a0 = 0;                        // wide bf16 plans: the activated prolog output, stored (act_store_kernel); 0 = not kept
  bool fp8_bwd = false;                 // fp8 plans at hidden 512 / 1024: sub-block data gradients on the f8f6f4 MFMA (TN_FP8_BWD=0: bf16)
  size_t ds8s = 0, dsexps = 0;          // ... and of the skip connection's layer (its dS is made while the last sub-block's is still pending)
  size_t ds8 = 0, dsexp = 0;            // ... their A operand: e4m3 dS [M][H] bytes + row exponent bytes (one layer at a time)
  size_t q8 = 0, fp8_table = 0;         // e4m3 copy of the current depthwise output [M][H] bytes; weight-cast descriptors
  bool fp8_wgrad = false;               // fp8 plans on the pipelined path: the sub-block pointwise weight gradients on the f8f6f4 MFMA
                                        // (pgemm_tn_f8_batched_kernel); delayed per-column scales: the FIRST backward of a plan has no
                                        // history and runs the bf16 contraction (fp8_hist_valid)
  bool fp8_hist_valid = false;
  bool fwd_q16_skipped = false;          // the last forward left the bf16 depthwise outputs of the sub-blocks unwritten (fp8 weight gradient with history)
  size_t tn_f8_table = 0, tn_skip_table = 0;   // PGemmTnF8Desc of the sub-block layers / PGemmTnDesc of the skip convs alone (backward order)
  int n_fp8 = 0;
  bool split_dw = false;                // wide models: depthwise producer as its own streaming kernel (forward)
  bool save_q = false;                  // forward stores the depthwise outputs (bf16 v2 path with batched weight gradients)
  size_t bzero_begin, bzero_bytes;      // region cleared at the start of every backward
  size_t step_state = 0;                // {uint64 step; uint32 word}: device-resident step counter / dropout word (hipGraph replay)
  std::vector<size_t> stats;            // per BN id: forward sums  float[NREP][2][C]
  std::vector<size_t> bsums;            // per BN id: backward sums float[NREP][2][C]
  size_t loss_acc;
  size_t Y0;
  std::vector<BlockWs> blk;
  size_t E, HID, EN, pooled, smax, sinv, qv, lin, emb, emb_norm, dlogits, dscale, logits, preds;
  WcRef wprolog, wepi, wwin, wwout;
  bool wide_dw_bwd = true;         // slab depthwise kernels of the wide models (off under TN_GENERIC)
  bool prolog_taps = false;        // bf16 plans: prolog GEMMs read a packed rows x n_mels copy of the input (ProdTaps)
  size_t x0 = 0, wprolog_taps = 0, prolog_gtmp = 0;
  size_t cast_table, bn_table, stats_ptr_table;
  int n_cast = 0;
  // backward scratch
  size_t dA[2];        // ping-pong grad wrt block outputs (rows x hidden, AT)
  size_t dYbn;         // rows x hidden AT
  size_t dD;           // rows x hidden AT
  size_t dZ;           // rows x hidden AT
  size_t dXs;          // rows x hidden AT (skip dgrad)
  size_t dE;           // rows x enc_out AT  (d energies)
  size_t dEbn;         // rows x enc_out AT
  size_t dHP;          // rows x attn AT
  size_t dpooled, dlin, demb;   // float
  size_t swz_table = 0; int n_swz = 0;   // (src, dst) pairs of the fragment-order copies, refreshed after every parameter cast
  size_t mu, dmu;               // float [B][D]: mean over time of the encoder output and its gradient (simple_pool)
  size_t slabs;        // split-K partial weight gradients
  size_t slab_bytes = 0;
  size_t bwd_table, bwd_table_eval, se_table;
  size_t wg2_desc, wg2_out, wg2_count, wg2_slabs;   // batched weight-gradient launch (v2); wg2_desc holds TWO tables of
                                                    // wg2_layers descriptors: [0] every unit from (dZ, Y), [1] the fused-tail
                                                    // flow (the last sub-block's unit reads the stored BatchNorm-backward'd dS)
  size_t tn_table = 0;          // wide models: PGemmTnDesc table of the mega blocks' pointwise layers in backward order (batched
                                // weight-gradient launch, one per gradient bucket), or 0
  size_t se_bacc = 0;           // ... its per-utterance partial sums [B][4][hidden] when several workgroups share an utterance
                                // (tail_parts > 1; inside the region zeroed at the start of every backward)
  size_t se_gu = 0;             // fused mega-block tail backward (combine_bwd1_v3): ga / ub [B][2][256] floats, one block at a time
  size_t dw_gacc, dw_table;                         // depthwise gradient accumulators [layer][NREP][KD+1][256] + finalize table
  int wg2_layers = 0, wg2_maxparts = 0, wg2_units_per_wg = 0, wg2_grid = 0, wg2_epi_slabs = 0, wg2_asp_units = 0;
  int wg2_upl = 1;              // units per pointwise layer: (hidden / 256)^2 output slabs of 256 x 256
  int wg2_parts3_u0 = 0, wg2_parts3_tail = 0;   // v2_tn plans: partial slabs of the compact table's units (block 0's skip conv / each tail unit)
  bool wide_wgrad = false;      // hidden = 512 / 1024 (TitaNet-M / -L), bf16: slab depthwise kernels + the pipelined GEMMs of tn_pgemm.h
  size_t bwd_table_bytes = 0;
  // gradient buckets in COMPLETION order (data-parallel overlap: bucket i's all-reduce starts when its event fires)
  struct GradBucket {
    int64_t begin = 0, end = 0;   // float range in the flat gradient buffer
    int blk_lo = 0, blk_hi = -1;  // mega blocks finalised with this bucket (inclusive), or empty
    bool tail = false;            // epilog conv, pooling, decoder tail, loss head
    bool prolog = false;          // prolog conv + BN
  };
  int prolog_cur = 0;             // which dA buffer holds the gradient wrt the prolog output (set by backward)
  int grad_groups = 1;            // 1: one bucket, every deferred weight gradient in one launch at the end of backward
  std::vector<GradBucket> buckets;
  std::vector<hipEvent_t> bucket_events;
  // independent launches of a mega block on a side stream (round 5): the skip conv beside the sub-block chain (forward), the
  // skip data gradient beside the last two sub-blocks' fused data-gradient launches (backward) — one kernel's ramp-up fills
  // the other's drain.  Fork / join with plan-owned events; works under stream capture (the side stream joins the capture).
  bool overlap = false;
  bool nt_skip = false;             // wide models: non-temporal stores for the skip conv's output and the skip data gradient
  bool skip_late = false;           // forward: the skip conv of a mega block is launched behind its sub-blocks (tn_api.hip)
  // attentive pooling without a stored energy tensor (asp_v2_kernel, tn_v2_wide_kernels.h): forward and backward recompute
  // the K = 128 product.  Decided at plan creation (TN_ASP_FUSED=0 keeps the stored-energies kernels; shapes outside the
  // kernel — frames > 320, several workgroups per utterance — always do): forward and backward must agree.
  bool asp_fused = false;
  // SE squeeze + residual combine of a mega block in one launch that keeps the utterance's Y3 in registers
  // (se_combine_fwd_v3_kernel, tn_v2_kernels.h): hidden 256, bf16, frames <= 320, one workgroup per utterance.  TN_SE_FUSED=0
  // (read at plan creation) keeps se_squeeze_v2 + combine_fwd_v2.
  bool se_fused = false;
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> ov_events;   // [block][4]: forward fork / join, backward fork / join
  // per-kernel event timing (tn_profile_*)
  int prof_class = 0;
  int prof_stride = 1;            // tn_profile_sample: bracket every n-th launch of the class
  unsigned prof_counter = 0;
  std::vector<hipEvent_t> prof_events;
  size_t prof_used = 0;
  // variable-length batches (tn_forward_masked): valid frames per utterance
  size_t lens = 0;                      // int32 [B] in the workspace
  // small batches of long utterances: the per-utterance kernels of the mega-block tail (SE squeeze, combine backward) run as
  // tail_parts workgroups per utterance; their partial sums meet in se_acc / dgate_acc ([block][B][hidden] floats, cleared
  // with the backward zero region; se_acc: [B][parts][hidden] partial sums, reused block after block)
  int tail_parts = 1;
  size_t dw_part = 0;           // wide models: partial tap-gradient records of dw_bwd_slab_kernel [layer][256 workgroups][K + 1][256] (dw_part_reduce_kernel)
  size_t dw_part_stride = 0;    // bytes per layer
  int rw_nt = 0;                // tuning (TN_RW_NT): 1 = non-temporal stores of the wide models' forward pointwise outputs, 2 = of their data gradients
  int se_parts = 1;             // workgroups per utterance of the one-launch SE squeeze (se_squeeze_fc_kernel mode 3), counters in se_cnt
  size_t se_cnt = 0;
  size_t se_acc = 0, dgate_acc = 0;
  // variable-length batches on the pipelined GEMMs: the 256-row tiles holding at least one valid frame (int32 list in the
  // workspace, rewritten with every set of lengths), their count and the rows they cover; skip_pad_tiles: the plan's kernels
  // all tolerate stale values in rows of skipped tiles (wide bf16 plans on the slab depthwise kernels)
  size_t rowtiles = 0;
  int n_rowtiles = 0, active_rows = 0;
  bool skip_pad_tiles = false;
  std::vector<int> lens_host;
  bool masked = false;                  // the last forward carried lengths
  int n_valid = 0;                      // sum of the lengths (rows that enter the [B*T]-row BatchNorm statistics)
  // state
  const float* last_input = nullptr;
  int last_training = -1;
  int last_has_loss = 0;
  uint64_t last_seed = 0;
};

int plan_forward(tn_plan* p, const float* spec, const int64_t* speakers, int training, uint64_t seed, float* emb_out,
                 int64_t* preds, float* loss, hipStream_t st);
RowMask plan_row_mask(const tn_plan* p);   // {device lengths, T} of the last forward, or {null, T}
int plan_backward(tn_plan* p, float grad_scale, const float* grad_scale_dev, const float* grad_emb, float* grad_input,
                  hipStream_t st);

int plan_upload_bwd_tables(tn_plan* p, hipStream_t st);
void plan_layout_tail(tn_plan* p);   // workspace regions whose size depends on grad_groups (end of the layout)

// bracket a launch with events when its class is being profiled
struct ProfScope {
  tn_plan* p;
  hipStream_t st;
  bool on;
  ProfScope(tn_plan* plan, int cls, hipStream_t s) : p(plan), st(s), on(plan && cls != 0 && plan->prof_class == cls) {
    if (on) on = (p->prof_counter++ % p->prof_stride) == 0;      // sampled: every prof_stride-th launch of the class
    if (on) {
      if (p->prof_used + 2 > p->prof_events.size()) {
        for (int i = 0; i < 256; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { on = false; return; } p->prof_events.push_back(e); }
      }
      (void)hipEventRecord(p->prof_events[p->prof_used], st);
    }
  }
  ~ProfScope() {
    if (on) { (void)hipEventRecord(p->prof_events[p->prof_used + 1], st); p->prof_used += 2; }
  }
};

// helpers shared by forward / backward orchestration
BnAct make_act(const tn_plan* p, const BnRef& bn, int rows, int training, int relu, float drop_p, uint64_t seed,
               int layer);
BnAct identity_act();
