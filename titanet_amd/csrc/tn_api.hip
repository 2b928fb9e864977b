// titanet_amd — C ABI, model layout, execution plan and the forward orchestration.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "tn_fwd_kernels.h"
#include "tn_gemm.h"
#include "tn_internal.h"
#include "tn_pgemm.h"
#include "tn_v2_kernels.h"
#include "tn_v2_wide_kernels.h"

// =============================================================================================
// model layout == reference state_dict (reference src/models.py:370-384, :432-455, :504-513,
// src/modules.py:65-78, :119-133, :166-171, src/losses.py:31, :68)
// =============================================================================================
namespace {

struct LayoutBuilder {
  tn_model* m;
  int64_t add(const std::string& name, int kind, std::initializer_list<int64_t> shape) {
    TensorInfo t;
    t.name = name;
    t.kind = kind;
    t.ndim = (int)shape.size();
    t.numel = 1;
    int i = 0;
    for (auto s : shape) { t.shape[i++] = s; t.numel *= s; }
    for (; i < 4; ++i) t.shape[i] = 1;
    int64_t* counter = (kind == TN_KIND_PARAM) ? &m->n_params : &m->n_buffers;
    if (kind == TN_KIND_NBT) {
      t.offset = m->n_bn;   // caller bumps n_bn
    } else {
      *counter = (*counter + 15) & ~(int64_t)15;   // 64-byte aligned tensors (float4 / 16-byte loads)
      t.offset = *counter;
      *counter += t.numel;
    }
    m->tensors.push_back(t);
    return t.offset;
  }
  BnRef bn(const std::string& prefix, int C) {
    BnRef r;
    r.C = C;
    r.gamma = add(prefix + ".weight", TN_KIND_PARAM, {C});
    r.beta = add(prefix + ".bias", TN_KIND_PARAM, {C});
    r.rmean = add(prefix + ".running_mean", TN_KIND_BUFFER, {C});
    r.rvar = add(prefix + ".running_var", TN_KIND_BUFFER, {C});
    add(prefix + ".num_batches_tracked", TN_KIND_NBT, {});
    r.id = m->n_bn++;
    m->all_bn.push_back(r);
    return r;
  }
};

int validate(const tn_config& c) {
  if (c.n_mels <= 0 || c.n_mega_blocks < 0 || c.n_sub_blocks < 1) return TN_E_BADARG;
  if (c.hidden % 8 || c.enc_out % 8 || c.emb % 8 || c.attn_hidden % 8 || c.hidden <= 0) return TN_E_UNSUPPORTED;
  if ((c.n_mels * c.prolog_kernel) % 8) return TN_E_UNSUPPORTED;
  if (c.kernel % 2 == 0 || c.prolog_kernel % 2 == 0 || c.epilog_kernel != 1) return TN_E_UNSUPPORTED;
  if (c.se_reduction <= 0 || c.hidden / c.se_reduction < 1) return TN_E_BADARG;
  if (c.hidden > 4096 || c.emb > 4096) return TN_E_UNSUPPORTED;
  if (c.loss_type != TN_LOSS_NONE && c.n_classes <= 0) return TN_E_BADARG;
  if (c.dropout < 0.f || c.dropout >= 1.f) return TN_E_BADARG;
  return 0;
}

}  // namespace

extern "C" int tn_model_create(const tn_config* cfg, tn_model** out) {
  if (!cfg || !out) return TN_E_BADARG;
  int rc = validate(*cfg);
  if (rc) return rc;
  tn_model* m = new tn_model();
  m->cfg = *cfg;
  LayoutBuilder L{m};
  const int H = cfg->hidden, D = cfg->enc_out, Hr = cfg->hidden / cfg->se_reduction;
  m->prolog_w = L.add("encoder.prolog.conv_block.0.weight", TN_KIND_PARAM, {H, cfg->n_mels, cfg->prolog_kernel});
  m->prolog_b = L.add("encoder.prolog.conv_block.0.bias", TN_KIND_PARAM, {H});
  m->prolog_bn = L.bn("encoder.prolog.conv_block.1", H);
  for (int i = 0; i < cfg->n_mega_blocks; ++i) {
    MegaBlockRef mb;
    const std::string p = "encoder.mega_blocks." + std::to_string(i);
    for (int j = 0; j < cfg->n_sub_blocks; ++j) {
      const std::string q = p + ".sub_blocks." + std::to_string(j) + ".conv_block";
      SubBlockRef s;
      s.wdw = L.add(q + ".0.conv.0.weight", TN_KIND_PARAM, {H, 1, cfg->kernel});
      s.bdw = L.add(q + ".0.conv.0.bias", TN_KIND_PARAM, {H});
      s.wpw = L.add(q + ".0.conv.1.weight", TN_KIND_PARAM, {H, H, 1});
      s.bpw = L.add(q + ".0.conv.1.bias", TN_KIND_PARAM, {H});
      s.bn = L.bn(q + ".1", H);
      mb.sub.push_back(s);
    }
    const std::string q = p + ".sub_blocks." + std::to_string(cfg->n_sub_blocks) + ".excitation";
    mb.se_w1 = L.add(q + ".0.weight", TN_KIND_PARAM, {Hr, H});
    mb.se_w2 = L.add(q + ".2.weight", TN_KIND_PARAM, {H, Hr});
    mb.wskip = L.add(p + ".skip_connection.0.weight", TN_KIND_PARAM, {H, H, 1});
    mb.bskip = L.add(p + ".skip_connection.0.bias", TN_KIND_PARAM, {H});
    mb.bnskip = L.bn(p + ".skip_connection.1", H);
    m->blocks.push_back(mb);
  }
  m->epi_w = L.add("encoder.epilog.conv_block.0.weight", TN_KIND_PARAM, {D, H, 1});
  m->epi_b = L.add("encoder.epilog.conv_block.0.bias", TN_KIND_PARAM, {D});
  m->epi_bn = L.bn("encoder.epilog.conv_block.1", D);
  if (cfg->simple_pool) {
    // reference src/models.py:497-502: nn.Sequential(AdaptiveAvgPool1d(1), Squeeze(-1), Linear(D, 2D)) -> keys decoder.pool.2.*
    m->pool2_w = L.add("decoder.pool.2.weight", TN_KIND_PARAM, {2 * D, D});
    m->pool2_b = L.add("decoder.pool.2.bias", TN_KIND_PARAM, {2 * D});
  } else {
    m->asp_win = L.add("decoder.pool.0.in_linear.weight", TN_KIND_PARAM, {cfg->attn_hidden, D});
    m->asp_bin = L.add("decoder.pool.0.in_linear.bias", TN_KIND_PARAM, {cfg->attn_hidden});
    m->asp_wout = L.add("decoder.pool.0.out_linear.weight", TN_KIND_PARAM, {D, cfg->attn_hidden});
    m->asp_bout = L.add("decoder.pool.0.out_linear.bias", TN_KIND_PARAM, {D});
    m->pool_bn = L.bn("decoder.pool.1", 2 * D);
  }
  m->lin_w = L.add("decoder.linear.0.weight", TN_KIND_PARAM, {cfg->emb, 2 * D});
  m->lin_b = L.add("decoder.linear.0.bias", TN_KIND_PARAM, {cfg->emb});
  m->lin_bn = L.bn("decoder.linear.1", cfg->emb);
  if (cfg->loss_type != TN_LOSS_NONE) {
    m->fc_w = L.add("loss_function.fc.weight", TN_KIND_PARAM, {cfg->n_classes, cfg->emb});
    if (cfg->loss_type == TN_LOSS_CE) m->fc_b = L.add("loss_function.fc.bias", TN_KIND_PARAM, {cfg->n_classes});
  }
  m->n_params = (m->n_params + 15) & ~(int64_t)15;
  m->n_buffers = (m->n_buffers + 15) & ~(int64_t)15;
  *out = m;
  return 0;
}

extern "C" void tn_model_destroy(tn_model* m) { delete m; }
extern "C" int64_t tn_model_param_floats(const tn_model* m) { return m ? m->n_params : 0; }
extern "C" int64_t tn_model_buffer_floats(const tn_model* m) { return m ? m->n_buffers : 0; }
extern "C" int32_t tn_model_num_bn(const tn_model* m) { return m ? m->n_bn : 0; }
extern "C" int32_t tn_model_num_tensors(const tn_model* m) { return m ? (int32_t)m->tensors.size() : 0; }
extern "C" int tn_model_tensor_info(const tn_model* m, int32_t index, char* name, int32_t* kind, int64_t* offset,
                                    int64_t* numel, int32_t* ndim, int64_t* shape) {
  if (!m || index < 0 || index >= (int32_t)m->tensors.size()) return TN_E_BADARG;
  const TensorInfo& t = m->tensors[index];
  if (name) { strncpy(name, t.name.c_str(), 159); name[159] = 0; }
  if (kind) *kind = t.kind;
  if (offset) *offset = t.offset;
  if (numel) *numel = t.numel;
  if (ndim) *ndim = t.ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = t.shape[i];
  return 0;
}
extern "C" const char* tn_version(void) { return "titanet_amd 0.1 (gfx950)"; }

// =============================================================================================
// plan
// =============================================================================================
namespace {
struct Bump {
  size_t off = 0;
  size_t take(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    size_t r = off;
    off += bytes;
    return r;
  }
};
}  // namespace

extern "C" int tn_plan_create(const tn_model* m, int32_t batch, int32_t frames, int32_t precision, tn_plan** out) {
  if (!m || !out || batch <= 0 || frames <= 0) return TN_E_BADARG;
  if (precision != TN_PREC_FP32 && precision != TN_PREC_BF16 && precision != TN_PREC_FP8 && precision != TN_PREC_FP8_FWD) return TN_E_BADARG;
  const bool fp8_fwd_only = precision == TN_PREC_FP8_FWD;
  if (fp8_fwd_only) precision = TN_PREC_FP8;
  if (precision == TN_PREC_FP8 && m->cfg.hidden % 16) return TN_E_UNSUPPORTED;
  if ((int64_t)batch * frames * std::max(m->cfg.enc_out, m->cfg.hidden) >= ((int64_t)1 << 32)) return TN_E_UNSUPPORTED;
  tn_plan* p = new tn_plan();
  p->model = m;
  p->B = batch; p->T = frames; p->M = batch * frames; p->prec = precision;
  p->fp8 = precision == TN_PREC_FP8;
  { const char* e = getenv("TN_FP8_BWD_EMU"); p->fp8_bwd_emu = p->fp8 && e && atoi(e) != 0; }
  {
    // fp8 data gradients of the sub-block pointwise convs (wide models on the pipelined GEMMs; a row of dS is one or two waves
    // of the BatchNorm-backward pass that quantises it)
    // (TN_PREC_FP8_FWD: forward GEMMs only; TN_FP8_BWD=0 in the environment does the same for A/B scripts.  Only plans on the
    //  pipelined GEMMs — the wide_wgrad condition below — have the kernels; elsewhere the flag and its buffers stay off)
    const char* e = getenv("TN_FP8_BWD");
    const char* eg = getenv("TN_GENERIC");
    const int Hh = m->cfg.hidden;
    p->fp8_bwd = p->fp8 && !fp8_fwd_only && !p->fp8_bwd_emu && (Hh == 512 || Hh == 1024) && ((size_t)batch * frames * (Hh / 8)) % 256 == 0 &&
                 m->cfg.enc_out % 256 == 0 && m->cfg.n_mega_blocks > 0 && !(eg && atoi(eg) != 0) && !(e && atoi(e) == 0);
  }
  if (p->fp8) precision = TN_PREC_BF16;      // storage, statistics and the backward pass are the bf16 plan's
  p->prec = precision;
  p->esz = precision == TN_PREC_BF16 ? 2 : 4;
  {
    // TN_GENERIC=1 (the one debug switch of the library, read at plan creation): every layer runs the generic kernel
    // templates — what the fp32 parity plans, masked batches and unusual shapes use anyway.  Tests cross-check the
    // specialised kernels against it (tests/test_v2_shapes_gpu.py, tests/test_model_sizes_gpu.py).
    const char* e = getenv("TN_GENERIC");
    const bool generic = e && atoi(e) != 0;
    p->generic = generic;
    // specialised kernels of the headline shape (hidden 256, 3 taps, bf16): tn_v2_kernels.h
    p->use_v2 = (precision == TN_PREC_BF16 && !p->fp8 && m->cfg.hidden == 256 && m->cfg.kernel == 3 && !generic) ? 1 : 0;
    // the forward keeps every depthwise output (the pointwise GEMM's operand) for the weight gradients
    p->save_q = true;
    // hidden >= 512: stand-alone depthwise producer + plain pointwise GEMM (tn_fwd_kernels.h)
    p->split_dw = !p->use_v2 && (p->fp8 || m->cfg.hidden >= 512) && m->cfg.hidden % 8 == 0;
  }
  const tn_config& c = m->cfg;
  const size_t M = p->M, H = c.hidden, D = c.enc_out, A = c.attn_hidden, Hr = c.hidden / c.se_reduction;
  const size_t e = p->esz;
  Bump b;
  // ---- region cleared at the start of every forward: statistics, loss accumulator
  p->zero_begin = b.take(0);
  p->stats.resize(m->n_bn);
  p->bsums.resize(m->n_bn);
  for (int i = 0; i < m->n_bn; ++i) p->stats[i] = b.take((size_t)TN_NREP * 2 * m->all_bn[i].C * sizeof(float));
  p->loss_acc = b.take(256);
  p->tail_parts = 1;
  // (utterances of >= 512 frames only: the summation tree of the SE mean changes with the number of parts, and the short
  //  fixed-length batches keep embeddings that do not depend on the batch they are computed in, bit for bit)
  if (batch * 2 <= 256 && frames >= 512) p->tail_parts = std::max(1, std::min(std::min(256 / batch, 16), frames / 128));
  p->se_parts = p->tail_parts;
  if (p->tail_parts > 1) {
    // TN_SE_ONE_LAUNCH=1: partial sums and the two mat-vecs in ONE launch (se_squeeze_fc_kernel mode 3) instead of two.
    // Measured on configs[3] (32 x <= 1969 frames, hidden 512): 10.37 ms either way — the last arrival's mat-vecs are the
    // same latency chain the second launch was, only the ~2 us launch gap goes away.  Off: the two-launch form is the tested default.
    const char* e1 = getenv("TN_SE_ONE_LAUNCH");
    if (e1 && atoi(e1) == 1) {
      p->se_parts = std::max(p->tail_parts, std::min(std::min(16, 512 / batch), frames / 128));
      p->se_cnt = b.take(sizeof(int) * (size_t)batch);
    }
  }
  p->zero_bytes = ((b.off + 255) & ~(size_t)255) - p->zero_begin;
  b.off = p->zero_begin + p->zero_bytes;
  // ---- region cleared at the start of every backward (so a backward can be repeated from one forward,
  //      loss.backward(retain_graph=True) twice): BN-backward sums, split-K counters, depthwise accumulators
  p->bzero_begin = b.take(0);
  for (int i = 0; i < m->n_bn; ++i) p->bsums[i] = b.take((size_t)TN_NREP * 2 * m->all_bn[i].C * sizeof(float));
  {
    const size_t hs = std::max<size_t>(H / 256, 1);
    p->wg2_count = b.take(sizeof(int) * (size_t)(c.n_mega_blocks * (c.n_sub_blocks + 1) * hs * hs + (D / 256 + 1) * hs + 2 * (D / 256) + 1));
  }
  p->dw_gacc = b.take((size_t)c.n_mega_blocks * c.n_sub_blocks * TN_NREP * (c.kernel + 1) * H * sizeof(float));
  {
    // fp8 weight gradient: this step's column maxima of |dS| per sub-block layer (zeroed with the region)
    const char* ew = getenv("TN_FP8_WGRAD");
    p->fp8_wgrad = p->fp8_bwd && !(ew && atoi(ew) == 0);
    if (p->fp8_wgrad) {
      p->blk.resize(c.n_mega_blocks);
      for (int i = 0; i < c.n_mega_blocks; ++i)
        for (int j = 0; j < c.n_sub_blocks; ++j) p->blk[i].amax_cur.push_back(b.take(H * sizeof(float)));
    }
  }
  if (p->tail_parts > 1) p->dgate_acc = b.take((size_t)c.n_mega_blocks * batch * H * sizeof(float));
  if (p->tail_parts > 1 && precision == TN_PREC_BF16) p->se_bacc = b.take((size_t)batch * 4 * H * sizeof(float));
  p->bzero_bytes = ((b.off + 255) & ~(size_t)255) - p->bzero_begin;
  b.off = p->bzero_begin + p->bzero_bytes;
  // ---- device-resident step state {uint64 step; uint32 word; ...}: cleared once at bind, advanced by tn_plan_step_tick
  p->step_state = b.take(64);
  p->lens = b.take(sizeof(int) * (size_t)batch);
  p->rowtiles = b.take(sizeof(int) * ((size_t)(M + 255) / 256 + 1));
  if (p->tail_parts > 1) p->se_acc = b.take((size_t)batch * p->se_parts * H * sizeof(float));      // one block at a time
  // ---- compute-precision weights
  auto wc = [&](size_t n, size_t k) {
    WcRef r; r.w = b.take(n * k * e); r.wt = b.take(n * k * e);
    if (p->use_v2 && ((n == 256 && k == 256) || (n % 256 == 0 && (k == 256 || k == 128)) || (k % 256 == 0 && n == 128))) {
      r.sw = b.take(n * k * e); r.swt = b.take(n * k * e);
    }
    return r;
  };
  p->wprolog = wc(H, (size_t)c.n_mels * c.prolog_kernel);
  p->wepi = wc(D, H);
  p->wwin = wc(A, D);
  p->wwout = wc(D, A);
  // ---- activations
  p->Y0 = b.take(M * H * e);
  {
    p->wide_dw_bwd = !p->generic;
    p->prolog_taps = precision == TN_PREC_BF16 && c.n_mels % 8 == 0;
    if (p->prolog_taps) {
      p->x0 = b.take(M * (size_t)c.n_mels * e + 64);
      p->wprolog_taps = b.take(H * (size_t)c.n_mels * c.prolog_kernel * e);
      p->prolog_gtmp = b.take(H * (size_t)c.n_mels * c.prolog_kernel * sizeof(float));
    }
  }
  p->blk.resize(c.n_mega_blocks);
  for (auto& bw : p->blk) {
    for (int j = 0; j < c.n_sub_blocks; ++j) { bw.Y.push_back(b.take(M * H * e)); bw.wpw.push_back(wc(H, H)); }
    if (p->save_q)
      for (int j = 0; j < c.n_sub_blocks; ++j) bw.Q.push_back(b.take(M * H * e));
    bw.wskip = wc(H, H);
    bw.S = b.take(M * H * e);
    bw.OUT = b.take(M * H * e);
    bw.m = b.take((size_t)batch * H * 4);
    bw.h = b.take((size_t)batch * Hr * 4);
    bw.g = b.take((size_t)batch * H * 4);
    bw.dpre2 = b.take((size_t)batch * H * 4);
    bw.dpre1 = b.take((size_t)batch * Hr * 4);
    bw.dgate = b.take((size_t)batch * H * 4);
    for (int j = 0; j < c.n_sub_blocks; ++j) bw.dY.push_back(b.take(M * H * e));
    bw.dZk = b.take(M * H * e);
  }
  if (p->fp8) {
    p->q8 = b.take(M * H);
    for (auto& bw : p->blk)
      for (int j = 0; j < c.n_sub_blocks; ++j) { bw.w8.push_back(b.take(H * H)); bw.w8s.push_back(b.take(H * sizeof(float))); }
    p->fp8_table = b.take(sizeof(Fp8CastDesc) * (size_t)std::max(1, c.n_mega_blocks * (2 * c.n_sub_blocks + 1)));
    if (p->fp8_wgrad) {
      for (auto& bw : p->blk)
        for (int j = 0; j < c.n_sub_blocks; ++j) { bw.Q8.push_back(b.take(M * H)); bw.dS8c.push_back(b.take(M * H)); bw.cexp.push_back(b.take(H)); }
      // (one contiguous run, like amax_cur: the end of backward rolls this step's maxima over with one copy)
      for (auto& bw : p->blk)
        for (int j = 0; j < c.n_sub_blocks; ++j) bw.amax_prev.push_back(b.take(H * sizeof(float)));
      p->tn_f8_table = b.take((size_t)c.n_mega_blocks * c.n_sub_blocks * 64);      // >= sizeof(PGemmTnF8Desc) each (checked at upload)
      p->tn_skip_table = b.take((size_t)c.n_mega_blocks * 64);
    }
    if (p->fp8_bwd) {
      p->ds8 = b.take(M * H);
      p->dsexp = b.take(((size_t)(M + 255) / 256) * 256);
      p->ds8s = b.take(M * H);
      p->dsexps = b.take(((size_t)(M + 255) / 256) * 256);
      for (auto& bw : p->blk) {
        for (int j = 0; j < c.n_sub_blocks; ++j) { bw.w8t.push_back(b.take(H * H)); bw.w8ts.push_back(b.take(H * sizeof(float))); }
        bw.w8t_skip = b.take(H * H); bw.w8ts_skip = b.take(H * sizeof(float));
      }
    }
  }
  p->E = b.take(M * D * e);
  p->HID = b.take(M * A * e + 512);      // + slack: the batched weight-gradient units read it 256 wide (row stride 128)
  p->EN = b.take(M * D * e);
  p->pooled = b.take((size_t)batch * 2 * D * 4);
  p->smax = b.take((size_t)batch * D * 4);
  p->sinv = b.take((size_t)batch * D * 4);
  p->qv = b.take((size_t)batch * D * 4);
  p->lin = b.take((size_t)batch * c.emb * 4);
  p->emb = b.take((size_t)batch * c.emb * 4);
  p->emb_norm = b.take((size_t)batch * c.emb * 4);
  const size_t nc = std::max(c.n_classes, 1);
  p->dlogits = b.take((size_t)batch * nc * 4);
  p->logits = b.take((size_t)batch * nc * 4);
  p->dscale = b.take((size_t)batch * 4);
  p->preds = b.take((size_t)batch * 8);
  // ---- backward scratch
  p->dA[0] = b.take(M * H * e);
  p->dA[1] = b.take(M * H * e);
  p->dYbn = b.take(M * H * e);
  p->dD = b.take(M * H * e);
  p->dZ = b.take(M * H * e);
  p->dXs = b.take(M * H * e);
  p->dE = b.take(M * D * e);
  p->dEbn = b.take(M * D * e);
  p->dHP = b.take(M * A * e + 512);
  p->dpooled = b.take((size_t)batch * 2 * D * 4);
  p->mu = b.take((size_t)batch * D * 4);
  p->dmu = b.take((size_t)batch * D * 4);
  p->dlin = b.take((size_t)batch * c.emb * 4);
  p->demb = b.take((size_t)batch * c.emb * 4);
  // split-K slabs for weight gradients: sized for the largest weight (see tn_bwd.hip)
  {
    size_t biggest = std::max({H * H, D * H, D * A, H * (size_t)c.n_mels * c.prolog_kernel});
    p->slab_bytes = biggest * sizeof(float) * 48;   // ~48 K-splits of the largest weight, more for smaller ones
    p->slabs = b.take(p->slab_bytes);
  }
  // hidden 512 / 1024 (TitaNet-M / -L), bf16: slab kernels for the depthwise convs, pipelined GEMMs (tn_pgemm.h)
  p->wide_wgrad = !p->use_v2 && !p->generic && precision == TN_PREC_BF16 && (H == 512 || H == 1024) && D % 256 == 0 && c.n_mega_blocks > 0;
  // (dropout > 0: the slab depthwise backward has no variant without it, and the generic one does not mask its dD operand)
  p->skip_pad_tiles = p->wide_wgrad && p->wide_dw_bwd && (c.kernel == 7 || c.kernel == 11) && !c.simple_pool && c.dropout > 0.f;
  if (p->use_v2) {
    const int hs = (int)(H / 256);
    p->wg2_upl = hs * hs;
    p->wg2_layers = c.n_mega_blocks * (c.n_sub_blocks + 1) * p->wg2_upl;
    // the epilog conv's weight gradient rides along as (D / 256) x (H / 256) slabs of 256 x 256
    p->wg2_epi_slabs = (c.n_mega_blocks > 0 && D % 256 == 0 && p->use_v2) ? (int)(D / 256) * hs : 0;
    p->wg2_layers += p->wg2_epi_slabs;
    // ... and the two weight gradients of the attentive pooling (D x 128 and 128 x D): one unit per 256-channel slab of D,
    // the 128-wide operand read as 256 (half of the unit's output is dropped by the reduction)
    p->wg2_asp_units = (p->wg2_epi_slabs > 0 && p->use_v2 && H == 256 && A == 128 && !c.simple_pool) ? 2 * (int)(D / 256) : 0;
    p->wg2_layers += p->wg2_asp_units;
    p->wg2_grid = 256;
    p->wg2_desc = b.take((size_t)3 * p->wg2_layers * 256);   // three tables, >= sizeof(WgradV2Desc) per unit (checked at upload)
    p->wg2_out = b.take((size_t)p->wg2_layers * 32);
    p->wg2_out3 = b.take((size_t)p->wg2_layers * 32);
    // the sub-block pointwise weight gradients as one pipelined TN contraction over stored operands (dS kept by dgrad_dw_v6,
    // the kept depthwise output): needs the one-pass tail (the last sub-block's dS) and one 256 x 256 slab per layer
    p->v2_tn = p->use_v2 && H == 256 && c.n_sub_blocks >= 2 && p->wg2_upl == 1;
    if (p->v2_tn) {
      for (auto& bw : p->blk) {
        bw.dS.assign(c.n_sub_blocks, 0);
        for (int j = 0; j + 1 < c.n_sub_blocks; ++j) bw.dS[j] = b.take(M * H * e);
        if (&bw != &p->blk[0]) bw.dS_skip = b.take(M * H * e);      // (block 0's skip conv reads the prolog output through its activation: stays in wgrad_batched_v2)
      }
      p->tn_table = b.take((size_t)c.n_mega_blocks * (c.n_sub_blocks + 1) * 64);
    }
    p->dw_table = b.take((size_t)c.n_mega_blocks * c.n_sub_blocks * 32);
  }
  if (p->wide_wgrad && p->wide_dw_bwd && (c.kernel == 7 || c.kernel == 11) && H % 256 == 0 && c.dropout > 0.f) {
    // wide models: the depthwise tap / bias gradients as stored partial sums + one reduction per gradient bucket
    // (TN_DW_PART=0: atomics on the gradient buffer, round 5)
    const char* ep = getenv("TN_DW_PART");
    if (!(ep && atoi(ep) == 0)) {
      p->dw_part_stride = (size_t)256 * (c.kernel + 1) * 256 * sizeof(float);
      p->dw_part = b.take((size_t)c.n_mega_blocks * c.n_sub_blocks * p->dw_part_stride);
      p->dw_table = b.take((size_t)c.n_mega_blocks * c.n_sub_blocks * 32);
    }
  }
  if (precision == TN_PREC_BF16) p->se_gu = b.take((size_t)batch * 2 * H * sizeof(float));      // fused mega-block tail backward
  if (p->wide_wgrad) p->a0 = b.take(M * H * e);      // the activated prolog output as a stored operand (first block's skip conv)
  if (p->wide_wgrad) p->tn_table = b.take((size_t)c.n_mega_blocks * (c.n_sub_blocks + 1) * 64);   // >= sizeof(PGemmTnDesc) each
  p->cast_table = b.take(sizeof(CastDesc) * (8 + (size_t)c.n_mega_blocks * (c.n_sub_blocks + 1)));
  p->swz_table = b.take(sizeof(SwzDesc) * (2 * (size_t)c.n_mega_blocks * (c.n_sub_blocks + 1) + 8));
  p->bn_table = b.take(sizeof(BnUpdateDesc) * m->n_bn);
  p->stats_ptr_table = b.take(sizeof(float*) * m->n_bn);
  p->bwd_table_bytes = (size_t)m->n_bn * 128;   // >= sizeof(BnGradDesc) each (checked at upload)
  p->bwd_table = b.take(p->bwd_table_bytes);
  p->bwd_table_eval = b.take(p->bwd_table_bytes);
  p->se_table = b.take((size_t)(c.n_mega_blocks + 1) * 64);
  p->ws_fixed_bytes = (b.off + 255) & ~(size_t)255;
  plan_layout_tail(p);
  {
    // OFF by default: measured round 5 (profiles/r05_ab_overlap.txt, same box, alternating): 9.01 ms with the side stream
    // against 8.81 without — two persistent 256-workgroup grids do not share the chip, the late workgroups of whichever
    // kernel found the CUs taken start when the other kernel's finish and run their whole tile list alone.  TN_OVERLAP=1 (read
    // at plan creation) turns it on for A/B runs.
    {
      const char* ef = getenv("TN_ASP_FUSED");
      const auto& cc = m->cfg;
      p->asp_fused = !(ef && atoi(ef) == 0) && precision == TN_PREC_BF16 && !p->generic && !cc.simple_pool && cc.enc_out % 256 == 0 &&
                     cc.attn_hidden == 128 && frames <= ASPV2_PR && p->tail_parts == 1;
    }
    {
      const char* es = getenv("TN_SE_FUSED");
      p->se_fused = !(es && atoi(es) == 0) && precision == TN_PREC_BF16 && p->use_v2 && frames <= 16 * SC3_MAXU && p->tail_parts == 1;
    }
    const char* eo = getenv("TN_OVERLAP");
    p->overlap = p->use_v2 && c.n_mega_blocks > 0 && eo && atoi(eo) != 0;
    {
      // (the late launch wants S stored with the default policy: it follows the compile-time store policy unless TN_SKIP_LATE says otherwise)
      const char* el = getenv("TN_SKIP_LATE");
      p->skip_late = el ? atoi(el) != 0 : (p->use_v2 && !(TN_NT_WGRAD_OPERANDS & 8));
    }
    {
      // wide models: the skip conv's output and the skip data gradient (pipelined GEMMs) stored non-temporal; TN_NT_SKIP=0 turns it off
      const char* en = getenv("TN_NT_SKIP");
      p->nt_skip = !(en && atoi(en) == 0);
      // the wide models' forward pointwise outputs (read by the next depthwise launch, then not before the backward pass):
      // same-box A/B (tools/env_ab.sh, TN_RW_NT=0 / 1): L bf16 20.02 -> 19.84 ms, L fp8 16.69 -> 16.64, M 14.37 -> 14.34, configs[3]
      // 10.09 -> 10.00; bit 2 (their data gradients) is worse on M (14.25 -> 14.51)
      { const char* er = getenv("TN_RW_NT"); p->rw_nt = er ? atoi(er) : 1; }
    }
    if (p->overlap) {
      if (hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) != hipSuccess) { p->side = nullptr; p->overlap = false; }
      for (int i = 0; p->overlap && i < 4 * c.n_mega_blocks; ++i) {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { p->overlap = false; break; }
        p->ov_events.push_back(ev);
      }
    }
  }
  *out = p;
  return 0;
}

// Gradient buckets + the regions sized by them.  grad_groups = 1: one bucket; G > 1: bucket 0 = everything from the epilog
// conv to the loss head (final first), then G groups of mega blocks from the last block down, the prolog riding with the
// group of block 0 (the flat buffer is in state_dict order, so every bucket is one contiguous range).
void plan_layout_tail(tn_plan* p) {
  const tn_model* m = p->model;
  const tn_config& c = m->cfg;
  const int nb = c.n_mega_blocks;
  p->buckets.clear();
  const int G = std::max(1, std::min(p->grad_groups, std::max(nb, 1)));
  if (G == 1 || nb == 0) {
    tn_plan::GradBucket bk;
    bk.begin = 0; bk.end = m->n_params; bk.blk_lo = 0; bk.blk_hi = nb - 1; bk.tail = true; bk.prolog = true;
    p->buckets.push_back(bk);
  } else {
    tn_plan::GradBucket tail;
    tail.begin = m->epi_w; tail.end = m->n_params; tail.tail = true;
    p->buckets.push_back(tail);
    int hi = nb - 1;
    for (int g = 0; g < G; ++g) {
      const int left = G - g, cnt = (hi + 1 + left - 1) / left;
      const int lo = hi - cnt + 1;
      tn_plan::GradBucket bk;
      bk.blk_lo = lo; bk.blk_hi = hi;
      bk.begin = lo == 0 ? 0 : m->blocks[lo].sub[0].wdw;
      bk.end = hi + 1 < nb ? m->blocks[hi + 1].sub[0].wdw : m->epi_w;
      bk.prolog = lo == 0;
      p->buckets.push_back(bk);
      hi = lo - 1;
    }
  }
  Bump b;
  b.off = p->ws_fixed_bytes;
  if (p->use_v2 && p->wg2_layers > 0) {
    // every group's launch cuts its (layer, 32-row chunk) units into one contiguous range per workgroup: the number of
    // partial slabs a layer can receive is bounded by the smallest group
    const int chunks = (p->M + 31) / 32;
    auto parts_for = [&](int count) {
      if (count <= 0) return 0;
      const long total = (long)count * chunks;
      const int upw = (int)((total + p->wg2_grid - 1) / p->wg2_grid);
      return (chunks + upw - 1) / upw + 1;
    };
    const int per_blk = (c.n_sub_blocks + 1) * p->wg2_upl;
    const int n_tail = p->wg2_epi_slabs + p->wg2_asp_units;
    // full layout (every unit of a bucket in its launch; uniform stride of wg2_maxparts slabs per unit)
    int maxparts = 1;
    for (const auto& bk : p->buckets) {
      const bool hasb = bk.blk_hi >= bk.blk_lo;
      const int layers = (hasb ? (bk.blk_hi - bk.blk_lo + 1) * per_blk : 0) + (bk.tail ? n_tail : 0);
      // (variable-length batches launch the group of block 0 without its first layer: both partitions must fit)
      maxparts = std::max(maxparts, parts_for(layers));
      if (hasb && bk.blk_lo == 0) maxparts = std::max(maxparts, parts_for(layers - p->wg2_upl));
    }
    p->wg2_maxparts = maxparts;
    size_t need = (size_t)p->wg2_layers * maxparts;
    // compact layout of v2_tn plans (third descriptor table: block 0's skip conv, then the epilog / pooling units): a bucket's
    // launch may hold ONE unit, cut over the whole grid — so its units get their own slab ranges inside the same region
    // (unit 0: parts3_u0 slabs, every tail unit: parts3_tail) instead of inflating the uniform stride of all ~86 units
    p->wg2_parts3_u0 = p->wg2_parts3_tail = 0;
    if (p->v2_tn) {
      for (const auto& bk : p->buckets) {
        const bool b0 = bk.blk_hi >= bk.blk_lo && bk.blk_lo == 0;
        const int cnt = (b0 ? 1 : 0) + (bk.tail ? n_tail : 0);
        const int parts = std::max(parts_for(cnt), b0 ? parts_for(cnt - 1) : 0);
        if (b0) p->wg2_parts3_u0 = std::max(p->wg2_parts3_u0, parts);
        if (bk.tail) p->wg2_parts3_tail = std::max(p->wg2_parts3_tail, parts);
      }
      need = std::max(need, (size_t)p->wg2_parts3_u0 + (size_t)n_tail * p->wg2_parts3_tail);
    }
    p->wg2_slabs = b.take(need * 256 * 256 * sizeof(float));
  }
  p->ws_bytes = (b.off + 255) & ~(size_t)255;
}

extern "C" int tn_plan_set_grad_groups(tn_plan* p, int32_t groups) {
  if (!p || groups < 1 || groups > 64) return TN_E_BADARG;
  const int old = p->grad_groups;
  p->grad_groups = groups;
  plan_layout_tail(p);
  if (!p->bound) return 0;
  // already bound: allowed when the new layout fits the bound workspace (the region sized by the groups is the last one);
  // re-uploads the descriptor tables (synchronises the given plan's stream 0: not for use inside a capture)
  if (p->ws_bytes > p->bound_bytes) {
    p->grad_groups = old;
    plan_layout_tail(p);
    return TN_E_STATE;
  }
  int rc = plan_upload_bwd_tables(p, 0);
  if (rc) return rc;
  while (p->bucket_events.size() < p->buckets.size()) {
    hipEvent_t e;
    TN_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    p->bucket_events.push_back(e);
  }
  return 0;
}
extern "C" int32_t tn_plan_num_grad_buckets(const tn_plan* p) { return p ? (int32_t)p->buckets.size() : 0; }
extern "C" int tn_plan_grad_bucket(const tn_plan* p, int32_t i, int64_t* begin, int64_t* end) {
  if (!p || i < 0 || i >= (int32_t)p->buckets.size() || !begin || !end) return TN_E_BADARG;
  *begin = p->buckets[i].begin; *end = p->buckets[i].end;
  return 0;
}
extern "C" int tn_plan_wait_grad_bucket(tn_plan* p, int32_t i, void* stream) {
  if (!p || i < 0 || i >= (int32_t)p->bucket_events.size()) return TN_E_BADARG;
  TN_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, p->bucket_events[i], 0));
  return 0;
}

extern "C" void tn_plan_destroy(tn_plan* p) {
  if (!p) return;
  for (auto e : p->prof_events) (void)hipEventDestroy(e);
  for (auto e : p->bucket_events) (void)hipEventDestroy(e);
  for (auto e : p->ov_events) (void)hipEventDestroy(e);
  if (p->side) (void)hipStreamDestroy(p->side);
  delete p;
}

extern "C" int tn_profile_begin(tn_plan* p, int32_t kernel_class) {
  if (!p || kernel_class < 0 || kernel_class > 4) return TN_E_BADARG;
  p->prof_class = kernel_class;
  p->prof_used = 0;
  return 0;
}

extern "C" int tn_profile_sample(tn_plan* p, int32_t every_n) {
  if (!p || every_n < 1) return TN_E_BADARG;
  p->prof_stride = every_n;
  p->prof_counter = 0;
  return 0;
}

extern "C" int tn_profile_read(tn_plan* p, double* total_ms, int64_t* launches) {
  if (!p || !total_ms || !launches) return TN_E_BADARG;
  double t = 0.0;
  for (size_t i = 0; i + 1 < p->prof_used; i += 2) {
    TN_CHECK_HIP(hipEventSynchronize(p->prof_events[i + 1]));
    float ms = 0.f;
    TN_CHECK_HIP(hipEventElapsedTime(&ms, p->prof_events[i], p->prof_events[i + 1]));
    t += ms;
  }
  *total_ms = t;
  *launches = (int64_t)(p->prof_used / 2);
  return 0;
}
extern "C" size_t tn_plan_workspace_bytes(const tn_plan* p) { return p ? p->ws_bytes : 0; }

extern "C" int tn_plan_bind(tn_plan* p, float* params, float* grads, float* bnbuf, int64_t* nbt, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (!p || !params || !bnbuf || !nbt || !workspace) return TN_E_BADARG;
  if (workspace_bytes < p->ws_bytes) return TN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  p->params = params; p->grads = grads; p->bnbuf = bnbuf; p->nbt = nbt; p->ws = (char*)workspace;
  p->bound_bytes = workspace_bytes;
  // every byte defined once: variable-length batches leave the rows of padding-only tiles untouched (PGemmNtArgs::rowtiles),
  // and what those rows hold is multiplied by zero weights later — it must never be a NaN / Inf bit pattern
  if (p->skip_pad_tiles) TN_CHECK_HIP(hipMemsetAsync(p->ws, 0, p->ws_bytes, st));
  TN_CHECK_HIP(hipMemsetAsync(p->ws + p->step_state, 0, 64, st));
  const tn_model* m = p->model;
  const tn_config& c = m->cfg;
  // cast table: bf16 needs the straight copy + the transpose; fp32 reads the masters directly and
  // only needs the transposes (for the data-gradient GEMMs)
  std::vector<CastDesc> cd;
  const bool bf = p->prec == TN_PREC_BF16;
  auto add = [&](int64_t off, const WcRef& r, int R, int C, bool need_t) {
    CastDesc d;
    d.src = params + off;
    d.dst = bf ? (void*)(p->ws + r.w) : nullptr;
    d.dstT = need_t ? (void*)(p->ws + r.wt) : nullptr;
    d.R = R; d.C = C;
    if (d.dst || d.dstT) cd.push_back(d);
  };
  add(m->prolog_w, p->wprolog, c.hidden, c.n_mels * c.prolog_kernel, false);
  add(m->epi_w, p->wepi, c.enc_out, c.hidden, true);
  if (!c.simple_pool) {
    add(m->asp_win, p->wwin, c.attn_hidden, c.enc_out, true);
    add(m->asp_wout, p->wwout, c.enc_out, c.attn_hidden, true);
  }
  for (int i = 0; i < c.n_mega_blocks; ++i) {
    for (int j = 0; j < c.n_sub_blocks; ++j) add(m->blocks[i].sub[j].wpw, p->blk[i].wpw[j], c.hidden, c.hidden, true);
    add(m->blocks[i].wskip, p->blk[i].wskip, c.hidden, c.hidden, true);
  }
  {
    std::vector<SwzDesc> sd;
    auto adds = [&](const WcRef& r) {
      if (!r.sw) return;
      sd.push_back(SwzDesc{(const bf16_t*)(p->ws + r.w), (uint4*)(p->ws + r.sw), 256, 256});
      sd.push_back(SwzDesc{(const bf16_t*)(p->ws + r.wt), (uint4*)(p->ws + r.swt), 256, 256});
    };
    // the wide kernels' weights: epilog conv [D][H], ASP energies [D][A], and W_in^T [D][A] for the backward
    // (the epilog conv's fragment-order copy is read whatever the pooling layer is: with `simple_pool` it used to be left
    //  unwritten, and the wide epilog kernel multiplied by whatever the workspace held — found by tools/fuzz_paths.py)
    if (p->use_v2 && c.hidden == 256 && c.enc_out % 256 == 0 && p->wepi.sw)
      sd.push_back(SwzDesc{(const bf16_t*)(p->ws + p->wepi.w), (uint4*)(p->ws + p->wepi.sw), c.enc_out, c.hidden});
    if (p->use_v2 && !c.simple_pool && c.hidden == 256 && c.enc_out % 256 == 0 && c.attn_hidden == 128) {
      if (p->wwout.sw) sd.push_back(SwzDesc{(const bf16_t*)(p->ws + p->wwout.w), (uint4*)(p->ws + p->wwout.sw), c.enc_out, c.attn_hidden});
      if (p->wwin.swt) sd.push_back(SwzDesc{(const bf16_t*)(p->ws + p->wwin.wt), (uint4*)(p->ws + p->wwin.swt), c.enc_out, c.attn_hidden});
    }
    for (int i = 0; i < c.n_mega_blocks; ++i) {
      for (int j = 0; j < c.n_sub_blocks; ++j) adds(p->blk[i].wpw[j]);
      adds(p->blk[i].wskip);
    }
    p->n_swz = (int)sd.size();
    if (p->n_swz) TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->swz_table, sd.data(), sd.size() * sizeof(SwzDesc), hipMemcpyHostToDevice, st));
  }
  if (p->fp8) {
    std::vector<Fp8CastDesc> fd;
    for (int i = 0; i < c.n_mega_blocks; ++i) {
      if (p->fp8_bwd)
        fd.push_back(Fp8CastDesc{nullptr, (uint8_t*)(p->ws + p->blk[i].w8t_skip), (float*)(p->ws + p->blk[i].w8ts_skip), c.hidden, c.hidden,
                                 (const bf16_t*)(p->ws + p->blk[i].wskip.wt)});
      for (int j = 0; j < c.n_sub_blocks; ++j)
      {
        fd.push_back(Fp8CastDesc{params + m->blocks[i].sub[j].wpw, (uint8_t*)(p->ws + p->blk[i].w8[j]), (float*)(p->ws + p->blk[i].w8s[j]),
                                 c.hidden, c.hidden, nullptr});
        if (p->fp8_bwd)      // rows of the transposed bf16 copy (cast_params_kernel runs first)
          fd.push_back(Fp8CastDesc{nullptr, (uint8_t*)(p->ws + p->blk[i].w8t[j]), (float*)(p->ws + p->blk[i].w8ts[j]), c.hidden, c.hidden,
                                   (const bf16_t*)(p->ws + p->blk[i].wpw[j].wt)});
      }
    }
    p->n_fp8 = (int)fd.size();
    if (p->n_fp8) TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->fp8_table, fd.data(), fd.size() * sizeof(Fp8CastDesc), hipMemcpyHostToDevice, st));
    TN_CHECK_HIP(hipStreamSynchronize(st));
  }
  p->n_cast = (int)cd.size();
  TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->cast_table, cd.data(), cd.size() * sizeof(CastDesc), hipMemcpyHostToDevice, st));
  std::vector<BnUpdateDesc> bd(m->n_bn);
  for (int i = 0; i < m->n_bn; ++i) {
    const BnRef& r = m->all_bn[i];
    bd[i].stats = (const float*)(p->ws + p->stats[i]);
    bd[i].rmean = bnbuf + r.rmean;
    bd[i].rvar = bnbuf + r.rvar;
    bd[i].C = r.C;
    bd[i].n = (i == m->pool_bn.id || i == m->lin_bn.id) ? p->B : p->M;
  }
  TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->bn_table, bd.data(), bd.size() * sizeof(BnUpdateDesc), hipMemcpyHostToDevice, st));
  std::vector<float*> sp(m->n_bn);
  for (int i = 0; i < m->n_bn; ++i) sp[i] = (float*)(p->ws + p->stats[i]);
  TN_CHECK_HIP(hipMemcpyAsync(p->ws + p->stats_ptr_table, sp.data(), sp.size() * sizeof(float*), hipMemcpyHostToDevice, st));
  TN_CHECK_HIP(hipStreamSynchronize(st));   // host vectors go out of scope
  int rc = plan_upload_bwd_tables(p, st);
  if (rc) return rc;
  while (p->bucket_events.size() < p->buckets.size()) {
    hipEvent_t e;
    TN_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    p->bucket_events.push_back(e);
  }
  p->fp8_hist_valid = false;      // (a freshly bound workspace holds no column maxima: the next backward records them, tn_bwd.hip)
  p->bound = true;
  return 0;
}

// =============================================================================================
// forward
// =============================================================================================
BnAct identity_act() {
  BnAct a;
  memset(&a, 0, sizeof(a));
  return a;
}
RowMask plan_row_mask(const tn_plan* p) {
  RowMask m;
  m.len = p->masked ? (const int*)(p->ws + p->lens) : nullptr;
  m.T = p->T;
  return m;
}
// identity activation of a stored, already activated [B*T]-row tensor (mega-block outputs): only the padding mask applies
static BnAct identity_rows(const tn_plan* p) {
  BnAct a = identity_act();
  a.rm = plan_row_mask(p);
  return a;
}

BnAct make_act(const tn_plan* p, const BnRef& bn, int rows, int training, int relu, float drop_p, uint64_t seed,
               int layer) {
  BnAct a;
  memset(&a, 0, sizeof(a));
  a.stats = (const float*)(p->ws + p->stats[bn.id]);
  a.gamma = p->params + bn.gamma;
  a.beta = p->params + bn.beta;
  a.inv_n = 1.f / (float)rows;
  if (p->masked && rows == p->M) {
    a.rm = plan_row_mask(p);
    // train: the statistics were summed over the valid rows only; eval: the synthetic sums of the running statistics
    // (bn_eval_prepare_kernel) are built for n = B*T, whatever the lengths
    if (training) a.inv_n = 1.f / (float)std::max(p->n_valid, 1);
  }
  a.eps = 1e-5f;
  a.mode = 1;
  a.relu = relu;
  if (training && drop_p > 0.f) {
    a.drop_thr = (uint32_t)lrintf(drop_p * 65536.f);
    a.drop_key = tn_layer_key(seed, (uint32_t)layer);
    a.inv_keep = 1.f / (1.f - drop_p);
    a.key_add = p->ws ? (const uint32_t*)(p->ws + p->step_state) + 2 : nullptr;
  }
  return a;
}

namespace {

template <typename AT, typename Prod, typename Epi = EpiStore>
int gemm_store(const GemmShape& g, const typename Prod::Args& pa, const EpiStoreArgs& ea, int KD, hipStream_t st) {
  if (g.N > 128) return launch_gemm<AT, 2, 4, Prod, Epi>(g, pa, ea, KD, st);
  return launch_gemm<AT, 2, 2, Prod, Epi>(g, pa, ea, KD, st);
}

// plain stored bf16 operand, big problem: the pipelined LDS-DMA GEMM (tn_pgemm.h); -1000 = not applicable
template <typename AT>
int gemm_plain_pipe(const tn_plan* p, const GemmShape& g, const void* X, int ldx, const BnAct& act, const EpiStoreArgs& ea, hipStream_t st,
                    bool zero_pad = true, bool nt_out = false) {
  // variable-length batches: every plain operand of this path is STORED with zero padding rows (dw_fwd_slab, combine_fwd),
  // so the masks of the activation / epilogue have nothing left to do except for the statistics: y == bias on those rows
  if (sizeof(AT) != 2 || p->generic || (p->masked && (g.M != p->M || !zero_pad))) return -1000;
  if (act.mode != 0 || act.relu || act.drop_thr) return -1000;
  if (g.K % 32 || g.N % 64 || g.N > 3072 || ldx % 8 || ea.ldy % 2) return -1000;
  if (g.K < 256) return -1000;
  const bool listed = p->masked && p->n_rowtiles > 0;
  PGemmNtArgs pa{(const bf16_t*)X, ldx, listed ? (const int*)(p->ws + p->rowtiles) : nullptr, listed ? p->n_rowtiles : 0};
  // (statistics: the padding rows INSIDE the computed tiles give y == bias; the skipped tiles add nothing)
  PGemmEpiArgs pe{(bf16_t*)ea.Y, ea.ldy, ea.bias, ea.stats, ea.colscale, p->masked ? (float)(p->active_rows - p->n_valid) : 0.f, nt_out ? 1 : 0};
  return launch_pgemm_nt(g, pa, pe, st);
}

template <typename AT>
const void* wsel(const tn_plan* p, int64_t master_off, const WcRef& r) {
  if (sizeof(AT) == 4) return p->params + master_off;
  return p->ws + r.w;
}

template <typename AT>
int forward_impl(tn_plan* p, const float* spec, const int64_t* speakers, int training, uint64_t seed, float* emb_out,
                 int64_t* preds, float* loss, hipStream_t st) {
  const tn_model* m = p->model;
  const tn_config& c = m->cfg;
  const int M = p->M, H = c.hidden, D = c.enc_out, A = c.attn_hidden, Hr = c.hidden / c.se_reduction, T = p->T, B = p->B;
  char* ws = p->ws;
  float* params = p->params;
  const float pd = c.dropout;
  auto statp = [&](const BnRef& bn) -> float* { return training ? (float*)(ws + p->stats[bn.id]) : nullptr; };
  // variable-length batches run the generic kernel templates (the mask lives in their activation-on-load and epilogues)
  // variable-length batches: the specialised forward kernels take the padding mask when an utterance spans at least one row
  // tile (T >= 64); padding rows are stored as zeros, and the BatchNorm statistics of a GEMM fed with all-zero rows (y == bias
  // there) are corrected right behind it
  const int use_v2 = (p->masked && T < 64) ? 0 : p->use_v2;
  const RowMask rm = plan_row_mask(p);
  if (p->masked && c.simple_pool) return TN_E_UNSUPPORTED;
  auto pad_fixup_on = [&](float* stats, const float* bias, int C, hipStream_t s2) {
    if (p->masked && training && stats && p->M > p->n_valid)
      hipLaunchKernelGGL(stats_pad_fixup_kernel<0>, dim3(1), dim3(256), 0, s2, stats, bias, (float)(p->M - p->n_valid), C, 1);
  };
  auto pad_fixup = [&](float* stats, const float* bias, int C) { pad_fixup_on(stats, bias, C, st); };
  // side stream (tn_plan::overlap): the skip conv of a mega block runs beside its sub-block chain
  const bool ov = p->overlap && use_v2 && p->side != nullptr;

  p->fwd_q16_skipped = false;
  TN_CHECK_HIP(tn_zero_async(ws + p->zero_begin, p->zero_bytes, st));
  if (p->n_cast > 0) {
    hipLaunchKernelGGL(cast_params_kernel<AT>, dim3(64, p->n_cast), dim3(256), 0, st, (const CastDesc*)(ws + p->cast_table));
    // every bf16 plan: cast_params_kernel leaves each matrix of >= TN_CAST_TILED_MIN elements (R, C multiples of 4) to the tiled
    // kernel, which returns at once for the others — the two per-descriptor tests must see the same set of launches, whatever the
    // model's widths (a hidden x enc_out test here once skipped the prolog / pooling matrices of narrow configurations)
    if (sizeof(AT) == 2)
      hipLaunchKernelGGL(cast_params_tiled_kernel<AT>, dim3(96, p->n_cast), dim3(256), 0, st, (const CastDesc*)(ws + p->cast_table));
    if (p->n_swz) hipLaunchKernelGGL(swizzle256_kernel<0>, dim3(16, p->n_swz), dim3(256), 0, st, (const SwzDesc*)(ws + p->swz_table));
  }
  if (p->fp8 && p->n_fp8 > 0)
    hipLaunchKernelGGL(cast_fp8_rows_kernel, dim3(64, p->n_fp8), dim3(64), 0, st, (const Fp8CastDesc*)(ws + p->fp8_table));
  if (!training) {
    // eval: BatchNorm uses the running statistics -> write the equivalent sums once for all layers
    hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(2, m->n_bn), dim3(256), 0, st, (const BnUpdateDesc*)(ws + p->bn_table),
                       (float* const*)(ws + p->stats_ptr_table));
  }
  // ---- prolog: dense k=3 conv as an im2col GEMM (reference src/models.py:370, :398)
  if (p->prolog_taps) {
    // packed operand: one transposing cast of the input, then the GEMM reads 16-byte vectors (a scalar gather from the
    // [B][C][T] float layout cost 100 us here, and 190 us in the weight gradient).  spec == null: the mel front end wrote
    // the packed operand itself (tn_mel_forward_batch_packed into tn_plan_prolog_input: no f32 spectrogram, no pack pass)
    const size_t tile = (size_t)c.n_mels * 65 * sizeof(float);
    if (spec)
      hipLaunchKernelGGL(prolog_pack_kernel<AT>, dim3((T + 63) / 64, B), dim3(256), tile, st, spec, c.n_mels, T, rm.len, (AT*)(ws + p->x0));
    hipLaunchKernelGGL(prolog_weight_taps_kernel<AT>, dim3(64), dim3(256), 0, st, params + m->prolog_w, H, c.n_mels, c.prolog_kernel,
                       (AT*)(ws + p->wprolog_taps));
    GemmShape g{M, H, c.n_mels * c.prolog_kernel, ws + p->wprolog_taps};
    ProdTaps::Args pa{ws + p->x0, c.n_mels, c.prolog_kernel, T};
    EpiStoreArgs ea{ws + p->Y0, H, params + m->prolog_b, statp(m->prolog_bn), rm};
    int rc = gemm_store<AT, ProdTaps>(g, pa, ea, 0, st);
    if (rc) return rc;
  } else {
    GemmShape g{M, H, c.n_mels * c.prolog_kernel, wsel<AT>(p, m->prolog_w, p->wprolog)};
    ProdIm2col::Args pa{spec, c.n_mels, c.prolog_kernel, T, rm.len};
    EpiStoreArgs ea{ws + p->Y0, H, params + m->prolog_b, statp(m->prolog_bn), rm};
    int rc = gemm_store<AT, ProdIm2col>(g, pa, ea, 0, st);
    if (rc) return rc;
  }
  const void* xin = ws + p->Y0;
  BnAct actx = make_act(p, m->prolog_bn, M, training, 1, 0.f, seed, 0);
  const bool keep_a0 = p->a0 != 0 && sizeof(AT) == 2 && c.n_mega_blocks > 0;
  if (keep_a0)
    hipLaunchKernelGGL(act_store_kernel, dim3(2048), dim3(256), (size_t)2 * H * sizeof(float), st, (const bf16_t*)xin, actx,
                       (bf16_t*)(ws + p->a0), M, H);
  for (int i = 0; i < c.n_mega_blocks; ++i) {
    const MegaBlockRef& mb = m->blocks[i];
    BlockWs& bw = p->blk[i];
    // skip connection: 1x1 conv (reference src/models.py:452-455).  tn_plan::skip_late: launched BEHIND the sub-block chain, right in
    // front of the combine that reads its output (the block input is still in the Infinity Cache then, and S is when the combine
    // reads it); the side-stream overlap keeps the early launch
    bool skip_on_side = false;
    const void* const blk_in = xin;
    const BnAct blk_act = actx;
    auto run_skip = [&]() -> int {
      const void* xin = blk_in;
      const BnAct actx = blk_act;
      int rc = -1000;
      if (keep_a0 && i == 0) {
        GemmShape g{M, H, H, wsel<AT>(p, mb.wskip, bw.wskip)};
        EpiStoreArgs ea{ws + bw.S, H, params + mb.bskip, statp(mb.bnskip), rm};
        rc = gemm_plain_pipe<AT>(p, g, ws + p->a0, H, identity_rows(p), ea, st, true, p->nt_skip);
      }
      if (use_v2) {
        SubFwdV2Args va{(const bf16_t*)xin, actx, nullptr, nullptr, (const bf16_t*)(ws + bw.wskip.w), params + mb.bskip,
                        (bf16_t*)(ws + bw.S), statp(mb.bnskip), M, T, 0,
                        bw.wskip.sw ? (const uint4*)(ws + bw.wskip.sw) : nullptr, nullptr};
        hipStream_t ss = st;
        if (ov) {
          TN_CHECK_HIP(hipEventRecord(p->ov_events[4 * i], st));            // fork: the block input is complete
          TN_CHECK_HIP(hipStreamWaitEvent(p->side, p->ov_events[4 * i], 0));
          ss = p->side;
        }
        rc = launch_sub_fwd_v4<1, false>(va, 256, ss);
        if (rc == 0) pad_fixup_on(statp(mb.bnskip), params + mb.bskip, H, ss);
        if (ov) TN_CHECK_HIP(hipEventRecord(p->ov_events[4 * i + 1], p->side));   // joined in front of the combine below
        if (ov && rc == -1000) TN_CHECK_HIP(hipStreamWaitEvent(st, p->ov_events[4 * i + 1], 0));
        skip_on_side = ov && rc == 0;
      }
      if (rc == -1000) {
        GemmShape g{M, H, H, wsel<AT>(p, mb.wskip, bw.wskip)};
        ProdPlain::Args pa{xin, H, actx};
        EpiStoreArgs ea{ws + bw.S, H, params + mb.bskip, statp(mb.bnskip), rm};
        rc = (H >= 512) ? gemm_plain_pipe<AT>(p, g, xin, H, actx, ea, st, true, p->nt_skip) : -1000;
        if (rc == -1000) rc = gemm_store<AT, ProdPlain>(g, pa, ea, 0, st);
      }
      return rc;
    };
    const bool skip_late = p->skip_late && !ov;
    if (!skip_late) { const int rcs = run_skip(); if (rcs) return rcs; }
    const void* cur = xin;
    BnAct acur = actx;
    for (int j = 0; j < c.n_sub_blocks; ++j) {
      const SubBlockRef& sb = mb.sub[j];
      GemmShape g{M, H, H, wsel<AT>(p, sb.wpw, bw.wpw[j])};
      ProdDw::Args pa{cur, H, acur, params + sb.wdw, params + sb.bdw, c.kernel, T,
                      (p->save_q && training) ? (void*)(ws + bw.Q[j]) : nullptr};
      EpiStoreArgs ea{ws + bw.Y[j], H, params + sb.bpw, statp(sb.bn), rm};
      int rc;
      {
        ProfScope ps(p, TN_PROF_FWD_SUBBLOCK, st);
        if (use_v2) {
          SubFwdV2Args va{(const bf16_t*)cur, acur, params + sb.wdw, params + sb.bdw, (const bf16_t*)(ws + bw.wpw[j].w),
                          params + sb.bpw, (bf16_t*)(ws + bw.Y[j]), statp(sb.bn), M, T, 0,
                          bw.wpw[j].sw ? (const uint4*)(ws + bw.wpw[j].sw) : nullptr,
                          (p->save_q && training) ? (bf16_t*)(ws + bw.Q[j]) : nullptr};
          rc = launch_sub_fwd_v5<3, true, 32>(va, 256, st);
          if (rc == 0) pad_fixup(statp(sb.bn), params + sb.bpw, H);
          if (rc == -1000) rc = gemm_store<AT, ProdDw>(g, pa, ea, c.kernel, st);
        } else if (p->split_dw && p->save_q) {
          // wide models: the depthwise output is produced once by a streaming kernel (it is kept for the weight gradients
          // anyway) and the pointwise GEMM reads it as a plain operand
          // (fp8 weight gradient: the e4m3 depthwise output is KEPT per layer — it is the contraction's second operand)
          uint8_t* q8 = (p->fp8 && sizeof(AT) == 2) ? (uint8_t*)(ws + ((p->fp8_wgrad && training && !bw.Q8.empty()) ? bw.Q8[j] : p->q8)) : nullptr;
          rc = -1000;
          bool q_clean = false;          // the slab kernel stores zeros on padding rows (what the pipelined GEMMs' pad_rows needs)
          if (sizeof(AT) == 2 && p->wide_dw_bwd) {
            DwFwdSlabArgs fa;
            memset(&fa, 0, sizeof(fa));
            fa.X = (const bf16_t*)cur; fa.act = acur; fa.wdw = params + sb.wdw; fa.bdw = params + sb.bdw;
            fa.Q = (bf16_t*)(ws + bw.Q[j]); fa.Q8 = q8; fa.M = M; fa.T = T; fa.C = H;
            // fp8 weight gradient with column maxima on record: the backward will contract the e4m3 copy — the bf16 one (2 of
            // the 3 bytes this pass writes per element) is not stored (a fallback to the generic producer below stores both)
            if (p->fp8_wgrad && p->fp8_hist_valid && training && q8 && !bw.Q8.empty()) { fa.Q = nullptr; p->fwd_q16_skipped = true; }
            if (p->masked && p->skip_pad_tiles && p->n_rowtiles > 0) { fa.rowtiles = (const int*)(ws + p->rowtiles); fa.n_rowtiles = p->n_rowtiles; }
            rc = launch_dw_fwd_slab(fa, c.kernel, st);
            if (rc > 0) return rc;
            q_clean = rc == 0;
          }
          if (rc == -1000)
            rc = launch_dw_fwd<AT>((const AT*)cur, acur, params + sb.wdw, params + sb.bdw, (AT*)(ws + bw.Q[j]), M, T, H, c.kernel, st, q8);
          if (rc) return rc;
          if (q8) {
            // TN_PREC_FP8: e4m3 x e4m3 -> f32 on the fp8 matrix cores, per-output-channel weight scales in the epilogue
            GemmShape g8{M, H, H, ws + bw.w8[j]};
            EpiStoreArgs e8 = ea;
            e8.colscale = (const float*)(ws + bw.w8s[j]);
            // the pipelined LDS-DMA GEMM on the 64-k e4m3 MFMA (tn_pgemm.h, F8): 1.25 PFLOP/s at 76800 x 1024 x 1024 vs 0.96
            rc = -1000;
            if (!p->generic && H % 64 == 0 && H >= 256 && (!p->masked || q_clean)) {
              const bool listed = p->masked && p->n_rowtiles > 0;
              PGemmNtArgs pa8{(const bf16_t*)q8, H, listed ? (const int*)(p->ws + p->rowtiles) : nullptr, listed ? p->n_rowtiles : 0};
              PGemmEpiArgs pe8{(bf16_t*)e8.Y, e8.ldy, e8.bias, e8.stats, e8.colscale, p->masked ? (float)(p->active_rows - p->n_valid) : 0.f};
              rc = launch_pgemm_nt_f8(g8, pa8, pe8, st);
            }
            if (rc == -1000) rc = launch_gemm_fp8<EpiStore>(g8, q8, e8, st);
          } else {
            ProdPlain::Args pq{ws + bw.Q[j], H, identity_act()};
            rc = gemm_plain_pipe<AT>(p, g, ws + bw.Q[j], H, identity_act(), ea, st, q_clean, (p->rw_nt & 1) != 0);
            if (rc == -1000) rc = gemm_store<AT, ProdPlain>(g, pq, ea, 0, st);
          }
        } else {
          rc = gemm_store<AT, ProdDw>(g, pa, ea, c.kernel, st);
        }
      }
      if (rc) return rc;
      cur = ws + bw.Y[j];
      acur = make_act(p, sb.bn, M, training, 1, pd, seed, i * (c.n_sub_blocks + 1) + j);
    }
    if (skip_late) { const int rcs = run_skip(); if (rcs) return rcs; }
    // SE gate + residual combine (reference src/modules.py:173-189, src/models.py:467-472)
    {
      const int CV = H / 8, TG = 512 / CV;
      auto join_skip = [&]() -> int {
        if (skip_on_side) { TN_CHECK_HIP(hipStreamWaitEvent(st, p->ov_events[4 * i + 1], 0)); skip_on_side = false; }
        return 0;
      };
      size_t smem = (size_t)(3 * H + ((Hr + 3) & ~3) + TG * H) * sizeof(float);
      int rc1 = -1000;
      bool combined = false;
      if (sizeof(AT) == 2 && H == V2_C && Hr == 16 && use_v2 && p->se_fused) {
        // squeeze + gate + combine in one launch that reads Y3 once (the utterance's rows stay in registers)
        { int rcj = join_skip(); if (rcj) return rcj; }
        SeCombineV3Args fa;
        memset(&fa, 0, sizeof(fa));
        fa.se.Y = (const bf16_t*)cur; fa.se.act = acur; fa.se.W1 = params + mb.se_w1; fa.se.W2 = params + mb.se_w2;
        fa.se.m_out = (float*)(ws + bw.m); fa.se.h_out = (float*)(ws + bw.h); fa.se.g_out = (float*)(ws + bw.g); fa.se.T = T; fa.se.len = rm.len;
        fa.S = (const bf16_t*)(ws + bw.S); fa.actS = make_act(p, mb.bnskip, M, training, 0, 0.f, seed, 0);
        fa.OUT = (bf16_t*)(ws + bw.OUT);
        if (training && pd > 0.f) {
          fa.drop_thr = (uint32_t)lrintf(pd * 65536.f);
          fa.drop_key = tn_layer_key(seed, (uint32_t)(i * (c.n_sub_blocks + 1) + c.n_sub_blocks));
          fa.inv_keep = 1.f / (1.f - pd);
        } else fa.inv_keep = 1.f;
        fa.key_add = (const uint32_t*)(ws + p->step_state) + 2;
        const int rcf = launch_se_combine_fwd_v3(fa, B, st);
        if (rcf > 0) return rcf;
        combined = rcf == 0;
      }
      if (!combined) {
      if (sizeof(AT) == 2 && H == V2_C && Hr == 16 && use_v2) {
        SeSqueezeV2Args sa;
        memset(&sa, 0, sizeof(sa));
        sa.Y = (const bf16_t*)cur; sa.act = acur; sa.W1 = params + mb.se_w1; sa.W2 = params + mb.se_w2;
        sa.m_out = (float*)(ws + bw.m); sa.h_out = (float*)(ws + bw.h); sa.g_out = (float*)(ws + bw.g); sa.T = T; sa.len = rm.len;
        rc1 = launch_se_squeeze_v2(sa, B, st);
        if (rc1 > 0) return rc1;
      }
      if (rc1 == -1000) {
        float* acc = p->tail_parts > 1 ? (float*)(ws + p->se_acc) : nullptr;
        // activation flags of the last sub-block's output as a template parameter where they are the usual ones
        const int flse = (acur.mode != 0 ? 1 : 0) | (acur.relu ? 2 : 0) | (acur.drop_thr ? 4 : 0);
        auto kse = flse == 7 ? se_squeeze_fc_kernel<AT, 7> : flse == 3 ? se_squeeze_fc_kernel<AT, 3> : se_squeeze_fc_kernel<AT, -1>;
        if (acc && p->se_cnt)     // partial column sums from se_parts workgroups per utterance; the last one to arrive finishes
          hipLaunchKernelGGL(kse, dim3(B, p->se_parts), dim3(512), smem, st, (const AT*)cur, acur, T, H, Hr,
                             params + mb.se_w1, params + mb.se_w2, (float*)(ws + bw.m), (float*)(ws + bw.h), (float*)(ws + bw.g), acc, 3, p->se_parts,
                             (int*)(ws + p->se_cnt));
        else {
        if (acc)     // partial column sums from tail_parts workgroups per utterance, then the two mat-vecs per utterance
          hipLaunchKernelGGL(kse, dim3(B, p->tail_parts), dim3(512), smem, st, (const AT*)cur, acur, T, H, Hr,
                             params + mb.se_w1, params + mb.se_w2, (float*)(ws + bw.m), (float*)(ws + bw.h), (float*)(ws + bw.g), acc, 1, p->tail_parts, (int*)nullptr);
        hipLaunchKernelGGL(kse, dim3(B), dim3(512), smem, st, (const AT*)cur, acur, T, H, Hr,
                           params + mb.se_w1, params + mb.se_w2, (float*)(ws + bw.m), (float*)(ws + bw.h), (float*)(ws + bw.g), acc, acc ? 2 : 0, p->tail_parts, (int*)nullptr);
        }
      }
      { int rcj = join_skip(); if (rcj) return rcj; }      // the combine reads S and the skip BatchNorm's statistics
      BnAct acts = make_act(p, mb.bnskip, M, training, 0, 0.f, seed, 0);
      uint32_t thr = 0, key = 0;
      float ik = 1.f;
      if (training && pd > 0.f) {
        thr = (uint32_t)lrintf(pd * 65536.f);
        key = tn_layer_key(seed, (uint32_t)(i * (c.n_sub_blocks + 1) + c.n_sub_blocks));
        ik = 1.f / (1.f - pd);
      }
      int rc2 = -1000;
      if (sizeof(AT) == 2 && H == V2_C && use_v2) {
        CombineFwdV2Args ca;
        memset(&ca, 0, sizeof(ca));
        ca.S = (const bf16_t*)(ws + bw.S); ca.actS = acts; ca.Y3 = (const bf16_t*)cur; ca.act3 = acur;
        ca.gate = (const float*)(ws + bw.g); ca.OUT = (bf16_t*)(ws + bw.OUT); ca.T = T; ca.parts = 4; ca.len = rm.len;
        ca.drop_thr = thr; ca.drop_key = key; ca.inv_keep = ik;
        ca.key_add = (const uint32_t*)(ws + p->step_state) + 2;
        rc2 = launch_combine_fwd_v2(ca, B, st);
        if (rc2 > 0) return rc2;
      }
      if (rc2 == -1000) {
        const int rpb = 64;
        auto kcomb = acur.rm.len ? combine_fwd_kernel<AT, true> : combine_fwd_kernel<AT, false>;
        // (variable-length batches on plans whose kernels tolerate stale rows in padding-only tiles: only the listed tiles)
        const bool listed = p->masked && p->skip_pad_tiles && p->n_rowtiles > 0 && acur.rm.len;
        hipLaunchKernelGGL(kcomb, dim3(listed ? p->n_rowtiles * (256 / rpb) : (M + rpb - 1) / rpb), dim3(256), (size_t)4 * H * sizeof(float), st,
                           (const AT*)(ws + bw.S), acts, (const AT*)cur, acur, (const float*)(ws + bw.g), (AT*)(ws + bw.OUT), M,
                           T, H, rpb, thr, key, ik, (const uint32_t*)(ws + p->step_state) + 2,
                           listed ? (const int*)(ws + p->rowtiles) : (const int*)nullptr);
      }
      }
    }
    xin = ws + bw.OUT;
    actx = identity_rows(p);
  }
  // ---- epilog 1x1 conv (reference src/models.py:384, :404)
  {
    const bool wide = sizeof(AT) == 2 && use_v2 && H == 256 && D % 256 == 0 && A == 128;
    if (wide && actx.mode == 0 && !actx.relu && !actx.drop_thr) {
      WideOutArgs wa;
      memset(&wa, 0, sizeof(wa));
      wa.X = (const bf16_t*)xin; wa.W = (const bf16_t*)wsel<AT>(p, m->epi_w, p->wepi); wa.bias = params + m->epi_b;
      wa.Wswz = p->wepi.sw ? (const uint4*)(ws + p->wepi.sw) : nullptr;
      wa.Y = (bf16_t*)(ws + p->E); wa.stats = statp(m->epi_bn); wa.M = M; wa.N = D;
      int rc = launch_wide_out_v2<256, 0>(wa, 256, st);
      if (rc) return rc;
      pad_fixup(statp(m->epi_bn), params + m->epi_b, D);
    } else {
      GemmShape g{M, D, H, wsel<AT>(p, m->epi_w, p->wepi)};
      ProdPlain::Args pa{xin, H, actx};
      EpiStoreArgs ea{ws + p->E, D, params + m->epi_b, statp(m->epi_bn), rm};
      int rc = (H >= 512) ? gemm_plain_pipe<AT>(p, g, xin, H, actx, ea, st) : -1000;
      if (rc == -1000) rc = gemm_store<AT, ProdPlain>(g, pa, ea, 0, st);
      if (rc) return rc;
    }
  }
  BnAct acte = make_act(p, m->epi_bn, M, training, 1, 0.f, seed, 0);
  // the attention GEMMs (1536 <-> 128) do not depend on the hidden width: the wide models run them on the same kernels
  const bool attn_v2 = sizeof(AT) == 2 && !p->generic && D % 256 == 0 && A == 128 && (size_t)M * D * 2 < ((size_t)1 << 31) && ((use_v2 && H == 256) || (H >= 512 && !use_v2 && p->wide_dw_bwd));
  if (c.simple_pool) {
    // ---- simple pool (reference src/models.py:497-502): mean over time, then Linear(D, 2D) in f32
    hipLaunchKernelGGL(mean_pool_fwd_kernel<AT>, dim3(B, (D + 511) / 512), dim3(256), 0, st, (const AT*)(ws + p->E), acte, T, D,
                       (float*)(ws + p->mu));
    GemmShape gp{B, 2 * D, D, params + m->pool2_w};
    ProdPlain::Args pap{ws + p->mu, D, identity_act()};
    EpiStoreArgs eap{ws + p->pooled, 2 * D, params + m->pool2_b, nullptr, RowMask{nullptr, 0}};
    int rc = gemm_store<float, ProdPlain>(gp, pap, eap, 0, st);
    if (rc) return rc;
  } else
  // ---- attentive statistics pooling (reference src/models.py:553-584)
  {
    int rc;
    if (attn_v2) {
      WideInArgs wa;
      memset(&wa, 0, sizeof(wa));
      wa.A = (const bf16_t*)(ws + p->E); wa.act = acte; wa.W = (const bf16_t*)wsel<AT>(p, m->asp_win, p->wwin);
      wa.bias = params + m->asp_bin; wa.Y = (bf16_t*)(ws + p->HID); wa.M = M; wa.KW = D;
      rc = launch_wide_in_v2<0>(wa, 256, st);
    } else {
      GemmShape g1{M, A, D, wsel<AT>(p, m->asp_win, p->wwin)};
      ProdPlain::Args pa1{ws + p->E, D, acte};
      EpiStoreArgs ea1{ws + p->HID, A, params + m->asp_bin, nullptr, RowMask{nullptr, 0}};
      rc = gemm_store<AT, ProdPlain, EpiStoreTanh>(g1, pa1, ea1, 0, st);
    }
    if (rc) return rc;
    bool pooled_done = false;
    if (attn_v2 && p->asp_fused) {
      // energies + softmax + statistics in one launch (no energy tensor)
      AspV2Args fa;
      memset(&fa, 0, sizeof(fa));
      fa.HID = (const bf16_t*)(ws + p->HID); fa.W = (const bf16_t*)wsel<AT>(p, m->asp_wout, p->wwout);
      fa.bias = params + m->asp_bout;
      fa.E = (const bf16_t*)(ws + p->E); fa.actE = acte;
      fa.pooled = (float*)(ws + p->pooled); fa.smax = (float*)(ws + p->smax); fa.sinv = (float*)(ws + p->sinv); fa.qv = (float*)(ws + p->qv);
      fa.stats = statp(m->pool_bn); fa.B = B; fa.T = T; fa.D = D; fa.eps = 1e-6f;
      rc = launch_asp_v2<0>(fa, st);
      if (rc) return rc;             // (the plan promised the shape: -1000 here is a bug, not a fallback)
      pooled_done = true;
    } else
    if (attn_v2) {
      WideOutArgs wa;
      memset(&wa, 0, sizeof(wa));
      wa.X = (const bf16_t*)(ws + p->HID); wa.W = (const bf16_t*)wsel<AT>(p, m->asp_wout, p->wwout); wa.bias = params + m->asp_bout;
      wa.Wswz = p->wwout.sw ? (const uint4*)(ws + p->wwout.sw) : nullptr;
      wa.Y = (bf16_t*)(ws + p->EN); wa.M = M; wa.N = D;
      rc = launch_wide_out_v2<128, 0>(wa, 256, st);
    } else {
      GemmShape g2{M, D, A, wsel<AT>(p, m->asp_wout, p->wwout)};
      ProdPlain::Args pa2{ws + p->HID, A, identity_act()};
      EpiStoreArgs ea2{ws + p->EN, D, params + m->asp_bout, nullptr, RowMask{nullptr, 0}};
      rc = gemm_store<AT, ProdPlain>(g2, pa2, ea2, 0, st);
    }
    if (rc) return rc;
    if (pooled_done) {
    } else if (p->tail_parts > 1) {
      // the frames of an utterance over tail_parts workgroups, partial records in the (idle) weight-gradient slabs
      const int P = p->slab_bytes >= (size_t)B * p->tail_parts * 4 * D * sizeof(float) ? p->tail_parts : 1;
      hipLaunchKernelGGL((asp_pool_fwd_kernel<AT, 16, 16>), dim3(B, (D + 127) / 128, P), dim3(256), 0, st, (const AT*)(ws + p->E), acte,
                         (const AT*)(ws + p->EN), T, D, 1e-6f, (float*)(ws + p->pooled), (float*)(ws + p->smax),
                         (float*)(ws + p->sinv), (float*)(ws + p->qv), statp(m->pool_bn), (float*)(ws + p->slabs));
      if (P > 1)
        hipLaunchKernelGGL(asp_pool_merge_kernel, dim3(B, (D + 255) / 256), dim3(256), 0, st, (const float*)(ws + p->slabs), P, D, 1e-6f,
                           (float*)(ws + p->pooled), (float*)(ws + p->smax), (float*)(ws + p->sinv), (float*)(ws + p->qv), statp(m->pool_bn));
    }
    else
      hipLaunchKernelGGL(asp_pool_fwd_kernel<AT>, dim3(B, (D + 511) / 512), dim3(256), 0, st, (const AT*)(ws + p->E), acte,
                         (const AT*)(ws + p->EN), T, D, 1e-6f, (float*)(ws + p->pooled), (float*)(ws + p->smax),
                         (float*)(ws + p->sinv), (float*)(ws + p->qv), statp(m->pool_bn));
  }
  // ---- decoder tail + loss head (reference src/models.py:504-513, src/losses.py)
  {
    BnAct actp = c.simple_pool ? identity_act() : make_act(p, m->pool_bn, B, training, 0, 0.f, seed, 0);
    if ((2 * D) % 1024 == 0 && p->slab_bytes >= (size_t)4 * D * sizeof(float)) {
      float* scsh = (float*)(ws + p->slabs);       // scratch: the weight-gradient slabs are idle in the forward pass
      hipLaunchKernelGGL(bn_scale_shift_kernel, dim3((2 * D + 255) / 256), dim3(256), 0, st, actp, 2 * D, scsh);
      // 4 utterances x 16 outputs per workgroup (round 5): 768 workgroups of 48 KB LDS, three per CU, two passes per wave.  The
      // kernel is a latency chain per workgroup, not L2 traffic: 4 x 64 (192 workgroups, 8 passes per wave) took 37 us, 8 x 32
      // (half the weight re-reads, 96 KB of LDS: one workgroup per CU) 33 us.  LDS > 128 KB (enc_out 3072 would need 96 KB: fits)
      constexpr int TL_NB = 4, TL_ET = 16;
      auto ktl = tail_linear_fwd2_kernel<TL_NB, TL_ET>;
      const size_t tl_smem = (size_t)TL_NB * 2 * D * sizeof(float);
      TN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ktl), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl_smem));
      hipLaunchKernelGGL(ktl, dim3((B + TL_NB - 1) / TL_NB, (c.emb + TL_ET - 1) / TL_ET), dim3(256), tl_smem, st,
                         (const float*)(ws + p->pooled), (const float*)scsh, B, 2 * D, c.emb, params + m->lin_w, params + m->lin_b,
                         (float*)(ws + p->lin), statp(m->lin_bn));
    } else {
      hipLaunchKernelGGL(tail_linear_fwd_kernel, dim3((B + 3) / 4, (c.emb + 63) / 64), dim3(256), (size_t)4 * 2 * D * sizeof(float), st,
                         (const float*)(ws + p->pooled), actp, B, 2 * D, c.emb, params + m->lin_w, params + m->lin_b,
                         (float*)(ws + p->lin), statp(m->lin_bn));
    }
    HeadArgs ha;
    memset(&ha, 0, sizeof(ha));
    ha.lin = (const float*)(ws + p->lin);
    ha.actL = make_act(p, m->lin_bn, B, training, 0, 0.f, seed, 0);
    ha.B = B; ha.E = c.emb; ha.NC = c.n_classes;
    ha.loss_type = speakers ? c.loss_type : TN_LOSS_NONE;
    if (speakers && c.loss_type == TN_LOSS_NONE) return TN_E_STATE;   // "Loss function should not be None in training mode"
    ha.W = m->fc_w >= 0 ? params + m->fc_w : nullptr;
    ha.bias = m->fc_b >= 0 ? params + m->fc_b : nullptr;
    ha.targets = speakers;
    ha.scale = c.scale; ha.has_scale = c.has_scale; ha.m1 = c.m1; ha.m2 = c.m2; ha.m3 = c.m3; ha.eps = c.loss_eps;
    ha.emb = (float*)(ws + p->emb);
    ha.emb_norm = (float*)(ws + p->emb_norm);
    ha.emb_user = emb_out;
    ha.preds = preds ? preds : (int64_t*)(ws + p->preds);
    ha.loss = (float*)(ws + p->loss_acc);
    ha.dlogits = (float*)(ws + p->dlogits);
    ha.dscale = (float*)(ws + p->dscale);
    ha.logits = (float*)(ws + p->logits);
    if (ha.loss_type == TN_LOSS_MARGIN) {
      hipLaunchKernelGGL(row_normalize_kernel, dim3((c.n_classes + 3) / 4), dim3(256), 0, st, params + m->fc_w, c.n_classes, c.emb);
    }
    hipLaunchKernelGGL(head_fwd_kernel, dim3(B), dim3(256), (size_t)(c.emb + std::max(c.n_classes, 1)) * sizeof(float), st, ha);
    if (speakers && loss) TN_CHECK_HIP(hipMemcpyAsync(loss, ws + p->loss_acc, sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  if (training) {
    hipLaunchKernelGGL(bn_running_update_kernel, dim3(2, m->n_bn), dim3(256), 0, st, (const BnUpdateDesc*)(ws + p->bn_table),
                       0.1f, p->nbt, m->n_bn, M, p->masked ? std::max(p->n_valid, 1) : M);
  }
  p->last_training = training;
  p->last_input = spec;
  p->last_has_loss = speakers != nullptr;
  p->last_seed = seed;
  return (int)hipGetLastError();
}

}  // namespace

int plan_forward(tn_plan* p, const float* spec, const int64_t* speakers, int training, uint64_t seed, float* emb_out,
                 int64_t* preds, float* loss, hipStream_t st) {
  if (p->prec == TN_PREC_BF16) return forward_impl<bf16_t>(p, spec, speakers, training, seed, emb_out, preds, loss, st);
  return forward_impl<float>(p, spec, speakers, training, seed, emb_out, preds, loss, st);
}

extern "C" int tn_forward(tn_plan* p, const float* spectrograms, const int64_t* speakers, int32_t training,
                          uint64_t seed, float* embeddings, int64_t* preds, float* loss, void* stream) {
  if (!p || !spectrograms) return TN_E_BADARG;
  if (!p->bound) return TN_E_NOTBOUND;
  if (training && p->B < 2) return TN_E_BADARG;   // BatchNorm1d raises on a batch of 1 in train mode
  p->masked = false;
  return plan_forward(p, spectrograms, speakers, training, seed, embeddings, preds, loss, (hipStream_t)stream);
}

// the valid-frame counts travel as KERNEL ARGUMENTS (copied at launch): an H2D copy from pageable host memory blocks the
// calling thread until the stream has drained (a sleep / wake-up round trip per step on the host), a pinned staging buffer
// would have to outlive the in-flight steps that still read it
struct LensChunk { int v[512]; };
__global__ void lens_write_kernel(LensChunk c, int n, int* __restrict__ dst) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = c.v[i];
}
static int plan_set_lengths(tn_plan* p, const int64_t* lengths_host, int training, hipStream_t st) {
  p->lens_host.resize(p->B);
  long total = 0;
  for (int b = 0; b < p->B; ++b) {
    if (lengths_host[b] < 1 || lengths_host[b] > p->T) return TN_E_BADARG;
    p->lens_host[b] = (int)lengths_host[b];
    total += lengths_host[b];
  }
  if (training && total < 2) return TN_E_BADARG;
  p->n_valid = (int)total;
  p->masked = true;
  for (int b0 = 0; b0 < p->B; b0 += 512) {
    LensChunk c;
    const int n = std::min(512, p->B - b0);
    memcpy(c.v, p->lens_host.data() + b0, sizeof(int) * (size_t)n);
    hipLaunchKernelGGL(lens_write_kernel, dim3(1), dim3(256), 0, st, c, n, (int*)(p->ws + p->lens) + b0);
  }
  p->n_rowtiles = 0;
  p->active_rows = p->M;
  if (p->skip_pad_tiles) {
    // 256-row tiles with at least one valid frame: utterance b covers rows [b T, b T + len_b)
    const int tiles = (p->M + 255) / 256;
    std::vector<int> act;
    act.reserve(tiles);
    int next = 0;                                  // first tile not yet listed
    for (int b = 0; b < p->B; ++b) {
      const long r0 = (long)b * p->T, r1 = r0 + p->lens_host[b] - 1;
      for (int t = std::max<int>(next, (int)(r0 / 256)); t <= (int)(r1 / 256); ++t) act.push_back(t);
      next = std::max<int>(next, (int)(r1 / 256) + 1);
    }
    if ((int)act.size() < tiles) {               // (nothing to skip otherwise: the kernels then run without a list)
      p->n_rowtiles = (int)act.size();
      long rows = 0;
      for (int t : act) rows += std::min(256, p->M - t * 256);
      p->active_rows = (int)rows;
      for (size_t i0 = 0; i0 < act.size(); i0 += 512) {
        LensChunk c;
        const int n = (int)std::min<size_t>(512, act.size() - i0);
        memcpy(c.v, act.data() + i0, sizeof(int) * (size_t)n);
        hipLaunchKernelGGL(lens_write_kernel, dim3(1), dim3(256), 0, st, c, n, (int*)(p->ws + p->rowtiles) + i0);
      }
    }
  }
  return (int)hipGetLastError();
}

extern "C" int tn_forward_masked(tn_plan* p, const float* spectrograms, const int64_t* lengths_host, const int64_t* speakers,
                                 int32_t training, uint64_t seed, float* embeddings, int64_t* preds, float* loss, void* stream) {
  if (!lengths_host) return tn_forward(p, spectrograms, speakers, training, seed, embeddings, preds, loss, stream);
  if (!p || !spectrograms) return TN_E_BADARG;
  if (!p->bound) return TN_E_NOTBOUND;
  if (training && p->B < 2) return TN_E_BADARG;
  { int rc = plan_set_lengths(p, lengths_host, training, (hipStream_t)stream); if (rc) return rc; }
  return plan_forward(p, spectrograms, speakers, training, seed, embeddings, preds, loss, (hipStream_t)stream);
}

extern "C" void* tn_plan_prolog_input(tn_plan* p) {
  if (!p || !p->bound || !p->prolog_taps || p->prec != TN_PREC_BF16) return nullptr;
  return p->ws + p->x0;
}

extern "C" int tn_forward_prepacked(tn_plan* p, const int64_t* lengths_host, const int64_t* speakers, int32_t training, uint64_t seed,
                                    float* embeddings, int64_t* preds, float* loss, void* stream) {
  if (!p) return TN_E_BADARG;
  if (!p->bound) return TN_E_NOTBOUND;
  if (!p->prolog_taps || p->prec != TN_PREC_BF16) return TN_E_UNSUPPORTED;
  if (training && p->B < 2) return TN_E_BADARG;
  p->masked = false;
  if (lengths_host) {
    int rc = plan_set_lengths(p, lengths_host, training, (hipStream_t)stream);
    if (rc) return rc;
  }
  return plan_forward(p, nullptr, speakers, training, seed, embeddings, preds, loss, (hipStream_t)stream);
}

extern "C" int tn_backward(tn_plan* p, float grad_scale, const float* grad_scale_dev, const float* grad_embeddings,
                           float* grad_input, void* stream) {
  if (!p) return TN_E_BADARG;
  if (!p->bound || !p->grads) return TN_E_NOTBOUND;
  if (p->last_training < 0) return TN_E_STATE;
  return plan_backward(p, grad_scale, grad_scale_dev, grad_embeddings, grad_input, (hipStream_t)stream);
}

// =============================================================================================
// Stand-alone loss heads: MetricLearningLoss.forward(inputs, targets) (reference src/losses.py:32-44, :77-132)
// =============================================================================================
extern "C" size_t tn_head_save_floats(int32_t batch, int32_t emb, int32_t n_classes) {
  return (size_t)batch * ((size_t)n_classes + 1 + 2 * (size_t)emb);
}
extern "C" int tn_head_forward(int32_t loss_type, int32_t batch, int32_t emb, int32_t n_classes, const float* inputs,
                               const int64_t* targets, float* fc_weight, const float* fc_bias, int32_t has_scale, float scale,
                               float m1, float m2, float m3, float eps, float* normalized, int64_t* preds, float* loss,
                               float* save, void* stream) {
  if (!inputs || !targets || !fc_weight || !normalized || !preds || !loss || !save) return TN_E_BADARG;
  if (batch <= 0 || emb <= 0 || n_classes <= 0) return TN_E_BADARG;
  if (loss_type != TN_LOSS_CE && loss_type != TN_LOSS_MARGIN) return TN_E_BADARG;
  if (loss_type == TN_LOSS_CE && !fc_bias) return TN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  HeadArgs ha;
  memset(&ha, 0, sizeof(ha));
  ha.lin = inputs; ha.actL = identity_act();
  ha.B = batch; ha.E = emb; ha.NC = n_classes; ha.loss_type = loss_type;
  ha.W = fc_weight; ha.bias = fc_bias; ha.targets = targets;
  ha.scale = scale; ha.has_scale = has_scale; ha.m1 = m1; ha.m2 = m2; ha.m3 = m3; ha.eps = eps;
  ha.dlogits = save;
  ha.dscale = save + (size_t)batch * n_classes;
  ha.emb = ha.dscale + batch;
  ha.emb_norm = ha.emb + (size_t)batch * emb;
  ha.emb_user = normalized; ha.preds = preds; ha.loss = loss; ha.logits = nullptr;
  TN_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(float), st));
  if (loss_type == TN_LOSS_MARGIN)
    hipLaunchKernelGGL(row_normalize_kernel, dim3((n_classes + 3) / 4), dim3(256), 0, st, fc_weight, n_classes, emb);
  hipLaunchKernelGGL(head_fwd_kernel, dim3(batch), dim3(256), (size_t)(emb + n_classes) * sizeof(float), st, ha);
  return (int)hipGetLastError();
}

// =============================================================================================
// Adam (torch.optim.Adam semantics, reference src/train.py:131-135)
// =============================================================================================
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                            float bc1, float bc2_sqrt, float gmult) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i] * gmult;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

extern "C" int tn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                            float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_mult,
                            void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || n <= 0 || step < 1) return TN_E_BADARG;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = sqrtf(1.f - powf(beta2, (float)step));
  int blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr,
                     beta1, beta2, eps, weight_decay, bc1, bc2, grad_mult);
  return (int)hipGetLastError();
}

// =============================================================================================
// device-resident step state: lets forward + backward + Adam replay as ONE hipGraph (nothing that changes from
// step to step is a kernel argument any more: the dropout word and the Adam step live in the workspace)
// =============================================================================================
__global__ void step_tick_kernel(uint64_t* st, uint64_t set_to, int do_set) {
  const uint64_t s = do_set ? set_to : st[0] + 1;
  st[0] = s;
  reinterpret_cast<uint32_t*>(st)[2] = s ? tn_mix32((uint32_t)s * 0x9E3779B9u + (uint32_t)(s >> 32) + 0x85ebca6bu) : 0u;
}
__global__ void mark_host_kernel(uint32_t* flag, uint32_t value) {
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
extern "C" int tn_mark_host(uint32_t* host_flag, uint32_t value, void* stream) {
  if (!host_flag) return TN_E_BADARG;
  hipLaunchKernelGGL(mark_host_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, host_flag, value);
  return (int)hipGetLastError();
}

extern "C" int tn_plan_step_tick(tn_plan* p, void* stream) {
  if (!p || !p->bound) return TN_E_STATE;
  hipLaunchKernelGGL(step_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (uint64_t*)(p->ws + p->step_state), (uint64_t)0, 0);
  return (int)hipGetLastError();
}
extern "C" int tn_plan_step_set(tn_plan* p, int64_t step, void* stream) {
  if (!p || !p->bound || step < 0) return TN_E_STATE;
  hipLaunchKernelGGL(step_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (uint64_t*)(p->ws + p->step_state), (uint64_t)step, 1);
  return (int)hipGetLastError();
}
__global__ void set_lr_kernel(float* lr_word, float lr) { *lr_word = lr; }
extern "C" int tn_plan_set_lr(tn_plan* p, float lr, void* stream) {
  if (!p || !p->bound || !(lr >= 0.f)) return TN_E_STATE;
  hipLaunchKernelGGL(set_lr_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (float*)(p->ws + p->step_state) + 3, lr);
  return (int)hipGetLastError();
}
__global__ void adam_dev_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                     int64_t n, float lr, float b1, float b2, float eps, float wd, const uint64_t* __restrict__ st,
                                     float gmul) {
  const float step = (float)st[0];
  if (lr < 0.f) lr = reinterpret_cast<const float*>(st)[3];   // schedulable: the plan's device lr word (tn_plan_set_lr)
  const float bc1 = 1.f - powf(b1, step), bc2 = sqrtf(1.f - powf(b2, step));
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i] * gmul;
    if (wd != 0.f) gi += wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2 + eps);
  }
}
extern "C" int tn_adam_step_plan(tn_plan* p, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, float grad_mult, void* stream) {
  if (!p || !p->bound || !params || !grads || !exp_avg || !exp_avg_sq || n <= 0) return TN_E_BADARG;
  int blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(adam_dev_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr,
                     beta1, beta2, eps, weight_decay, (const uint64_t*)(p->ws + p->step_state), grad_mult);
  return (int)hipGetLastError();
}

// =============================================================================================
// debug fetch (tests): internal rows-x-channels tensors -> float32 reference layout
// =============================================================================================
template <typename AT>
__global__ void fetch_bct_kernel(const AT* __restrict__ src, BnAct act, int M, int T, int C, float* __restrict__ dst) {
  const size_t n = (size_t)M * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / C), c = (int)(i % C);
    float v = Elem<AT>::to_f(src[i]);
    if (act.mode != 0) { float sc, sh; bn_scale_shift(act, C, c, sc, sh); v = v * sc + sh; }
    if (act.relu) v = fmaxf(v, 0.f);
    if (act.drop_thr && !tn_keep_elem((uint32_t)i, tn_act_key(act), act.drop_thr)) v = 0.f;   // (1/(1-p) is in sc, sh)
    const int b = row / T, t = row % T;
    dst[((size_t)b * C + c) * T + t] = v;
  }
}

extern "C" int tn_debug_fetch(tn_plan* p, const char* what, float* dst, int64_t dst_floats, void* stream) {
  if (!p || !what || !dst || !p->bound || p->last_training < 0) return TN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const tn_model* m = p->model;
  const tn_config& c = m->cfg;
  std::string w(what);
  auto copyf = [&](size_t off, int64_t n) -> int {
    if (dst_floats < n) return TN_E_BADARG;
    TN_CHECK_HIP(hipMemcpyAsync(dst, p->ws + off, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
  };
  auto fetch = [&](size_t off, const BnAct& act, int C) -> int {
    if (dst_floats < (int64_t)p->M * C) return TN_E_BADARG;
    if (p->prec == TN_PREC_BF16)
      hipLaunchKernelGGL(fetch_bct_kernel<bf16_t>, dim3(1024), dim3(256), 0, st, (const bf16_t*)(p->ws + off), act, p->M, p->T, C, dst);
    else
      hipLaunchKernelGGL(fetch_bct_kernel<float>, dim3(1024), dim3(256), 0, st, (const float*)(p->ws + off), act, p->M, p->T, C, dst);
    return (int)hipGetLastError();
  };
  if (w == "logits") return copyf(p->logits, (int64_t)p->B * c.n_classes);
  if (w == "embeddings_raw") return copyf(p->emb, (int64_t)p->B * c.emb);
  if (w == "pooled") return copyf(p->pooled, (int64_t)p->B * 2 * c.enc_out);
  if (w == "prolog_out") return fetch(p->Y0, make_act(p, m->prolog_bn, p->M, p->last_training, 1, 0.f, 0, 0), c.hidden);
  if (w == "epilog_out") return fetch(p->E, make_act(p, m->epi_bn, p->M, p->last_training, 1, 0.f, 0, 0), c.enc_out);
  // after tn_backward: the two encoder-output-sized gradient tensors of the decoder side (the attention energies' gradient as
  // the pooling backward wrote it; the gradient wrt the epilog BatchNorm output with the attention data gradient added) — what
  // an element-wise comparison of two pooling paths needs (a corrupted 4-byte piece does not move a cosine)
  if (w == "d_energies") return fetch(p->dE, identity_act(), c.enc_out);
  if (w == "d_epilog_bn") return fetch(p->dEbn, identity_act(), c.enc_out);
  if (w.rfind("block_out:", 0) == 0) {
    int i = atoi(w.c_str() + 10);
    if (i < 0 || i >= c.n_mega_blocks) return TN_E_BADARG;
    return fetch(p->blk[i].OUT, identity_act(), c.hidden);
  }
  if (w.rfind("se_gate:", 0) == 0) {
    int i = atoi(w.c_str() + 8);
    if (i < 0 || i >= c.n_mega_blocks) return TN_E_BADARG;
    return copyf(p->blk[i].g, (int64_t)p->B * c.hidden);
  }
  return TN_E_BADARG;
}
