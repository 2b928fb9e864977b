/* titanet_amd.h — C ABI of libtitanet_amd.so: the MI355X (gfx950 / CDNA4) hot path of
 * Wadaboa/titanet — TitaNet encoder (prolog, depthwise-separable MegaBlocks with SE, epilog),
 * attentive-statistics-pooling decoder and the CE / angular-margin loss heads, forward AND
 * backward, as hand-written HIP kernels.
 *
 * The reference has no FFI: its boundary for this path is the nn.Module surface
 *   TitaNet.forward(spectrograms, speakers=None)            reference src/models.py:318-339
 *   TitaNet.get_titanet(...) / TitaNet.__init__(...)        reference src/models.py:175-219, :262-316
 *   LOSSES[name](embedding_size, n_classes, ...).forward    reference src/losses.py:22-44, :47-132
 *   model.state_dict() key names / shapes                   (SURVEY.md §8b listing)
 * Every entry point below names the reference call it stands behind.  All pointers are plain
 * device pointers (HIP), no framework types.  Every function returns 0 on success, a positive
 * hipError_t, or a negative TN_E_* code; none of them synchronises the device or throws.
 *
 * Memory model: the caller owns four flat device buffers and binds them to a plan once:
 *   params  float[tn_model_param_floats]   all learnable tensors, state_dict order
 *   grads   float[tn_model_param_floats]   same layout (d loss / d param)
 *   bnbuf   float[tn_model_buffer_floats]  BatchNorm running_mean / running_var
 *   nbt     int64[tn_model_num_bn]         BatchNorm num_batches_tracked
 *   workspace  char[tn_plan_workspace_bytes] activations saved for backward, statistics, scratch
 * tn_model_tensor_info() maps every reference state_dict key to (buffer kind, offset, shape).
 */
#ifndef TITANET_AMD_H
#define TITANET_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TN_E_BADARG (-1)
#define TN_E_UNSUPPORTED (-2)
#define TN_E_NOTBOUND (-3)
#define TN_E_STATE (-4)

#define TN_PREC_FP32 0 /* f32 storage, v_mfma_f32_32x32x2_f32: the 1e-3 parity path */
#define TN_PREC_BF16 1 /* bf16 activations/weights, v_mfma_f32_32x32x16_bf16, f32 accumulate/statistics */
#define TN_PREC_FP8 2  /* BASELINE.json configs[4] (TitaNet-L): the pointwise (1x1) convs of the mega blocks — reference
                          src/modules.py:76-78, 91 % of the model's FLOPs — on the fp8 matrix cores
                          (v_mfma_scale_f32_32x32x64_f8f6f4, OCP e4m3 operands, f32 accumulation):
                            forward   the sub-blocks' GEMMs: the depthwise output as produced, weights with one scale per
                                      output channel (unit block scales);
                            backward  (hidden 512 / 1024) the data gradients dS * W of the sub-blocks and of the skip
                                      connections: dS rows as e4m3 with one power-of-two scale per row, passed as the MFMA's
                                      own block-scale operand; W^T rows with one scale per input channel;
                                      the WEIGHT gradients dS^T * Q of the sub-blocks: dS as e4m3 with one power-of-two scale per
                                      COLUMN (delayed scaling: the column maxima of the plan's previous backward; its first
                                      backward runs this product in bf16 and records them), the kept e4m3 depthwise outputs.
                          Storage, statistics, the skip connections' weight gradients, the depthwise / SE arithmetic and the
                          1536-wide decoder GEMMs are the bf16 plan's. */
#define TN_PREC_FP8_FWD 3 /* TN_PREC_FP8 with the whole backward pass in bf16 (forward GEMMs only on the fp8 matrix cores) */

#define TN_LOSS_NONE 0
#define TN_LOSS_CE 1     /* reference src/losses.py:22-44  (fc has a bias)                     */
#define TN_LOSS_MARGIN 2 /* reference src/losses.py:47-132 (ArcFace/CosFace/SphereFace; no bias) */

/* Mirrors TitaNet.__init__ (reference src/models.py:175-192) + the loss constructor arguments
 * (reference src/losses.py:56-75). */
typedef struct tn_config {
  int32_t n_mels;        /* 80 */
  int32_t n_mega_blocks; /* 17 (parameters.yml:54) */
  int32_t n_sub_blocks;  /* 3 */
  int32_t hidden;        /* encoder_hidden_size: 256 / 512 / 1024 */
  int32_t enc_out;       /* encoder_output_size: 1536 */
  int32_t emb;           /* embedding_size: 192 */
  int32_t kernel;        /* mega_block_kernel_size: 3 / 7 / 11 */
  int32_t prolog_kernel; /* 3 */
  int32_t epilog_kernel; /* 1 (only 1 is supported) */
  int32_t attn_hidden;   /* attention_hidden_size: 128 */
  int32_t se_reduction;  /* 16 */
  int32_t loss_type;     /* TN_LOSS_* */
  int32_t n_classes;     /* fc rows (0 when loss_type == TN_LOSS_NONE) */
  int32_t has_scale;     /* margin loss: 0 => scale = ||x|| (reference scale=None) */
  float dropout;         /* p of every Dropout / F.dropout in the mega blocks */
  float scale, m1, m2, m3, loss_eps; /* margin loss: s, m1, m2, m3, eps (1e-6) */
  int32_t simple_pool;   /* Decoder(simple_pool=True), reference src/models.py:497-502: mean over time -> Linear(D, 2D)
                            instead of attentive statistics pooling + BatchNorm (state_dict keys decoder.pool.2.*) */
} tn_config;

typedef struct tn_model tn_model; /* parameter layout for one architecture */
typedef struct tn_plan tn_plan;   /* execution plan for one (batch, frames, precision) */

/* kind of a state_dict tensor */
#define TN_KIND_PARAM 0  /* offset in floats into params / grads */
#define TN_KIND_BUFFER 1 /* offset in floats into bnbuf */
#define TN_KIND_NBT 2    /* offset in int64 into nbt */

/* ---- model layout (TitaNet.__init__ / state_dict; reference src/models.py:193-219) ---------- */
int tn_model_create(const tn_config* cfg, tn_model** out);
void tn_model_destroy(tn_model* m);
int64_t tn_model_param_floats(const tn_model* m);
int64_t tn_model_buffer_floats(const tn_model* m);
int32_t tn_model_num_bn(const tn_model* m);
int32_t tn_model_num_tensors(const tn_model* m);
/* name: >=160 bytes; shape: 4 entries.  Tensors come in the reference's state_dict order. */
int tn_model_tensor_info(const tn_model* m, int32_t index, char* name, int32_t* kind, int64_t* offset,
                         int64_t* numel, int32_t* ndim, int64_t* shape);

/* ---- plan ----------------------------------------------------------------------------------- */
int tn_plan_create(const tn_model* m, int32_t batch, int32_t frames, int32_t precision, tn_plan** out);
void tn_plan_destroy(tn_plan* p);
size_t tn_plan_workspace_bytes(const tn_plan* p);
/* Bind the caller-owned device buffers (grads may be NULL for inference-only use).  Uploads the
 * small descriptor tables; call outside stream capture. */
int tn_plan_bind(tn_plan* p, float* params, float* grads, float* bnbuf, int64_t* nbt, void* workspace,
                 size_t workspace_bytes, void* stream);

/* ---- TitaNet.forward (reference src/models.py:318-339) --------------------------------------
 * spectrograms: float32 [batch][n_mels][frames] (the reference's [B, M, T] layout).
 * speakers:     int64 [batch] or NULL (inference: loss head skipped).
 * training != 0: BatchNorm uses batch statistics and updates running stats, dropout active
 *                (model.train()); activations are kept in the workspace for tn_backward.
 * embeddings:   float32 [batch][emb]  L2-normalised embeddings (what the reference returns first).
 * preds: int64 [batch], loss: float32 scalar (device) — written only when speakers != NULL.
 * seed: dropout stream key for this step (counter-based, regenerated in backward). */
int tn_forward(tn_plan* p, const float* spectrograms, const int64_t* speakers, int32_t training, uint64_t seed,
               float* embeddings, int64_t* preds, float* loss, void* stream);

/* ---- variable-length batches (new: SURVEY.md 8f3 / BASELINE.json configs[3]; the reference zero-pads to the batch maximum
 * and ignores the lengths, src/datasets.py:63-73, src/learn.py:88) ---------------------------------------------------------
 * lengths_host: HOST int64 [batch] (what collate_fn returns), 1 <= lengths[b] <= frames, or NULL (== tn_forward).
 * Frames t >= lengths[b] of utterance b are padding: every layer sees zeros there (the zero padding an un-padded utterance
 * would get at its end), BatchNorm statistics, the SE mean and the attentive softmax run over the valid frames only, and
 * padded frames receive no gradient — a padded batch gives every utterance the embedding it has on its own (eval mode
 * exactly; train mode with the BatchNorm statistics pooled over all valid frames of the batch).  With every length equal
 * to `frames` the results equal tn_forward's.  Runs the generic kernel templates; the following tn_backward uses the same
 * lengths.  Not for stream capture (host lengths). */
int tn_forward_masked(tn_plan* p, const float* spectrograms, const int64_t* lengths_host, const int64_t* speakers,
                      int32_t training, uint64_t seed, float* embeddings, int64_t* preds, float* loss, void* stream);
/* The bf16 plans read the input as a packed bf16 [batch * frames][n_mels] operand in their workspace (device pointer, NULL
 * for fp32 plans / unbound plans).  A producer on the same stream (tn_mel_forward_batch_packed) may fill it directly;
 * tn_forward_prepacked == tn_forward / tn_forward_masked (lengths_host NULL / given) on that operand, without a spectrogram
 * argument (frames beyond an utterance's length must be zero there, as the mel front end writes them; d loss / d
 * spectrograms is not available on this path). */
void* tn_plan_prolog_input(tn_plan* p);
int tn_forward_prepacked(tn_plan* p, const int64_t* lengths_host, const int64_t* speakers, int32_t training, uint64_t seed,
                         float* embeddings, int64_t* preds, float* loss, void* stream);

/* ---- loss.backward() through the module (reference src/learn.py:117) -------------------------
 * Must follow a tn_forward(training or eval) on the same plan.  Writes d loss / d params into
 * the bound grads buffer (overwrite, not accumulate), scaled by grad_scale (d out / d loss, 1.0
 * for loss.backward()) times *grad_scale_dev when that device scalar is given (autograd hands
 * the upstream gradient over as a device tensor; reading it on the host would force a sync).
 * grad_embeddings (float32 [batch][emb], may be NULL) is an additional
 * upstream gradient on the returned normalised embeddings.  grad_input (float32
 * [batch][n_mels][frames], may be NULL) receives d / d spectrograms (utils.chart_dependencies,
 * reference src/utils.py:451-468). */
int tn_backward(tn_plan* p, float grad_scale, const float* grad_scale_dev, const float* grad_embeddings,
                float* grad_input, void* stream);

/* ---- gradient buckets: overlapping the data-parallel all-reduce with backward (new; the reference is single-device) ----
 * The flat gradient buffer is in state_dict order, and backward finishes it from the END: loss head, decoder, epilog,
 * then the mega blocks from the last one down, the prolog last.  tn_plan_set_grad_groups(G > 1) (before
 * tn_plan_workspace_bytes / tn_plan_bind) splits it into 1 + G contiguous buckets in COMPLETION order — bucket 0 = epilog
 * conv .. loss head, buckets 1..G = groups of mega blocks from the top (the prolog rides with block 0's group) — and makes
 * tn_backward finalise each bucket as soon as its layers are done (its own balanced weight-gradient launch) and record an
 * event.  A trainer enqueues, per bucket i, tn_plan_wait_grad_bucket(i, comm_stream) + ncclAllReduce on comm_stream: the
 * collective of bucket i then runs under the backward of the blocks below it (titanet_amd/trainer.py).  G = 1 (default):
 * one bucket, every deferred weight gradient in one launch at the end of backward (fastest on a single GPU). */
int tn_plan_set_grad_groups(tn_plan* p, int32_t groups); /* after tn_plan_bind: only if the new layout fits the bound workspace */
int32_t tn_plan_num_grad_buckets(const tn_plan* p);
int tn_plan_grad_bucket(const tn_plan* p, int32_t i, int64_t* begin_float, int64_t* end_float);
/* hipStreamWaitEvent(stream, event of bucket i of the last tn_backward) */
int tn_plan_wait_grad_bucket(tn_plan* p, int32_t i, void* stream);

/* ---- optim.Adam step on the flat buffers (reference src/train.py:131-135: Adam, lr 1e-3) -----
 * grads are multiplied by grad_mult first (1/world_size after a sum all-reduce). */
int tn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                 float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_mult,
                 void* stream);

/* ---- device-resident step state: one hipGraph per training step (new; the reference dispatches ~400 eager ops per step,
 * reference src/learn.py:88-135) -----------------------------------------------------------------------------------------
 * Everything that changes from step to step and used to be a kernel argument — the dropout stream and the Adam step
 * count — can instead live in the plan's workspace: tn_plan_step_tick advances a device counter and derives the
 * per-step dropout word from it (added to every layer's key(seed, layer); the word is 0 until the first tick, so plain
 * tn_forward calls keep the documented key(seed, layer) streams), and tn_adam_step_plan reads its bias-correction step
 * from the same counter.  A capture of { tn_plan_step_tick; tn_forward(seed fixed); tn_backward; tn_adam_step_plan } on
 * one stream is then replayable as a single hipGraph with no per-step host arguments (titanet_amd/trainer.py). */
int tn_plan_step_tick(tn_plan* p, void* stream);

/* ---- stream progress visible to the host WITHOUT a runtime call (new; the reference's training loop synchronises every
 * step through `loss.item()`, reference src/learn.py:110) -------------------------------------------------------------------
 * Enqueues a one-thread kernel that stores `value` (system scope, release) to `host_flag`, a 4-byte word of PINNED host
 * memory: everything enqueued on `stream` before the call has completed when the host reads `value` there.  The trainer
 * bounds the steps in flight with it and the mel front end recycles its pinned staging slots with it: waiting on HIP
 * events from a host that is several steps ahead of the GPU (hipEventQuery / hipEventSynchronize) left the GPU idle for
 * 50-80 ms at a time in the middle of a step (BASELINE configs[3] leg, round 4), polling a plain word does not. */
int tn_mark_host(uint32_t* host_flag, uint32_t value, void* stream);
int tn_plan_step_set(tn_plan* p, int64_t step, void* stream);   /* step 0 = word 0 (the ungraphed convention) */
int tn_adam_step_plan(tn_plan* p, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                      float beta1, float beta2, float eps, float weight_decay, float grad_mult, void* stream);
/* Schedulable learning rate for a captured step (reference src/learn.py:257-258 steps a CosineAnnealingLR): lr < 0 in
 * tn_adam_step_plan means "read the plan's device lr word", which tn_plan_set_lr writes (a one-thread kernel enqueued
 * OUTSIDE the captured graph, before each replay). */
int tn_plan_set_lr(tn_plan* p, float lr, void* stream);

/* ---- stand-alone loss heads: MetricLearningLoss.forward(inputs, targets) -> (normalised inputs, preds, loss)
 * (reference src/losses.py:32-44 CELoss, :77-132 AngularMarginLoss) for callers that use a loss object outside
 * TitaNet.forward.  loss_type: TN_LOSS_CE (fc_bias required) or TN_LOSS_MARGIN (fc_weight is row-normalised IN PLACE, as
 * the reference does on .data, src/losses.py:86).  inputs: float32 [batch][emb]; targets int64 [batch];
 * normalized: float32 [batch][emb]; preds int64 [batch]; loss float32 scalar (device).  save: tn_head_save_floats()
 * floats of scratch that tn_head_backward reads (d logits, d scale, embeddings).  An out-of-range target poisons the
 * loss with NaN (F.cross_entropy would raise; a kernel cannot). */
size_t tn_head_save_floats(int32_t batch, int32_t emb, int32_t n_classes);
int tn_head_forward(int32_t loss_type, int32_t batch, int32_t emb, int32_t n_classes, const float* inputs,
                    const int64_t* targets, float* fc_weight, const float* fc_bias, int32_t has_scale, float scale, float m1,
                    float m2, float m3, float eps, float* normalized, int64_t* preds, float* loss, float* save, void* stream);
/* grad_scale * (*grad_loss_dev if given) = d out / d loss; grad_normalized (may be NULL) = upstream gradient on the
 * returned normalised inputs.  Writes (overwrites) grad_inputs [batch][emb], grad_weight [n_classes][emb], grad_bias
 * [n_classes] (CE; may be NULL). */
int tn_head_backward(int32_t loss_type, int32_t batch, int32_t emb, int32_t n_classes, const float* fc_weight,
                     const float* save, float grad_scale, const float* grad_loss_dev, const float* grad_normalized,
                     float* grad_inputs, float* grad_weight, float* grad_bias, void* stream);

/* ---- mel front end: MelSpectrogram.__call__ (reference src/transforms.py:158-203) -----------------------
 * Spectrogram(n_fft, win_length, hop_length, power=None) -> |.|^2 -> MelScale(n_mels, sample_rate) ->
 * AmplitudeToDB() -> F.normalize(dim=1) -> SpecAugment frequency/time masks, for a batch of equal-length
 * waveforms.  waves: float32 [batch][n_samples]; out: float32 [batch][n_mels][frames],
 * frames = 1 + n_samples / hop_length (center=True), i.e. exactly the tensor tn_forward consumes.
 * masks: int32 [batch][4] = {f_start, f_end, t_start, t_end} (zeros = no mask) or NULL; the mask draws
 * (torchaudio.functional.mask_along_axis) stay on the host.  Time stretch, several masks per axis and ragged batches:
 * tn_mel_forward_batch below. */
typedef struct tn_mel tn_mel;
int tn_mel_create(int32_t sample_rate, int32_t n_fft, int32_t win_length, int32_t hop_length, int32_t n_mels, tn_mel** out);
void tn_mel_destroy(tn_mel* m);
int64_t tn_mel_num_frames(const tn_mel* m, int64_t n_samples);
int tn_mel_forward(tn_mel* m, const float* waves, int32_t batch, int64_t n_samples, const int32_t* masks, float* out,
                   void* stream);
/* Batched form for variable-length utterances with the full SpecAugment branch (reference src/transforms.py:168-201):
 * waves float32 [batch][n_samples_max] (utterance b uses its first lengths[b] samples: its own frame count
 * 1 + lengths[b] / hop and reflect padding); lengths int64 [batch] DEVICE or NULL (= n_samples_max);
 * rates float64 [batch] DEVICE or NULL: time-stretch rate of torchaudio.transforms.TimeStretch — the reference takes
 * .abs().pow(2) of the phase vocoder's output, so the magnitude interpolation of the vocoder is what is computed;
 * utterance b then has ceil(frames / rate) frames; freq_mask uint8 [batch][n_mels], time_mask uint8 [batch][frames_out]
 * DEVICE or NULL: non-zero = masked (the union of any number of mask_along_axis intervals, drawn on the host);
 * out float32 [batch][n_mels][frames_out], zero beyond an utterance's last frame (the collate_fn layout). */
int tn_mel_forward_batch(tn_mel* m, const float* waves, int32_t batch, int64_t n_samples_max, const int64_t* lengths,
                         const double* rates, const uint8_t* freq_mask, const uint8_t* time_mask, int32_t frames_out, float* out,
                         void* stream);
/* The front end FUSED into the prolog's producer (BASELINE.json configs[3]; replaces reference src/transforms.py:158-203 +
 * the spectrograms.to(device) of src/learn.py:95 + the first operand of the prolog conv, src/models.py:370): the same
 * arithmetic, but the result is written as the prolog conv's packed operand — bf16 [batch * frames_out][n_mels], row =
 * b * frames_out + frame — straight into a bf16 plan's input buffer (tn_plan_prolog_input), so no float32
 * [batch][n_mels][frames] tensor and no packing pass exist; tn_forward_prepacked then runs the network on it.
 * out_or_null: optionally ALSO the float32 [batch][n_mels][frames_out] tensor (tests). */
int tn_mel_forward_batch_packed(tn_mel* m, const float* waves, int32_t batch, int64_t n_samples_max, const int64_t* lengths,
                                const double* rates, const uint8_t* freq_mask, const uint8_t* time_mask, int32_t frames_out,
                                void* packed_bf16, float* out_or_null, void* stream);

/* ---- per-kernel timing with HIP events on the launch stream (roofline measurement) ----------------
 * Kernel classes: the heavy kernels of one mega-block sub-block (there are n_mega_blocks*n_sub_blocks
 * launches of each per step). */
#define TN_PROF_NONE 0
#define TN_PROF_FWD_SUBBLOCK 1 /* act-on-load + depthwise stencil + pointwise MFMA GEMM + BN statistics */
#define TN_PROF_BWD_WGRAD 2    /* pointwise weight gradient (TN MFMA GEMM, recomputes the depthwise output) */
#define TN_PROF_BWD_DGRAD 3    /* BN-backward-on-load + pointwise data-gradient MFMA GEMM */
#define TN_PROF_BWD_DW 4       /* depthwise backward stencil + activation backward + BN backward sums; on the 256-wide bf16
                                  path fused with the sub-block's pointwise data gradient (dgrad_dw_v6), class 3 is then the
                                  skip-connection data gradient only */
/* Start bracketing every launch of `kernel_class` with hipEvents (TN_PROF_NONE stops). */
int tn_profile_begin(tn_plan* p, int32_t kernel_class);
/* Bracket only every `every_n`-th launch of the class (default 1): two event records per launch cost ~2.5 us of stream time,
 * which a class with ~50 launches per step would otherwise add to the step it is measuring. */
int tn_profile_sample(tn_plan* p, int32_t every_n);
/* Waits for the recorded events; returns the summed kernel time and launch count since begin. */
int tn_profile_read(tn_plan* p, double* total_ms, int64_t* launches);

/* ---- introspection for tests / roofline -------------------------------------------------------*/
/* Copies a named internal tensor of the last forward as float32 in the REFERENCE layout.
 * what: "logits" [B][n_classes]; "embeddings_raw" [B][emb]; "pooled" [B][2*enc_out];
 *       "block_out:<i>" [B][hidden][T]; "prolog_out" [B][hidden][T]; "epilog_out" [B][enc_out][T];
 *       "se_gate:<i>" [B][hidden]; after tn_backward also "d_energies" and "d_epilog_bn" [B][enc_out][T] (the attention
 *       energies' gradient / the gradient wrt the epilog BatchNorm output: element-wise comparisons of pooling paths).
 *       dst: device float buffer of the right size. */
int tn_debug_fetch(tn_plan* p, const char* what, float* dst, int64_t dst_floats, void* stream);
const char* tn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TITANET_AMD_H */
