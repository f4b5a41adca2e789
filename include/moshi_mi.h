/* moshi_mi.h — C ABI of libmoshi_mi.so: the MI355X-native (gfx950) engine for the 12.5 Hz
 * full-duplex frame step of kyutai-labs/moshi:
 *
 *      MimiModel.encode  ->  LMGen.step  ->  MimiModel.decode
 *
 * Each entry point replaces one method of the reference's PyTorch model API (the reference has
 * no FFI of its own for this path; its boundary is the Python surface of
 * moshi/moshi/models/compression.py and moshi/moshi/models/lm.py).  The citation on every
 * function is the reference method it stands in for (path:line under the reference checkout).
 *
 * Conventions
 *   - Plain C, no torch types.  All tensor pointers are DEVICE pointers on the current HIP device
 *     unless the comment says "host".  The caller owns every I/O buffer; a handle owns its packed
 *     weights and all streaming state (conv histories, conv-transpose partials, ring KV caches,
 *     delay token ring, offsets, exec masks).
 *   - Every function returns 0 (MMI_OK) or a negative mmi_status; nothing throws across the ABI.
 *     mmi_last_error() gives a thread-local human readable message for the last failure.
 *   - `stream` is a hipStream_t passed as void*.  Work is enqueued on it; nothing synchronises
 *     the device (same contract as the reference: "no hidden syncs in step", sampling.py:32-46).
 *   - One handle = one GPU = one caller at a time (not re-entrant), like a reference model
 *     instance (server.py:45,57 serialises sessions with a lock).  Independent handles on
 *     different GPUs may be driven from different processes (one process per GPU).
 *   - Integer tensors at the boundary are int64, as in the reference API.
 */
#ifndef MOSHI_MI_H_
#define MOSHI_MI_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMI_ABI_VERSION 3

typedef enum mmi_status {
    MMI_OK = 0,
    MMI_ERR_INVALID = -1,        /* bad argument (null pointer, negative size...)                       */
    MMI_ERR_SHAPE = -2,          /* batch/shape mismatch: reference raises AssertionError (lm.py:679-686) */
    MMI_ERR_STATE = -3,          /* not streaming: reference raises RuntimeError (lm.py:673-676)          */
    MMI_ERR_HIP = -4,            /* a HIP runtime call failed                                           */
    MMI_ERR_MISSING_WEIGHT = -5, /* a state-dict key required by the config was not supplied             */
    MMI_ERR_UNSUPPORTED = -6,    /* config outside what the kernels implement                           */
    MMI_ERR_BUSY = -7,           /* batcher: no free slot / a channel's buffer is full                    */
    MMI_ERR_NO_CHANNEL = -8      /* batcher: the channel id names no open channel (closed, or re-opened under a new id) */
} mmi_status;

typedef enum mmi_dtype { MMI_F32 = 0, MMI_BF16 = 1, MMI_I64 = 2, MMI_F16 = 3, MMI_I8 = 4, MMI_F8E4M3 = 5 /* OCP e4m3fn */ } mmi_dtype;

typedef void* mmi_stream; /* hipStream_t */

/* One state-dict entry, named exactly as in the reference checkpoints
 * (loaders.py:356-358,413-422).  `data` is a device pointer. */
typedef struct mmi_tensor_desc {
    const char* name;
    const void* data;
    int32_t dtype; /* mmi_dtype */
    int32_t ndim;
    int64_t shape[4];
} mmi_tensor_desc;

int mmi_version(void);
const char* mmi_last_error(void);

/* ------------------------------------------------------------------------------------------ */
/* Mimi codec                                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* Mirrors loaders._seanet_kwargs / _quantizer_kwargs / _transformer_kwargs / _mimi_config
 * (loaders.py:38-88).  Fixed by the kernels: causal, pad_mode "constant", true_skip, ELU(1.0),
 * norm "none", n_residual_layers == 1 (dilation 1), transformer norm "layer_norm", gating "none",
 * positional_embedding "rope", learnt conv resampling with the channel-wise upsample. */
typedef struct mmi_mimi_cfg {
    int32_t sample_rate;      /* 24000 */
    int32_t frame_size;       /* samples per 12.5 Hz frame = 1920 (compression.py:244-246) */
    int32_t channels;         /* 1 */
    int32_t dimension;        /* 512 */
    int32_t n_filters;        /* 64 */
    int32_t n_ratios;         /* 4 */
    int32_t ratios[8];        /* decoder order {8,6,5,4}; the encoder uses them reversed (seanet.py:154) */
    int32_t kernel_size;      /* 7 */
    int32_t last_kernel_size; /* 3 */
    int32_t residual_kernel_size; /* 3 */
    int32_t compress;         /* 2 */
    int32_t resample_stride;  /* encoder_frame_rate / frame_rate = 2 (compression.py:197-207) */
    int32_t tr_d_model;       /* 512 */
    int32_t tr_num_heads;     /* 8 */
    int32_t tr_num_layers;    /* 8 */
    int32_t tr_dim_feedforward; /* 2048 */
    int32_t tr_context;       /* 250 */
    float tr_max_period;      /* 10000 */
    int32_t q_dimension;      /* 256 */
    int32_t q_bins;           /* 2048 */
    int32_t q_n_q;            /* 32 total codebooks */
    int32_t q_n_q_semantic;   /* 1 */
} mmi_mimi_cfg;

typedef struct mmi_mimi mmi_mimi;

/* loaders.get_mimi (loaders.py:323-363): build the model from state-dict tensors (fp32).
 * Weights are repacked into MFMA fragment order on the device; the descs may be freed after. */
int mmi_mimi_create(const mmi_mimi_cfg* cfg, const mmi_tensor_desc* weights, int32_t n_weights,
                    int32_t max_batch, mmi_mimi** out);
void mmi_mimi_destroy(mmi_mimi* m);

/* MimiModel.set_num_codebooks (compression.py:258-260). 1 <= n <= q_n_q. */
int mmi_mimi_set_num_codebooks(mmi_mimi* m, int32_t n);
int mmi_mimi_num_codebooks(const mmi_mimi* m);

/* StreamingModule.streaming(batch) enter / exit (streaming.py:131-137, 110-129): allocate and
 * zero the streaming state of `batch` rows; exec mask all ones. */
int mmi_mimi_streaming_start(mmi_mimi* m, int32_t batch, mmi_stream stream);
int mmi_mimi_streaming_stop(mmi_mimi* m);
/* Rows of the current stream (StreamingModule._streaming_state.batch_size); 0 when not streaming. */
int mmi_mimi_streaming_batch(const mmi_mimi* m);

/* StreamingModule.set_exec_mask (streaming.py:183-211): mask = device uint8[batch]. */
int mmi_mimi_set_exec_mask(mmi_mimi* m, const uint8_t* mask, mmi_stream stream);
/* StreamingModule.reset_streaming (streaming.py:139-156): mask = device uint8[batch] or NULL (= all rows). */
int mmi_mimi_reset(mmi_mimi* m, const uint8_t* mask_or_null, mmi_stream stream);

/* StreamingModule.get_streaming_state / set_streaming_state (streaming.py:158-181): the complete streaming state of the
 * current stream as one opaque device buffer of mmi_mimi_state_bytes() bytes (conv histories, conv-transpose partials, KV
 * rings, offsets, masks).  A snapshot can be loaded back into the same handle while it streams with the same batch. */
int64_t mmi_mimi_state_bytes(const mmi_mimi* m);
int mmi_mimi_state_save(mmi_mimi* m, void* dst, int64_t bytes, mmi_stream stream);
int mmi_mimi_state_load(mmi_mimi* m, const void* src, int64_t bytes, mmi_stream stream);

/* MimiModel.encode in streaming mode (compression.py:376-388, 338-374):
 * pcm f32 [batch,1,n_frames*frame_size] -> codes i64 [batch,num_codebooks,n_frames]. */
int mmi_mimi_encode_step(mmi_mimi* m, const float* pcm, int64_t* codes, int32_t batch, int32_t n_frames,
                         mmi_stream stream);
/* MimiModel.encode_to_latent(x, quantize=False) (compression.py:390-404):
 * pcm -> unquantized latent f32 [batch,dimension,n_frames]. */
int mmi_mimi_encode_latent_step(mmi_mimi* m, const float* pcm, float* latent, int32_t batch, int32_t n_frames,
                                mmi_stream stream);
/* SplitResidualVectorQuantizer.encode (vq.py:269-279) on a given latent (stateless):
 * latent f32 [batch,dimension,n_frames] -> codes i64 [batch,num_codebooks,n_frames]. */
int mmi_mimi_quantize(mmi_mimi* m, const float* latent, int64_t* codes, int32_t batch, int32_t n_frames,
                      mmi_stream stream);
/* MimiModel.decode_latent (compression.py:431-433; vq.py:281-287) (stateless):
 * codes i64 [batch,K,n_frames] -> latent f32 [batch,dimension,n_frames]. */
int mmi_mimi_decode_latent(mmi_mimi* m, const int64_t* codes, float* latent, int32_t batch, int32_t n_codebooks,
                           int32_t n_frames, mmi_stream stream);
/* MimiModel.decode in streaming mode (compression.py:406-429):
 * codes i64 [batch,K,n_frames] (values in [0,bins)) -> pcm f32 [batch,1,n_frames*frame_size]. */
int mmi_mimi_decode_step(mmi_mimi* m, const int64_t* codes, float* pcm, int32_t batch, int32_t n_codebooks,
                         int32_t n_frames, mmi_stream stream);
/* The same for a strided view: codes[b * batch_stride + k * n_frames + f].  What `mimi.decode(tokens[:, 1:])` is in the
 * reference's serving loop (server.py:144-146: the audio columns of LMGen.step's [B, 1 + dep_q, 1] output) - the view is read
 * in place, batch_stride = (1 + dep_q) * n_frames there. */
int mmi_mimi_decode_step_strided(mmi_mimi* m, const int64_t* codes, int64_t batch_stride, float* pcm, int32_t batch,
                                 int32_t n_codebooks, int32_t n_frames, mmi_stream stream);

/* ------------------------------------------------------------------------------------------ */
/* Moshi LM (Temporal + Depth transformer) and LMGen                                          */
/* ------------------------------------------------------------------------------------------ */

/* Mirrors loaders._lm_kwargs (loaders.py:90-119) for the options LMGen.step exercises on Moshi.
 * Fixed by the kernels: norm "rms_norm_f32" (eps 1e-8), gating "silu", positional_embedding "rope"
 * (interleaved) for the temporal transformer and "none" for the depformer, causal, no fuser / no CFG,
 * depformer_multi_linear + depformer_weights_per_step, kv_repeat 1. */
typedef struct mmi_lm_cfg {
    int32_t dim;            /* 4096 */
    int32_t num_heads;      /* 32 */
    int32_t num_layers;     /* 32 */
    int32_t ffn_hidden;     /* gating hidden size: 11264 (gating.py:55-58) */
    int32_t context;        /* 3000 */
    float max_period;       /* 10000 */
    int32_t n_q;            /* 16 audio streams seen by the temporal transformer */
    int32_t dep_q;          /* 8 generated by the depformer */
    int32_t card;           /* 2048 */
    int32_t text_card;      /* 32000 */
    int32_t text_card_out;  /* 32000 */
    int32_t depformer_dim;        /* 1024 */
    int32_t depformer_num_heads;  /* 16 */
    int32_t depformer_num_layers; /* 6 */
    int32_t depformer_ffn_hidden; /* 2816 */
    int32_t delays[64];     /* num_codebooks = n_q + 1 entries (text first) */
    int32_t existing_text_padding_id; /* 3 */
    int32_t extra_heads_num_heads;    /* 0: nn.Linear(dim, extra_heads_dim) heads on the transformer output (lm.py:101-102, 224-226) */
    int32_t extra_heads_dim;          /* 6 */
    int32_t kv_cache_dtype;           /* 0 / MMI_BF16: the reference's bf16 ring (transformer.py:453-455); MMI_F8E4M3: e4m3 ring -
                                         half the attention stream and half the per-session state (SURVEY.md 8d C5 "fp8 KV") */
    int32_t cross_attention;          /* 1: every temporal layer has a cross-attention block (`cross_attention.{in,out}_projs.0`,
                                         `norm_cross.{weight,bias}`; transformer.py:727-732, 779-786) fed by mmi_guidance.condition_cross */
} mmi_lm_cfg;

/* LMGen constructor arguments that change what step computes (lm.py:557-574). */
typedef struct mmi_sampling {
    int32_t use_sampling; /* 0 = greedy argmax */
    float temp;           /* audio temperature 0.8 */
    float temp_text;      /* 0.7 */
    int32_t top_k;        /* 250 */
    int32_t top_k_text;   /* 25 */
    uint64_t seed;        /* seed of the on-device counter RNG used when no noise is supplied */
} mmi_sampling;

typedef struct mmi_lm mmi_lm;

/* loaders.get_moshi_lm (loaders.py:366-446): build LMModel from state-dict tensors - bf16, or with the linears in the
 * reference's quantised storage (utils/quantize.py:13-22): `<linear>.weight` MMI_I8 [out,in] row-wise absmax codes plus
 * `<linear>.weight_scb` MMI_F32 [out] row absmax; or as fp8 (BASELINE configs[4], run on the fp8 MFMA): `<linear>.weight`
 * MMI_F8E4M3 [out,in] codes, `<linear>.weight_scale` MMI_F32 [out] (W ~= code * scale) and an optional scalar
 * `<linear>.input_scale` MMI_F32 (static activation scale: x8 = e4m3(x / input_scale), default 1).  Embeddings and norms
 * stay bf16.  The linears must be all bf16, all int8 or all fp8. */
int mmi_lm_create(const mmi_lm_cfg* cfg, const mmi_tensor_desc* weights, int32_t n_weights,
                  int32_t max_batch, mmi_lm** out);
void mmi_lm_destroy(mmi_lm* lm);

/* LMGen.streaming(batch) enter/exit (lm.py:605-666). */
int mmi_lm_streaming_start(mmi_lm* lm, int32_t batch, const mmi_sampling* sampling, mmi_stream stream);
/* LMGen's classifier-free guidance and conditioning arguments (lm.py:566-574, 612-651).  With
 * cfg_coef != 1 the model runs two rows per session (conditioned, unconditioned) - 2 * batch must fit max_batch - and every
 * sampling site draws from logits_null + (logits - logits_null) * cfg_coef (lm.py:727-733, 828-832).
 *   cfg_is_masked_until  host i64 [batch] or NULL: the unconditioned row reads zero tokens until offset > delay + value (lm.py:713-721)
 *   cfg_is_no_text       the unconditioned row never reads text tokens and the text logits stay unguided (lm.py:724-732)
 *   condition_sum        device bf16 [model rows, dim] or NULL: ConditionFuser.get_sum of the condition tensors, added to the
 *                        input embeddings of every step (lm.py:621-628, 399-400); model rows = batch, or 2 * batch when guided
 *                        (conditioned rows first).
 *   condition_cross      device bf16 [model rows, cross_len, dim] or NULL: ConditionFuser.get_cross of the condition tensors
 *                        (conditioners/base.py:392-409), the source of every temporal layer's cross-attention block; required
 *                        when the model was created with cfg.cross_attention.  Its keys / values are projected once, here
 *                        (the reference caches them on the first step, transformer.py:521-531). */
typedef struct mmi_guidance {
    float cfg_coef;
    int32_t cfg_is_no_text;
    const int64_t* cfg_is_masked_until;
    const void* condition_sum;
    const void* condition_cross;
    int32_t cross_len;
} mmi_guidance;
int mmi_lm_streaming_start_guided(mmi_lm* lm, int32_t batch, const mmi_sampling* sampling, const mmi_guidance* guide_or_null,
                                  mmi_stream stream);
int mmi_lm_model_rows(const mmi_lm* lm);
/* Sessions of the current stream (rows of the caller's tensors); 0 when not streaming. */
int mmi_lm_streaming_batch(const mmi_lm* lm);
/* A handle binds to the HIP device that is current when it is created (weights, state, streams and graphs live there); every
 * entry point switches the calling thread to that device for the call and restores the caller's.  Pointers passed in must be
 * on that device.  mmi_*_device return the ordinal. */
int mmi_lm_device(const mmi_lm* lm);
int mmi_mimi_device(const mmi_mimi* m);
/* Engine counters for tests / diagnostics.  which = 0: GEMM launches (or captured graph nodes) that took the LDS-resident
 * kernel (k_gemm_xlds); 1: bit v set = step program v (short-ring / deep-ring decode attention) is captured and instantiated -
 * both are from the stream's first step on, so that the switch is never a capture inside a live session; 2: the host's bound on
 * the ring depth (steps since streaming_start / seek / the offsets of a restored snapshot); 3: the depth transformer's MFMA tile
 * (16 at <= 32 sessions with bf16 weights, else 32). */
int64_t mmi_lm_stat(const mmi_lm* lm, int32_t which);
/* LMGen.step_with_extra_heads (lm.py:793-807): softmax(extra_head(transformer_out)) of the LAST step for every head:
 * probs f32 [model rows, extra_heads_num_heads, extra_heads_dim]. */
int mmi_lm_extra_heads(mmi_lm* lm, float* probs, mmi_stream stream);
/* get / set_streaming_state of LMGen (streaming.py:158-181; _LMGenState lm.py:520-547): KV rings, token ring, offsets, masks,
 * RNG counter as one opaque device buffer; host_word carries the host-side step counter (`offset_cpu`). */
int64_t mmi_lm_state_bytes(const mmi_lm* lm);
int mmi_lm_state_save(mmi_lm* lm, void* dst, int64_t bytes, int64_t* host_word, mmi_stream stream);
int mmi_lm_state_load(mmi_lm* lm, const void* src, int64_t bytes, int64_t host_word, mmi_stream stream);
int mmi_lm_streaming_stop(mmi_lm* lm);
int mmi_lm_set_exec_mask(mmi_lm* lm, const uint8_t* mask, mmi_stream stream);        /* lm.py:544-547 */
int mmi_lm_reset(mmi_lm* lm, const uint8_t* mask_or_null, mmi_stream stream);        /* lm.py:537-542 */

/* LMGen.step (lm.py:785-791, 668-783).
 *   user_codes  i64 [batch, n_user(>= n_q - dep_q), 1]; extra rows are ignored (lm.py:688-689)
 *   out_tokens  i64 [batch, dep_q + 1, 1]; rows not yet valid hold -2 (lm.py:781-782)
 *   opt_text_logits  f32 [batch, text_card_out] or NULL   } parity taps: the logits the tokens were
 *   opt_audio_logits f32 [batch, dep_q, card]   or NULL   } sampled from (bf16 values widened to f32)
 *   opt_noise   f32 [batch, 1 + dep_q, max(top_k, top_k_text)] or NULL: Exp(1) draws used instead of the
 *               on-device RNG, indexed by rank in the descending top-k (sampling.py:40-47,59-63)
 *   valid (host int*): 0 while offset_cpu <= max_delay, i.e. where the reference returns None (lm.py:774-776)
 */
int mmi_lm_step(mmi_lm* lm, const int64_t* user_codes, int32_t n_user, int64_t* out_tokens,
                float* opt_text_logits, float* opt_audio_logits, const float* opt_noise, int32_t batch,
                int32_t* valid, mmi_stream stream);

/* Phase callback (no reference counterpart; used by the duplex pipeline below): a host function that every following
 * mmi_lm_step calls, on the calling thread, between the temporal transformer + text head (chip-filling, HBM-bound GEMMs) and
 * the depth transformer (dep_q x 33 small dependent launches that leave most CUs idle), with the step's stream: what it
 * enqueues there marks the point from which other streams can run beside the step without costing it.  A non-zero return
 * aborts the step.  NULL clears. */
int mmi_lm_set_phase_callback(mmi_lm* lm, int (*fn)(void* user, mmi_stream stream), void* user);

/* LMGen's per-step hooks (lm.py:568-570, 734-747): host callbacks between the stages of a step, each of which may READ and
 * MODIFY IN PLACE what the reference's hook receives (the TTS wrapper forces text tokens this way, models/tts.py):
 *   on_text_logits   after the (guided) text logits are final, before the text token is sampled      (lm.py:734-735)
 *   on_text_token    after the text token is sampled, before the depth transformer reads it          (lm.py:746-747)
 *   on_audio_tokens  after the dep_q audio tokens are sampled (or replaced), before they enter the ring (lm.py:756-757)
 * With any hook set, mmi_lm_step launches the step in segments (no graph replay) on `stream` and calls the hooks on the
 * calling thread, without synchronising: a hook works on the same stream through mmi_lm_hook_io.  A non-zero return aborts the
 * step with MMI_ERR_INVALID.  Pass NULL to clear. */
typedef struct mmi_lm_hooks {
    int (*on_text_logits)(void* user);
    int (*on_text_token)(void* user);
    int (*on_audio_tokens)(void* user);
    void* user;
} mmi_lm_hooks;
int mmi_lm_set_hooks(mmi_lm* lm, const mmi_lm_hooks* hooks_or_null);
/* Inside a hook: copy one of the step's tensors out (write = 0) or back in (write = 1), stream-ordered on `stream`:
 *   which 0  text logits   bf16 [batch, text_card_out]   (the model dtype, as the reference's hook sees them)
 *   which 1  text token    i64  [batch]
 *   which 2  audio tokens  i64  [batch, dep_q]
 * nbytes = the size of `buf`; anything but the tensor's size is refused (MMI_ERR_SHAPE). */
int mmi_lm_hook_io(mmi_lm* lm, int32_t which, int32_t write, void* buf, int64_t nbytes, mmi_stream stream);
int32_t mmi_lm_has_hooks(const mmi_lm* lm);         /* 1 while any of the three hooks is set */

/* Teacher forcing for the NEXT step only: tokens i64 [batch, 1 + dep_q] (text, then the dep_q audio codebooks);
 * entries >= 0 replace the sampled token at that site (the logits taps are still produced), entries < 0 keep
 * sampling.  Covers LMGen.step's `depformer_replace_tokens` argument (lm.py:751-755) and lets the parity tests
 * replay the reference's token history exactly. */
int mmi_lm_force_next_tokens(mmi_lm* lm, const int64_t* tokens, mmi_stream stream);

/* Parity tap of the residual stream (test aid; the reference has none - it is what a forward hook on
 * `transformer.layers[0]` / `layers[-1]` would see, transformer.py:814-929): switched on BEFORE streaming_start, every step also
 * keeps the hidden state after the first and after the last temporal layer; get copies bf16 [2][model rows][dim] (row-major)
 * out, stream-ordered.  Off by default: no launch is added. */
int mmi_lm_set_hidden_taps(mmi_lm* lm, int32_t on);
int mmi_lm_get_hidden_taps(mmi_lm* lm, void* buf_bf16, int64_t nbytes, mmi_stream stream);

/* Parity tap (test aid; no reference counterpart - it is `F.linear(x, weight)` / `QLinear.forward(x)` of ONE module,
 * utils/quantize.py:24-40, on rows the caller supplies): runs the linear stored under the state-dict key `weight_name`
 * ("transformer.layers.0.gating.linear_in.weight", "depformer.layers.2.self_attn.out_projs.5.weight", "linears.3.weight", ...)
 * through the kernels the step uses for it and returns its bf16 output, so that a test can hold the engine's int8 x int8
 * arithmetic to the oracle BIT FOR BIT, per linear, instead of through the logits of a whole network.
 *   x         device bf16 [rows][in_features], row-major; rows <= max_batch
 *   out       device bf16 [rows][out_features]; a gated linear_in returns [rows][hidden] = silu(gate) * value, what the step hands on
 *   path      MMI_DBG_PLAIN       (int8 x int8 models: k_quant_rows_i8, then) the weight-streaming GEMM as planned for the
 *                                 shape and batch (k_gemm_xp / k_gemm_xlds), store epilogue
 *             MMI_DBG_SPLITK      the same operand through the split-K form + the fold of the next norm launch
 *                                 (out = bf16(sum of the partials)); MMI_ERR_UNSUPPORTED if the engine does not split this GEMM
 *             MMI_DBG_FUSED       k_gemm_q8: the row quantisation inside the GEMM (int8 x int8 models, rows of <= 88 entries)
 *             MMI_DBG_NORM        RMSNorm with the vector stored under `alpha_name` first, as the step's norm launch does it
 *                                 (k_resid_rmsnorm + its int8 copy), then the GEMM; norm_out (optional, bf16 [rows][in_features])
 *                                 receives the normalised rows
 *             MMI_DBG_NORM_FUSED  the norm inside the GEMM (k_gemm_xp_norm / k_gemm_q8<NORM>; rows of <= 1024 features)
 *   codes, absmax (optional; int8 x int8 models, paths PLAIN / SPLITK / NORM): int8 [rows][in_features] and fp32 [rows] - the
 *             row-wise quantisation the GEMM consumed (bitsandbytes' CA / SCA)
 * Does not touch a running stream's state (own scratch); synchronises `stream`. */
enum { MMI_DBG_PLAIN = 0, MMI_DBG_SPLITK = 1, MMI_DBG_FUSED = 2, MMI_DBG_NORM = 3, MMI_DBG_NORM_FUSED = 4 };
int mmi_lm_debug_linear(mmi_lm* lm, const char* weight_name, const char* alpha_name_or_null, int32_t path, const void* x_bf16,
                        int32_t rows, void* out_bf16, int8_t* codes_or_null, float* absmax_or_null, void* norm_out_or_null,
                        mmi_stream stream);

/* The launch list of one frame step, recorded while the step ran for the first time: one line "site<TAB>kernel[<TAB>weight bytes]" per kernel
 * launch in launch order (sites: "L.in_proj", "L.ffn_in", "dep.out_proj", "text_linear", ...).  scripts/rocpd_sites.py joins
 * it with a rocprofv3 kernel trace by position inside the step, which is how per-site durations of kernels that share one
 * name are recomputed under profiles/.  Returns the bytes needed including the final NUL (call with buf = NULL to size);
 * 0 before the first step.  mmi_mimi_launch_list: which = 0 encoder step, 1 decoder step. */
int64_t mmi_lm_launch_list(const mmi_lm* lm, char* buf, int64_t cap);
int64_t mmi_mimi_launch_list(const mmi_mimi* m, int32_t which, char* buf, int64_t cap);

/* Test / benchmark aid (no reference counterpart): move every session to stream position offsets[b] (host i64 [batch]) WITHOUT
 * touching the KV ring - the positions skipped read whatever the ring holds (zeros after streaming_start).  Lets the ring
 * wrap at the real capacity (context 3000) be exercised in seconds, and a full-context step be timed without 3000 warm-up
 * steps.  The host step counter becomes max(offsets). */
int mmi_lm_seek(mmi_lm* lm, const int64_t* offsets, mmi_stream stream);

/* Dominant-kernel timing tap for bench.py's roofline object: when enabled, steps run un-graphed and
 * every launch of the widest weight-streaming GEMM is bracketed by hipEvents on `stream`. */
int mmi_lm_profile_begin(mmi_lm* lm);
/* Per-site timings of the steps run since mmi_lm_profile_begin: one line "site<TAB>ops<TAB>total ms<TAB>weight bytes per op"
 * per site of the launch list (hipEvent pairs around every op of the un-graphed steps; dispatch gaps included).  Call before
 * mmi_lm_profile_end; synchronises the stream.  Returns the bytes needed including the final NUL (buf = NULL to size). */
int64_t mmi_lm_profile_sites(mmi_lm* lm, char* buf, int64_t cap);
/* Returns the mean duration (ms) and launch count since mmi_lm_profile_begin, plus the algorithmic
 * bytes one such launch streams (packed weight bytes + activations in/out). Synchronises the stream. */
int mmi_lm_profile_end(mmi_lm* lm, double* mean_ms, int64_t* n_launches, int64_t* bytes_per_launch,
                       const char** kernel_name);

/* Architecture the handle was created with (what callers read as attributes: frame_size, num_codebooks, dep_q ...). */
int mmi_mimi_get_cfg(const mmi_mimi* m, mmi_mimi_cfg* out);
int mmi_lm_get_cfg(const mmi_lm* lm, mmi_lm_cfg* out);

/* ------------------------------------------------------------------------------------------ */
/* Duplex pipeline: the frame step  encode -> LMGen.step -> decode  software-pipelined        */
/* ------------------------------------------------------------------------------------------ */
/* The reference's serving loop runs the three calls of a frame back to back on one stream (server.py:132-146).  Their
 * streaming states are disjoint: encode(t+1) depends on encode(t) only, decode(t) on LMGen.step(t) and decode(t-1).  A duplex
 * pipeline owns three HIP streams (encoder, LM, decoder) and the events between them, so that - when frames are submitted
 * back to back (offline inference, several session groups per GPU, a loaded server) - encode(t+1) and decode(t-1) run in the
 * shadow of LMGen.step(t), whose depth-transformer phase leaves most CUs idle.  Results are bit-identical to the serial
 * schedule (same kernels, same per-stream order).  The DEVICE is never synchronised, but the HOST is: mmi_duplex_submit(t)
 * blocks the calling thread (hipEventSynchronize) until LMGen.step(t-2) has completed (flow control: two steps in flight at
 * most) and until LMGen.step(t-1) has reached its depth-transformer phase (the gate behind which the codec work of frame t is
 * enqueued, csrc/duplex.hip) - in steady state it returns about one LM temporal phase after it was called.
 *
 *   mmi_duplex_submit   frame t: pcm_in f32 [batch, 1, frame_size] -> pcm_out f32 [batch, 1, frame_size] (written when the
 *                       frame's decode ran; untouched while *valid == 0, i.e. where LMGen.step returns None) and, optionally,
 *                       tokens_out i64 [batch, 1 + dep_q, 1] (LMGen.step's output, -2 rows included).  Work submitted on
 *                       `caller` before the call is ordered before the frame (inputs, set_exec_mask / reset_streaming issued
 *                       on the handles with that stream after a mmi_duplex_join).  At most two frames are in flight: the call
 *                       blocks the host until frame t-2 has completed.  pcm_in is copied into the pipeline's own input ring in
 *                       `caller`'s stream order: the caller may refill it as soon as the call returns (as after
 *                       MimiModel.encode).  pcm_out / tokens_out stay unread until a mmi_duplex_join on the consuming stream.
 *                       `batch` must equal the streaming batch (MMI_ERR_SHAPE otherwise, like mmi_lm_step).
 *                       MMI_ERR_UNSUPPORTED while the LM has per-step hooks installed (mmi_lm_set_hooks): the hooks work on the
 *                       caller's stream, the pipeline steps the LM on its own - a hooked LMGen goes through mmi_lm_step.
 *   mmi_duplex_join     makes `caller` wait (device side) for every frame submitted so far.
 * Both handles must be streaming with the same batch before mmi_duplex_create and must not be driven through their own
 * step entry points between a submit and the next join.  A call that fails half-way (a launch error inside a frame) leaves the
 * three stream orders inconsistent: the pipeline is dead from then on - submit / join / flush return MMI_ERR_STATE, waiters are
 * released - and must be destroyed. */
typedef struct mmi_duplex mmi_duplex;
int mmi_duplex_create(mmi_mimi* mimi, mmi_lm* lm, mmi_duplex** out);
void mmi_duplex_destroy(mmi_duplex* d);
int mmi_duplex_submit(mmi_duplex* d, const float* pcm_in, float* pcm_out, int64_t* tokens_out_or_null, int32_t batch,
                      int32_t* valid, mmi_stream caller);
int32_t mmi_duplex_batch(const mmi_duplex* d);      /* the streaming batch the pipeline was created with */
int mmi_duplex_join(mmi_duplex* d, mmi_stream caller);
/* The host-side form: blocks the calling thread until every submitted frame has completed (its outputs are then readable from any
 * stream).  Unlike mmi_duplex_join it leaves no waiter on the device while the frames run. */
int mmi_duplex_flush(mmi_duplex* d);
/* Diagnostics: with the timeline on, every submit records timestamps around the frame's three phases on their streams;
 * mmi_duplex_get_timeline synchronises the pipeline and returns ms since the LAST submit reached the caller's stream:
 * {encode begin, encode end, LM begin, -, LM end, decode begin, decode end} (host f32[7]); entries of phases that did not run
 * are -1.  One set of events serves every frame, so the seven entries belong to ONE frame only when a mmi_duplex_flush follows
 * every submit (one frame alone in the pipeline: what bench.py's p50 / p95 measure); with frames in flight the decode entries are
 * those of frame t-2.  Per-frame figures in steady state: mmi_duplex_get_stamps. */
int mmi_duplex_set_timeline(mmi_duplex* d, int32_t on);
int mmi_duplex_get_timeline(mmi_duplex* d, float* ms7);
/* Diagnostics, finer: with the timeline on the pipeline also stamps the device's constant-rate clock at its hand-off points
 * (one-thread kernels on the three streams).  Returns ms40[4][10] - frames t & 3 of the last four submits; per frame: input
 * published, encode begin / end, the LM's wait for the encoder begin / end, LM begin, depth-transformer phase, LM end, decode
 * begin / end - in ms since the oldest stamp held (-1: not stamped; the clock's rate is hipDeviceAttributeWallClockRate), and the
 * number of the last submitted frame. */
int mmi_duplex_get_stamps(mmi_duplex* d, double* ms40, int64_t* last_frame);

/* ------------------------------------------------------------------------------------------ */
/* Session batcher: many live dialogue sessions on one GPU                                    */
/* ------------------------------------------------------------------------------------------ */
/* SURVEY.md 8f-1.  The reference's Python server runs ONE session under a lock (server.py:45,57,154-169);
 * its Rust server packs up to `batch_size` live channels into one batched model step with a per-row stream
 * mask and a per-row reset when a channel is (re)opened (rust/moshi-server/src/batched_asr.rs:188-276 model
 * loop, :279-374 pre_process, :376-437 post_process; py_module.rs:443-470 slot allocation).  This is that
 * model loop for the full-duplex path, on top of the entry points above:
 *
 *   every mmi_batcher_step:  per slot, take one 80 ms frame from the channel's PCM FIFO if it holds one (exec mask),
 *   reset the rows of channels opened since the last step, then  Mimi encode -> LMGen.step -> Mimi decode  on the
 *   whole batch, and hand each executed row's (text token, audio tokens, PCM frame) to its channel's output FIFO.
 *
 * Host pointers here are HOST memory: the batcher owns pinned staging buffers and its own HIP stream.
 * open/close/push/pop may be called from any thread; step from one thread at a time (the model loop). */
typedef struct mmi_batcher_cfg {
    int32_t slots;                          /* rows of the batch = concurrent channels; <= both handles' max_batch */
    int32_t reset_codec_after_first_frame;  /* server.py:135-141: the first input frame's encoder state is dropped  */
    int32_t max_buffered_frames;            /* per-channel cap on queued input frames and on un-popped output frames */
    mmi_sampling sampling;                  /* LMGen constructor arguments (lm.py:557-574)                          */
    mmi_guidance guidance;                  /* cfg_coef = 0 or 1: none.  Otherwise as mmi_lm_streaming_start_guided, shared by
                                               every channel (server.py:53-54: one condition for the model type); the LM
                                               handle then needs max_batch >= 2 * slots                               */
} mmi_batcher_cfg;

typedef struct mmi_batcher_stats {
    int64_t steps;            /* model steps run (iterations that had at least one row with data)          */
    int64_t frames;           /* session-frames executed (sum of active rows over steps)                   */
    int64_t dropped_frames;   /* output frames dropped because a channel's output FIFO was full            */
    int32_t used_slots, total_slots;
    float last_step_ms;       /* device time of the last step (hipEvents on the batcher's stream)          */
} mmi_batcher_stats;

typedef struct mmi_batcher mmi_batcher;

/* Puts both models into streaming mode with batch = cfg->slots (streaming_forever, server.py:59-60).  The handles must
 * outlive the batcher and must not be driven by anyone else meanwhile. */
int mmi_batcher_create(mmi_mimi* mimi, mmi_lm* lm, const mmi_batcher_cfg* cfg, mmi_batcher** out);
void mmi_batcher_destroy(mmi_batcher* b);
/* Claim a free slot (py_module.rs:443-470); MMI_ERR_BUSY when all are taken.  The row's streaming state is reset at
 * the next step (handle_chat: mimi.reset_streaming(); lm_gen.reset_streaming(), server.py:163-164). */
int mmi_batcher_open(mmi_batcher* b, int64_t* channel_id);
int mmi_batcher_close(mmi_batcher* b, int64_t channel_id);
/* Append 24 kHz mono PCM (host f32) to the channel's FIFO (batched_asr.rs:77-90 extend_data). */
int mmi_batcher_push_pcm(mmi_batcher* b, int64_t channel_id, const float* pcm, int32_t n_samples);
/* One iteration of the model loop.  n_active (host) = rows that had a frame; 0 means nothing was run. */
int mmi_batcher_step(mmi_batcher* b, int32_t* n_active);
/* Next un-popped output frame of a channel: pcm f32[frame_size], tokens i64[1 + dep_q] (text, audio...), got = 0/1.
 * With an ASR-style LM (dep_q = 0, the batched_asr.rs case proper) nothing is decoded: tokens = the text token, pcm is zeros.
 * Frames produced while the LM's delay ring was still filling (tokens -2, lm.py:781-782) are never queued, exactly
 * like the reference server skips `None` (server.py:144-146). */
int mmi_batcher_pop(mmi_batcher* b, int64_t channel_id, float* pcm, int64_t* tokens, int32_t* got);
int mmi_batcher_get_stats(mmi_batcher* b, mmi_batcher_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* MOSHI_MI_H_ */
