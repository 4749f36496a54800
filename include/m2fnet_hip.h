/* m2fnet_hip.h - C ABI of the MI355X-native M2FNet fusion-transformer training path.
 *
 * Shared library: multimodal-emotion-recognition_amd/csrc/libm2fnet_hip.so (gfx950 only).
 * Plain pointers, sizes and a hipStream_t only - no torch types.  All device buffers are owned by the
 * caller (the Python host allocates them as torch tensors; any hipMalloc'ed memory works).  Every
 * function returns 0 on success and a non-zero code on failure (m2f_last_error() has the text); the
 * Python binding turns non-zero into an exception.  One stream per rank, no internal threads, no
 * allocation inside the library.
 *
 * The reference (iosonopersia/Multimodal-Emotion-Recognition) has no FFI: its "interface" for this path
 * is the Python surface of src/model.py / src/train.py.  Each entry point below names the reference
 * code it stands in for (paths relative to the reference root).
 */
#ifndef M2FNET_HIP_H
#define M2FNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* m2f_stream_t;            /* hipStream_t */
typedef struct m2f_plan m2f_plan;

/* Plain-value mirror of the reference's `config.model` sub-tree (src/config.yaml:31-54, consumed at
 * src/model.py:28-56).  dim_ff / ln_eps are the torch defaults the reference inherits (2048, 1e-5). */
typedef struct m2f_config {
    int32_t audio_enabled, text_enabled, fam_enabled;
    int32_t d_audio, d_text, d_fam;
    int32_t nhead_audio, nhead_text, nhead_fam;
    int32_t nlayers_audio, nlayers_text, nlayers_fam;   /* n_encoder_layers, n_encoder_layers, FAM.n_layers */
    int32_t ntrans_audio, ntrans_text;                  /* n_transformers */
    int32_t cls_hidden, cls_out, cls_layers;
    int32_t dim_ff;
    float dropout;
    float ln_eps;
} m2f_config;

enum { M2F_F32 = 0, M2F_BF16 = 1 };    /* GEMM operand precision: exact-fp32 MFMA | bf16 MFMA (fp32 accumulate) */

/* Buffers inside the caller-provided workspace that the host reads / writes (m2f_plan_buffer). */
enum {
    M2F_BUF_TEXT = 0,       /* float [B*L, pad8(d_text)]  input (batch["text"], src/train.py:222); rows padded to x8  */
    M2F_BUF_AUDIO = 1,      /* float [B*L, pad8(d_audio)] input (batch["audio"], src/train.py:223); pad columns stay 0 */
    M2F_BUF_KEYPAD = 2,     /* uint8 [B*L]           input  (batch["padding_mask"], 1 = pad, :225)      */
    M2F_BUF_LABELS = 3,     /* int64 [B*L]           input  (batch["emotion"], -1 = ignore, :224)       */
    M2F_BUF_CLASSW = 4,     /* float [16]            input  optional class weights (src/train.py:45-48) */
    M2F_BUF_LOGITS = 5,     /* float [B*L, cls_out]  output (M2FNet.forward, src/model.py:145)          */
    M2F_BUF_LOSS = 6,       /* float [4]: loss, denominator, numerator, - ; for train plans this IS grads[total..] */
    M2F_BUF_DLOGITS = 7,    /* float [B*L, cls_out]  d loss / d logits (written by m2f_loss, or by host) */
    M2F_BUF_FAM0_OUT = 8,   /* float [B*L, pad8(d_fam)] first fusion layer output (kernel-level parity)  */
    M2F_BUF_CU_SEQLENS = 9, /* int32 [B+1]           input of PACKED plans: dialogue b owns token rows cu[b] .. cu[b+1]-1 */
    M2F_BUF_COUNT = 10
};

const char* m2f_last_error(void);
int m2f_device_check(void);            /* 0 iff the current HIP device is gfx950 */

/* Flat parameter layout = reference state_dict order (src/model.py:24-100), unique tensors
 * only, each padded to 64 floats.  Fills offsets/numels (elements) for up to max_entries tensors and
 * *total (flat length in elements); returns the number of unique tensors, or <0 on error. */
int m2f_param_layout(const m2f_config* cfg, int64_t* offsets, int64_t* numels, int max_entries, int64_t* total);

/* Workspace size (bytes) a plan for (cfg, B dialogues, L utterances) needs. */
int64_t m2f_workspace_bytes(const m2f_config* cfg, int B, int L, int train);

/* A plan = the launch list of one M2FNet step for fixed (cfg, B, L, precision, train/eval) bound to the
 * caller's flat parameter buffer, flat gradient buffer (may be NULL for eval plans), workspace and
 * dropout RNG state (4 x uint32 in device memory: seed_lo, seed_hi, step_lo, step_hi).
 * The gradient buffer must hold total + 64 floats (total from m2f_param_layout): the 64-float tail receives
 * (loss, denominator, numerator) so that a data-parallel all-reduce of the whole buffer also sums the
 * valid-utterance denominators. */
m2f_plan* m2f_plan_create(const m2f_config* cfg, int B, int L, int precision, int train,
                          float* params, float* grads, void* workspace, int64_t workspace_bytes,
                          uint32_t* rng_state);
/* PACKED ("varlen") plan: the T token rows of every buffer belong to B dialogues of 1 .. L utterances each, dialogue b owning rows
 * cu[b] .. cu[b+1]-1 of M2F_BUF_CU_SEQLENS (int32 [B+1], cu[0] = 0, cu[B] <= T, written by the caller before each step; rows from
 * cu[B] on are padding: label -1, finite inputs).  No pad slots inside dialogues, so a ragged batch (reference collate_fn,
 * src/dataset.py:69-89, pads every dialogue to the longest) costs its valid utterances only.  M2F_BUF_KEYPAD is not read.
 * Same arithmetic per valid utterance as the padded plan of the same dialogues; B <= T <= B * L. */
int64_t m2f_workspace_bytes_packed(const m2f_config* cfg, int B, int L, int T, int train);
m2f_plan* m2f_plan_create_packed(const m2f_config* cfg, int B, int L, int T, int precision, int train,
                                 float* params, float* grads, void* workspace, int64_t workspace_bytes,
                                 uint32_t* rng_state);
/* SHARED PARAMETER SHADOWS (bf16 mode).  The GEMMs stage bf16 copies of every 2-D parameter (W [rows][pad8(cols)] and W^T
 * [cols][pad8(rows)]); a plan of m2f_plan_create keeps its own copies in its workspace and refreshes them with cast launches at
 * the head of every forward.  With m2f_plan_create_shared all plans of a model use ONE caller-owned buffer of
 * m2f_param_shadow_elems(cfg) uint16 (256-byte aligned; initialise it once with m2f_param_shadow_init), and an optimizer step
 * through m2f_adam_step_shadowed writes the shadows of the parameters it has just updated.  The caller then declares them
 * current with m2f_plan_params_fresh(plan, 1) and the forward skips its parameter casts (2 x 87 us of 2.7 ms at C3); after
 * any OTHER write to the parameters (load_state_dict, a foreign optimizer) it must pass 0 again - a forward that ran the casts
 * leaves the shadows current, too.  T = 0: padded plan, T > 0: packed plan of T token rows (as m2f_plan_create_packed).
 * No counterpart in the reference: torch keeps no low-precision parameter copies (src/train.py:56,231 is all it does). */
int64_t m2f_param_shadow_elems(const m2f_config* cfg);
int m2f_param_shadow_init(const m2f_config* cfg, uint16_t* param_shadow, m2f_stream_t stream);
int64_t m2f_workspace_bytes_shared(const m2f_config* cfg, int B, int L, int T, int train);
m2f_plan* m2f_plan_create_shared(const m2f_config* cfg, int B, int L, int T, int precision, int train,
                                 float* params, float* grads, void* workspace, int64_t workspace_bytes,
                                 uint32_t* rng_state, uint16_t* param_shadow);
int m2f_plan_params_fresh(m2f_plan* plan, int fresh);
void m2f_plan_destroy(m2f_plan* plan);
void* m2f_plan_buffer(m2f_plan* plan, int which);
int m2f_plan_num_launches(m2f_plan* plan, int phase);   /* 0 fwd, 1 loss, 2 bwd */
/* Kept for callers of rounds 2-3, when a plan could run its launch lists as two persistent ("strip-dataflow") launches (measured
 * slower than the lists and removed in round 4; git history keeps csrc/mega.hip): always 0 = launch lists. */
int m2f_plan_persistent(m2f_plan* plan);
/* Status record of a plan's kernels: out8 is zeroed and 0 returned - no kernel of the launch lists waits on another workgroup, so
 * none can give up (the persistent kernels left their bounded-wait record here).  Fails on a NULL / destroyed plan. */
int m2f_plan_status(m2f_plan* plan, uint32_t* out8);

/* M2FNet.forward (src/model.py:102-145): inputs read from M2F_BUF_TEXT/AUDIO/KEYPAD, logits -> M2F_BUF_LOGITS. */
int m2f_forward(m2f_plan* plan, m2f_stream_t stream);
/* criterion(outputs.permute(0,2,1), emotion) (src/train.py:229; CrossEntropyLoss(ignore_index=-1,
 * label_smoothing) of :48-50): labels from M2F_BUF_LABELS, loss -> M2F_BUF_LOSS, dlogits -> M2F_BUF_DLOGITS.
 * normalise=1: gradient of the mean-over-valid loss; 0: gradient of the SUM (data-parallel path divides
 * by the global denominator after the all-reduce). */
int m2f_loss(m2f_plan* plan, float label_smoothing, int use_class_weights, int normalise, m2f_stream_t stream);
/* loss.backward() (src/train.py:230): consumes M2F_BUF_DLOGITS, OVERWRITES the flat gradient buffer. */
int m2f_backward(m2f_plan* plan, m2f_stream_t stream);
/* Fused train-step body of src/train.py:228-230 (forward + criterion + backward) with the dropout RNG
 * advanced on the device; use_graph=1 captures the launch list into a hipGraph once and replays it. */
int m2f_step(m2f_plan* plan, float label_smoothing, int use_class_weights, int normalise, int use_graph,
             m2f_stream_t stream);

/* The same step in TWO parts, for data-parallel overlap (no counterpart in the reference, which is single-process):
 *   part 0 = dropout-RNG advance + forward + criterion + the backward chain of the classifier and the fusion stack + every weight
 *            gradient whose operands that chain completes;   part 1 = the encoders' backward + the remaining weight gradients.
 * After part 0 the flat gradient buffer is final from element m2f_plan_split_offset(plan) on (fusion stack + classifier = its
 * tail, and the 64-float loss tail behind it), so a rank can put that bucket's all-reduce on the wire and run part 1 under it.
 * m2f_plan_split_offset returns 0 for plans that cannot be split (fp32 mode, eval plans); parts 0 and 1 must alternate. */
int64_t m2f_plan_split_offset(m2f_plan* plan);
int m2f_step_part(m2f_plan* plan, int part, float label_smoothing, int use_class_weights, int normalise, int use_graph,
                  m2f_stream_t stream);

/* Device-side dialogue batcher: Dataset.__getitem__ + collate_fn / apply_padding (src/dataset.py:32-89, src/utils.py:15-31)
 * on device-resident embedding tables.  Token slot t receives row rows[t] of each table; rows[t] < 0 marks a padded slot
 * (features 0, label -1, key_pad 1).  Outputs may be a plan's staging buffers (row strides ld_text / ld_audio). */
int m2f_gather_dialogues(const float* text_table, int d_text, const float* audio_table, int d_audio,
                         const int64_t* label_table, const int32_t* rows, int T, float* text_out, int ld_text,
                         float* audio_out, int ld_audio, uint8_t* key_pad_out, int64_t* labels_out, m2f_stream_t stream);

/* Measurement aid: one EAGER m2f_step with a hipEvent pair recorded on `stream` around every launch.  Fills, per
 * launch, kinds[] (0/1/2 = grouped GEMM forward/dgrad/wgrad form, 3/4 attention fwd/bwd, 5/6 LayerNorm fwd/bwd,
 * 7 dropout-mask, 8 criterion, 9 LayerNorm-parameter reduce, 10 bf16 cast / token-transpose copies, 11 / 12 unused (the removed persistent kernels);
 * chain launches carry + 32 x their part of the model: 0 modality encoders, 1 fusion stack (FusionAttentionModule, src/model.py:13-20), 2 classifier), ms[]
 * (device time) and flops[] (algorithmic FLOPs of the launch, 0 for row-wise kernels).  Synchronises the stream.  Returns the number of launches, or <0. */
int m2f_step_timed(m2f_plan* plan, float label_smoothing, int use_class_weights, int normalise, m2f_stream_t stream,
                   int max_entries, int* kinds, float* ms, double* flops);

/* Calibration of m2f_step_timed's intervals: the mean device time between the two hipEventRecords of a pair with
 * NOTHING between them (*empty_pair_ms) and with a one-thread kernel between them (*trivial_kernel_pair_ms; needs
 * scratch_rng_state = 4 device uint32, may be NULL to skip), over `pairs` pairs on `stream`.  rocprofv3 reports the
 * kernel's own begin..end; an event interval adds this record/dispatch overhead to it.  Synchronises the stream. */
int m2f_event_overhead(uint32_t* scratch_rng_state, int pairs, float* empty_pair_ms, float* trivial_kernel_pair_ms,
                       m2f_stream_t stream);

/* Advances the dropout RNG state by one step on the device (what nn.Dropout's generator advance is to the
 * reference; m2f_step does it itself). */
int m2f_rng_advance(uint32_t* rng_state, m2f_stream_t stream);

/* optimizer.step() of torch.optim.Adam(lr, weight_decay) (src/train.py:56,231): coupled L2, bias-corrected,
 * over flat buffers of n floats (n % 4 == 0).  grad_scale_ptr (device, nullable): g <- g / *grad_scale_ptr. */
int m2f_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  const float* grad_scale_ptr, m2f_stream_t stream);

/* Same update with the gradients given as bf16 (n values, 8-byte aligned): the data-parallel path can exchange
 * gradients in bf16 over xGMI (half the bytes of the fp32 all-reduce) and feed the reduced buffer straight to the
 * optimizer; parameters and both moments stay fp32.  No counterpart in the reference (single process). */
int m2f_adam_step_g16(float* params, const uint16_t* grads_bf16, float* exp_avg, float* exp_avg_sq, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                      const float* grad_scale_ptr, m2f_stream_t stream);

/* The same optimizer step over the WHOLE flat buffers of a model (cfg gives the tensor table), walking the 2-D parameters in
 * 64 x 64 tiles so that the kernel also writes their bf16 shadows (W and W^T) into the shared buffer of m2f_param_shadow_init:
 * 28 B of optimizer traffic + 4 B of shadow writes per parameter instead of 28 B + a separate 8 B cast pass per forward. */
int m2f_adam_step_shadowed(const m2f_config* cfg, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                           uint16_t* param_shadow, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                           const float* grad_scale_ptr, m2f_stream_t stream);
/* ... over the parameter tensors at flat offsets [first, end) only (both the offset of a tensor; end < 0: to the last one), reading the
 * gradients as fp32 or (grads_bf16 != 0) as bf16 with the same indexing.  This is what lets the data-parallel path - which steps
 * bucket by bucket behind each bucket's all-reduce, on the reduced bf16 buffer when the exchange is bf16 - keep the parameter
 * shadows current as well (dp.GradReducer aligns its buckets to tensor boundaries; optim.FusedAdam.step_ranges). */
int m2f_adam_step_shadowed_range(const m2f_config* cfg, float* params, const void* grads, int grads_bf16, float* exp_avg,
                                 float* exp_avg_sq, uint16_t* param_shadow, int64_t first, int64_t end, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, int step, const float* grad_scale_ptr, m2f_stream_t stream);

/* bf16-mode plans write every activation twice - fp32 and the bf16 shadow the GEMMs / attention kernels stage from.  When a plan
 * is built, the readers of every workspace buffer are enumerated from its final launch lists; a copy nobody reads is not written
 * (fp32 of QKV projections, attention outputs, their gradients and the FFN hidden gradients; the shadows of results that are only
 * residual terms or LayerNorm inputs), and its buffer is filled with NaNs once so
 * that an unknown reader cannot go unnoticed.  Returns how many copies this plan skips (0: fp32 mode, or M2F_SKIP_F32=0 when the
 * plan was built).  Results are bit-identical either way.  No counterpart in the reference (autocast keeps one copy per tensor). */
int m2f_plan_skipped_copies(m2f_plan* plan);

/* ---- in-loop text encoder (BASELINE config C5) -------------------------------------------------
 * The reference computes its text embeddings with transformers' RobertaModel (src/feature_extractors/text/model.py:16-21,
 * [CLS] pooling at text/embeddings.py:83) in a separate stage; these entry points are the pieces that model needs beyond
 * the GEMM / LayerNorm kernels below, so the encoder can run in the training loop on the same device buffers. */

/* Outputs of m2f_gemm / m2f_layernorm_fwd / m2f_embed_layernorm / m2f_attention_long_fwd that lie inside
 * [ws_base, ws_base + floats) are ALSO written as bf16 at the same element index of `shadow` (the operand copies the
 * bf16 GEMM stages from).  NULL, NULL, 0 switches it off.  Thread-local. */
int m2f_set_shadow_map(const float* ws_base, uint16_t* shadow, int64_t floats);

/* RobertaEmbeddings.forward in eval mode: out[t] = LayerNorm(word_emb[input_ids[t]] + pos_emb[position_ids[t]] +
 * token_type_emb[0]) for T tokens of width d (d % 4 == 0, d <= 2048). */
int m2f_embed_layernorm(int T, int d, const int64_t* input_ids, const int64_t* position_ids, const float* word_emb,
                        const float* pos_emb, const float* type_emb_row0, const float* gamma, const float* beta, float eps,
                        float* out, int ld_out, m2f_stream_t stream);

/* Token-level multi-head self-attention, forward only, any sequence length S (RobertaSelfAttention in eval mode):
 * q/k/v rows are tokens t = b*S + i, head h in columns [h*hd, (h+1)*hd), hd <= 128; key_pad [B, S] (1 = padded key,
 * nullable); softmax(q k^T / sqrt(hd) + mask) v with an online softmax over 64-key blocks. */
int m2f_attention_long_fwd(int B, int S, int H, int hd, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                           const uint8_t* key_pad, float* out, int ldo, m2f_stream_t stream);

/* The same attention on bf16 operands (round 4; the encoder's bf16 mode): q / k / v are the bf16 result of the packed projection
 * GEMM as it is (leading dimensions in elements; hd, the leading dimensions and the addresses multiples of 8 elements), the products run on
 * v_mfma_f32_16x16x16_bf16 with fp32 accumulation and an fp32 online softmax, probabilities rounded to bf16 for the P V product (the
 * denominator sums the rounded values); out16 (bf16) is always written, out32 (fp32, same indexing) when not NULL.  hd <= 128. */
int m2f_attention_long_fwd_bf16(int B, int S, int H, int hd, const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v,
                                int ldv, const uint8_t* key_pad, uint16_t* out16, float* out32, int ldo, m2f_stream_t stream);

/* ... with a third output (nullable like the others; at least one must be given): out8 = OCP e4m3 bytes of value * out8_scale,
 * saturating at +-448 - the operand the fp8 output projection stages (no fp32 copy, no quantise pass); needs hd and ldo in multiples of 16. */
int m2f_attention_long_fwd_bf16_out8(int B, int S, int H, int hd, const uint16_t* q, int ldq, const uint16_t* k, int ldk,
                                     const uint16_t* v, int ldv, const uint8_t* key_pad, uint16_t* out16, float* out32, uint8_t* out8,
                                     float out8_scale, int ldo, m2f_stream_t stream);

/* Diagnostic (tools/ln_stats_ab.py): one LayerNorm-forward launch over 1..4 problems of T rows, as the plans merge them; pre = 1 reads
 * (mean, rstd) from `stats` instead of computing them - the cost of a LayerNorm whose statistics came out of the preceding GEMM's epilogue. */
int m2f_layernorm_fwd_diag(int T, int n_prob, const int* d, const float* const* x, const float* const* gamma, const float* const* beta, float* const* out,
                           float* const* stats, float eps, int pre, m2f_stream_t stream);

/* m2f_layernorm_fwd that ALSO writes its result as e4m3(value * out8_scale), saturating, into out8 [T, d] (d % 4 == 0): the fp8 text
 * encoder's LayerNorm outputs are GEMM operands (round 4: replaces a quantise pass over the fp32 result). */
int m2f_layernorm_fwd_out8(int T, int d, const float* x, const float* gamma, const float* beta, const float* res, float* out,
                           float* stats, float eps, uint8_t* out8, float out8_scale, m2f_stream_t stream);

/* Results of the following m2f_gemm calls of this thread that have a bf16 shadow (m2f_set_shadow_map) have NO fp32 reader: kernels
 * that know how (the chip-filling bf16 forms) write the shadow only and leave the fp32 buffer untouched; edge tiles and the other
 * forms still write both.  0 switches it off.  (The plans decide this per buffer from their launch lists: m2f_plan_skipped_copies.) */
int m2f_set_shadow_only(int on);

/* fp8 GEMM of the in-loop text encoder (BASELINE C5 asks for fp8 MFMA): C[M,N] = act(acc_scale * A8 B8^T + bias) + res with
 * A8 [M,K], B8 [N,K] row-major OCP e4m3 bytes (K, lda, ldb multiples of 16; 16-byte aligned), fp32 accumulate on
 * v_mfma_f32_32x32x16_fp8_fp8; acc_scale = 1 / (scale_a * scale_b) undoes the per-tensor quantisation scales.
 * activation: 0 none, 1 ReLU, 2 GELU.  c8 (nullable): the result is written as e4m3(result * c8_scale) at c8[m*ldc + n]
 * INSTEAD of fp32 c (an activation whose only reader is the next fp8 GEMM, e.g. the FFN hidden layer).  Forward only. */
int m2f_gemm_fp8(int M, int N, int K, const uint8_t* a8, int lda, const uint8_t* b8, int ldb, float acc_scale, float* c, int ldc,
                 const float* bias, const float* res, int ldres, int activation, uint8_t* c8, float c8_scale,
                 m2f_stream_t stream);

/* dst[i] = e4m3(clamp(src[i] * scale, +-448)), n % 4 == 0: operand quantisation for m2f_gemm_fp8. */
int m2f_quantize_fp8(const float* src, uint8_t* dst, int64_t n, float scale, m2f_stream_t stream);

/* ---- kernel-level entry points (used by the parity tests; same kernels the plan launches) ---------- */
/* Number of bf16 GEMM launches this process has issued in the RING form (csrc/gemm.hip, m2f_gemm16_ring_kernel: LDS-direct
 * operand ring, 128x128 tiles; taken by k-contiguous launches of at least M2F_RING_MIN = 200 such tiles unless M2F_RING=0).
 * Diagnostic: lets a test assert that the form it means to check actually ran. */
long long m2f_gemm_ring_launches(void);

/* C[M,N] = epilogue(A x B); layout 0: C = A[M,K] B[N,K]^T (nn.Linear forward), 1: C = A[M,K] B[K,N]
 * (input gradient), 2: C = A[K,M]^T B[K,N] (weight gradient; bias_grad[M] = column sums of A).
 * Optional second operand segment (a1/b1, k1) = never-materialised torch.cat along the reduction dim. */
int m2f_gemm(int precision, int layout, int M, int N, int K0, int K1,
             const float* a0, int lda0, const float* a1, int lda1,
             const float* b0, int ldb0, const float* b1, int ldb1,
             float* c, int ldc, const float* bias, const float* res, int ldres,
             const float* gate, int ldgate, float gate_scale, float* bias_grad,
             int relu_a, int relu_b, int relu_out /* 0 none, 1 ReLU, 2 exact GELU */, int accumulate,
             uint32_t drop_site, float drop_p, const uint32_t* rng_state, int tile,
             float* splitk_ws, uint32_t* splitk_tickets, int splitk_max_tiles,
             const uint16_t* a0_bf16, int lda0_bf16, const uint16_t* a1_bf16, int lda1_bf16,
             const uint16_t* b0_bf16, int ldb0_bf16, const uint16_t* b1_bf16, int ldb1_bf16, m2f_stream_t stream);
/* a*_bf16 / b*_bf16 (nullable): bf16 copies of the operands (same logical elements; leading dimensions multiples of 8,
 * pad columns zero).  In bf16 mode a launch whose operands all have one stages from them (half the bytes per CU).
 * splitk_ws / splitk_tickets (nullable): scratch for in-launch split-K of launches too small to fill the chip:
 * splitk_max_tiles * 4 * 64*64 floats and splitk_max_tiles ZEROED uint32 tickets (re-armed by the kernel). */
/* ---- optimizer inside the step (round 4) ----------------------------------------------------------------------
 * torch.optim.Adam.step (src/train.py:56,231) applied where the weight gradient is born: the weight-gradient launch of a bf16 train
 * plan (eight-phase table form, M2F_TABLE_TILE=132) updates p / exp_avg / exp_avg_sq and both bf16 parameter shadows of the elements
 * whose dW it holds in registers, and one launch of the shadow-writing Adam kernel updates everything else (biases, LayerNorm, the
 * few matrices outside the table) - all inside m2f_step's captured graph.  dW of the table's matrices is NOT written to `grads`.
 * Same arithmetic as m2f_adam_step_shadowed on the same gradients: bit-identical parameters, moments and shadows.
 * setup: buffers as for m2f_adam_step_shadowed (params = the plan's parameter buffer, param_shadow = the buffer the plan was created
 * with); hyper_dev = 8 device floats refreshed by m2f_adam_hyper BEFORE every step (lr, betas, eps, weight decay, step count t >= 1:
 * lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t) change every step and a replayed graph cannot take them as arguments);
 * grad_scale_ptr (nullable): device scalar the gradients are divided by (m2f_step(normalise = 0)).
 * m2f_plan_fused_adam(plan, 1 | 0) switches the form of the NEXT m2f_step (re-captures the graph on a change). */
int m2f_plan_fused_adam_setup(m2f_plan* plan, float* params, float* exp_avg, float* exp_avg_sq, uint16_t* param_shadow,
                              const float* hyper_dev, const float* grad_scale_ptr);
int m2f_plan_fused_adam(m2f_plan* plan, int on);
int m2f_adam_hyper(float* hyper_dev, float lr, float beta1, float beta2, float eps, float weight_decay, int step, m2f_stream_t stream);

/* Gradients left as bf16 (round 4; the data-parallel bf16 exchange, multimodal-emotion-recognition_amd/dp.py): after m2f_plan_grad_bf16(plan, g16)
 * a step writes EVERY gradient, rounded once to bf16, at its element index of g16 (n_params uint16, 16-byte aligned) - the weight
 * gradients of the table launch directly (no fp32 dW: -2 bytes per parameter written, and no rounding pass over the fp32 buffer before
 * the all-reduce), all others through one cast launch behind the backward.  The fp32 buffer then holds only those others (and the loss
 * tail).  m2f_plan_grad_bf16(plan, NULL) restores fp32 gradients.  Same bits as rounding the fp32 gradients of a plain step. */
int m2f_plan_grad_bf16(m2f_plan* plan, uint16_t* grads_bf16);

/* The 256x256-tile bf16 GEMM on the eight-phase schedule (csrc/gemm_p8.h; round 4), bf16 operands handed over directly - the kernel
 * the weight-gradient table launch (rc = 1) and the text encoder's launches (rc = 0) run, for kernel-level tests and measurements.
 *   rc = 0: C[M,N] = act(A[M,K] B[N,K]^T + bias) + res   (nn.Linear forward: src/feature_extractors/text/model.py:16-21's encoder
 *           layers); K % 64 == 0; act 0 none, 1 ReLU, 2 GELU
 *   rc = 1: C[M,N] = A[K,M]^T B[K,N]   (weight gradient dW = dY^T X of every nn.Linear in src/model.py: reduction over the token rows),
 *           relu_a / relu_b on the operands, bias_grad[M] = column sums of A (nullable); runs as a one-problem table launch whose table,
 *           tile records and per-workgroup ranges are written to `scratch` (device memory, >= 64 KiB + 4 bytes per tile) for n_wg
 *           workgroups (<= 0: 256); scratch_bytes < 0: `scratch` (of -scratch_bytes bytes) still holds the tables of an identical earlier call.
 * Returns 0, or < 0 when the shape / alignment is not this kernel's (no fallback). */
int m2f_gemm_p8(int rc, int M, int N, int K, const uint16_t* a, int lda, const uint16_t* b, int ldb, float* c, int ldc,
                const float* bias, const float* res, int ldres, int act, int relu_a, int relu_b, float* bias_grad,
                void* scratch, int64_t scratch_bytes, int n_wg, m2f_stream_t stream);
/* softmax(q k^T / sqrt(hd) + key_padding_mask) v per (dialogue, head) (nn.MultiheadAttention inside
 * src/model.py:8,14,61,73); probs receives P^T per head, padded to Lp = 16*ceil(L/16). */
int m2f_attention_fwd(int B, int L, int H, int hd, const float* q, int ldq, const float* k, int ldk,
                      const float* v, int ldv, const uint8_t* key_pad, float* out, int ldo, float* probs,
                      uint32_t drop_site, float drop_p, const uint32_t* rng_state, m2f_stream_t stream);
int m2f_attention_bwd(int B, int L, int H, int hd, const float* q, int ldq, const float* k, int ldk,
                      const float* v, int ldv, const uint8_t* key_pad, const float* out, int ldo,
                      const float* probs, const float* dout, int lddo, float* dq, int lddq, float* dk,
                      int lddk, float* dv, int lddv, uint32_t drop_site, float drop_p,
                      const uint32_t* rng_state, m2f_stream_t stream);
int64_t m2f_attention_probs_elems(int B, int H, int L);
/* out = (res ? res : 0) + LayerNorm(x) (nn.LayerNorm, eps), stats[T,2] = (mean, rstd). */
int m2f_layernorm_fwd(int T, int d, const float* x, const float* gamma, const float* beta, const float* res,
                      float* out, float* stats, float eps, m2f_stream_t stream);
/* dx = LayerNorm backward (+extra); dgamma/dbeta via per-block partials (partial: [ceil(T/4), 2, d]). */
int m2f_layernorm_bwd(int T, int d, const float* x, const float* gamma, const float* stats, const float* dy,
                      const float* extra, float* dx, float* partial, float* dgamma, float* dbeta,
                      m2f_stream_t stream);
/* CrossEntropyLoss(ignore_index=-1, label_smoothing[, weight]) + gradient; loss_out[0..2] = loss, den, num. */
int m2f_cross_entropy(int T, int C, const float* logits, const int64_t* labels, const float* class_w,
                      float label_smoothing, int normalise, float* loss_terms, float* dlogits, float* loss_out,
                      m2f_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* M2FNET_HIP_H */
