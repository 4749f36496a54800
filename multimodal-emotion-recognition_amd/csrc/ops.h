// Internal launch interfaces of the gfx950 kernels (host side). The public C ABI is include/m2fnet_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ------------------------------------------------------------------------------------------------
// Grouped GEMM:  C[M,N] = epilogue( sum_k A(m,k) * B(n,k) )
//   operand layouts   KC: element (row, k) at p[row*ld + k]   (k contiguous)
//                     RC: element (row, k) at p[k*ld + row]   (row contiguous)
//   forward  Y  = X W^T        A = X  KC, B = W  KC
//   dgrad    dX = dY W         A = dY KC, B = W  RC   (reduction over W's row index)
//   wgrad    dW = dY^T X       A = dY RC, B = X  RC   (reduction over tokens)
//   Each operand may be the concatenation of two segments along k (cat(x, text) of the FAM layer,
//   src/model.py:16, is never materialised).
//   epilogue order: +bias[n] -> relu -> dropout(site) -> +res[m,n] -> *(gate[m,n] > 0 ? gate_scale : 0)
//                   -> (C += | C =)
// ------------------------------------------------------------------------------------------------
enum { M2F_PREC_F32 = 0, M2F_PREC_BF16 = 1 };
enum { M2F_LAYOUT_NT = 0, M2F_LAYOUT_NN = 1, M2F_LAYOUT_TN = 2 };   // fwd / dgrad / wgrad
enum {
    GF_RELU_A = 1, GF_RELU_B = 2, GF_RELU_OUT = 4, GF_ACCUM = 8,
    GF_VEC_A = 16, GF_VEC_B = 32,    // set by the launcher when 16-byte loads are legal
    GF_GELU_OUT = 64,                // exact (erf) GELU instead of ReLU in the epilogue (RoBERTa's intermediate.dense)
    GF_NO_BF16 = 256,                // bf16 mode: nobody stages C from its bf16 shadow (a residual term, a LayerNorm input): the ring epilogue skips it
    GF_NO_F32 = 128                  // bf16 mode: nobody reads C as fp32 (plan.hip::mark_unread_fp32) - a kernel that writes C's bf16 shadow
                                     // may leave the fp32 copy unwritten (the ring epilogue does; the other kernels ignore the flag)
    // (fp8 launches, m2f_launch_gemm_fp8: a.q / b.q point at e4m3 bytes, k / ldq count BYTE PAIRS, the accumulator is
    //  multiplied by acc_scale = 1 / (scale_a * scale_b) before the epilogue terms)
};

struct GemmOperand {
    const float* p[2];
    int ld[2];
    int k[2];          // reduction length of each segment (k[1] == 0: single segment)
    // bf16 shadow of the same data (bf16 mode): same logical element (row, k), own leading dimension; the columns
    // between the logical width and ldq are zero.  null = no shadow (the launch then stages from fp32).
    const uint16_t* q[2];
    int ldq[2];
    // B operand of the dgrad form only: bf16 shadow of the TRANSPOSED matrix (element (row, k) at qt[row*ldqt + k]),
    // which lets the launch run as the k-contiguous (forward) form.  null = not available.
    const uint16_t* qt[2];
    int ldqt[2];
};

// bf16 shadows of workspace activations: element i of the fp32 workspace has its shadow at shadow[i] (activation
// buffers use leading dimensions that are multiples of 8, so shadows keep 16-byte row alignment).  Producer kernels
// write both copies; ws_base == null disables shadow writes.
struct ShadowMap {
    const float* ws_base;
    uint16_t* shadow;
    size_t ws_floats;
};

struct GemmProblem {
    GemmOperand a, b;
    float* c;
    const float* bias;        // [N] or null
    const float* res;         // [M, N] (ldres) or null
    const float* gate;        // [M, N] (ldgate) or null
    float* bias_grad;         // wgrad only: [M] column sums of A over the reduction dim, or null
    int M, N, ldc, ldres, ldgate;
    float gate_scale;
    float acc_scale;          // fp8 launches only: de-quantisation factor applied to the accumulator
    uint8_t* c8;              // fp8 launches only (nullable): the result goes out as e4m3(v * c8_scale) at c8[m*ldc + n]
    float c8_scale;           //   INSTEAD of fp32 C (an activation whose only reader is the next fp8 GEMM)
    uint32_t drop_site;       // 0 = no dropout in the epilogue
    uint32_t flags;
    int tile_begin, tiles_n;  // filled by the launcher
    int splitk, slab_begin, cnt_begin;   // filled by the launcher (k-slices per tile, first partial slab, first ticket)
};

#define M2F_GEMM_MAX_PROBLEMS 8
// What a workgroup of the bf16-source kernel needs before it can issue its first load, packed at the FRONT of the kernarg
// block (filled by the launcher from pr[]): the tile -> problem search reads one 32-byte array instead of eight fields in
// eight different 256-byte structs, the producer's descriptors sit in 96 contiguous bytes instead of five cache lines of a
// cold kernarg segment.
struct GemmHot {
    const uint16_t* aq[2]; const uint16_t* bq[2];
    int M, N, k[2], ldaq[2], ldbq[2];
    uint32_t flags; int tile_begin; int has_bias_grad; int pad_[3];
};
struct GemmBatch {
    int tb[M2F_GEMM_MAX_PROBLEMS];           // tile_begin of problem i (INT_MAX for unused slots)
    GemmHot hot[M2F_GEMM_MAX_PROBLEMS];
    GemmProblem pr[M2F_GEMM_MAX_PROBLEMS];
    int count;
    const uint32_t* rng;      // dropout RNG state (device), may be null when no problem has drop_site
    uint32_t drop_thresh;     // p * 2^32
    float drop_scale;         // 1 / (1 - p)
    // split-K scratch (optional; null = never split): partial-tile slabs [splitk_max_tiles * 4][64*64] floats and one
    // zero-initialised ticket per tile (the last arriver re-arms it)
    float* splitk_ws;
    unsigned* splitk_cnt;
    int splitk_max_tiles;
    ShadowMap sh;             // C (when it lies in the workspace) is also written as bf16
    // TABLE form (m2f_launch_gemm_table): the problems live in device memory and a launch of at most chip-filling size
    // walks the whole tile list (persistent workgroups).  table[i] has tile_begin / tiles_n filled in for `table_tile`;
    // tile_prob[t] = index of the problem tile t belongs to.  pr[] / count are unused in this form.
    const GemmProblem* table;
    const uint16_t* tile_prob;
    int total_tiles;
    int table_tile;           // 64, 128, 256 (256x128) or 129 (128x128, ring form)
    // RING table forms (129 / 130): every workgroup walks its OWN tile list, tile_rec[wg_begin[b] .. wg_begin[b + 1]) for
    // workgroup b of a grid of wg_count; a record = problem | m-tile << 16 | n-tile << 24 (m2f_gemm_table_walk)
    const uint32_t* tile_rec;
    const int* wg_begin;
    int wg_count;
    // eight-phase form (gemm_p8.h): start skew - estimated cycles per k-tile (0 = off; bit 30: also workgroups without slack) and the
    // longest tile list of any workgroup of the launch
    int p8_skew, p8_max_tiles;
    // eight-phase table form with the optimizer in its epilogue (EPI 3, round 4): the tile that holds a weight gradient in registers
    // applies torch.optim.Adam's update to its parameter elements - dW never reaches memory.  Device pointer (scalar loads), null = off.
    // Per problem: res = (float*) bf16 shadow W of the problem's first element, gate = (float*) shadow W^T of it, ldres / ldgate their leading
    // dimensions (pad8(cols), pad8(rows of the whole tensor)).
    const struct M2FAdamFuse* adam;
};
struct M2FAdamFuse {
    float* p; const float* g_base; float* m; float* v;      // flat parameter / gradient / moment buffers, same indexing; the problems' c points into g_base
    const float* hyper;                                     // device: lr / bc1, beta1, beta2, eps, weight_decay, 1 / sqrt(bc2) (m2f_launch_adam_hyper)
    const float* gs_ptr;                                    // nullable device scalar: gradients are divided by it (global valid-utterance denominator)
};
#define M2F_SPLITK_MAX_TILES 512

// RING form of the k-contiguous bf16 GEMM (gemm_ring.h): bm x bn = 128x128 or 128x64; the table form walks gb.table.
int m2f_launch_gemm_skinny(const GemmBatch& gb, int prec, int layout, hipStream_t stream);      // skinny.hip: 1 launched, 0 not skinny, < 0 error
bool m2f_gemm_stages_bf16(const GemmBatch& gb, int layout);      // host: does the bf16-mode launch read bf16 shadows only?
bool m2f_gemm_ring_ok(const GemmBatch& gb);
bool m2f_gemm_ring256_ok(const GemmBatch& gb);       // 256x128 tiles: bias / ReLU / GELU / residual epilogues only
hipError_t m2f_launch_gemm_ring(GemmBatch& gb, int bm, int bn, hipStream_t stream);
hipError_t m2f_launch_gemm_ring_table(const GemmBatch& gb, hipStream_t stream);
// eight-phase 256x256 form (gemm_p8.h): forward-form launches with the 256x128 ring form's epilogue set, single segment, k % 64 == 0
bool m2f_gemm_p8_ok(const GemmBatch& gb);
hipError_t m2f_p8_launch_kc(GemmBatch& gb, hipStream_t stream);
hipError_t m2f_p8_launch_kc_fp8(GemmBatch& gb, hipStream_t stream);
hipError_t m2f_p8_launch_table_rc(const GemmBatch& gb, hipStream_t stream);
hipError_t m2f_p8_launch_table_rc_adam(const GemmBatch& gb, hipStream_t stream);     // Adam in the epilogue (gb.adam)

// Launches one grouped GEMM. Returns hipSuccess or the launch error. `tile` = 0 (auto), 64 or 128.
hipError_t m2f_launch_gemm(GemmBatch& gb, int prec, int layout, int tile, hipStream_t stream);
// TABLE form, bf16 mode, k-contiguous (NT) operands staged from gb.table[i].{a,b}.q.  Host-side preparation of a table:
// m2f_gemm_table_layout fills tile_begin / tiles_n of every problem for `tile` and returns the tile -> problem map.
hipError_t m2f_launch_gemm_table(const GemmBatch& gb, hipStream_t stream);
// fp8 (OCP e4m3) operands, forward form only, single problem: C = act(acc_scale * A8 B8^T + bias) + res.  K % 16 == 0,
// lda / ldb % 16 == 0, 16-byte aligned operands.  (SURVEY 8-f4 / BASELINE C5: the text encoder's GEMMs.)
hipError_t m2f_launch_gemm_fp8(GemmBatch& gb, hipStream_t stream);
#ifdef __cplusplus
#include <vector>
int m2f_gemm_table_layout(std::vector<GemmProblem>& prs, int tile, std::vector<uint16_t>& tile_prob, bool operand_options = false);
bool m2f_gemm_p8_table_ok(const std::vector<GemmProblem>& prs);      // table form of gemm_p8.h: no ReLU on A, single segment
// Per-workgroup tile lists of the ring table forms (tile_m x tile_n tiles: 128x128, 256x128, 256x256) for a grid of n_wg workgroups (workgroup b runs on XCD
// b % 8 under round-robin placement - speed only).  walk = 0: the order of m2f_gemm_table_layout (every round spreads 256
// consecutive tiles - usually of ONE problem - over all eight XCDs, so each L2 pulls its own copy of that problem's
// operands); walk = 1: the tile list is cut into eight contiguous ranges of whole 8 x 4 super-tiles, one per XCD, so a
// problem's operand panels are fetched by one L2 (two at a range boundary) and the 32 tiles an XCD multiplies at a time
// share 8 row panels and 4 column panels.  Returns the number of records (= total tiles), -1 if a problem has more than
// 255 tiles along a dimension.
int m2f_gemm_table_walk(const std::vector<GemmProblem>& prs, int walk, int n_wg, int tile_m, int tile_n, std::vector<uint32_t>& tile_rec, std::vector<int>& wg_begin,
                        const std::vector<int>* only = nullptr);
#endif

// ------------------------------------------------------------------------------------------------
// Attention (one wavefront per (dialogue, head); Q/K/V tiles staged in LDS, fp32 MFMA 16x16x4,
// wavefront-shuffle softmax).  Rows of q/k/v/out are token-major: token t = b*L + i.
// ------------------------------------------------------------------------------------------------
struct AttnProblem {
    const float* q; const float* k; const float* v;   // head h occupies columns [h*hd, (h+1)*hd)
    int ldq, ldk, ldv;
    float* out; int ldo;                               // [T, H*hd]
    float* probs;                                      // [B*H, Lp, Lp] saved P^T (pre-dropout), Lp = 16*ceil(L/16)
    // backward only
    const float* dout; int lddo;
    float* dq; float* dk; float* dv; int lddq, lddk, lddv;
    int H, hd;
    uint32_t drop_site;
    int block_begin;                                   // filled by the launcher
    uint32_t no_f32;                                   // bf16 mode: out (fwd) / dq, dk, dv (bwd) have no fp32 reader - only their bf16 shadows are written
};
#define M2F_ATTN_MAX_PROBLEMS 4
struct AttnBatch {
    int bb[M2F_ATTN_MAX_PROBLEMS];      // block_begin of problem i (INT_MAX for unused slots): one line for the block -> problem search
    AttnProblem pr[M2F_ATTN_MAX_PROBLEMS];
    int count;
    int B, L;
    const uint8_t* key_pad;    // [B, L], 1 = padded key
    const int* cu;             // PACKED layout (nullable): dialogue b owns token rows cu[b] .. cu[b+1]-1 (at most L of them); key_pad unused
    int T;                     // PACKED layout: token rows of the buffers; rows cu[B] .. T-1 belong to no dialogue and are written as zeros
    const uint32_t* rng;
    uint32_t drop_thresh;
    float drop_scale;
    ShadowMap sh;              // out (fwd) / dq, dk, dv (bwd) also written as bf16
    int bwd_fast;              // set by the launcher: LDS holds the fifth (O) slab of the one-round-trip backward path
    int bf16_math;             // bf16 mode, bit mask: 1 = the head-dim contractions (Q K^T, dO V^T) round their operands to bf16 and run
                               // on v_mfma_f32_16x16x32_bf16 (fp32 accumulate): 1/8 of the MFMA instructions at 1/2 the cycles each;
                               // 2 / 4 / 8 / 16 / 32 = the Q / K / V / dO / O slab is staged from the operand's bf16 shadow (half the bytes)
};
int m2f_attn_shadow_only_bits(const AttnBatch& ab, int pi, bool bwd);     // host: operands this launch stages from bf16 shadows only
hipError_t m2f_launch_attn_fwd(AttnBatch& ab, hipStream_t stream);
// Long-sequence forward (S unbounded, hd <= 128): token-level self-attention of the in-loop text encoder (inference).
// q/k/v rows are token-major (token t = b*S + i), head h in columns [h*hd, (h+1)*hd); key_pad [B, S] (1 = padded, nullable).
hipError_t m2f_launch_attn_long_fwd_bf16(const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v, int ldv,
                                         const uint8_t* key_pad, uint16_t* out16, float* out32, uint8_t* out8, float out8_scale, int ldo,
                                         int B, int S, int H, int hd, hipStream_t stream);
hipError_t m2f_launch_attn_long_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                    const uint8_t* key_pad, float* out, int ldo, int B, int S, int H, int hd, ShadowMap sh,
                                    hipStream_t stream);
hipError_t m2f_launch_attn_bwd(AttnBatch& ab, hipStream_t stream);
size_t m2f_attn_probs_elems(int B, int H, int L);

// ------------------------------------------------------------------------------------------------
// Row-wise kernels
// ------------------------------------------------------------------------------------------------
struct LnProblem {
    // forward: y = LN(x)*gamma + beta ; out = (res ? res : 0) + y ; optional dropout(out)
    const float* x; const float* gamma; const float* beta; const float* res;
    float* out; float* stats;          // stats [T, 2] = (mean, rstd)
    int d;
    int ld;                            // row stride of x / res / out / dy / extra / dx / dx_masked (0 = d)
    uint32_t drop_site;
    // backward: dx = LNbwd(dy) (+ extra) ; optional second output dx_masked = LNbwd(dy) * keep(site2)/(1-p)
    const float* dy; const float* extra;
    float* dx; float* dx_masked;
    float* partial;                    // [n_row_blocks, 2, d] partial (dgamma, dbeta)
    uint32_t drop_site2;
    int block_begin;
    // bf16 mode, set by plan.hip::mark_unread_fp32 - copies nobody reads: 1 = dx_masked as fp32 (only GEMMs stage it, from its
    // shadow), 2 = dx as bf16 (it is only a residual term / a LayerNorm input)
    uint32_t skip;
};
#define M2F_LN_MAX_PROBLEMS 4
struct LnBatch {
    int bb[M2F_LN_MAX_PROBLEMS];        // block_begin of problem i (INT_MAX for unused slots)
    LnProblem pr[M2F_LN_MAX_PROBLEMS];
    int count;
    int T;
    float eps;
    const uint32_t* rng; uint32_t drop_thresh; float drop_scale;
    ShadowMap sh;                      // out (fwd) / dx, dx_masked (bwd) also written as bf16
    int pre_stats;                     // forward, diagnostic (tools/ln_stats_ab.py): 1 = mean / rstd are READ from `stats` instead of computed (what a LayerNorm
                                       // whose statistics came out of the preceding GEMM's epilogue would cost)
    uint8_t* out8; float out8_scale;   // forward, problem 0 only (nullable): out ALSO as OCP e4m3 bytes of value * out8_scale, saturating, row stride d
                                       // (the operand the text encoder's fp8 GEMMs stage: no quantise pass; d % 4 == 0)
};
#define M2F_LN_ROWS_PER_BLOCK 4
hipError_t m2f_launch_ln_fwd(LnBatch& lb, hipStream_t stream);
hipError_t m2f_launch_ln_bwd(LnBatch& lb, hipStream_t stream);
static inline int m2f_ln_row_blocks(int T) { return (T + M2F_LN_ROWS_PER_BLOCK - 1) / M2F_LN_ROWS_PER_BLOCK; }

// dgamma/dbeta = sum over row blocks of the partials, for many LayerNorms in one launch.
struct LnReduceItem { const float* partial; float* dgamma; float* dbeta; int d; int nblk; };
#define M2F_LNRED_MAX_ITEMS 32
struct LnReduceBatch { LnReduceItem it[M2F_LNRED_MAX_ITEMS]; int count; };
hipError_t m2f_launch_ln_param_reduce(const LnReduceBatch& rb, hipStream_t stream);

// Criterion of src/train.py:48-50 on logits [T, C] (C <= 16): CrossEntropyLoss(ignore_index=-1,
// label_smoothing, optional class weights).  Per token it writes the loss numerator / denominator
// terms and the UNNORMALISED gradient d(sum of numerators)/dlogits.
struct CeArgs {
    const float* logits; int T, C;
    const int64_t* labels;                 // [T], ignore_index = -1
    const float* class_w;                  // [C] or null
    float label_smoothing;
    float* loss_terms;                     // [T, 2] (numerator, denominator)
    float* dlogits;                        // [T, C]
};
hipError_t m2f_launch_ce(const CeArgs& a, hipStream_t stream);
// loss_out[0] = num/den, loss_out[1] = den, loss_out[2] = num.  normalise != 0: dlogits *= 1/den
// (single-process mean-over-valid loss); normalise == 0 leaves the sum-gradient for the data-parallel
// path, which divides by the GLOBAL denominator after the all-reduce.
hipError_t m2f_launch_loss_finalize(const float* loss_terms, int T, int C, float* dlogits, float* loss_out,
                                    int normalise, hipStream_t stream);

// fp32 -> bf16 (round to nearest even) of up to M2F_CAST_MAX_ITEMS 2-D blocks in one launch: dst[r*ldd + c] =
// bf16(src[r*lds + c]) for c < cols (pad columns of dst are left untouched = zero).
struct CastItem { const float* src; uint16_t* dst; int rows, cols, lds, ldd; uint16_t* dst_t; int ldd_t; };   // dst_t: transposed copy [cols][rows] (nullable)
#define M2F_CAST_MAX_ITEMS 48
struct CastBatch { CastItem it[M2F_CAST_MAX_ITEMS]; int count; };
hipError_t m2f_launch_cast(const CastBatch& cb, hipStream_t stream);

// Token-transposed bf16 copies of activations for the weight-gradient GEMMs: dst[f*ldt + t] = bf16(relu?(src[t*ld + f]))
// for f < F, t < T; columns T..ldt-1 of dst are written as zero.  With both operands of dW = dY^T X stored
// [feature][token] the weight gradient runs as the k-contiguous (forward-form) GEMM.  colsum (nullable): [F] sums over
// tokens of the fp32 source = the bias gradient when src is dY (fixed summation order: deterministic).
// The item table lives in device memory; block_item[b] = item of workgroup b, items[i].block_begin = its first workgroup
// (one workgroup per 64 features).
struct TransItem { const float* src; int ld, F; uint16_t* dst; int ldt; float* colsum; int relu; int block_begin; };
struct TransBatch { const TransItem* items; const uint16_t* block_item; int blocks; int T; };
hipError_t m2f_launch_transpose_tokens(const TransBatch& tb, hipStream_t stream);

// Dialogue batcher (replaces Dataset.__getitem__ + collate_fn, reference src/dataset.py:32-89, on the device): token slot t
// takes row rows[t] of the two device-resident embedding tables (rows[t] < 0: padded slot -> zeros, label -1, key_pad 1).
struct GatherArgs {
    const float* text_table; const float* audio_table; const int64_t* label_table;
    const int32_t* rows;
    int T, d_text, d_audio, ld_text, ld_audio;
    float* text_out; float* audio_out; uint8_t* key_pad; int64_t* labels;
};
hipError_t m2f_launch_gather(const GatherArgs& a, hipStream_t stream);

// RoBERTa embeddings + LayerNorm (in-loop text encoder): out[t] = LN(word[ids[t]] + pos[pos_ids[t]] + type0)
hipError_t m2f_launch_embed_ln(const int64_t* ids, const int64_t* pos_ids, const float* word, const float* pos, const float* type0,
                               const float* gamma, const float* beta, float eps, float* out, int ld, int T, int d, ShadowMap sh,
                               hipStream_t stream);

// dst[i] = e4m3(clamp(src[i] * scale, +-448)) for n values (n % 4 == 0): operand quantisation of the fp8 GEMMs
hipError_t m2f_launch_quant_fp8(const float* src, uint8_t* dst, int64_t n, float scale, hipStream_t stream);

// in-place: x[t, c] *= keep(site, t*d + c) / (1 - p)
// (x2 != null: a second buffer of the same shape with its own site, same launch)
hipError_t m2f_launch_dropout_inplace2(float* x, float* x2, int T, int d, int ld, uint32_t site, uint32_t site2, const uint32_t* rng,
                                       uint32_t thresh, float scale, ShadowMap sh, hipStream_t stream);
hipError_t m2f_launch_dropout_inplace(float* x, int T, int d, int ld, uint32_t site, const uint32_t* rng,
                                      uint32_t thresh, float scale, ShadowMap sh, hipStream_t stream);
// rng.step += 1 (device side, graph-replay safe)
hipError_t m2f_launch_rng_advance(uint32_t* rng, hipStream_t stream);

// Fused Adam with coupled L2 (torch.optim.Adam semantics, src/train.py:56) over the flat buffers.
// grad_scale_ptr (device, may be null): gradients are multiplied by 1 / *grad_scale_ptr first (the
// global valid-utterance denominator under data parallelism).
// g_is_bf16: g points at bf16 gradients (data-parallel bf16 exchange) instead of fp32.
hipError_t m2f_launch_adam(float* p, const void* g, int g_is_bf16, float* m, float* v, int64_t n, float lr, float beta1,
                           float beta2, float eps, float weight_decay, int step, const float* grad_scale_ptr,
                           hipStream_t stream);

// Fused Adam + parameter-shadow refresh (bf16 mode, single process): the same update as m2f_launch_adam, walked matrix by matrix
// in 64x64 tiles so that the kernel that has the new fp32 parameter in registers also writes its bf16 shadows - W [rows][pad8(cols)]
// and W^T [cols][pad8(rows)] through an LDS tile - which the forward / input-gradient GEMMs stage from.  The cast launches at
// the head of the forward (8 B of traffic per parameter, 2 x 87 us at C3) disappear.  Items live in device memory:
// rows > 0: a 2-D parameter (tiles of 64 x 64); rows == 0: `cols` consecutive elements (1-D parameters incl. their pads; tiles
// of 4096 elements).
struct AdamItem { long long off, soff, soff_t; int rows, cols, tile_begin, tiles_c; };
#define M2F_ADAM_MAX_ITEMS 1024
hipError_t m2f_launch_adam_shadowed(float* p, const void* g, int g_is_bf16, float* m, float* v, uint16_t* shadow, const AdamItem* items,
                                    const int* tile_begin, int n_items, int tile_first, int total_tiles, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, int step, const float* grad_scale_ptr,
                                    hipStream_t stream);

hipError_t m2f_launch_cast_items(const float* src, uint16_t* dst, const AdamItem* items, const int* tile_begin, int n_items, int total_tiles, hipStream_t stream);
hipError_t m2f_launch_adam_hyper(float* hyper_dev, float lr, float beta1, float beta2, float eps, float weight_decay, int step, hipStream_t stream);
hipError_t m2f_launch_adam_shadowed_dev(float* p, const float* g, float* m, float* v, uint16_t* shadow, const AdamItem* items, const int* tile_begin,
                                        int n_items, int total_tiles, const float* hyper_dev, const float* grad_scale_ptr, hipStream_t stream);

#ifdef __HIPCC__
// shadow address of a workspace element, or null (no shadows / pointer outside the workspace, e.g. the gradient buffer)
__device__ __forceinline__ uint16_t* m2f_shadow_of(const ShadowMap& sh, const float* p) {
    if (!sh.ws_base) return nullptr;
    const ptrdiff_t i = p - sh.ws_base;
    return (i >= 0 && (size_t)i < sh.ws_floats) ? sh.shadow + i : nullptr;
}
#endif
