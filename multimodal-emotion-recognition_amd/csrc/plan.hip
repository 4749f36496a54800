// Plan builder / executor and the C ABI (include/m2fnet_hip.h) of the M2FNet training step on gfx950.
//
// A plan is the complete launch list of one step (forward, criterion, backward) for fixed
// (config, B, L, precision, train/eval), bound to caller-owned device buffers.  Built once, then replayed
// either eagerly or as one hipGraph.  MI355X-first decisions encoded here:
//   * the two modality branches (audio / text encoders) are independent until the fusion stack, so their
//     launch chains are MERGED pairwise into grouped launches (longest-common-subsequence alignment):
//     half the launches, twice the workgroups per launch on a chip with 256 CUs;
//   * every weight-gradient GEMM is DEFERRED: with 288 GB of HBM all dY / X operands stay resident, so the
//     ~85 wgrad problems run after the latency-bound input-gradient chain instead of being interleaved with
//     it - in bf16 mode as ONE persistent launch over a device-resident problem table, fed by one launch that
//     writes token-transposed bf16 copies of the operands (fp32 mode: a handful of grouped launches);
//   * torch.cat is never materialised (two-segment GEMM operands), bias/ReLU/dropout/residual/ReLU-gate
//     live in GEMM epilogues, dropout masks are regenerated from a counter RNG in backward.
// Math follows the reference: src/model.py:102-145 (M2FNet.forward), :13-20 (FusionAttentionModule),
// torch TransformerEncoderLayer (post-LN, ReLU, eps 1e-5, dim_feedforward 2048), src/train.py:48-50 (CE).
#include "../../include/m2fnet_hip.h"
#include "common.h"
#include "ops.h"

#ifndef M2F_TABLE_TILE_DEFAULT
#define M2F_TABLE_TILE_DEFAULT 132      // weight-gradient table launch: see build_plan (M2F_TABLE_TILE); 132 = 256 x 256 tiles, eight-phase schedule (round 4)
#endif

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return 1; }
int hipfail(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return 2;
}
#define M2F_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hipfail(e_, #x); } while (0)

// ---------------------------------------------------------------------------------------------------
// parameter map (mirror of layout.py / reference state_dict order)
// ---------------------------------------------------------------------------------------------------
struct EncLayerP { size_t in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b; };
struct ModalityP { std::vector<std::vector<EncLayerP>> enc; size_t norm_w = 0, norm_b = 0, proj_w = 0, proj_b = 0; };
struct FamP { size_t in_w, in_b, out_w, out_b, lin_w, lin_b; };
struct LinP { size_t w, b; };
struct ParamMap {
    ModalityP audio, text;
    std::vector<FamP> fam;
    std::vector<LinP> cls;               // Linear0, extra hidden..., last
    std::vector<int64_t> offsets, numels;
    size_t total = 0;
    struct Mat { size_t off; int rows, cols; size_t soff, soff_t; };   // 2-D tensors + offsets of their padded bf16 shadow
    std::vector<Mat> mats;                                             // and of the shadow of their transpose
    size_t shadow_elems = 0;
};

size_t pm_add(ParamMap& pm, size_t n, int rows = 0, int cols = 0) {
    const size_t off = pm.total;
    pm.offsets.push_back((int64_t)off);
    pm.numels.push_back((int64_t)n);
    pm.total = (off + n + 63) / 64 * 64;
    if (rows > 0 && cols > 0) {
        const size_t so = pm.shadow_elems;
        pm.shadow_elems += ((size_t)rows * ((cols + 7) & ~7) + 63) / 64 * 64;
        pm.mats.push_back({off, rows, cols, so, pm.shadow_elems});
        pm.shadow_elems += ((size_t)cols * ((rows + 7) & ~7) + 63) / 64 * 64;
    }
    return off;
}

void pm_modality(ParamMap& pm, ModalityP& m, int d, int ntrans, int nlayers, int dff, int dfam) {
    m.enc.resize(ntrans);
    for (int e = 0; e < ntrans; ++e) {
        for (int l = 0; l < nlayers; ++l) {
            EncLayerP p;
            p.in_w = pm_add(pm, (size_t)3 * d * d, 3 * d, d);
            p.in_b = pm_add(pm, (size_t)3 * d);
            p.out_w = pm_add(pm, (size_t)d * d, d, d);
            p.out_b = pm_add(pm, d);
            p.l1_w = pm_add(pm, (size_t)dff * d, dff, d);
            p.l1_b = pm_add(pm, dff);
            p.l2_w = pm_add(pm, (size_t)d * dff, d, dff);
            p.l2_b = pm_add(pm, d);
            p.n1_w = pm_add(pm, d);
            p.n1_b = pm_add(pm, d);
            p.n2_w = pm_add(pm, d);
            p.n2_b = pm_add(pm, d);
            m.enc[e].push_back(p);
        }
        if (e == 0) {                       // the final norm object is shared by all encoders of a modality
            m.norm_w = pm_add(pm, d);
            m.norm_b = pm_add(pm, d);
        }
    }
    m.proj_w = pm_add(pm, (size_t)dfam * d, dfam, d);
    m.proj_b = pm_add(pm, dfam);
}

int cls_in_width(const m2f_config& c) { return (c.audio_enabled && c.text_enabled) ? 2 * c.d_fam : c.d_fam; }

int check_config(const m2f_config& c) {
    if (!c.audio_enabled && !c.text_enabled) return fail("At least one of audio and text must be enabled!");
    if (c.fam_enabled && !(c.audio_enabled && c.text_enabled))
        return fail("Fusion Attention Module can only be used with both audio and text enabled!");
    if (c.audio_enabled && (c.nhead_audio < 1 || c.d_audio % c.nhead_audio)) return fail("AUDIO: embed_dim must be divisible by num_heads");
    if (c.text_enabled && (c.nhead_text < 1 || c.d_text % c.nhead_text)) return fail("TEXT: embed_dim must be divisible by num_heads");
    if (c.fam_enabled && (c.nhead_fam < 1 || c.d_fam % c.nhead_fam)) return fail("FAM: embed_dim must be divisible by num_heads");
    if (c.cls_out < 1 || c.cls_out > 16) return fail("CLASSIFIER.output_size must be in [1,16]");
    if (c.dropout < 0.f || c.dropout >= 1.f) return fail("dropout must be in [0,1)");
    if (c.dim_ff < 1 || c.cls_hidden < 1 || c.d_fam < 1) return fail("bad widths");
    const int dmax = std::max(std::max(c.audio_enabled ? c.d_audio : 0, c.text_enabled ? c.d_text : 0), c.d_fam);
    if (dmax > 2048) return fail("embedding sizes above 2048 are not supported by the LayerNorm kernels");
    return 0;
}

int build_param_map(const m2f_config& c, ParamMap& pm) {
    if (check_config(c)) return 1;
    if (c.audio_enabled) pm_modality(pm, pm.audio, c.d_audio, c.ntrans_audio, c.nlayers_audio, c.dim_ff, c.d_fam);
    if (c.text_enabled) pm_modality(pm, pm.text, c.d_text, c.ntrans_text, c.nlayers_text, c.dim_ff, c.d_fam);
    if (c.fam_enabled) {
        const size_t E = c.d_fam;
        for (int i = 0; i < c.nlayers_fam; ++i) {
            FamP f;
            f.in_w = pm_add(pm, 3 * E * E, (int)(3 * E), (int)E);
            f.in_b = pm_add(pm, 3 * E);
            f.out_w = pm_add(pm, E * E, (int)E, (int)E);
            f.out_b = pm_add(pm, E);
            f.lin_w = pm_add(pm, E * 2 * E, (int)E, (int)(2 * E));
            f.lin_b = pm_add(pm, E);
            pm.fam.push_back(f);
        }
    }
    const size_t h = c.cls_hidden, in = cls_in_width(c);
    LinP l0;
    l0.w = pm_add(pm, h * in, (int)h, (int)in);
    l0.b = pm_add(pm, h);
    pm.cls.push_back(l0);
    for (int j = 0; j < std::max(c.cls_layers - 2, 0); ++j) {
        LinP l;
        l.w = pm_add(pm, h * h, (int)h, (int)h);
        l.b = pm_add(pm, h);
        pm.cls.push_back(l);
    }
    LinP ll;
    ll.w = pm_add(pm, (size_t)c.cls_out * h, c.cls_out, (int)h);
    ll.b = pm_add(pm, c.cls_out);
    pm.cls.push_back(ll);
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// ops, launches
// ---------------------------------------------------------------------------------------------------
enum OpKind { OP_GEMM, OP_ATTN_FWD, OP_ATTN_BWD, OP_LN_FWD, OP_LN_BWD, OP_DROPOUT };

struct Op {
    int kind = OP_GEMM;
    int layout = M2F_LAYOUT_NT;
    std::vector<GemmProblem> gp;
    std::vector<AttnProblem> ap;
    std::vector<LnProblem> lp;
    float* dptr = nullptr; int dT = 0, dd = 0, dld = 0; uint32_t dsite = 0;      // OP_DROPOUT
    std::vector<std::pair<int, int>> src;       // (chain id, index in that chain) of the ops merged into this one
    int group = 0;                              // 0 modality encoders + projections, 1 fusion stack (reference src/model.py:13-20,129-131), 2 classifier
};

struct Launch {
    int kind = OP_GEMM;
    int layout = M2F_LAYOUT_NT;
    GemmBatch gb;
    AttnBatch ab;
    LnBatch lb;
    float* dptr = nullptr; int dT = 0, dd = 0, dld = 0; uint32_t dsite = 0;
    float* dptr2 = nullptr; uint32_t dsite2 = 0;      // OP_DROPOUT: a second buffer of the same shape in the same launch
    std::vector<std::pair<int, int>> src;
    int group = 0;
};

struct Arena {
    char* base = nullptr;
    size_t off = 0;
    template <typename T> T* alloc(size_t n) {
        off = (off + 255) / 256 * 256;
        T* p = reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return p;
    }
    float* f(size_t n) { return alloc<float>(n); }
};

GemmProblem gp_make(const float* a, int lda, const float* b, int ldb, int M, int N, int K, float* c, int ldc) {
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.a.p[0] = a; p.a.ld[0] = lda; p.a.k[0] = K;
    p.b.p[0] = b; p.b.ld[0] = ldb; p.b.k[0] = K;
    p.M = M; p.N = N; p.c = c; p.ldc = ldc; p.gate_scale = 1.f;
    return p;
}
void gp_seg2(GemmProblem& p, const float* a1, int lda1, const float* b1, int ldb1, int K1) {
    p.a.p[1] = a1; p.a.ld[1] = lda1; p.a.k[1] = K1;
    p.b.p[1] = b1; p.b.ld[1] = ldb1; p.b.k[1] = K1;
}
Op op_gemm(int layout, const GemmProblem& p) { Op o; o.kind = OP_GEMM; o.layout = layout; o.gp.push_back(p); return o; }

struct EncLayerBuf {
    const float* y_in; float *qkv, *probs, *att, *s1, *st1, *y1, *h, *s2, *st2, *y2;
    uint32_t site_attn, site_d1, site_ff, site_d2;
};
struct ModBuf {
    int d = 0, H = 0, nl = 0, nt = 0;
    const float* x_in = nullptr;
    std::vector<std::vector<EncLayerBuf>> L;
    std::vector<float*> xe, stf;
    float* xp = nullptr;
    uint32_t site_pre = 0, site_post = 0;
};
struct FamBuf { const float* t_in; float *qv, *k, *probs, *att, *x, *t_out; uint32_t site_attn, site_out; };

}  // namespace

struct m2f_plan {
    m2f_config cfg;
    int B = 0, L = 0, T = 0, prec = 0, train = 0;
    float* params = nullptr;
    float* grads = nullptr;
    bool packed = false;             // T token rows owned by B dialogues through cu_seqlens (m2f_plan_create_packed)
    int* cu = nullptr;               // device int32 [B + 1] (packed plans)
    uint32_t* rng = nullptr;
    uint32_t drop_thresh = 0;
    float drop_scale = 1.f;
    bool use_dropout = false;
    ParamMap pm;
    void* bufs[M2F_BUF_COUNT] = {nullptr};
    float* loss_terms = nullptr;
    // bf16 mode: shadows of the workspace activations (same element index) and of the 2-D parameters (padded rows)
    ShadowMap sh = {nullptr, nullptr, 0};
    uint16_t* wshadow = nullptr;
    std::vector<CastBatch> param_casts;  // bf16 mode, at the start of a forward: every 2-D parameter -> its W and W^T shadows ...
    std::vector<CastBatch> input_casts;  // ... and the two input staging buffers -> their activation shadows
    // The parameter shadows may live OUTSIDE the plan (m2f_plan_create_shared: one buffer shared by all plans of a model) and be
    // kept current by the optimizer itself (m2f_adam_step_shadowed); the caller then declares them fresh
    // (m2f_plan_params_fresh) and the forward skips the parameter casts.
    uint16_t* ext_wshadow = nullptr;
    bool params_fresh = false;
    std::vector<Launch> fwd, bwd;
    // deferred weight-gradient launches, run after the backward chain on the same stream.  (Overlapping them with the
    // chain on a second stream was measured SLOWER inside the captured graph: every fork costs the chain a 12-16 us
    // cross-queue gap and the chain's small GEMMs run ~2x longer next to a chip-filling launch; 2.85 -> 2.77 ms/step.)
    std::vector<Launch> wg;
    std::vector<Launch> wg_rest;     // weight gradients the table launch cannot take (row-major table form only)
    std::vector<CastBatch> wg_casts;  // ... and the narrow operands it takes as padded bf16 copies made in front of it
    // bf16 mode: all weight gradients as ONE persistent k-contiguous GEMM launch over a device-resident problem table,
    // fed by token-transposed bf16 copies of dY / X made by one transposing launch (which also sums the bias gradients)
    bool wg_nt = false;
    TransBatch wg_trans = {nullptr, nullptr, 0, 0};
    GemmBatch wg_tab;
    double wg_flops = 0.0;
    // SPLIT backward (m2f_step_part; data-parallel overlap): part 0 = forward + criterion + the classifier / fusion-stack backward
    // chain (bwd[0, bwd_head)) + the weight gradients whose operands that chain completes; part 1 = the encoders' backward +
    // the rest.  The fusion stack's and the classifier's gradients are the TAIL of the flat buffer (from split_offset on): their
    // all-reduce can travel under part 1.  Same table, two tile lists.
    size_t bwd_head = 0;
    bool split_ok = false;
    int n_no_f32 = 0;                    // outputs whose fp32 copy is not written (mark_unread_fp32)
    int64_t split_offset = 0;
    GemmBatch wg_tab_part[2];
    // Optimizer in the weight-gradient launch's epilogue (round 4; m2f_plan_fused_adam_setup / m2f_plan_fused_adam): the table launch in its
    // Adam form (gemm_p8.h EPI 3) + ONE launch of the shadow-writing Adam kernel over the tensors the table does not cover (1-D
    // parameters, weight gradients of wg_rest), both reading their step-dependent factors from device memory so that the captured
    // graph stays valid from step to step.
    std::vector<GemmProblem> tprobs_host;        // the table's problems as uploaded (host copy)
    GemmBatch wg_tab_adam;
    bool fused_ready = false, fused_on = false;
    void* fused_dev = nullptr;                   // one hipMalloc: M2FAdamFuse | GemmProblem[] | AdamItem[] | int tile_begin[]
    float* fz_p = nullptr; float* fz_m = nullptr; float* fz_v = nullptr; uint16_t* fz_sh = nullptr;
    const float* fz_hyper = nullptr; const float* fz_gs = nullptr;
    const AdamItem* fz_items = nullptr; const int* fz_tb = nullptr; int fz_n_items = 0, fz_tiles = 0;
    int g_fused = -1;
    // gradients left as bf16 (m2f_plan_grad_bf16): the table launch writes dW as bf16 into `g16` (same element index as the fp32 buffer),
    // one cast launch at the end of the backward rounds every other gradient into it
    GemmBatch wg_tab_g16;
    bool g16_ready = false, g16_on = false;
    void* g16_dev = nullptr;                     // GemmProblem[] | AdamItem[] | int tile_begin[]
    uint16_t* g16_buf = nullptr;
    const AdamItem* g16_items = nullptr; const int* g16_tb = nullptr; int g16_n_items = 0, g16_tiles = 0;
    int g_g16 = -1;
    size_t wg_rest_head = 0;             // wg_rest[0, wg_rest_head) belong to part 0
    hipGraphExec_t gexec_part[2] = {nullptr, nullptr};
    float gp_ls[2] = {0.f, 0.f}; int gp_cw[2] = {-1, -1}, gp_norm[2] = {-1, -1}, gp_fresh[2] = {-1, -1};
    std::vector<LnReduceBatch> lnred;
    size_t ws_used = 0;
    // graph cache for m2f_step
    hipGraphExec_t gexec = nullptr;
    float g_ls = 0.f; int g_cw = -1, g_norm = -1, g_fresh = -1;
    bool warmed = false;          // one eager step (sets kernel attributes) before the first capture
    ~m2f_plan() {
        if (gexec) (void)hipGraphExecDestroy(gexec);
        for (hipGraphExec_t g : gexec_part) if (g) (void)hipGraphExecDestroy(g);
        if (fused_dev) (void)hipFree(fused_dev);
        if (g16_dev) (void)hipFree(g16_dev);
    }
};

namespace {

struct Builder {
    m2f_plan& P;
    Arena ar;
    uint32_t next_site = 1;
    std::vector<Op> chain_f;                   // sequential forward ops after the branches (FAM, classifier)
    std::vector<Op> br_f[2], br_b[2];          // per-branch (0 audio, 1 text) forward / backward chains
    std::vector<Op> chain_b;                   // classifier + FAM backward (before the branches)
    std::vector<GemmProblem> wgrads;           // deferred weight-gradient problems (TN)
    std::vector<std::pair<int, int>> wdeps;    // per wgrad problem: (chain id, index) of the last op emitted before it
    int cur_chain = 2;                         // chain being built: 0 audio branch, 1 text branch, 2 head (chain_b)
    std::vector<LnReduceItem> lnitems;
    ModBuf mod[2];
    std::vector<FamBuf> fam;
    int T;

    explicit Builder(m2f_plan& p) : P(p), T(p.T) {}

    static int pad8(int w) { return (w + 7) & ~7; }     // activation leading dimensions: multiples of 8 (bf16 shadows stay 16-byte aligned)
    uint32_t site() { return P.use_dropout ? next_site++ : 0u; }
    const float* W(size_t off) const { return P.params + off; }
    float* G(size_t off) const { return P.grads ? P.grads + off : nullptr; }
    float gscale() const { return P.use_dropout ? P.drop_scale : 1.f; }

    Op op_ln_fwd(const float* x, size_t gw, size_t gb, const float* res, float* out, float* stats, int d, int ld, uint32_t s) {
        Op o; o.kind = OP_LN_FWD;
        LnProblem p; memset(&p, 0, sizeof(p));
        p.x = x; p.gamma = W(gw); p.beta = W(gb); p.res = res; p.out = out; p.stats = stats; p.d = d; p.ld = ld; p.drop_site = s;
        o.lp.push_back(p);
        return o;
    }
    // shared_partial != null: the LayerNorm object is shared (final encoder norm): the caller owns one partial
    // buffer for all its uses and registers a single reduce item covering every slice.
    Op op_ln_bwd(const float* x, size_t gw, size_t gb, const float* stats, const float* dy, const float* extra,
                 float* dx, float* dx_masked, int d, int ld, uint32_t s2, float* shared_partial = nullptr) {
        Op o; o.kind = OP_LN_BWD;
        LnProblem p; memset(&p, 0, sizeof(p));
        p.x = x; p.gamma = W(gw); p.stats = const_cast<float*>(stats); p.dy = dy; p.extra = extra; p.dx = dx;
        p.dx_masked = dx_masked; p.d = d; p.ld = ld; p.drop_site2 = s2;
        const int nblk = m2f_ln_row_blocks(T);
        p.partial = shared_partial ? shared_partial : ar.f((size_t)nblk * 2 * d);
        o.lp.push_back(p);
        if (!shared_partial) {
            LnReduceItem it; it.partial = p.partial; it.dgamma = G(gw); it.dbeta = G(gb); it.d = d; it.nblk = nblk;
            lnitems.push_back(it);
        }
        return o;
    }
    void wgrad(const float* dy, int lddy, const float* x, int ldx, int Nw, int Kw, size_t w_off, int ldw, long b_off,
               bool relu_b = false) {
        // dW[Nw, Kw] = dY[T, Nw]^T X[T, Kw]; bias grad = column sums of dY
        GemmProblem p = gp_make(dy, lddy, x, ldx, Nw, Kw, T, G(w_off), ldw);
        if (b_off >= 0) p.bias_grad = G((size_t)b_off);
        if (relu_b) p.flags |= GF_RELU_B;
        wgrads.push_back(p);
        const std::vector<Op>& cur = cur_chain == 2 ? chain_b : br_b[cur_chain];
        wdeps.push_back({cur_chain, (int)cur.size() - 1});
    }

    // ---------------- modality branch: encoders + projection --------------------------------------
    void build_modality(int bi, const ModalityP& mp, int d, int H, int nl, int nt, const float* x_in) {
        ModBuf& m = mod[bi];
        m.d = d; m.H = H; m.nl = nl; m.nt = nt; m.x_in = x_in;
        const int F = P.cfg.dim_ff, E = P.cfg.d_fam;
        const int dp = pad8(d), d3p = pad8(3 * d), Fp = pad8(F), Ep = pad8(E);   // activation leading dimensions
        std::vector<Op>& f = br_f[bi];
        const float* x = x_in;
        m.L.resize(nt);
        for (int e = 0; e < nt; ++e) {
            const float* y = x;
            for (int l = 0; l < nl; ++l) {
                const EncLayerP& p = mp.enc[e][l];
                EncLayerBuf b;
                b.y_in = y;
                b.qkv = ar.f((size_t)T * d3p);
                b.probs = ar.f(m2f_attn_probs_elems(P.B, H, P.L));
                b.att = ar.f((size_t)T * dp);
                b.s1 = ar.f((size_t)T * dp);
                b.st1 = ar.f((size_t)T * 2);
                b.y1 = ar.f((size_t)T * dp);
                b.h = ar.f((size_t)T * Fp);
                b.s2 = ar.f((size_t)T * dp);
                b.st2 = ar.f((size_t)T * 2);
                b.y2 = ar.f((size_t)T * dp);
                b.site_attn = site(); b.site_d1 = site(); b.site_ff = site(); b.site_d2 = site();
                {   // packed QKV in-projection
                    GemmProblem g = gp_make(y, dp, W(p.in_w), d, T, 3 * d, d, b.qkv, d3p);
                    g.bias = W(p.in_b);
                    f.push_back(op_gemm(M2F_LAYOUT_NT, g));
                }
                {
                    Op o; o.kind = OP_ATTN_FWD;
                    AttnProblem a; memset(&a, 0, sizeof(a));
                    a.q = b.qkv; a.k = b.qkv + d; a.v = b.qkv + 2 * d; a.ldq = a.ldk = a.ldv = d3p;
                    a.out = b.att; a.ldo = dp; a.probs = b.probs; a.H = H; a.hd = d / H; a.drop_site = b.site_attn;
                    o.ap.push_back(a);
                    f.push_back(o);
                }
                {   // out-proj -> dropout1 -> + residual
                    GemmProblem g = gp_make(b.att, dp, W(p.out_w), d, T, d, d, b.s1, dp);
                    g.bias = W(p.out_b); g.drop_site = b.site_d1; g.res = y; g.ldres = dp;
                    f.push_back(op_gemm(M2F_LAYOUT_NT, g));
                }
                f.push_back(op_ln_fwd(b.s1, p.n1_w, p.n1_b, nullptr, b.y1, b.st1, d, dp, 0));
                {   // linear1 -> relu -> dropout
                    GemmProblem g = gp_make(b.y1, dp, W(p.l1_w), d, T, F, d, b.h, Fp);
                    g.bias = W(p.l1_b); g.flags |= GF_RELU_OUT; g.drop_site = b.site_ff;
                    f.push_back(op_gemm(M2F_LAYOUT_NT, g));
                }
                {   // linear2 -> dropout2 -> + residual
                    GemmProblem g = gp_make(b.h, Fp, W(p.l2_w), F, T, d, F, b.s2, dp);
                    g.bias = W(p.l2_b); g.drop_site = b.site_d2; g.res = b.y1; g.ldres = dp;
                    f.push_back(op_gemm(M2F_LAYOUT_NT, g));
                }
                f.push_back(op_ln_fwd(b.s2, p.n2_w, p.n2_b, nullptr, b.y2, b.st2, d, dp, 0));
                y = b.y2;
                m.L[e].push_back(b);
            }
            // x_e = x_{e-1} + LNf(y); the last one feeds `dropout -> proj` (src/model.py:111,123)
            float* xe = ar.f((size_t)T * dp);
            float* stf = ar.f((size_t)T * 2);
            const uint32_t s = (e == nt - 1) ? (m.site_pre = site()) : 0u;
            f.push_back(op_ln_fwd(y, mp.norm_w, mp.norm_b, x, xe, stf, d, dp, s));
            m.xe.push_back(xe);
            m.stf.push_back(stf);
            x = xe;
        }
        m.xp = ar.f((size_t)T * Ep);
        m.site_post = site();
        {
            GemmProblem g = gp_make(x, dp, W(mp.proj_w), d, T, E, d, m.xp, Ep);
            g.bias = W(mp.proj_b); g.drop_site = m.site_post;
            f.push_back(op_gemm(M2F_LAYOUT_NT, g));
        }
    }

    // backward of a modality branch given d(xp) (already masked by the post-proj dropout)
    void build_modality_bwd(int bi, const ModalityP& mp, const float* dxp) {
        ModBuf& m = mod[bi];
        const int d = m.d, H = m.H, nl = m.nl, nt = m.nt, F = P.cfg.dim_ff, E = P.cfg.d_fam;
        const int dp = pad8(d), d3p = pad8(3 * d), Fp = pad8(F), Ep = pad8(E);
        std::vector<Op>& bw = br_b[bi];
        const int saved_chain = cur_chain;
        {   // the projection wgrad reads d(xp), produced by the head chain (everything emitted so far)
            cur_chain = 2;
        }
        const float* x_last = nt > 0 ? m.xe[nt - 1] : m.x_in;
        wgrad(dxp, Ep, x_last, dp, E, d, mp.proj_w, d, (long)mp.proj_b);
        cur_chain = bi;
        if (nt == 0) { cur_chain = saved_chain; return; }
        float* dx = ar.f((size_t)T * dp);
        {
            GemmProblem g = gp_make(dxp, Ep, W(mp.proj_w), d, T, d, E, dx, dp);
            g.drop_site = m.site_pre;
            bw.push_back(op_gemm(M2F_LAYOUT_NN, g));
        }
        const float* dxe = dx;
        const int nblk = m2f_ln_row_blocks(T);
        float* normp = ar.f((size_t)nt * nblk * 2 * d);       // one slice per use of the shared final norm
        {
            LnReduceItem it; it.partial = normp; it.dgamma = G(mp.norm_w); it.dbeta = G(mp.norm_b); it.d = d; it.nblk = nt * nblk;
            lnitems.push_back(it);
        }
        for (int e = nt - 1; e >= 0; --e) {
            const float* yN = nl > 0 ? m.L[e][nl - 1].y2 : (e > 0 ? m.xe[e - 1] : m.x_in);
            float* g_cur = ar.f((size_t)T * dp);
            bw.push_back(op_ln_bwd(yN, mp.norm_w, mp.norm_b, m.stf[e], dxe, nullptr, g_cur, nullptr, d, dp, 0,
                                   normp + (size_t)e * nblk * 2 * d));
            const float* gy = g_cur;
            const bool need_in = e > 0;          // gradient w.r.t. this encoder's input is needed
            for (int l = nl - 1; l >= 0; --l) {
                const EncLayerP& p = mp.enc[e][l];
                const EncLayerBuf& b = m.L[e][l];
                const bool first = (l == 0);
                // LN2
                float* ds2 = ar.f((size_t)T * dp);
                float* ds2m = b.site_d2 ? ar.f((size_t)T * dp) : nullptr;
                bw.push_back(op_ln_bwd(b.s2, p.n2_w, p.n2_b, b.st2, gy, nullptr, ds2, ds2m, d, dp, b.site_d2));
                const float* ds2g = ds2m ? ds2m : ds2;
                wgrad(ds2g, dp, b.h, Fp, d, F, p.l2_w, F, (long)p.l2_b);
                float* dh = ar.f((size_t)T * Fp);
                {
                    GemmProblem g = gp_make(ds2g, dp, W(p.l2_w), F, T, F, d, dh, Fp);
                    g.gate = b.h; g.ldgate = Fp; g.gate_scale = b.site_ff ? P.drop_scale : 1.f;
                    bw.push_back(op_gemm(M2F_LAYOUT_NN, g));
                }
                wgrad(dh, Fp, b.y1, dp, F, d, p.l1_w, d, (long)p.l1_b);
                float* dy1 = ar.f((size_t)T * dp);
                {
                    GemmProblem g = gp_make(dh, Fp, W(p.l1_w), d, T, d, F, dy1, dp);
                    g.res = ds2; g.ldres = dp;
                    bw.push_back(op_gemm(M2F_LAYOUT_NN, g));
                }
                // LN1 (+ the encoder-level residual d x_{e-1} += d x_e on the first layer)
                const float* extra = (first && need_in) ? dxe : nullptr;
                float* ds1 = ar.f((size_t)T * dp);
                float* ds1m = (b.site_d1 || extra) ? ar.f((size_t)T * dp) : nullptr;
                bw.push_back(op_ln_bwd(b.s1, p.n1_w, p.n1_b, b.st1, dy1, extra, ds1, ds1m, d, dp, b.site_d1));
                const float* ds1g = ds1m ? ds1m : ds1;
                wgrad(ds1g, dp, b.att, dp, d, d, p.out_w, d, (long)p.out_b);
                float* datt = ar.f((size_t)T * dp);
                bw.push_back(op_gemm(M2F_LAYOUT_NN, gp_make(ds1g, dp, W(p.out_w), d, T, d, d, datt, dp)));
                float* dqkv = ar.f((size_t)T * d3p);
                {
                    Op o; o.kind = OP_ATTN_BWD;
                    AttnProblem a; memset(&a, 0, sizeof(a));
                    a.q = b.qkv; a.k = b.qkv + d; a.v = b.qkv + 2 * d; a.ldq = a.ldk = a.ldv = d3p;
                    a.out = b.att; a.ldo = dp; a.probs = b.probs; a.H = H; a.hd = d / H; a.drop_site = b.site_attn;
                    a.dout = datt; a.lddo = dp;
                    a.dq = dqkv; a.dk = dqkv + d; a.dv = dqkv + 2 * d; a.lddq = a.lddk = a.lddv = d3p;
                    o.ap.push_back(a);
                    bw.push_back(o);
                }
                wgrad(dqkv, d3p, b.y_in, dp, 3 * d, d, p.in_w, d, (long)p.in_b);
                if (!first || need_in) {
                    float* gn = ar.f((size_t)T * dp);
                    GemmProblem g = gp_make(dqkv, d3p, W(p.in_w), d, T, d, 3 * d, gn, dp);
                    g.res = ds1; g.ldres = dp;            // ds1 already carries the encoder-level residual
                    bw.push_back(op_gemm(M2F_LAYOUT_NN, g));
                    gy = gn;
                }
            }
            dxe = gy;
        }
        cur_chain = saved_chain;
    }

    // ---------------- fusion stack + classifier ------------------------------------------------------
    void build_head() {
        const m2f_config& c = P.cfg;
        const int E = c.d_fam, Hf = c.nhead_fam;
        const int Ep = pad8(E), E2p = pad8(2 * E);
        const bool a_on = c.audio_enabled, t_on = c.text_enabled;
        const float* a = a_on ? mod[0].xp : nullptr;
        const float* t = t_on ? mod[1].xp : nullptr;
        if (c.fam_enabled) {
            // all K projections read the same (layer-invariant) audio tensor: one grouped launch per 8 layers
            std::vector<float*> kbuf;
            for (int i = 0; i < c.nlayers_fam; ++i) kbuf.push_back(ar.f((size_t)T * Ep));
            for (int i0 = 0; i0 < c.nlayers_fam; i0 += M2F_GEMM_MAX_PROBLEMS) {
                Op o; o.kind = OP_GEMM; o.layout = M2F_LAYOUT_NT;
                for (int i = i0; i < std::min(c.nlayers_fam, i0 + M2F_GEMM_MAX_PROBLEMS); ++i) {
                    const FamP& p = P.pm.fam[i];
                    GemmProblem g = gp_make(a, Ep, W(p.in_w + (size_t)E * E), E, T, E, E, kbuf[i], Ep);
                    g.bias = W(p.in_b + E);
                    o.gp.push_back(g);
                }
                chain_f.push_back(o);
            }
            for (int i = 0; i < c.nlayers_fam; ++i) {
                const FamP& p = P.pm.fam[i];
                FamBuf b;
                b.t_in = t;
                b.qv = ar.f((size_t)T * E2p);
                b.k = kbuf[i];
                b.probs = ar.f(m2f_attn_probs_elems(P.B, Hf, P.L));
                b.att = ar.f((size_t)T * Ep);
                b.x = ar.f((size_t)T * Ep);
                b.t_out = ar.f((size_t)T * Ep);
                b.site_attn = site(); b.site_out = site();
                {   // q = t Wq^T + bq -> qv[:, :E] ; v = t Wv^T + bv -> qv[:, E:]
                    Op o; o.kind = OP_GEMM; o.layout = M2F_LAYOUT_NT;
                    GemmProblem gq = gp_make(t, Ep, W(p.in_w), E, T, E, E, b.qv, E2p);
                    gq.bias = W(p.in_b);
                    GemmProblem gv = gp_make(t, Ep, W(p.in_w + (size_t)2 * E * E), E, T, E, E, b.qv + E, E2p);
                    gv.bias = W(p.in_b + 2 * E);
                    o.gp.push_back(gq); o.gp.push_back(gv);
                    chain_f.push_back(o);
                }
                {
                    Op o; o.kind = OP_ATTN_FWD;
                    AttnProblem at; memset(&at, 0, sizeof(at));
                    at.q = b.qv; at.ldq = E2p; at.k = b.k; at.ldk = Ep; at.v = b.qv + E; at.ldv = E2p;
                    at.out = b.att; at.ldo = Ep; at.probs = b.probs; at.H = Hf; at.hd = E / Hf; at.drop_site = b.site_attn;
                    o.ap.push_back(at);
                    chain_f.push_back(o);
                }
                {
                    GemmProblem g = gp_make(b.att, Ep, W(p.out_w), E, T, E, E, b.x, Ep);
                    g.bias = W(p.out_b);
                    chain_f.push_back(op_gemm(M2F_LAYOUT_NT, g));
                }
                {   // relu(Linear(relu(cat(x, t)))) -> dropout   (src/model.py:16-19,131)
                    GemmProblem g = gp_make(b.x, Ep, W(p.lin_w), 2 * E, T, E, E, b.t_out, Ep);
                    gp_seg2(g, t, Ep, W(p.lin_w) + E, 2 * E, E);
                    g.bias = W(p.lin_b); g.flags |= GF_RELU_A | GF_RELU_OUT; g.drop_site = b.site_out;
                    chain_f.push_back(op_gemm(M2F_LAYOUT_NT, g));
                }
                t = b.t_out;
                fam.push_back(b);
            }
            P.bufs[M2F_BUF_FAM0_OUT] = fam.empty() ? nullptr : fam[0].t_out;
        }
        for (Op& o : chain_f) o.group = 1;                  // everything so far = the fusion stack
        const size_t cls_f0 = chain_f.size();
        // classifier (src/model.py:89-100,143): Linear0, [ReLU, Linear]*(n-2), ReLU, Dropout, Linear
        const int hid = c.cls_hidden, C = c.cls_out;
        const int hidp = pad8(hid);
        const int nhid = (int)P.pm.cls.size() - 1;           // hidden activations h[0..nhid-1]
        std::vector<float*> hbuf;
        std::vector<uint32_t> hsite;
        const float* s0 = (a_on && t_on) ? a : (t_on ? t : a);
        const float* s1 = (a_on && t_on) ? t : nullptr;
        for (int j = 0; j < nhid; ++j) {
            float* h = ar.f((size_t)T * hidp);
            const uint32_t s = (j == nhid - 1) ? site() : 0u;
            const LinP& lp = P.pm.cls[j];
            GemmProblem g;
            if (j == 0) {
                const int ldw = cls_in_width(c);
                g = gp_make(s0, Ep, W(lp.w), ldw, T, hid, E, h, hidp);
                if (s1) gp_seg2(g, s1, Ep, W(lp.w) + E, ldw, E);
            } else {
                g = gp_make(hbuf[j - 1], hidp, W(lp.w), hid, T, hid, hid, h, hidp);
            }
            g.bias = W(lp.b); g.flags |= GF_RELU_OUT; g.drop_site = s;
            chain_f.push_back(op_gemm(M2F_LAYOUT_NT, g));
            hbuf.push_back(h);
            hsite.push_back(s);
        }
        float* logits = ar.f((size_t)T * C);
        P.bufs[M2F_BUF_LOGITS] = logits;
        {
            const LinP& lp = P.pm.cls[nhid];
            GemmProblem g = gp_make(hbuf[nhid - 1], hidp, W(lp.w), hid, T, C, hid, logits, C);
            g.bias = W(lp.b);
            chain_f.push_back(op_gemm(M2F_LAYOUT_NT, g));
        }
        for (size_t i = cls_f0; i < chain_f.size(); ++i) chain_f[i].group = 2;
        float* dlogits = ar.f((size_t)T * C);
        P.bufs[M2F_BUF_DLOGITS] = dlogits;
        P.loss_terms = ar.f((size_t)T * 2);
        // train plans keep (loss, den, num) in the TAIL of the flat gradient buffer, so the data-parallel all-reduce
        // carries the denominators along with the gradients and nothing has to be copied
        {
            float* ws_loss = ar.f(4);
            P.bufs[M2F_BUF_LOSS] = (P.train && P.grads) ? P.grads + P.pm.total : ws_loss;
        }
        if (!P.train) return;

        // ------------------------------ backward -----------------------------------------------------
        const float gs = gscale();
        {   // last linear
            const LinP& lp = P.pm.cls[nhid];
            wgrad(dlogits, C, hbuf[nhid - 1], hidp, C, hid, lp.w, hid, (long)lp.b);
        }
        float* dh = ar.f((size_t)T * hidp);
        {
            const LinP& lp = P.pm.cls[nhid];
            GemmProblem g = gp_make(dlogits, C, W(lp.w), hid, T, hid, C, dh, hidp);
            g.gate = hbuf[nhid - 1]; g.ldgate = hidp; g.gate_scale = hsite[nhid - 1] ? gs : 1.f;
            chain_b.push_back(op_gemm(M2F_LAYOUT_NN, g));
        }
        for (int j = nhid - 1; j >= 1; --j) {
            const LinP& lp = P.pm.cls[j];
            wgrad(dh, hidp, hbuf[j - 1], hidp, hid, hid, lp.w, hid, (long)lp.b);
            float* dprev = ar.f((size_t)T * hidp);
            GemmProblem g = gp_make(dh, hidp, W(lp.w), hid, T, hid, hid, dprev, hidp);
            g.gate = hbuf[j - 1]; g.ldgate = hidp; g.gate_scale = 1.f;
            chain_b.push_back(op_gemm(M2F_LAYOUT_NN, g));
            dh = dprev;
        }
        // Linear0: input = cat(a, t) | t | a
        const LinP& l0 = P.pm.cls[0];
        const int ldw0 = cls_in_width(c);
        float* d_a = a_on ? ar.f((size_t)T * Ep) : nullptr;
        float* d_t = t_on ? ar.f((size_t)T * Ep) : nullptr;
        {
            Op o; o.kind = OP_GEMM; o.layout = M2F_LAYOUT_NN;
            if (a_on && t_on) {
                wgrad(dh, hidp, a, Ep, hid, E, l0.w, ldw0, (long)l0.b);
                wgrad(dh, hidp, t, Ep, hid, E, l0.w + E, ldw0, -1);
                o.gp.push_back(gp_make(dh, hidp, W(l0.w), ldw0, T, E, hid, d_a, Ep));
                GemmProblem g = gp_make(dh, hidp, W(l0.w) + E, ldw0, T, E, hid, d_t, Ep);
                if (c.fam_enabled) { g.gate = t; g.ldgate = Ep; g.gate_scale = fam.back().site_out ? gs : 1.f; }
                o.gp.push_back(g);
            } else {
                const float* x = t_on ? t : a;
                float* dx = t_on ? d_t : d_a;
                wgrad(dh, hidp, x, Ep, hid, E, l0.w, ldw0, (long)l0.b);
                o.gp.push_back(gp_make(dh, hidp, W(l0.w), ldw0, T, E, hid, dx, Ep));
            }
            chain_b.push_back(o);
        }
        for (Op& o : chain_b) o.group = 2;                  // classifier backward
        const size_t fam_b0 = chain_b.size();
        // fusion layers, last to first.  `dz` = gradient w.r.t. the pre-ReLU output of layer i (the
        // ReLU/dropout gate was applied by the producer's epilogue).
        const float* dz = d_t;
        for (int i = (int)fam.size() - 1; i >= 0; --i) {
            const FamP& p = P.pm.fam[i];
            const FamBuf& b = fam[i];
            wgrad(dz, Ep, b.x, Ep, E, E, p.lin_w, 2 * E, (long)p.lin_b, true);
            wgrad(dz, Ep, b.t_in, Ep, E, E, p.lin_w + E, 2 * E, -1, true);
            float* dx = ar.f((size_t)T * Ep);
            float* dtA = ar.f((size_t)T * Ep);
            {
                Op o; o.kind = OP_GEMM; o.layout = M2F_LAYOUT_NN;
                GemmProblem g1 = gp_make(dz, Ep, W(p.lin_w), 2 * E, T, E, E, dx, Ep);
                g1.gate = b.x; g1.ldgate = Ep;
                GemmProblem g2 = gp_make(dz, Ep, W(p.lin_w) + E, 2 * E, T, E, E, dtA, Ep);
                g2.gate = b.t_in; g2.ldgate = Ep;
                o.gp.push_back(g1); o.gp.push_back(g2);
                chain_b.push_back(o);
            }
            wgrad(dx, Ep, b.att, Ep, E, E, p.out_w, E, (long)p.out_b);
            float* datt = ar.f((size_t)T * Ep);
            chain_b.push_back(op_gemm(M2F_LAYOUT_NN, gp_make(dx, Ep, W(p.out_w), E, T, E, E, datt, Ep)));
            float* dqv = ar.f((size_t)T * E2p);
            float* dk = ar.f((size_t)T * Ep);
            {
                Op o; o.kind = OP_ATTN_BWD;
                AttnProblem at; memset(&at, 0, sizeof(at));
                at.q = b.qv; at.ldq = E2p; at.k = b.k; at.ldk = Ep; at.v = b.qv + E; at.ldv = E2p;
                at.out = b.att; at.ldo = Ep; at.probs = b.probs; at.H = Hf; at.hd = E / Hf; at.drop_site = b.site_attn;
                at.dout = datt; at.lddo = Ep;
                at.dq = dqv; at.lddq = E2p; at.dv = dqv + E; at.lddv = E2p; at.dk = dk; at.lddk = Ep;
                o.ap.push_back(at);
                chain_b.push_back(o);
            }
            wgrad(dqv, E2p, b.t_in, Ep, E, E, p.in_w, E, (long)p.in_b);                                         // dWq, dbq
            wgrad(dk, Ep, a, Ep, E, E, p.in_w + (size_t)E * E, E, (long)(p.in_b + E));                          // dWk, dbk
            wgrad(dqv + E, E2p, b.t_in, Ep, E, E, p.in_w + (size_t)2 * E * E, E, (long)(p.in_b + 2 * E));       // dWv, dbv
            float* dt = ar.f((size_t)T * Ep);
            {
                Op o; o.kind = OP_GEMM; o.layout = M2F_LAYOUT_NN;
                // d t_in = dq Wq + dv Wv + dtA   (then the previous layer's ReLU/dropout gate)
                GemmProblem g = gp_make(dqv, E2p, W(p.in_w), E, T, E, E, dt, Ep);
                gp_seg2(g, dqv + E, E2p, W(p.in_w + (size_t)2 * E * E), E, E);
                g.res = dtA; g.ldres = Ep;
                if (i > 0) { g.gate = b.t_in; g.ldgate = Ep; g.gate_scale = fam[i - 1].site_out ? gs : 1.f; }
                // d audio += dk Wk
                GemmProblem ga = gp_make(dk, Ep, W(p.in_w + (size_t)E * E), E, T, E, E, d_a, Ep);
                ga.flags |= GF_ACCUM;
                o.gp.push_back(g); o.gp.push_back(ga);
                chain_b.push_back(o);
            }
            dz = dt;
        }
        d_t = const_cast<float*>(dz);
        for (size_t i = fam_b0; i < chain_b.size(); ++i) chain_b[i].group = 1;
        // gradient through the post-projection dropout (src/model.py:113,125)
        for (int bi = 0; bi < 2; ++bi) {
            float* dxp = bi == 0 ? d_a : d_t;
            if (!dxp) continue;
            if (mod[bi].site_post) {
                Op o; o.kind = OP_DROPOUT; o.dptr = dxp; o.dT = T; o.dd = E; o.dld = Ep; o.dsite = mod[bi].site_post;
                chain_b.push_back(o);
            }
            build_modality_bwd(bi, bi == 0 ? P.pm.audio : P.pm.text, dxp);
        }
    }
};

// ---- merge two independent op chains into grouped launches (LCS on compatibility) -------------------
bool compatible(const Op& a, const Op& b) {
    if (a.kind != b.kind) return false;
    switch (a.kind) {
        case OP_GEMM: return a.layout == b.layout && a.gp.size() + b.gp.size() <= M2F_GEMM_MAX_PROBLEMS;
        case OP_ATTN_FWD: case OP_ATTN_BWD: return a.ap.size() + b.ap.size() <= M2F_ATTN_MAX_PROBLEMS;
        case OP_LN_FWD: case OP_LN_BWD: return a.lp.size() + b.lp.size() <= M2F_LN_MAX_PROBLEMS;
        default: return false;
    }
}
Op merged(const Op& a, const Op& b) {
    Op o = a;
    o.src.insert(o.src.end(), b.src.begin(), b.src.end());
    o.gp.insert(o.gp.end(), b.gp.begin(), b.gp.end());
    o.ap.insert(o.ap.end(), b.ap.begin(), b.ap.end());
    o.lp.insert(o.lp.end(), b.lp.begin(), b.lp.end());
    return o;
}
std::vector<Op> merge_chains(const std::vector<Op>& A, const std::vector<Op>& B) {
    const size_t n = A.size(), m = B.size();
    if (n == 0) return B;
    if (m == 0) return A;
    std::vector<std::vector<int>> dp(n + 1, std::vector<int>(m + 1, 0));
    for (size_t i = n; i-- > 0;)
        for (size_t j = m; j-- > 0;) {
            int best = std::max(dp[i + 1][j], dp[i][j + 1]);
            if (compatible(A[i], B[j])) best = std::max(best, dp[i + 1][j + 1] + 1);
            dp[i][j] = best;
        }
    std::vector<Op> out;
    size_t i = 0, j = 0;
    while (i < n && j < m) {
        if (compatible(A[i], B[j]) && dp[i][j] == dp[i + 1][j + 1] + 1) { out.push_back(merged(A[i], B[j])); ++i; ++j; }
        else if (dp[i + 1][j] >= dp[i][j + 1]) out.push_back(A[i++]);
        else out.push_back(B[j++]);
    }
    while (i < n) out.push_back(A[i++]);
    while (j < m) out.push_back(B[j++]);
    return out;
}

void to_launches(const m2f_plan& P, const std::vector<Op>& ops, std::vector<Launch>& out) {
    for (const Op& o : ops) {
        Launch l;
        l.kind = o.kind; l.layout = o.layout; l.src = o.src; l.group = o.group;
        memset(&l.gb, 0, sizeof(l.gb)); memset(&l.ab, 0, sizeof(l.ab)); memset(&l.lb, 0, sizeof(l.lb));
        switch (o.kind) {
            case OP_GEMM:
                l.gb.count = (int)o.gp.size();
                for (size_t i = 0; i < o.gp.size(); ++i) l.gb.pr[i] = o.gp[i];
                l.gb.rng = P.rng; l.gb.drop_thresh = P.drop_thresh; l.gb.drop_scale = P.drop_scale;
                // in-launch split-K stays OFF in plans: measured 2x SLOWER on the 96-tile GEMMs (the agent-scope
                // release/acquire seam costs more than the spread saves; same verdict as the guide's M=256 block)
                l.gb.splitk_ws = nullptr; l.gb.splitk_cnt = nullptr; l.gb.splitk_max_tiles = 0;
                break;
            case OP_ATTN_FWD: case OP_ATTN_BWD:
                l.ab.count = (int)o.ap.size();
                for (size_t i = 0; i < o.ap.size(); ++i) l.ab.pr[i] = o.ap[i];
                l.ab.B = P.B; l.ab.L = P.L; l.ab.key_pad = static_cast<const uint8_t*>(P.bufs[M2F_BUF_KEYPAD]);
                l.ab.cu = P.packed ? P.cu : nullptr; l.ab.T = P.T;
                l.ab.rng = P.rng; l.ab.drop_thresh = P.drop_thresh; l.ab.drop_scale = P.drop_scale;
                {   // M2F_ATTN_BF16=<mask> (read when a plan is built; AttnBatch::bf16_math): 0 keeps the fp32 slabs / contractions in bf16 mode too
                    const char* e = getenv("M2F_ATTN_BF16");
                    l.ab.bf16_math = P.prec == M2F_PREC_BF16 ? (e ? atoi(e) & 63 : 63) : 0;
                }
                break;
            case OP_LN_FWD: case OP_LN_BWD:
                l.lb.count = (int)o.lp.size();
                for (size_t i = 0; i < o.lp.size(); ++i) l.lb.pr[i] = o.lp[i];
                l.lb.T = P.T; l.lb.eps = P.cfg.ln_eps;
                l.lb.rng = P.rng; l.lb.drop_thresh = P.drop_thresh; l.lb.drop_scale = P.drop_scale;
                break;
            case OP_DROPOUT:
                l.dptr = o.dptr; l.dT = o.dT; l.dd = o.dd; l.dld = o.dld; l.dsite = o.dsite;
                // two in-place dropouts in a row over buffers of one shape (the modalities' post-projection gradients): one launch
                if (!out.empty() && out.back().kind == OP_DROPOUT && !out.back().dptr2 && out.back().group == l.group &&
                    out.back().dT == l.dT && out.back().dd == l.dd && out.back().dld == l.dld && out.back().dptr != l.dptr) {
                    out.back().dptr2 = l.dptr; out.back().dsite2 = l.dsite;
                    continue;
                }
                break;
        }
        out.push_back(l);
    }
}



// bf16 mode: which fp32 results does nobody read?  Every GEMM / attention output in the workspace is written twice - fp32 and
// its bf16 shadow - but GEMM operands, the weight-gradient table and (M2F_ATTN_BF16 bits) the attention slabs are staged from the
// shadows: the fp32 copy of a QKV projection, an attention output, an attention input gradient or an FFN hidden gradient is
// dead weight (half a gigabyte of writes per C3 step).  The readers are enumerated from the FINAL launch lists - whatever takes a
// workspace pointer as fp32 marks its extent in a map of 64-float blocks - and an output whose extent holds no mark gets
// GF_NO_F32 / AttnProblem::no_f32.  Its fp32 buffer is then filled with NaNs, once: a reader this walk does not know about
// cannot go unnoticed.  M2F_SKIP_F32=0 (read when a plan is built) keeps every fp32 store.
int mark_unread_fp32(m2f_plan& P, const float* wsf, size_t ws_floats, const std::vector<GemmProblem>& table_probs, bool poison) {
    const char* e = getenv("M2F_SKIP_F32");
    if (e && atoi(e) == 0) return 0;
    std::vector<uint8_t> read32((ws_floats + 63) / 64, 0), read16((ws_floats + 63) / 64, 0);      // fp32 readers / bf16-shadow readers
    auto span = [&](const float* p, size_t rows, size_t ld, size_t cols, size_t& lo, size_t& hi) {
        if (!p || p < wsf || p >= wsf + ws_floats || rows == 0) return false;
        lo = (size_t)(p - wsf); hi = std::min(ws_floats, lo + (rows - 1) * ld + cols);
        return hi > lo;
    };
    auto mark = [&](const float* p, size_t rows, size_t ld, size_t cols) {
        size_t lo, hi;
        if (span(p, rows, ld, cols, lo, hi)) for (size_t b = lo / 64; b <= (hi - 1) / 64; ++b) read32[b] = 1;
    };
    const uint16_t* sh0 = P.sh.shadow;
    auto mark16 = [&](const uint16_t* q, size_t rows, size_t ld, size_t cols) {      // q: a pointer into the activation shadow
        if (!q || !sh0 || q < sh0 || q >= sh0 + ws_floats) return;
        size_t lo, hi;
        if (span(wsf + (q - sh0), rows, ld, cols, lo, hi)) for (size_t b = lo / 64; b <= (hi - 1) / 64; ++b) read16[b] = 1;
    };
    auto unread16 = [&](const float* p, size_t rows, size_t ld, size_t cols) {
        size_t lo, hi;
        if (!span(p, rows, ld, cols, lo, hi)) return false;
        for (size_t b = lo / 64; b <= (hi - 1) / 64; ++b) if (read16[b]) return false;
        return true;
    };
    auto unread = [&](const float* p, size_t rows, size_t ld, size_t cols) {
        size_t lo, hi;
        if (!span(p, rows, ld, cols, lo, hi)) return false;
        for (size_t b = lo / 64; b <= (hi - 1) / 64; ++b) if (read32[b]) return false;
        return true;
    };
    const size_t T = (size_t)P.T;
    // buffers the caller sees (m2f_plan_buffer): always fp32-read
    auto pad8 = [](int x) { return (size_t)((x + 7) & ~7); };
    mark(static_cast<const float*>(P.bufs[M2F_BUF_TEXT]), T, pad8(P.cfg.d_text), pad8(P.cfg.d_text));
    mark(static_cast<const float*>(P.bufs[M2F_BUF_AUDIO]), T, pad8(P.cfg.d_audio), pad8(P.cfg.d_audio));
    mark(static_cast<const float*>(P.bufs[M2F_BUF_LOGITS]), T, (size_t)P.cfg.cls_out, (size_t)P.cfg.cls_out);
    mark(static_cast<const float*>(P.bufs[M2F_BUF_DLOGITS]), T, (size_t)P.cfg.cls_out, (size_t)P.cfg.cls_out);
    mark(static_cast<const float*>(P.bufs[M2F_BUF_FAM0_OUT]), T, pad8(P.cfg.d_fam), pad8(P.cfg.d_fam));
    auto operand = [&](const GemmOperand& o, size_t rows_hint, bool launch16) {
        for (int sg = 0; sg < 2; ++sg)
            if (o.k[sg] > 0 && o.p[sg]) {
                const size_t rows = std::max(rows_hint, (size_t)o.k[sg]);
                if (!launch16 || !o.q[sg]) mark(o.p[sg], rows, (size_t)o.ld[sg], (size_t)o.ld[sg]);      // staged from fp32 (extent: generous)
                else mark16(o.q[sg], rows, (size_t)o.ldq[sg], (size_t)o.ldq[sg]);
            }
    };
    std::vector<std::vector<Launch>*> lists = {&P.fwd, &P.bwd, &P.wg_rest};
    for (std::vector<Launch>* ls : lists)
        for (Launch& l : *ls) {
            switch (l.kind) {
                case OP_GEMM: {
                    const bool launch16 = m2f_gemm_stages_bf16(l.gb, l.layout);      // (one operand without a shadow sends the whole launch to fp32 staging)
                    for (int i = 0; i < l.gb.count; ++i) {
                        const GemmProblem& g = l.gb.pr[i];
                        const size_t big = (size_t)std::max(std::max(g.M, g.N), (int)T);
                        operand(g.a, big, launch16); operand(g.b, big, launch16);
                        mark(g.res, (size_t)g.M, (size_t)g.ldres, (size_t)g.N);
                        mark(g.gate, (size_t)g.M, (size_t)g.ldgate, (size_t)g.N);
                        if (g.flags & GF_ACCUM) mark(g.c, (size_t)g.M, (size_t)g.ldc, (size_t)g.N);
                    }
                    break;
                }
                case OP_ATTN_FWD: case OP_ATTN_BWD:
                    for (int i = 0; i < l.ab.count; ++i) {
                        const AttnProblem& a = l.ab.pr[i];
                        const size_t w = (size_t)a.H * a.hd;
                        const int bits = m2f_attn_shadow_only_bits(l.ab, i, l.kind == OP_ATTN_BWD);
                        auto rd = [&](const float* p, int ld, int bit) {
                            if (!p) return;
                            if (bits & bit) mark16(sh0 + (p - wsf), T, (size_t)ld, w); else mark(p, T, (size_t)ld, w);
                        };
                        rd(a.q, a.ldq, 2); rd(a.k, a.ldk, 4); rd(a.v, a.ldv, 8);
                        if (l.kind == OP_ATTN_BWD) { rd(a.dout, a.lddo, 16); rd(a.out, a.ldo, 32); }
                    }
                    break;
                case OP_LN_FWD: case OP_LN_BWD:
                    for (int i = 0; i < l.lb.count; ++i) {
                        const LnProblem& q = l.lb.pr[i];
                        const size_t ld = (size_t)(q.ld ? q.ld : q.d);
                        mark(q.x, T, ld, (size_t)q.d);
                        mark(q.res, T, ld, (size_t)q.d);
                        if (l.kind == OP_LN_BWD) { mark(q.dy, T, ld, (size_t)q.d); mark(q.extra, T, ld, (size_t)q.d); }
                    }
                    break;
                case OP_DROPOUT: mark(l.dptr, (size_t)l.dT, (size_t)l.dld, (size_t)l.dd); mark(l.dptr2, (size_t)l.dT, (size_t)l.dld, (size_t)l.dd); break;
                default: break;
            }
        }
    for (const CastBatch& cb : P.wg_casts)
        for (int i = 0; i < cb.count; ++i) mark(cb.it[i].src, (size_t)cb.it[i].rows, (size_t)cb.it[i].lds, (size_t)cb.it[i].cols);
    // (the weight-gradient table stages bf16 shadows only - what it cannot take went to wg_rest / wg_casts above; the criterion
    // reads the logits, marked with the caller-visible buffers)
    for (const GemmProblem& g : table_probs) {
        mark16(g.a.q[0], T, (size_t)g.a.ldq[0], (size_t)g.a.ldq[0]);
        mark16(g.b.q[0], T, (size_t)g.b.ldq[0], (size_t)g.b.ldq[0]);
    }
    int n = 0;
    auto poison_buf = [&](float* p, size_t rows, size_t ld, size_t cols) {
        size_t lo, hi;
        if (!poison || !span(p, rows, ld, cols, lo, hi)) return 0;
        return hipMemset(p, 0xFF, (hi - lo) * sizeof(float)) == hipSuccess ? 0 : 1;
    };
    auto poison16 = [&](const float* p, size_t rows, size_t ld, size_t cols) {       // ... and a shadow that is no longer written
        size_t lo, hi;
        if (!poison || !sh0 || !span(p, rows, ld, cols, lo, hi)) return 0;
        return hipMemset(const_cast<uint16_t*>(sh0) + lo, 0xFF, (hi - lo) * sizeof(uint16_t)) == hipSuccess ? 0 : 1;
    };
    for (std::vector<Launch>* ls : {&P.fwd, &P.bwd})
        for (Launch& l : *ls) {
            if (l.kind == OP_GEMM) {
                for (int i = 0; i < l.gb.count; ++i) {
                    GemmProblem& g = l.gb.pr[i];
                    if (unread16(g.c, (size_t)g.M, (size_t)g.ldc, (size_t)g.N)) {      // nobody stages C's shadow
                        g.flags |= GF_NO_BF16; ++n;
                        if (poison16(g.c, (size_t)g.M, (size_t)g.ldc, (size_t)g.N)) return fail("mark_unread_fp32: hipMemset");
                    }
                    if ((g.flags & (GF_ACCUM | GF_NO_BF16)) || g.c8 || !unread(g.c, (size_t)g.M, (size_t)g.ldc, (size_t)g.N)) continue;
                    g.flags |= GF_NO_F32; ++n;
                    if (poison_buf(g.c, (size_t)g.M, (size_t)g.ldc, (size_t)g.N)) return fail("mark_unread_fp32: hipMemset");
                }
            } else if (l.kind == OP_LN_BWD) {
                for (int i = 0; i < l.lb.count; ++i) {
                    LnProblem& q = l.lb.pr[i];
                    const size_t ld = (size_t)(q.ld ? q.ld : q.d);
                    if (q.dx_masked && unread(q.dx_masked, T, ld, (size_t)q.d)) {
                        q.skip |= 1u; ++n;
                        if (poison_buf(q.dx_masked, T, ld, (size_t)q.d)) return fail("mark_unread_fp32: hipMemset");
                    }
                    if (unread16(q.dx, T, ld, (size_t)q.d)) {
                        q.skip |= 2u; ++n;
                        if (poison16(q.dx, T, ld, (size_t)q.d)) return fail("mark_unread_fp32: hipMemset");
                    }
                }
            } else if (l.kind == OP_ATTN_FWD) {
                for (int i = 0; i < l.ab.count; ++i) {
                    AttnProblem& a = l.ab.pr[i];
                    const size_t w = (size_t)a.H * a.hd;
                    if (!unread(a.out, T, (size_t)a.ldo, w)) continue;
                    a.no_f32 = 1; ++n;
                    if (poison_buf(a.out, T, (size_t)a.ldo, w)) return fail("mark_unread_fp32: hipMemset");
                }
            } else if (l.kind == OP_ATTN_BWD) {
                for (int i = 0; i < l.ab.count; ++i) {
                    AttnProblem& a = l.ab.pr[i];
                    const size_t w = (size_t)a.H * a.hd;
                    if (!unread(a.dq, T, (size_t)a.lddq, w) || !unread(a.dk, T, (size_t)a.lddk, w) || !unread(a.dv, T, (size_t)a.lddv, w)) continue;
                    a.no_f32 = 1; ++n;
                    if (poison_buf(a.dq, T, (size_t)a.lddq, w) || poison_buf(a.dk, T, (size_t)a.lddk, w) || poison_buf(a.dv, T, (size_t)a.lddv, w))
                        return fail("mark_unread_fp32: hipMemset");
                }
            }
        }
    P.n_no_f32 = n;
    return 0;
}

int build_plan(m2f_plan& P, char* ws_base) {
    const m2f_config& c = P.cfg;
    Builder bld(P);
    bld.ar.base = ws_base;
    const int T = P.T;
    // host-visible input staging
    // input staging rows are padded to a multiple of 8 floats (like every activation buffer)
    P.bufs[M2F_BUF_TEXT] = bld.ar.f((size_t)T * Builder::pad8(std::max(c.d_text, 1)));
    P.bufs[M2F_BUF_AUDIO] = bld.ar.f((size_t)T * Builder::pad8(std::max(c.d_audio, 1)));
    P.bufs[M2F_BUF_KEYPAD] = bld.ar.alloc<uint8_t>((size_t)T);
    P.cu = bld.ar.alloc<int>((size_t)P.B + 1);
    P.bufs[M2F_BUF_CU_SEQLENS] = P.cu;
    P.bufs[M2F_BUF_LABELS] = bld.ar.alloc<int64_t>((size_t)T);
    P.bufs[M2F_BUF_CLASSW] = bld.ar.f(16);
    if (c.audio_enabled)
        bld.build_modality(0, P.pm.audio, c.d_audio, c.nhead_audio, c.nlayers_audio, c.ntrans_audio,
                           static_cast<const float*>(P.bufs[M2F_BUF_AUDIO]));
    if (c.text_enabled)
        bld.build_modality(1, P.pm.text, c.d_text, c.nhead_text, c.nlayers_text, c.ntrans_text,
                           static_cast<const float*>(P.bufs[M2F_BUF_TEXT]));
    bld.build_head();
    // forward: merged branches, then fusion + classifier
    to_launches(P, merge_chains(bld.br_f[0], bld.br_f[1]), P.fwd);
    to_launches(P, bld.chain_f, P.fwd);
    if (P.train) {
        for (size_t i = 0; i < bld.chain_b.size(); ++i) bld.chain_b[i].src.push_back({2, (int)i});
        for (int b = 0; b < 2; ++b)
            for (size_t i = 0; i < bld.br_b[b].size(); ++i) bld.br_b[b][i].src.push_back({b, (int)i});
        to_launches(P, bld.chain_b, P.bwd);
        P.bwd_head = P.bwd.size();
        to_launches(P, merge_chains(bld.br_b[0], bld.br_b[1]), P.bwd);
        // where did (chain, index) end up in the final launch order?
        auto final_index = [&](std::pair<int, int> tag) {
            if (tag.second < 0) return tag.first == 2 ? -1 : (int)bld.chain_b.size() - 1;   // before the branch: after the head
            for (size_t i = 0; i < P.bwd.size(); ++i)
                for (const auto& sidx : P.bwd[i].src)
                    if (sidx == tag) return (int)i;
            return (int)P.bwd.size() - 1;
        };
        // deferred weight gradients: chip-filling grouped launches, in dependency order
        std::vector<std::pair<int, size_t>> order;
        for (size_t i = 0; i < bld.wgrads.size(); ++i) order.push_back({final_index(bld.wdeps[i]), i});
        std::stable_sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        // groups of up to 8 problems; a problem that cannot be staged from bf16 shadows (leading dimension not a
        // multiple of 8: the [T, n_classes] criterion gradient) gets a launch of its own instead of dragging seven
        // others onto the fp32-source path
        std::vector<Op> wops;
        auto shadowable = [](const GemmProblem& g) { return !(g.a.ld[0] & 7) && !(g.b.ld[0] & 7); };
        size_t i = 0;
        while (i < order.size()) {
            Op o; o.kind = OP_GEMM; o.layout = M2F_LAYOUT_TN;
            const bool kind0 = shadowable(bld.wgrads[order[i].second]);
            while (i < order.size() && o.gp.size() < M2F_GEMM_MAX_PROBLEMS && shadowable(bld.wgrads[order[i].second]) == kind0) {
                o.gp.push_back(bld.wgrads[order[i].second]);
                ++i;
            }
            wops.push_back(o);
        }
        to_launches(P, wops, P.wg);
        for (size_t i = 0; i < bld.lnitems.size(); i += M2F_LNRED_MAX_ITEMS) {
            LnReduceBatch rb;
            memset(&rb, 0, sizeof(rb));
            for (size_t j = i; j < std::min(bld.lnitems.size(), i + M2F_LNRED_MAX_ITEMS); ++j) rb.it[rb.count++] = bld.lnitems[j];
            P.lnred.push_back(rb);
        }
    }
    // ---- bf16 shadows: activation shadow = one bf16 per fp32 workspace element; parameter shadow = padded matrices
    bld.ar.off = (bld.ar.off + 255) / 256 * 256;
    const size_t ws_floats = bld.ar.off / 4;
    uint16_t* shadow = bld.ar.alloc<uint16_t>(ws_floats);
    uint16_t* wshadow = P.ext_wshadow ? P.ext_wshadow : bld.ar.alloc<uint16_t>(P.pm.shadow_elems + 64);
    // ---- weight-gradient table (bf16 mode): token-transposed operand copies, problem table, tile map ------------------
    std::vector<TransItem> titems;
    std::vector<uint16_t> tblock;
    std::vector<GemmProblem> tprobs;
    std::vector<int> tprob_head, tprob_rest;   // indices into tprobs by backward part (m2f_step_part)
    size_t rest_head = 0;
    std::vector<uint16_t> tile_prob;
    // M2F_WGRAD_TABLE=0 (read when a plan is built) keeps the grouped row-contiguous launches in bf16 mode too: the parity
    // tests compare the two weight-gradient paths against each other
    const char* wg_env = getenv("M2F_WGRAD_TABLE");
    bool table_ok = P.train && !bld.wgrads.empty() && !(wg_env && wg_env[0] == '0');
    const int ldt = (T + 7) & ~7;
    // M2F_TABLE_TILE (read when a plan is built): 64 | 128 | 256 = register-staged builds of the table kernel on token-
    // transposed operand copies (256 = 256x128 tiles); 129 = ring form (gemm_ring.h, 128x128 tiles) on the same copies;
    // 130 = ring form on the ROW-MAJOR bf16 shadows the chain already maintains - no transposing launch, no copies, the
    // kernel sums the bias gradients itself; 131 (default) = the same with 256 (M) x 128 (N) tiles: the launch is bound by
    // the bytes its 256 workgroups pull through the L2s together (26.0 M L1->L2 requests per launch at C3 with 128x128
    // tiles, 19.5 M with 256x128; 361 -> 290 us, profiles/r03_*).
    const char* tt_env = getenv("M2F_TABLE_TILE");
    const int tt = tt_env ? atoi(tt_env) : 0;
    int table_tile = (tt == 64 || tt == 128 || (tt >= 129 && tt <= 132) || tt == 256) ? tt : M2F_TABLE_TILE_DEFAULT;
    const bool table_rc = table_tile >= 130 && table_tile <= 132;          // 131: 256 x 128 tiles; 132: 256 x 256 tiles on the eight-phase schedule (gemm_p8.h)
    // (256 x 256 tiles - eight waves that all load and multiply, no producer waves: the accumulators fill the register file -
    //  were built and measured SLOWER, 342 vs 296 us: commit c3f1730)
    if (table_ok && table_rc) {
        // operands = shadows of the fp32 activations (same element index); every one of them is also the A operand of a
        // forward-form chain launch, so its shadow is current when the backward chain has run
        const float* wsf = reinterpret_cast<const float*>(ws_base);
        auto shadow_ptr = [&](const float* ptr) -> const uint16_t* {
            if (!ws_base) return reinterpret_cast<const uint16_t*>(16);           // sizing pass: any non-null, aligned value
            const ptrdiff_t i = ptr - wsf;
            return (i >= 0 && (size_t)i < ws_floats) ? shadow + i : nullptr;
        };
        std::vector<Op> rest_ops;
        CastBatch wg_cast;
        memset(&wg_cast, 0, sizeof(wg_cast));
        for (size_t gi = 0; gi < bld.wgrads.size(); ++gi) {
            const GemmProblem& g = bld.wgrads[gi];
            const bool head = bld.wdeps[gi].first == 2;        // both operands exist once the classifier / fusion backward chain has run
            const uint16_t* qa = shadow_ptr(g.a.p[0]);
            const uint16_t* qb = shadow_ptr(g.b.p[0]);
            if (g.a.k[1] != 0 || g.b.k[1] != 0) { table_ok = false; break; }
            int lda = g.a.ld[0];
            const bool b_ok = qb && !(g.b.ld[0] & 7) && !(reinterpret_cast<uintptr_t>(qb) & 15) && (size_t)T * g.b.ld[0] * 2 < 0x80000000ull;
            if (b_ok && g.M <= 8 && !(g.flags & GF_RELU_A) && wg_cast.count < M2F_CAST_MAX_ITEMS) {
                // a narrow dY without a 16-byte-stageable shadow (the [T, n_classes] criterion gradient): a bf16 copy with
                // 8-column rows, made by a cast launch in front of the table launch (the pad column is zeroed once, here)
                uint16_t* pad = bld.ar.alloc<uint16_t>((size_t)T * 8 + 256);      // + one 512-byte fragment row (256-row tiles) past the end
                if (ws_base && hipMemset(pad, 0, ((size_t)T * 8 + 256) * sizeof(uint16_t)) != hipSuccess) { table_ok = false; break; }
                CastItem& ci = wg_cast.it[wg_cast.count++];
                ci.src = g.a.p[0]; ci.dst = pad; ci.rows = T; ci.cols = g.M; ci.lds = g.a.ld[0]; ci.ldd = 8; ci.dst_t = nullptr; ci.ldd_t = 0;
                qa = pad; lda = 8;
            }
            if (!qa || !b_ok || (lda & 7) || (reinterpret_cast<uintptr_t>(qa) & 15) || (size_t)T * lda * 2 >= 0x80000000ull) {
                // no 16-byte-stageable shadow (the [T, n_classes] criterion gradient): a grouped launch of its own
                Op o; o.kind = OP_GEMM; o.layout = M2F_LAYOUT_TN;
                o.gp.push_back(g);
                if (head) { rest_ops.insert(rest_ops.begin() + (ptrdiff_t)rest_head, o); ++rest_head; } else rest_ops.push_back(o);
                continue;
            }
            GemmProblem q;
            memset(&q, 0, sizeof(q));
            q.a.q[0] = qa; q.a.ldq[0] = lda; q.a.k[0] = T;
            q.b.q[0] = qb; q.b.ldq[0] = g.b.ld[0]; q.b.k[0] = T;
            q.M = g.M; q.N = g.N; q.c = g.c; q.ldc = g.ldc; q.gate_scale = 1.f;
            q.flags = g.flags & (uint32_t)(GF_RELU_A | GF_RELU_B);
            q.bias_grad = g.bias_grad;
            (head ? tprob_head : tprob_rest).push_back((int)tprobs.size());
            tprobs.push_back(q);
            P.wg_flops += 2.0 * g.M * g.N * (double)T;
        }
        if (table_ok) { to_launches(P, rest_ops, P.wg_rest); P.wg_rest_head = rest_head; }
        if (table_ok && wg_cast.count) P.wg_casts.push_back(wg_cast);
    }
    if (table_ok && !table_rc) {
        auto item_of = [&](const float* src, int ld, int F, int relu) {
            for (size_t k = 0; k < titems.size(); ++k)
                if (titems[k].src == src && titems[k].ld == ld && titems[k].F == F && titems[k].relu == relu) return (int)k;
            TransItem it;
            it.src = src; it.ld = ld; it.F = F; it.relu = relu; it.colsum = nullptr; it.ldt = ldt;
            it.dst = bld.ar.alloc<uint16_t>((size_t)F * ldt);
            it.block_begin = (int)tblock.size();
            for (int b = 0; b < (F + 63) / 64; ++b) tblock.push_back((uint16_t)titems.size());
            titems.push_back(it);
            return (int)titems.size() - 1;
        };
        for (const GemmProblem& g : bld.wgrads) {
            if (g.a.k[1] != 0 || g.b.k[1] != 0 || titems.size() > 60000) { table_ok = false; break; }
            const int ia = item_of(g.a.p[0], g.a.ld[0], g.M, (g.flags & GF_RELU_A) ? 1 : 0);
            const int ib = item_of(g.b.p[0], g.b.ld[0], g.N, (g.flags & GF_RELU_B) ? 1 : 0);
            if (g.bias_grad) {
                if (titems[ia].colsum && titems[ia].colsum != g.bias_grad) { table_ok = false; break; }
                titems[ia].colsum = g.bias_grad;
            }
            GemmProblem q;
            memset(&q, 0, sizeof(q));
            q.a.q[0] = titems[ia].dst; q.a.ldq[0] = ldt; q.a.k[0] = T;
            q.b.q[0] = titems[ib].dst; q.b.ldq[0] = ldt; q.b.k[0] = T;
            q.M = g.M; q.N = g.N; q.c = g.c; q.ldc = g.ldc; q.gate_scale = 1.f;
            q.flags = g.flags & ~(uint32_t)(GF_RELU_A | GF_RELU_B);
            tprobs.push_back(q);
            P.wg_flops += 2.0 * g.M * g.N * (double)T;
        }
        if (tblock.size() > 65535) table_ok = false;
    }
    int total_tiles = 0;
    // Tile choice, measured: 256x128 register-staged tiles beat 128x128 ones on the transposed copies (a quarter fewer operand
    // bytes through L1: 145 vs 183 us at C2) and also the ring form on the same copies (129: C3 step 3.707 vs 3.655 ms - this
    // launch keeps all 256 CUs streaming at once and is bound by what the L2s can pull together, so bytes per FLOP decide).
    if (table_tile == 132 && table_ok && !m2f_gemm_p8_table_ok(tprobs)) table_tile = 131;      // (a problem with ReLU on its A operand: the eight-phase form has no loop copy for it)
    const int walk_m = table_tile >= 131 ? 256 : 128, walk_n = table_tile == 132 ? 256 : 128;
    if (table_ok) total_tiles = m2f_gemm_table_layout(tprobs, (table_tile == 129 || table_tile == 130) ? 128 : (table_tile >= 131 && table_tile <= 132 ? 256 : table_tile), tile_prob, table_rc);
    if (total_tiles <= 0) table_ok = false;
    // ring table forms: per-workgroup tile lists (m2f_gemm_table_walk).  M2F_TABLE_WALK=0 (read when a plan is built) keeps the
    // order of the tile list; default 1 = every XCD walks its own problems in 8 x 4 super-tiles
    std::vector<uint32_t> tile_rec;
    std::vector<int> wg_begin;
    const bool table_ring = table_tile >= 129 && table_tile <= 132;
    int wg_count = 0;
    if (table_ok && table_ring) {
        const char* walk_env = getenv("M2F_TABLE_WALK");
        if (m2f_gemm_table_walk(tprobs, walk_env ? atoi(walk_env) : 1, 256, walk_m, walk_n, tile_rec, wg_begin) <= 0) table_ok = false;
        total_tiles = (int)tile_rec.size();                     // (the tile count of THIS tiling; tile_prob above serves the register-staged forms only)
        wg_count = std::min(total_tiles, 256);
        if (table_ok && wg_count < 256 && m2f_gemm_table_walk(tprobs, walk_env ? atoi(walk_env) : 1, wg_count, walk_m, walk_n, tile_rec, wg_begin) <= 0) table_ok = false;
    }
    std::vector<uint32_t> part_rec[2];
    std::vector<int> part_begin[2];
    int part_wg[2] = {0, 0};
    bool split = table_ok && table_ring && table_rc && P.bwd_head > 0 && P.bwd_head < P.bwd.size() && !tprob_head.empty() && !tprob_rest.empty();
    if (split) {
        const char* walk_env = getenv("M2F_TABLE_WALK");
        for (int part = 0; part < 2 && split; ++part) {
            const std::vector<int>& sub = part == 0 ? tprob_head : tprob_rest;
            if (m2f_gemm_table_walk(tprobs, walk_env ? atoi(walk_env) : 1, 256, walk_m, walk_n, part_rec[part], part_begin[part], &sub) <= 0) split = false;
            part_wg[part] = std::min((int)part_rec[part].size(), 256);
            if (split && part_wg[part] < 256 &&
                m2f_gemm_table_walk(tprobs, walk_env ? atoi(walk_env) : 1, part_wg[part], walk_m, walk_n, part_rec[part], part_begin[part], &sub) <= 0) split = false;
        }
    }
    GemmProblem* d_table = table_ok ? bld.ar.alloc<GemmProblem>(tprobs.size()) : nullptr;
    uint16_t* d_tile_prob = table_ok ? bld.ar.alloc<uint16_t>(tile_prob.size()) : nullptr;
    uint32_t* d_tile_rec = table_ok && table_ring ? bld.ar.alloc<uint32_t>(tile_rec.size()) : nullptr;
    int* d_wg_begin = table_ok && table_ring ? bld.ar.alloc<int>(wg_begin.size()) : nullptr;
    uint32_t* d_part_rec[2] = {nullptr, nullptr};
    int* d_part_begin[2] = {nullptr, nullptr};
    for (int part = 0; part < 2; ++part) {                          // (same allocations in the sizing pass: `split` does not depend on pointers)
        d_part_rec[part] = split ? bld.ar.alloc<uint32_t>(part_rec[part].size()) : nullptr;
        d_part_begin[part] = split ? bld.ar.alloc<int>(part_begin[part].size()) : nullptr;
    }
    TransItem* d_items = table_ok ? bld.ar.alloc<TransItem>(titems.size()) : nullptr;
    uint16_t* d_tblock = table_ok ? bld.ar.alloc<uint16_t>(tblock.size()) : nullptr;
    P.ws_used = bld.ar.off;
    if (table_ok && P.prec == M2F_PREC_BF16 && ws_base != nullptr) {
        bool ok = hipMemcpy(d_table, tprobs.data(), tprobs.size() * sizeof(GemmProblem), hipMemcpyHostToDevice) == hipSuccess;
        P.tprobs_host = tprobs;
        ok = ok && hipMemcpy(d_tile_prob, tile_prob.data(), tile_prob.size() * sizeof(uint16_t), hipMemcpyHostToDevice) == hipSuccess;
        if (table_ring) {
            ok = ok && hipMemcpy(d_tile_rec, tile_rec.data(), tile_rec.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
            ok = ok && hipMemcpy(d_wg_begin, wg_begin.data(), wg_begin.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
        }
        if (!titems.empty()) {
            ok = ok && hipMemcpy(d_items, titems.data(), titems.size() * sizeof(TransItem), hipMemcpyHostToDevice) == hipSuccess;
            ok = ok && hipMemcpy(d_tblock, tblock.data(), tblock.size() * sizeof(uint16_t), hipMemcpyHostToDevice) == hipSuccess;
        }
        for (int part = 0; part < 2 && split; ++part) {
            ok = ok && hipMemcpy(d_part_rec[part], part_rec[part].data(), part_rec[part].size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
            ok = ok && hipMemcpy(d_part_begin[part], part_begin[part].data(), part_begin[part].size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
        }
        if (ok) {
            P.wg_nt = true;
            P.wg_trans = {d_items, d_tblock, (int)tblock.size(), T};
            memset(&P.wg_tab, 0, sizeof(P.wg_tab));
            P.wg_tab.table = d_table; P.wg_tab.tile_prob = d_tile_prob; P.wg_tab.total_tiles = total_tiles; P.wg_tab.table_tile = table_tile;
            P.wg_tab.tile_rec = d_tile_rec; P.wg_tab.wg_begin = d_wg_begin; P.wg_tab.wg_count = wg_count;
            auto longest = [](const std::vector<int>& beg) { int m = 0; for (size_t w = 0; w + 1 < beg.size(); ++w) m = std::max(m, beg[w + 1] - beg[w]); return m; };
            P.wg_tab.p8_max_tiles = longest(wg_begin);
            P.wg_tab.rng = P.rng; P.wg_tab.drop_thresh = P.drop_thresh; P.wg_tab.drop_scale = P.drop_scale;
            for (int part = 0; part < 2 && split; ++part) {
                P.wg_tab_part[part] = P.wg_tab;
                P.wg_tab_part[part].tile_rec = d_part_rec[part]; P.wg_tab_part[part].wg_begin = d_part_begin[part];
                P.wg_tab_part[part].wg_count = part_wg[part]; P.wg_tab_part[part].total_tiles = (int)part_rec[part].size();
                P.wg_tab_part[part].p8_max_tiles = longest(part_begin[part]);
            }
            // the fusion stack's (else the classifier's) first parameter: everything from there on is complete after part 0
            P.split_ok = split;
            P.split_offset = (int64_t)(c.fam_enabled && !P.pm.fam.empty() ? P.pm.fam[0].in_w : P.pm.cls[0].w);
        }
    }
    if (P.prec == M2F_PREC_BF16 && ws_base != nullptr) {
        const float* wsf = reinterpret_cast<const float*>(ws_base);
        P.sh = {wsf, shadow, ws_floats};
        P.wshadow = wshadow;
        P.wg_tab.sh = P.sh;
        P.wg_tab_part[0].sh = P.sh; P.wg_tab_part[1].sh = P.sh;
        auto map_q = [&](GemmOperand& o) {
            for (int sgm = 0; sgm < 2; ++sgm) {
                o.q[sgm] = nullptr; o.ldq[sgm] = 0; o.qt[sgm] = nullptr; o.ldqt[sgm] = 0;
                const float* p = o.p[sgm];
                if (o.k[sgm] == 0 || !p) continue;
                const float* dl = static_cast<const float*>(P.bufs[M2F_BUF_DLOGITS]);
                if (p >= dl && p < dl + (size_t)T * c.cls_out) continue;          // written by the criterion kernel: no shadow
                if (p >= wsf && p < wsf + ws_floats) { o.q[sgm] = shadow + (p - wsf); o.ldq[sgm] = o.ld[sgm]; continue; }
                if (p >= P.params && p < P.params + P.pm.total) {
                    const size_t idx = (size_t)(p - P.params);
                    for (const ParamMap::Mat& m : P.pm.mats) {
                        if (idx >= m.off && idx < m.off + (size_t)m.rows * m.cols && o.ld[sgm] == m.cols) {
                            const size_t r = (idx - m.off) / m.cols, cc = (idx - m.off) % m.cols;
                            o.ldq[sgm] = (m.cols + 7) & ~7;
                            o.q[sgm] = wshadow + m.soff + r * o.ldq[sgm] + cc;
                            o.ldqt[sgm] = (m.rows + 7) & ~7;                       // W^T shadow: [cols][pad8(rows)]
                            o.qt[sgm] = wshadow + m.soff_t + cc * o.ldqt[sgm] + r;
                            break;
                        }
                    }
                }
            }
        };
        for (std::vector<Launch>* ls : {&P.fwd, &P.bwd, &P.wg})
            for (Launch& l : *ls) {
                l.gb.sh = P.sh; l.ab.sh = P.sh; l.lb.sh = P.sh;
                if (l.kind == OP_GEMM)
                    for (int i = 0; i < l.gb.count; ++i) { map_q(l.gb.pr[i].a); map_q(l.gb.pr[i].b); }
            }
        // casts at the start of a forward: every 2-D parameter into its padded shadow, then the two input buffers
        CastBatch cb;
        memset(&cb, 0, sizeof(cb));
        std::vector<CastBatch>* dst = &P.param_casts;
        auto flush = [&]() { if (cb.count) { dst->push_back(cb); memset(&cb, 0, sizeof(cb)); } };
        for (const ParamMap::Mat& m : P.pm.mats) {
            cb.it[cb.count++] = {P.params + m.off, wshadow + m.soff, m.rows, m.cols, m.cols, (m.cols + 7) & ~7,
                                 wshadow + m.soff_t, (m.rows + 7) & ~7};
            if (cb.count == M2F_CAST_MAX_ITEMS) flush();
        }
        flush();
        dst = &P.input_casts;
        if (c.text_enabled) {
            const int dp = Builder::pad8(c.d_text);
            float* x = static_cast<float*>(P.bufs[M2F_BUF_TEXT]);
            cb.it[cb.count++] = {x, shadow + (x - wsf), T, c.d_text, dp, dp, nullptr, 0};
        }
        if (c.audio_enabled) {
            const int dp = Builder::pad8(c.d_audio);
            float* x = static_cast<float*>(P.bufs[M2F_BUF_AUDIO]);
            cb.it[cb.count++] = {x, shadow + (x - wsf), T, c.d_audio, dp, dp, nullptr, 0};
        }
        flush();
    }
    if (P.prec == M2F_PREC_BF16 && ws_base != nullptr && (!P.train || (table_ok && table_rc))) {
        if (int r = mark_unread_fp32(P, reinterpret_cast<const float*>(ws_base), ws_floats, tprobs, true)) return r;
    }
    P.ws_used = bld.ar.off;
    return 0;
}

// Optional per-launch timing (m2f_step_timed): hipEvents recorded on the launch stream around every launch.
struct Profiler {
    std::vector<hipEvent_t> ev;
    std::vector<int> kind;
    std::vector<double> flops;
    hipStream_t s = nullptr;
    void begin(int k, double f) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        ev.push_back(a); ev.push_back(b); kind.push_back(k); flops.push_back(f);
        (void)hipEventRecord(a, s);
    }
    void end() { (void)hipEventRecord(ev.back(), s); }
};
thread_local Profiler* g_prof = nullptr;

double gemm_flops(const GemmBatch& gb) {
    double f = 0;
    for (int i = 0; i < gb.count; ++i)
        f += 2.0 * gb.pr[i].M * gb.pr[i].N * ((double)gb.pr[i].a.k[0] + gb.pr[i].a.k[1]);
    return f;
}
double attn_flops(const AttnBatch& ab, bool bwd) {
    double f = 0;
    for (int i = 0; i < ab.count; ++i) f += (bwd ? 10.0 : 4.0) * ab.B * ab.pr[i].H * (double)ab.L * ab.L * ab.pr[i].hd;
    return f;
}

int run_launches(m2f_plan& P, std::vector<Launch>& ls, hipStream_t s, size_t first = 0, size_t count = (size_t)-1) {
    for (size_t li = first; li < ls.size() && li - first < count; ++li) {
        Launch& l = ls[li];
        hipError_t e = hipSuccess;
        if (g_prof) {
            int k = l.kind == OP_GEMM ? l.layout : (l.kind + 2);     // 0..2 gemm NT/NN/TN, 3 attn fwd, 4 attn bwd, 5 ln fwd, 6 ln bwd, 7 dropout
            k += 32 * l.group;                                       // + 32 x (0 encoders, 1 fusion stack, 2 classifier)
            double f = l.kind == OP_GEMM ? gemm_flops(l.gb) : (l.kind == OP_ATTN_FWD ? attn_flops(l.ab, false) : (l.kind == OP_ATTN_BWD ? attn_flops(l.ab, true) : 0.0));
            g_prof->begin(k, f);
        }
        switch (l.kind) {
            case OP_GEMM: e = m2f_launch_gemm(l.gb, P.prec, l.layout, 0, s); break;
            case OP_ATTN_FWD: e = m2f_launch_attn_fwd(l.ab, s); break;
            case OP_ATTN_BWD: e = m2f_launch_attn_bwd(l.ab, s); break;
            case OP_LN_FWD: e = m2f_launch_ln_fwd(l.lb, s); break;
            case OP_LN_BWD: e = m2f_launch_ln_bwd(l.lb, s); break;
            case OP_DROPOUT: e = m2f_launch_dropout_inplace2(l.dptr, l.dptr2, l.dT, l.dd, l.dld, l.dsite, l.dsite2, P.rng, P.drop_thresh, P.drop_scale, P.sh, s); break;
        }
        if (g_prof) g_prof->end();
        if (e != hipSuccess) return hipfail(e, "kernel launch");
    }
    return 0;
}

int do_loss(m2f_plan& P, float ls, int use_cw, int normalise, hipStream_t s) {
    CeArgs a;
    a.logits = static_cast<const float*>(P.bufs[M2F_BUF_LOGITS]);
    a.T = P.T; a.C = P.cfg.cls_out;
    a.labels = static_cast<const int64_t*>(P.bufs[M2F_BUF_LABELS]);
    a.class_w = use_cw ? static_cast<const float*>(P.bufs[M2F_BUF_CLASSW]) : nullptr;
    a.label_smoothing = ls;
    a.loss_terms = P.loss_terms;
    a.dlogits = static_cast<float*>(P.bufs[M2F_BUF_DLOGITS]);
    if (g_prof) g_prof->begin(8, 0.0);
    M2F_HIP(m2f_launch_ce(a, s));
    M2F_HIP(m2f_launch_loss_finalize(P.loss_terms, P.T, P.cfg.cls_out, a.dlogits, static_cast<float*>(P.bufs[M2F_BUF_LOSS]), normalise, s));
    if (g_prof) g_prof->end();
    return 0;
}

// part 0 / 1 of the split backward (see m2f_plan::bwd_head)
int do_backward_part(m2f_plan& P, int part, hipStream_t s) {
    if (!P.split_ok) return fail("m2f_step_part: this plan has no split backward (bf16 train plans with the row-major weight-gradient table only)");
    if (part == 0) {
        if (int r = run_launches(P, P.bwd, s, 0, P.bwd_head)) return r;
        for (const CastBatch& cb : P.wg_casts) M2F_HIP(m2f_launch_cast(cb, s));
        M2F_HIP(m2f_launch_gemm_table(P.wg_tab_part[0], s));
        return run_launches(P, P.wg_rest, s, 0, P.wg_rest_head);
    }
    if (int r = run_launches(P, P.bwd, s, P.bwd_head)) return r;
    M2F_HIP(m2f_launch_gemm_table(P.wg_tab_part[1], s));
    if (int r = run_launches(P, P.wg_rest, s, P.wg_rest_head)) return r;
    for (const LnReduceBatch& rb : P.lnred) M2F_HIP(m2f_launch_ln_param_reduce(rb, s));
    return 0;
}

int do_backward(m2f_plan& P, hipStream_t s) {
    if (!P.train || !P.grads) return fail("m2f_backward: plan was created without train=1 / gradient buffer");
    if (int r = run_launches(P, P.bwd, s)) return r;
    if (P.wg_nt) {
        if (g_prof) g_prof->begin(10, 0.0);
        if (P.wg_trans.blocks > 0) M2F_HIP(m2f_launch_transpose_tokens(P.wg_trans, s));
        for (const CastBatch& cb : P.wg_casts) M2F_HIP(m2f_launch_cast(cb, s));
        if (g_prof) { g_prof->end(); g_prof->begin(M2F_LAYOUT_TN, P.wg_flops); }
        if (P.fused_on) M2F_HIP(m2f_p8_launch_table_rc_adam(P.wg_tab_adam, s));      // dW stays in registers: Adam in the epilogue
        else if (P.g16_on) M2F_HIP(m2f_launch_gemm_table(P.wg_tab_g16, s));            // dW leaves as bf16 (the bf16 gradient exchange's buffer)
        else M2F_HIP(m2f_launch_gemm_table(P.wg_tab, s));
        if (g_prof) g_prof->end();
        if (int r = run_launches(P, P.wg_rest, s)) return r;
    } else {
        if (int r = run_launches(P, P.wg, s)) return r;
    }
    for (const LnReduceBatch& rb : P.lnred) {
        if (g_prof) g_prof->begin(9, 0.0);
        M2F_HIP(m2f_launch_ln_param_reduce(rb, s));
        if (g_prof) g_prof->end();
    }
    if (P.g16_on && !P.fused_on) {                     // ... and every other gradient rounded into the bf16 buffer
        if (g_prof) g_prof->begin(10, 0.0);
        M2F_HIP(m2f_launch_cast_items(P.grads, P.g16_buf, P.g16_items, P.g16_tb, P.g16_n_items, P.g16_tiles, s));
        if (g_prof) g_prof->end();
    }
    if (P.fused_on) {                                  // every gradient the table launch did not consume: biases, LayerNorm, wg_rest's matrices
        if (g_prof) g_prof->begin(10, 0.0);
        M2F_HIP(m2f_launch_adam_shadowed_dev(P.fz_p, P.grads, P.fz_m, P.fz_v, P.fz_sh, P.fz_items, P.fz_tb, P.fz_n_items, P.fz_tiles,
                                             P.fz_hyper, P.fz_gs, s));
        if (g_prof) g_prof->end();
    }
    return 0;
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

const char* m2f_last_error(void) { return g_err.c_str(); }

int m2f_device_check(void) {
    int dev = 0;
    M2F_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    M2F_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(std::string("not a gfx950 device: ") + prop.gcnArchName);
    return 0;
}

int m2f_param_layout(const m2f_config* cfg, int64_t* offsets, int64_t* numels, int max_entries, int64_t* total) {
    ParamMap pm;
    if (build_param_map(*cfg, pm)) return -1;
    const int n = (int)pm.offsets.size();
    for (int i = 0; i < n && i < max_entries; ++i) { offsets[i] = pm.offsets[i]; numels[i] = pm.numels[i]; }
    if (total) *total = (int64_t)pm.total;
    return n;
}

static m2f_plan* plan_new(const m2f_config* cfg, int B, int L, int precision, int train, int T_packed = 0) {
    if (B < 1 || L < 1 || L > 64) { fail("B >= 1 and 1 <= L <= 64 required (L = utterances per dialogue)"); return nullptr; }
    if (T_packed < 0 || (T_packed > 0 && (T_packed < B || (int64_t)T_packed > (int64_t)B * L))) {
        fail("packed plan: B <= T <= B * L token rows required"); return nullptr;
    }
    if (precision != M2F_F32 && precision != M2F_BF16) { fail("bad precision"); return nullptr; }
    m2f_plan* p = new m2f_plan();
    p->cfg = *cfg;
    p->B = B; p->L = L; p->T = T_packed > 0 ? T_packed : B * L; p->packed = T_packed > 0; p->prec = precision; p->train = train;
    if (build_param_map(*cfg, p->pm)) { delete p; return nullptr; }
    if ((cfg->audio_enabled && (cfg->nlayers_audio < 1 || cfg->ntrans_audio < 1)) ||
        (cfg->text_enabled && (cfg->nlayers_text < 1 || cfg->ntrans_text < 1)) ||
        (cfg->fam_enabled && cfg->nlayers_fam < 1)) {
        fail("layer / transformer counts must be >= 1");
        delete p;
        return nullptr;
    }
    p->use_dropout = train && cfg->dropout > 0.f;
    if (p->use_dropout) {
        p->drop_thresh = (uint32_t)std::min(4294967295.0, std::floor((double)cfg->dropout * 4294967296.0));
        p->drop_scale = 1.0f / (1.0f - cfg->dropout);
    }
    return p;
}

static int64_t workspace_bytes_impl(const m2f_config* cfg, int B, int L, int train, int T_packed, bool shared = false);
int64_t m2f_workspace_bytes(const m2f_config* cfg, int B, int L, int train) { return workspace_bytes_impl(cfg, B, L, train, 0); }
int64_t m2f_workspace_bytes_packed(const m2f_config* cfg, int B, int L, int T, int train) {
    if (T < 1) { fail("packed plan: T >= 1 required"); return -1; }
    return workspace_bytes_impl(cfg, B, L, train, T);
}
int64_t m2f_workspace_bytes_shared(const m2f_config* cfg, int B, int L, int T, int train) {
    if (T < 0) { fail("m2f_workspace_bytes_shared: T >= 0 required (0 = padded plan)"); return -1; }
    return workspace_bytes_impl(cfg, B, L, train, T, true);
}
static int64_t workspace_bytes_impl(const m2f_config* cfg, int B, int L, int train, int T_packed, bool shared) {
    m2f_plan* p = plan_new(cfg, B, L, M2F_F32, train, T_packed);
    if (!p) return -1;
    p->params = nullptr; p->grads = nullptr;
    if (shared) p->ext_wshadow = reinterpret_cast<uint16_t*>(256);        // sizing pass: the plan will not hold parameter shadows
    build_plan(*p, nullptr);
    const int64_t n = (int64_t)p->ws_used + 4096;
    delete p;
    return n;
}

static m2f_plan* plan_create_impl(const m2f_config* cfg, int B, int L, int T_packed, int precision, int train, float* params,
                                  float* grads, void* workspace, int64_t workspace_bytes, uint32_t* rng_state, uint16_t* param_shadow = nullptr);
m2f_plan* m2f_plan_create_shared(const m2f_config* cfg, int B, int L, int T, int precision, int train, float* params, float* grads,
                                 void* workspace, int64_t workspace_bytes, uint32_t* rng_state, uint16_t* param_shadow) {
    if (T < 0) { fail("m2f_plan_create_shared: T >= 0 required (0 = padded plan)"); return nullptr; }
    if (!param_shadow || (reinterpret_cast<uintptr_t>(param_shadow) & 255)) { fail("m2f_plan_create_shared: param_shadow must be a 256-byte aligned device buffer of m2f_param_shadow_elems() uint16"); return nullptr; }
    return plan_create_impl(cfg, B, L, T, precision, train, params, grads, workspace, workspace_bytes, rng_state, param_shadow);
}
m2f_plan* m2f_plan_create(const m2f_config* cfg, int B, int L, int precision, int train, float* params, float* grads,
                          void* workspace, int64_t workspace_bytes, uint32_t* rng_state) {
    return plan_create_impl(cfg, B, L, 0, precision, train, params, grads, workspace, workspace_bytes, rng_state);
}
m2f_plan* m2f_plan_create_packed(const m2f_config* cfg, int B, int L, int T, int precision, int train, float* params, float* grads,
                                 void* workspace, int64_t workspace_bytes, uint32_t* rng_state) {
    if (T < 1) { fail("packed plan: T >= 1 required"); return nullptr; }
    return plan_create_impl(cfg, B, L, T, precision, train, params, grads, workspace, workspace_bytes, rng_state);
}
static m2f_plan* plan_create_impl(const m2f_config* cfg, int B, int L, int T_packed, int precision, int train, float* params,
                                  float* grads, void* workspace, int64_t workspace_bytes, uint32_t* rng_state, uint16_t* param_shadow) {
    m2f_plan* p = plan_new(cfg, B, L, precision, train, T_packed);
    if (!p) return nullptr;
    p->ext_wshadow = param_shadow;
    if (!params || !workspace) { fail("params / workspace must not be NULL"); delete p; return nullptr; }
    if (train && !grads) { fail("train plan needs a gradient buffer"); delete p; return nullptr; }
    if (p->use_dropout && !rng_state) { fail("dropout > 0 in train mode needs an rng_state"); delete p; return nullptr; }
    if ((reinterpret_cast<uintptr_t>(workspace) & 255) || (reinterpret_cast<uintptr_t>(params) & 255) ||
        (grads && (reinterpret_cast<uintptr_t>(grads) & 255))) {
        fail("params / grads / workspace must be 256-byte aligned"); delete p; return nullptr;
    }
    p->params = params; p->grads = grads; p->rng = rng_state;
    {   // size the workspace with a dry build BEFORE anything is written into it
        const int64_t need = workspace_bytes_impl(cfg, B, L, train, T_packed, param_shadow != nullptr);
        if (need < 0 || need > workspace_bytes) {
            fail("workspace too small: need " + std::to_string(need) + " bytes");
            delete p;
            return nullptr;
        }
    }
    build_plan(*p, static_cast<char*>(workspace));
    if ((int64_t)p->ws_used > workspace_bytes) {
        fail("workspace too small: need " + std::to_string(p->ws_used) + " bytes");
        delete p;
        return nullptr;
    }
    return p;
}

void m2f_plan_destroy(m2f_plan* plan) { delete plan; }

void* m2f_plan_buffer(m2f_plan* plan, int which) {
    if (!plan || which < 0 || which >= M2F_BUF_COUNT) return nullptr;
    return plan->bufs[which];
}

int m2f_plan_persistent(m2f_plan*) { return 0; }        // (the persistent strip-dataflow kernels of rounds 2-3 are gone: git history)

/* diagnostic (not part of the ABI header): where the activation shadows of a bf16 plan live */
int m2f_dbg_shadow_map(m2f_plan* plan, const float** ws_base, uint16_t** shadow, int64_t* floats) {
    if (!plan) return 1;
    *ws_base = plan->sh.ws_base; *shadow = plan->sh.shadow; *floats = (int64_t)plan->sh.ws_floats;
    return 0;
}

int m2f_plan_params_fresh(m2f_plan* plan, int fresh) {
    if (!plan) return fail("m2f_plan_params_fresh: NULL plan (destroyed?)");
    if (fresh && !plan->ext_wshadow) return fail("m2f_plan_params_fresh: only plans created with m2f_plan_create_shared can skip their parameter casts");
    plan->params_fresh = fresh != 0;
    return 0;
}

// ---- parameter shadows shared by the plans of one model + the optimizer that keeps them current -------------------------
namespace {
constexpr int64_t ADAM_TABLE_BYTES = 64 * 1024;          // behind the shadows: AdamItem[n] | int tile_begin[n + 1]
struct AdamTable { std::vector<AdamItem> items; std::vector<int> tile_begin; int total_tiles = 0; size_t shadow_elems = 0; };
int adam_table(const m2f_config& cfg, AdamTable& t) {
    ParamMap pm;
    if (build_param_map(cfg, pm)) return 1;
    t.shadow_elems = pm.shadow_elems + 64;
    size_t mi = 0;
    for (size_t i = 0; i < pm.offsets.size(); ++i) {
        AdamItem it; memset(&it, 0, sizeof(it));
        it.off = pm.offsets[i]; it.tile_begin = t.total_tiles;
        if (mi < pm.mats.size() && pm.mats[mi].off == (size_t)pm.offsets[i]) {
            const ParamMap::Mat& m = pm.mats[mi++];
            it.rows = m.rows; it.cols = m.cols; it.soff = (long long)m.soff; it.soff_t = (long long)m.soff_t;
            it.tiles_c = (m.cols + 63) / 64;
            t.total_tiles += ((m.rows + 63) / 64) * it.tiles_c;
        } else {
            const int64_t next = i + 1 < pm.offsets.size() ? pm.offsets[i + 1] : (int64_t)pm.total;
            it.rows = 0; it.cols = (int)(next - pm.offsets[i]);            // the tensor and its pad up to the next one (multiples of 64)
            it.tiles_c = 1;
            t.total_tiles += (it.cols + 4095) / 4096;
        }
        t.tile_begin.push_back(it.tile_begin);
        t.items.push_back(it);
    }
    t.tile_begin.push_back(t.total_tiles);
    if (mi != pm.mats.size()) return fail("adam_table: parameter map walk lost a matrix");
    if (t.items.size() > M2F_ADAM_MAX_ITEMS || t.items.size() * sizeof(AdamItem) + t.tile_begin.size() * sizeof(int) > (size_t)ADAM_TABLE_BYTES)
        return fail("too many parameter tensors for the fused optimizer table");
    return 0;
}
}  // namespace

int64_t m2f_param_shadow_elems(const m2f_config* cfg) {
    AdamTable t;
    if (adam_table(*cfg, t)) return -1;
    return (int64_t)((t.shadow_elems + 127) / 128 * 128) + ADAM_TABLE_BYTES / 2;
}

int m2f_param_shadow_init(const m2f_config* cfg, uint16_t* param_shadow, m2f_stream_t stream) {
    AdamTable t;
    if (adam_table(*cfg, t)) return 1;
    if (!param_shadow || (reinterpret_cast<uintptr_t>(param_shadow) & 255)) return fail("m2f_param_shadow_init: 256-byte aligned buffer required");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t sh = (t.shadow_elems + 127) / 128 * 128;
    M2F_HIP(hipMemsetAsync(param_shadow, 0, sh * sizeof(uint16_t) + ADAM_TABLE_BYTES, s));      // the pad columns of the shadows stay zero for good
    char* tab = reinterpret_cast<char*>(param_shadow + sh);
    M2F_HIP(hipMemcpyAsync(tab, t.items.data(), t.items.size() * sizeof(AdamItem), hipMemcpyHostToDevice, s));
    M2F_HIP(hipMemcpyAsync(tab + t.items.size() * sizeof(AdamItem), t.tile_begin.data(), t.tile_begin.size() * sizeof(int), hipMemcpyHostToDevice, s));
    M2F_HIP(hipStreamSynchronize(s));                           // (the host vectors go out of scope)
    return 0;
}

int m2f_adam_step_shadowed_range(const m2f_config* cfg, float* params, const void* grads, int grads_bf16, float* exp_avg,
                                 float* exp_avg_sq, uint16_t* param_shadow, int64_t first, int64_t end, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, int step, const float* grad_scale_ptr, m2f_stream_t stream) {
    thread_local m2f_config cached_cfg;
    thread_local AdamTable cached;
    thread_local bool have = false;
    if (!have || memcmp(&cached_cfg, cfg, sizeof(m2f_config)) != 0) {
        AdamTable t;
        if (adam_table(*cfg, t)) return 1;
        cached = t; cached_cfg = *cfg; have = true;
    }
    // parameters [first, end) of the flat buffers = a run of whole table items (end < 0, or past the last tensor: to the end)
    const int n = (int)cached.items.size();
    int i0 = 0, i1 = n;
    while (i0 < n && cached.items[i0].off < first) ++i0;
    if (end >= 0) { i1 = i0; while (i1 < n && cached.items[i1].off < end) ++i1; }
    if (i0 >= n || cached.items[i0].off != first || i1 <= i0)
        return fail("m2f_adam_step_shadowed_range: [first, end) must start at a parameter tensor and hold at least one");
    if (end >= 0 && i1 < n && cached.items[i1].off != end)
        return fail("m2f_adam_step_shadowed_range: `end` must be the offset of a parameter tensor (or < 0)");
    const size_t sh = (cached.shadow_elems + 127) / 128 * 128;
    const char* tab = reinterpret_cast<const char*>(param_shadow + sh);
    const AdamItem* items = reinterpret_cast<const AdamItem*>(tab);
    const int* tile_begin = reinterpret_cast<const int*>(tab + cached.items.size() * sizeof(AdamItem));
    M2F_HIP(m2f_launch_adam_shadowed(params, grads, grads_bf16, exp_avg, exp_avg_sq, param_shadow, items + i0, tile_begin + i0, i1 - i0,
                                     cached.tile_begin[i0], cached.tile_begin[i1], lr, beta1, beta2, eps, weight_decay, step,
                                     grad_scale_ptr, static_cast<hipStream_t>(stream)));
    return 0;
}

int m2f_adam_step_shadowed(const m2f_config* cfg, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                           uint16_t* param_shadow, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                           const float* grad_scale_ptr, m2f_stream_t stream) {
    return m2f_adam_step_shadowed_range(cfg, params, grads, 0, exp_avg, exp_avg_sq, param_shadow, 0, -1, lr, beta1, beta2, eps,
                                        weight_decay, step, grad_scale_ptr, stream);
}

namespace {
// The table's problems with their parameter-shadow pointers (res / gate / ldres / ldgate: the Adam epilogue's; `shadow` may be null) and the
// optimizer-table items the table does NOT cover (1-D parameters, matrices of wg_rest), re-tiled from 0.  Fails when a problem is not a block
// of one 2-D parameter or a parameter is only partly covered.
int table_coverage(m2f_plan& P, uint16_t* param_shadow, std::vector<GemmProblem>& tp, std::vector<AdamItem>& items, std::vector<int>& tb, int& tiles) {
    AdamTable at;
    if (adam_table(P.cfg, at)) return 1;
    // every table problem writes a block of whole rows of ONE 2-D parameter's gradient: find the tensor, its shadows, the first row
    tp = P.tprobs_host;
    std::vector<long long> covered(P.pm.mats.size(), 0);
    for (GemmProblem& q : tp) {
        const long long e = q.c - P.grads;
        size_t mi = P.pm.mats.size();
        for (size_t i = 0; i < P.pm.mats.size(); ++i) {
            const ParamMap::Mat& m = P.pm.mats[i];
            if (e >= (long long)m.off && e < (long long)m.off + (long long)m.rows * m.cols) { mi = i; break; }
        }
        if (mi == P.pm.mats.size()) return fail("weight-gradient table coverage: a weight-gradient problem does not write a 2-D parameter's gradient");
        const ParamMap::Mat& m = P.pm.mats[mi];
        const long long rel = e - (long long)m.off;
        // (a block of rows r0.. x columns c0.. of the parameter: whole matrices, the q / k / v row blocks of a fusion layer's in-projection,
        //  the column halves of a Linear over a never-materialised torch.cat)
        const int r0 = (int)(rel / m.cols), c0 = (int)(rel % m.cols), ldd = (m.cols + 7) & ~7, ldt = (m.rows + 7) & ~7;
        if (q.ldc != m.cols || c0 + q.N > m.cols || r0 + q.M > m.rows || q.res || q.gate)
            return fail("weight-gradient table coverage: a weight-gradient problem is not a block of its parameter");
        q.res = param_shadow ? reinterpret_cast<const float*>(param_shadow + m.soff + (size_t)r0 * ldd + c0) : nullptr;
        q.gate = param_shadow ? reinterpret_cast<const float*>(param_shadow + m.soff_t + (size_t)c0 * ldt + r0) : nullptr;
        q.ldres = ldd; q.ldgate = ldt;
        covered[mi] += (long long)q.M * q.N;
    }
    // what is left for the shadow-writing Adam kernel: every item of the optimizer's tensor table except the fully covered matrices
    items.clear(); tb.clear(); tiles = 0;
    for (const AdamItem& it0 : at.items) {
        bool fused = false;
        if (it0.rows > 0)
            for (size_t i = 0; i < P.pm.mats.size(); ++i)
                if ((long long)P.pm.mats[i].off == it0.off) {
                    const long long all = (long long)P.pm.mats[i].rows * P.pm.mats[i].cols;
                    if (covered[i] == all) fused = true;
                    else if (covered[i] != 0) return fail("weight-gradient table coverage: a parameter is only partly covered by the weight-gradient table");
                }
        if (fused) continue;
        AdamItem it = it0;
        it.tile_begin = tiles;
        tb.push_back(tiles);
        tiles += it.rows > 0 ? ((it.rows + 63) / 64) * it.tiles_c : (it.cols + 4095) / 4096;
        items.push_back(it);
    }
    tb.push_back(tiles);
    if (items.empty() || items.size() > M2F_ADAM_MAX_ITEMS) return fail("weight-gradient table coverage: no / too many residual tensors");
    return 0;
}
}  // namespace

/* Optimizer inside the step (round 4).  See include/m2fnet_hip.h. */
int m2f_plan_fused_adam_setup(m2f_plan* plan, float* params, float* exp_avg, float* exp_avg_sq, uint16_t* param_shadow,
                              const float* hyper_dev, const float* grad_scale_ptr) {
    if (!plan) return fail("m2f_plan_fused_adam_setup: NULL plan (destroyed?)");
    m2f_plan& P = *plan;
    P.fused_ready = false; P.fused_on = false;
    if (!P.train || !P.grads || P.prec != M2F_PREC_BF16 || !P.wg_nt || P.wg_tab.table_tile != 132 || P.tprobs_host.empty())
        return fail("m2f_plan_fused_adam_setup: needs a bf16 train plan whose weight-gradient table runs in the eight-phase form (M2F_TABLE_TILE=132)");
    if (!P.ext_wshadow || P.ext_wshadow != param_shadow) return fail("m2f_plan_fused_adam_setup: the plan must share the model's parameter-shadow buffer (m2f_plan_create_shared)");
    if (!params || !exp_avg || !exp_avg_sq || !hyper_dev) return fail("m2f_plan_fused_adam_setup: NULL buffer");
    if (params != P.params) return fail("m2f_plan_fused_adam_setup: `params` is not the plan's parameter buffer");
    std::vector<GemmProblem> tp;
    std::vector<AdamItem> items;
    std::vector<int> tb;
    int tiles = 0;
    if (int r = table_coverage(P, param_shadow, tp, items, tb, tiles)) return r;
    const size_t o_tab = 256, o_items = o_tab + ((tp.size() * sizeof(GemmProblem) + 255) & ~(size_t)255),
                 o_tb = o_items + ((items.size() * sizeof(AdamItem) + 255) & ~(size_t)255), bytes = o_tb + tb.size() * sizeof(int) + 256;
    if (P.fused_dev) { (void)hipFree(P.fused_dev); P.fused_dev = nullptr; }
    M2F_HIP(hipMalloc(&P.fused_dev, bytes));
    char* d = static_cast<char*>(P.fused_dev);
    M2FAdamFuse af;
    af.p = params; af.g_base = P.grads; af.m = exp_avg; af.v = exp_avg_sq; af.hyper = hyper_dev; af.gs_ptr = grad_scale_ptr;
    M2F_HIP(hipMemcpy(d, &af, sizeof(af), hipMemcpyHostToDevice));
    M2F_HIP(hipMemcpy(d + o_tab, tp.data(), tp.size() * sizeof(GemmProblem), hipMemcpyHostToDevice));
    M2F_HIP(hipMemcpy(d + o_items, items.data(), items.size() * sizeof(AdamItem), hipMemcpyHostToDevice));
    M2F_HIP(hipMemcpy(d + o_tb, tb.data(), tb.size() * sizeof(int), hipMemcpyHostToDevice));
    P.wg_tab_adam = P.wg_tab;
    P.wg_tab_adam.table = reinterpret_cast<const GemmProblem*>(d + o_tab);
    P.wg_tab_adam.adam = reinterpret_cast<const M2FAdamFuse*>(d);
    P.fz_p = params; P.fz_m = exp_avg; P.fz_v = exp_avg_sq; P.fz_sh = param_shadow; P.fz_hyper = hyper_dev; P.fz_gs = grad_scale_ptr;
    P.fz_items = reinterpret_cast<const AdamItem*>(d + o_items); P.fz_tb = reinterpret_cast<const int*>(d + o_tb);
    P.fz_n_items = (int)items.size(); P.fz_tiles = tiles;
    P.fused_ready = true;
    return 0;
}

/* Gradients left as bf16 (round 4).  See include/m2fnet_hip.h. */
int m2f_plan_grad_bf16(m2f_plan* plan, uint16_t* grads_bf16) {
    if (!plan) return fail("m2f_plan_grad_bf16: NULL plan (destroyed?)");
    m2f_plan& P = *plan;
    if (!grads_bf16) { P.g16_on = false; return 0; }
    if (!P.train || !P.grads || P.prec != M2F_PREC_BF16 || !P.wg_nt || P.wg_tab.table_tile != 132 || P.tprobs_host.empty())
        return fail("m2f_plan_grad_bf16: needs a bf16 train plan whose weight-gradient table runs in the eight-phase form (M2F_TABLE_TILE=132)");
    if (reinterpret_cast<uintptr_t>(grads_bf16) & 15) return fail("m2f_plan_grad_bf16: 16-byte aligned buffer required");
    if (!P.g16_ready || P.g16_buf != grads_bf16) {
        std::vector<GemmProblem> tp;
        std::vector<AdamItem> items;
        std::vector<int> tb;
        int tiles = 0;
        if (int r = table_coverage(P, nullptr, tp, items, tb, tiles)) return r;
        tb.clear(); tiles = 0;                               // the cast kernel walks every item as a flat range in tiles of 4,096 elements
        for (AdamItem& it : items) {
            it.tile_begin = tiles; tb.push_back(tiles);
            const long long n = it.rows > 0 ? (long long)it.rows * it.cols : (long long)it.cols;
            tiles += (int)((n + 4095) / 4096);
        }
        tb.push_back(tiles);
        for (GemmProblem& q : tp) { q.res = reinterpret_cast<const float*>(grads_bf16 + (q.c - P.grads)); q.gate = nullptr; q.ldres = 0; q.ldgate = 0; }
        const size_t o_items = (tp.size() * sizeof(GemmProblem) + 255) & ~(size_t)255, o_tb = o_items + ((items.size() * sizeof(AdamItem) + 255) & ~(size_t)255),
                     bytes = o_tb + tb.size() * sizeof(int) + 256;
        if (P.g16_dev) { (void)hipFree(P.g16_dev); P.g16_dev = nullptr; }
        M2F_HIP(hipMalloc(&P.g16_dev, bytes));
        char* d = static_cast<char*>(P.g16_dev);
        M2F_HIP(hipMemcpy(d, tp.data(), tp.size() * sizeof(GemmProblem), hipMemcpyHostToDevice));
        M2F_HIP(hipMemcpy(d + o_items, items.data(), items.size() * sizeof(AdamItem), hipMemcpyHostToDevice));
        M2F_HIP(hipMemcpy(d + o_tb, tb.data(), tb.size() * sizeof(int), hipMemcpyHostToDevice));
        P.wg_tab_g16 = P.wg_tab;
        P.wg_tab_g16.table = reinterpret_cast<const GemmProblem*>(d);
        P.g16_items = reinterpret_cast<const AdamItem*>(d + o_items); P.g16_tb = reinterpret_cast<const int*>(d + o_tb);
        P.g16_n_items = (int)items.size(); P.g16_tiles = tiles;
        P.g16_buf = grads_bf16; P.g16_ready = true;
        if (P.gexec) { (void)hipGraphExecDestroy(P.gexec); P.gexec = nullptr; }        // (another buffer: the captured launches hold the old pointers)
    }
    P.g16_on = true;
    return 0;
}

int m2f_plan_fused_adam(m2f_plan* plan, int on) {
    if (!plan) return fail("m2f_plan_fused_adam: NULL plan (destroyed?)");
    if (on && !plan->fused_ready) return fail("m2f_plan_fused_adam: call m2f_plan_fused_adam_setup first");
    plan->fused_on = on != 0;
    return 0;
}

int m2f_adam_hyper(float* hyper_dev, float lr, float beta1, float beta2, float eps, float weight_decay, int step, m2f_stream_t stream) {
    if (!hyper_dev || step < 1) return fail("m2f_adam_hyper: NULL buffer / step < 1");
    M2F_HIP(m2f_launch_adam_hyper(hyper_dev, lr, beta1, beta2, eps, weight_decay, step, static_cast<hipStream_t>(stream)));
    return 0;
}

int m2f_plan_skipped_copies(m2f_plan* plan) { return plan ? plan->n_no_f32 : -1; }

int m2f_plan_status(m2f_plan* plan, uint32_t* out8) {
    if (!plan || !out8) return fail("m2f_plan_status: NULL plan (destroyed?) or output");
    for (int i = 0; i < 8; ++i) out8[i] = 0u;
    return 0;            // (no kernel of the launch lists can give up: nothing waits on another workgroup)
}

/* diagnostic (tools/chain_floor.py): the grouped GEMM launches of the forward list and of the backward chain, in order, as
 * [phase (0 fwd, 2 bwd), layout, part of the model, problems, then M, N, K0 + K1 per problem] records; returns the ints written (< 0: too small) */
int m2f_plan_gemm_shapes(m2f_plan* plan, int* out, int max_ints) {
    if (!plan || !out) return -1;
    int n = 0;
    for (int phase = 0; phase < 2; ++phase)
        for (const Launch& l : phase == 0 ? plan->fwd : plan->bwd) {
            if (l.kind != OP_GEMM) continue;
            if (n + 4 + 3 * l.gb.count > max_ints) return -2;
            out[n++] = phase == 0 ? 0 : 2; out[n++] = l.layout; out[n++] = l.group; out[n++] = l.gb.count;
            for (int i = 0; i < l.gb.count; ++i) {
                out[n++] = l.gb.pr[i].M; out[n++] = l.gb.pr[i].N; out[n++] = l.gb.pr[i].a.k[0] + l.gb.pr[i].a.k[1];
            }
        }
    return n;
}

int m2f_plan_num_launches(m2f_plan* plan, int phase) {
    if (!plan) return -1;
    if (phase == 0) return (int)(plan->fwd.size() + (plan->params_fresh ? 0 : plan->param_casts.size()) + plan->input_casts.size());
    if (phase == 1) return 2;
    return (int)(plan->bwd.size() + (plan->wg_nt ? (plan->wg_trans.blocks > 0 ? 2 : 1) + plan->wg_rest.size() + plan->wg_casts.size() : plan->wg.size()) + plan->lnred.size());
}

static int do_forward(m2f_plan& P, hipStream_t s) {
    for (const std::vector<CastBatch>* cl : {&P.param_casts, &P.input_casts}) {      // bf16 mode only: refresh the parameter / input shadows
        if (cl == &P.param_casts && P.params_fresh) continue;                         // (the optimizer wrote them: m2f_adam_step_shadowed)
        for (const CastBatch& cb : *cl) {
            if (g_prof) g_prof->begin(10, 0.0);
            M2F_HIP(m2f_launch_cast(cb, s));
            if (g_prof) g_prof->end();
        }
    }
    return run_launches(P, P.fwd, s);
}

int m2f_forward(m2f_plan* plan, m2f_stream_t stream) {
    if (!plan) return fail("m2f_forward: NULL plan (destroyed?)");
    return do_forward(*plan, static_cast<hipStream_t>(stream));
}

int m2f_loss(m2f_plan* plan, float label_smoothing, int use_class_weights, int normalise, m2f_stream_t stream) {
    if (!plan) return fail("m2f_loss: NULL plan (destroyed?)");
    return do_loss(*plan, label_smoothing, use_class_weights, normalise, static_cast<hipStream_t>(stream));
}

int m2f_backward(m2f_plan* plan, m2f_stream_t stream) {
    if (!plan) return fail("m2f_backward: NULL plan (destroyed?)");
    return do_backward(*plan, static_cast<hipStream_t>(stream));
}

static int step_body(m2f_plan& P, float ls, int cw, int normalise, hipStream_t s) {
    if (P.use_dropout) M2F_HIP(m2f_launch_rng_advance(P.rng, s));
    if (int r = do_forward(P, s)) return r;
    if (int r = do_loss(P, ls, cw, normalise, s)) return r;
    return do_backward(P, s);
}

int m2f_step(m2f_plan* plan, float label_smoothing, int use_class_weights, int normalise, int use_graph,
             m2f_stream_t stream) {
    if (!plan) return fail("m2f_step: NULL plan (destroyed?)");
    m2f_plan& P = *plan;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!P.train) return fail("m2f_step needs a train plan");
    if (!use_graph || !P.warmed) { P.warmed = true; return step_body(P, label_smoothing, use_class_weights, normalise, s); }
    if (P.gexec && (P.g_ls != label_smoothing || P.g_cw != use_class_weights || P.g_norm != normalise || P.g_fresh != (int)P.params_fresh ||
                    P.g_fused != (int)P.fused_on || P.g_g16 != (int)P.g16_on)) {
        (void)hipGraphExecDestroy(P.gexec);
        P.gexec = nullptr;
    }
    if (!P.gexec) {
        hipGraph_t graph = nullptr;
        M2F_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int r = step_body(P, label_smoothing, use_class_weights, normalise, s);
        hipError_t e = hipStreamEndCapture(s, &graph);
        if (r) { if (graph) (void)hipGraphDestroy(graph); return r; }
        if (e != hipSuccess) return hipfail(e, "hipStreamEndCapture");
        e = hipGraphInstantiate(&P.gexec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { P.gexec = nullptr; return hipfail(e, "hipGraphInstantiate"); }
        P.g_ls = label_smoothing; P.g_cw = use_class_weights; P.g_norm = normalise; P.g_fresh = (int)P.params_fresh; P.g_fused = (int)P.fused_on; P.g_g16 = (int)P.g16_on;
    }
    M2F_HIP(hipGraphLaunch(P.gexec, s));
    return 0;
}

int64_t m2f_plan_split_offset(m2f_plan* plan) { return plan && plan->split_ok ? plan->split_offset : 0; }

int m2f_step_part(m2f_plan* plan, int part, float label_smoothing, int use_class_weights, int normalise, int use_graph,
                  m2f_stream_t stream) {
    if (!plan) return fail("m2f_step_part: NULL plan (destroyed?)");
    m2f_plan& P = *plan;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!P.train || part < 0 || part > 1) return fail("m2f_step_part needs a train plan and part 0 or 1");
    if (!P.split_ok) return fail("m2f_step_part: this plan has no split backward");
    auto body = [&]() -> int {
        if (part == 0) {
            if (P.use_dropout) M2F_HIP(m2f_launch_rng_advance(P.rng, s));
            if (int r = do_forward(P, s)) return r;
            if (int r = do_loss(P, label_smoothing, use_class_weights, normalise, s)) return r;
        }
        return do_backward_part(P, part, s);
    };
    if (!use_graph || !P.warmed) {
        const int r = body();
        if (part == 1) P.warmed = true;
        return r;
    }
    hipGraphExec_t& ge = P.gexec_part[part];
    if (ge && (P.gp_ls[part] != label_smoothing || P.gp_cw[part] != use_class_weights || P.gp_norm[part] != normalise ||
               P.gp_fresh[part] != (int)P.params_fresh)) {
        (void)hipGraphExecDestroy(ge);
        ge = nullptr;
    }
    if (!ge) {
        hipGraph_t graph = nullptr;
        M2F_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int r = body();
        hipError_t e = hipStreamEndCapture(s, &graph);
        if (r) { if (graph) (void)hipGraphDestroy(graph); return r; }
        if (e != hipSuccess) return hipfail(e, "hipStreamEndCapture");
        e = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { ge = nullptr; return hipfail(e, "hipGraphInstantiate"); }
        P.gp_ls[part] = label_smoothing; P.gp_cw[part] = use_class_weights; P.gp_norm[part] = normalise; P.gp_fresh[part] = (int)P.params_fresh;
    }
    M2F_HIP(hipGraphLaunch(ge, s));
    return 0;
}

int m2f_step_timed(m2f_plan* plan, float label_smoothing, int use_class_weights, int normalise, m2f_stream_t stream,
                   int max_entries, int* kinds, float* ms, double* flops) {
    if (!plan) { fail("m2f_step_timed: NULL plan (destroyed?)"); return -1; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    Profiler prof;
    prof.s = s;
    g_prof = &prof;
    const int r = step_body(*plan, label_smoothing, use_class_weights, normalise, s);
    g_prof = nullptr;
    hipError_t e = hipStreamSynchronize(s);
    const int n = (int)prof.kind.size();
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        if (e == hipSuccess) (void)hipEventElapsedTime(&t, prof.ev[2 * i], prof.ev[2 * i + 1]);
        if (i < max_entries) { kinds[i] = prof.kind[i]; ms[i] = t; flops[i] = prof.flops[i]; }
        (void)hipEventDestroy(prof.ev[2 * i]); (void)hipEventDestroy(prof.ev[2 * i + 1]);
    }
    if (r) return -1;
    if (e != hipSuccess) { hipfail(e, "hipStreamSynchronize"); return -1; }
    return n;
}

int m2f_event_overhead(uint32_t* scratch_rng_state, int pairs, float* empty_pair_ms, float* trivial_kernel_pair_ms,
                       m2f_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (pairs < 1 || pairs > 4096) return fail("m2f_event_overhead: pairs must be in [1, 4096]");
    std::vector<hipEvent_t> ev(4 * (size_t)pairs);
    for (auto& e : ev) M2F_HIP(hipEventCreate(&e));
    for (int i = 0; i < pairs; ++i) {                       // the interval of m2f_step_timed with nothing inside
        (void)hipEventRecord(ev[2 * i], s);
        (void)hipEventRecord(ev[2 * i + 1], s);
    }
    for (int i = 0; i < pairs; ++i) {                       // ... and with a one-thread kernel inside
        (void)hipEventRecord(ev[2 * (pairs + i)], s);
        if (scratch_rng_state) (void)m2f_launch_rng_advance(scratch_rng_state, s);
        (void)hipEventRecord(ev[2 * (pairs + i) + 1], s);
    }
    hipError_t e = hipStreamSynchronize(s);
    double acc[2] = {0, 0};
    for (int i = 0; i < 2 * pairs; ++i) {
        float t = 0.f;
        if (e == hipSuccess) (void)hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]);
        acc[i / pairs] += t;
    }
    for (auto& x : ev) (void)hipEventDestroy(x);
    if (e != hipSuccess) { hipfail(e, "hipStreamSynchronize"); return -1; }
    if (empty_pair_ms) *empty_pair_ms = (float)(acc[0] / pairs);
    if (trivial_kernel_pair_ms) *trivial_kernel_pair_ms = (float)(acc[1] / pairs);
    return 0;
}

int m2f_gather_dialogues(const float* text_table, int d_text, const float* audio_table, int d_audio,
                         const int64_t* label_table, const int32_t* rows, int T, float* text_out, int ld_text,
                         float* audio_out, int ld_audio, uint8_t* key_pad_out, int64_t* labels_out, m2f_stream_t stream) {
    GatherArgs a;
    a.text_table = text_table; a.audio_table = audio_table; a.label_table = label_table; a.rows = rows;
    a.T = T; a.d_text = d_text; a.d_audio = d_audio; a.ld_text = ld_text; a.ld_audio = ld_audio;
    a.text_out = text_out; a.audio_out = audio_out; a.key_pad = key_pad_out; a.labels = labels_out;
    M2F_HIP(m2f_launch_gather(a, static_cast<hipStream_t>(stream)));
    return 0;
}

int m2f_rng_advance(uint32_t* rng_state, m2f_stream_t stream) {
    M2F_HIP(m2f_launch_rng_advance(rng_state, static_cast<hipStream_t>(stream)));
    return 0;
}

int m2f_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, const float* grad_scale_ptr, m2f_stream_t stream) {
    M2F_HIP(m2f_launch_adam(params, grads, 0, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale_ptr,
                            static_cast<hipStream_t>(stream)));
    return 0;
}

int m2f_adam_step_g16(float* params, const uint16_t* grads_bf16, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                      float beta1, float beta2, float eps, float weight_decay, int step, const float* grad_scale_ptr,
                      m2f_stream_t stream) {
    M2F_HIP(m2f_launch_adam(params, grads_bf16, 1, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale_ptr,
                            static_cast<hipStream_t>(stream)));
    return 0;
}

// ---- kernel-level entry points -----------------------------------------------------------------------
// Optional bf16 shadow map for the kernel-level entry points (the plans carry their own): outputs that lie inside
// [ws_base, ws_base + floats) are also written as bf16 at the same element index of `shadow`.
static thread_local ShadowMap g_sh = {nullptr, nullptr, 0};
int m2f_set_shadow_map(const float* ws_base, uint16_t* shadow, int64_t floats) {
    if ((ws_base == nullptr) != (shadow == nullptr) || floats < 0) return fail("m2f_set_shadow_map: base / shadow must both be set or both be NULL");
    g_sh = {ws_base, shadow, (size_t)floats};
    return 0;
}

static thread_local int g_shadow_only = 0;
int m2f_set_shadow_only(int on) { g_shadow_only = on != 0; return 0; }

static void drop_params(float p, uint32_t* thresh, float* scale) {
    *thresh = (uint32_t)std::min(4294967295.0, std::floor((double)p * 4294967296.0));
    *scale = 1.0f / (1.0f - p);
}

int m2f_gemm(int precision, int layout, int M, int N, int K0, int K1, const float* a0, int lda0, const float* a1, int lda1,
             const float* b0, int ldb0, const float* b1, int ldb1, float* c, int ldc, const float* bias, const float* res,
             int ldres, const float* gate, int ldgate, float gate_scale, float* bias_grad, int relu_a, int relu_b,
             int relu_out, int accumulate, uint32_t drop_site, float drop_p, const uint32_t* rng_state, int tile,
             float* splitk_ws, uint32_t* splitk_tickets, int splitk_max_tiles,
             const uint16_t* a0q, int ldaq0, const uint16_t* a1q, int ldaq1,
             const uint16_t* b0q, int ldbq0, const uint16_t* b1q, int ldbq1, m2f_stream_t stream) {
    GemmBatch gb;
    memset(&gb, 0, sizeof(gb));
    GemmProblem p = gp_make(a0, lda0, b0, ldb0, M, N, K0, c, ldc);
    if (K1 > 0) gp_seg2(p, a1, lda1, b1, ldb1, K1);
    p.bias = bias; p.res = res; p.ldres = ldres; p.gate = gate; p.ldgate = ldgate; p.gate_scale = gate_scale;
    p.bias_grad = bias_grad; p.drop_site = drop_site;
    p.flags = (relu_a ? GF_RELU_A : 0) | (relu_b ? GF_RELU_B : 0) | (relu_out == 1 ? GF_RELU_OUT : 0) |
              (relu_out == 2 ? GF_GELU_OUT : 0) | (accumulate ? GF_ACCUM : 0) | (g_shadow_only && !accumulate ? GF_NO_F32 : 0);
    p.a.q[0] = a0q; p.a.ldq[0] = ldaq0; p.a.q[1] = a1q; p.a.ldq[1] = ldaq1;
    p.b.q[0] = b0q; p.b.ldq[0] = ldbq0; p.b.q[1] = b1q; p.b.ldq[1] = ldbq1;
    p.a.qt[0] = p.a.qt[1] = p.b.qt[0] = p.b.qt[1] = nullptr;
    gb.pr[0] = p; gb.count = 1; gb.rng = rng_state; gb.sh = g_sh;
    gb.splitk_ws = splitk_ws; gb.splitk_cnt = splitk_tickets; gb.splitk_max_tiles = splitk_ws && splitk_tickets ? splitk_max_tiles : 0;
    if (drop_site) drop_params(drop_p, &gb.drop_thresh, &gb.drop_scale);
    M2F_HIP(m2f_launch_gemm(gb, precision, layout, tile, static_cast<hipStream_t>(stream)));
    return 0;
}

int m2f_quantize_fp8(const float* src, uint8_t* dst, int64_t n, float scale, m2f_stream_t stream) {
    M2F_HIP(m2f_launch_quant_fp8(src, dst, n, scale, static_cast<hipStream_t>(stream)));
    return 0;
}

int m2f_gemm_fp8(int M, int N, int K, const uint8_t* a8, int lda, const uint8_t* b8, int ldb, float acc_scale, float* c, int ldc,
                 const float* bias, const float* res, int ldres, int activation, uint8_t* c8, float c8_scale,
                 m2f_stream_t stream) {
    GemmBatch gb;
    memset(&gb, 0, sizeof(gb));
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.a.q[0] = reinterpret_cast<const uint16_t*>(a8); p.a.ldq[0] = lda; p.a.k[0] = K;
    p.b.q[0] = reinterpret_cast<const uint16_t*>(b8); p.b.ldq[0] = ldb; p.b.k[0] = K;
    p.M = M; p.N = N; p.c = c; p.ldc = ldc; p.bias = bias; p.res = res; p.ldres = ldres;
    p.gate_scale = 1.f; p.acc_scale = acc_scale; p.c8 = c8; p.c8_scale = c8_scale;
    if (!c && !c8) return fail("m2f_gemm_fp8: no output buffer");
    p.flags = (activation == 1 ? GF_RELU_OUT : 0) | (activation == 2 ? GF_GELU_OUT : 0) | (g_shadow_only && c && !c8 ? GF_NO_F32 : 0);
    gb.pr[0] = p; gb.count = 1; gb.sh = g_sh;
    M2F_HIP(m2f_launch_gemm_fp8(gb, static_cast<hipStream_t>(stream)));
    return 0;
}

static void attn_fill(AttnBatch& ab, int B, int L, int H, int hd, const float* q, int ldq, const float* k, int ldk,
                      const float* v, int ldv, const uint8_t* key_pad, float* out, int ldo, float* probs, uint32_t drop_site,
                      float drop_p, const uint32_t* rng_state) {
    memset(&ab, 0, sizeof(ab));
    AttnProblem& p = ab.pr[0];
    p.q = q; p.k = k; p.v = v; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.out = out; p.ldo = ldo; p.probs = probs;
    p.H = H; p.hd = hd; p.drop_site = drop_site;
    ab.count = 1; ab.B = B; ab.L = L; ab.key_pad = key_pad; ab.rng = rng_state;
    ab.drop_scale = 1.f;
    ab.sh = g_sh;                                              // (m2f_set_shadow_map: operands / results inside that range have bf16 copies)
    {   // kernel-level tests of the bf16-mode forms: M2F_ATTN_BF16_KERNEL=<mask> (AttnBatch::bf16_math), read per call
        const char* e = getenv("M2F_ATTN_BF16_KERNEL");
        ab.bf16_math = e ? atoi(e) & 63 : 0;
    }
    if (drop_site) drop_params(drop_p, &ab.drop_thresh, &ab.drop_scale);
}

int m2f_attention_fwd(int B, int L, int H, int hd, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                      const uint8_t* key_pad, float* out, int ldo, float* probs, uint32_t drop_site, float drop_p,
                      const uint32_t* rng_state, m2f_stream_t stream) {
    AttnBatch ab;
    attn_fill(ab, B, L, H, hd, q, ldq, k, ldk, v, ldv, key_pad, out, ldo, probs, drop_site, drop_p, rng_state);
    M2F_HIP(m2f_launch_attn_fwd(ab, static_cast<hipStream_t>(stream)));
    return 0;
}

int m2f_attention_bwd(int B, int L, int H, int hd, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                      const uint8_t* key_pad, const float* out, int ldo, const float* probs, const float* dout, int lddo,
                      float* dq, int lddq, float* dk, int lddk, float* dv, int lddv, uint32_t drop_site, float drop_p,
                      const uint32_t* rng_state, m2f_stream_t stream) {
    AttnBatch ab;
    attn_fill(ab, B, L, H, hd, q, ldq, k, ldk, v, ldv, key_pad, const_cast<float*>(out), ldo, const_cast<float*>(probs),
              drop_site, drop_p, rng_state);
    AttnProblem& p = ab.pr[0];
    p.dout = dout; p.lddo = lddo; p.dq = dq; p.dk = dk; p.dv = dv; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    M2F_HIP(m2f_launch_attn_bwd(ab, static_cast<hipStream_t>(stream)));
    return 0;
}

int64_t m2f_attention_probs_elems(int B, int H, int L) { return (int64_t)m2f_attn_probs_elems(B, H, L); }

int m2f_layernorm_fwd(int T, int d, const float* x, const float* gamma, const float* beta, const float* res, float* out,
                      float* stats, float eps, m2f_stream_t stream) {
    LnBatch lb;
    memset(&lb, 0, sizeof(lb));
    LnProblem& p = lb.pr[0];
    p.x = x; p.gamma = gamma; p.beta = beta; p.res = res; p.out = out; p.stats = stats; p.d = d;
    lb.count = 1; lb.T = T; lb.eps = eps; lb.drop_scale = 1.f; lb.sh = g_sh;
    M2F_HIP(m2f_launch_ln_fwd(lb, static_cast<hipStream_t>(stream)));
    return 0;
}

/* diagnostic (tools/ln_stats_ab.py; not in include/m2fnet_hip.h's product surface): the LayerNorm forward over `count` problems merged in one launch
 * as the plans do (x / out of problem i at x + i * T * ld ... the caller lays them out), statistics computed (pre = 0) or read from `stats` (pre = 1) */
int m2f_layernorm_fwd_diag(int T, int n_prob, const int* d, const float* const* x, const float* const* gamma, const float* const* beta, float* const* out,
                           float* const* stats, float eps, int pre, m2f_stream_t stream) {
    if (n_prob < 1 || n_prob > M2F_LN_MAX_PROBLEMS) return fail("m2f_layernorm_fwd_diag: 1..4 problems");
    LnBatch lb;
    memset(&lb, 0, sizeof(lb));
    for (int i = 0; i < n_prob; ++i) {
        LnProblem& p = lb.pr[i];
        p.x = x[i]; p.gamma = gamma[i]; p.beta = beta[i]; p.out = out[i]; p.stats = stats[i]; p.d = d[i];
    }
    lb.count = n_prob; lb.T = T; lb.eps = eps; lb.drop_scale = 1.f; lb.sh = g_sh; lb.pre_stats = pre;
    M2F_HIP(m2f_launch_ln_fwd(lb, static_cast<hipStream_t>(stream)));
    return 0;
}

int m2f_layernorm_fwd_out8(int T, int d, const float* x, const float* gamma, const float* beta, const float* res, float* out,
                           float* stats, float eps, uint8_t* out8, float out8_scale, m2f_stream_t stream) {
    if (!out8 || (d & 3) || (reinterpret_cast<uintptr_t>(out8) & 3)) return fail("m2f_layernorm_fwd_out8: e4m3 output needs d % 4 == 0 and a 4-byte aligned buffer");
    LnBatch lb;
    memset(&lb, 0, sizeof(lb));
    LnProblem& p = lb.pr[0];
    p.x = x; p.gamma = gamma; p.beta = beta; p.res = res; p.out = out; p.stats = stats; p.d = d;
    lb.count = 1; lb.T = T; lb.eps = eps; lb.drop_scale = 1.f; lb.sh = g_sh; lb.out8 = out8; lb.out8_scale = out8_scale;
    M2F_HIP(m2f_launch_ln_fwd(lb, static_cast<hipStream_t>(stream)));
    return 0;
}

int m2f_embed_layernorm(int T, int d, const int64_t* input_ids, const int64_t* position_ids, const float* word_emb,
                        const float* pos_emb, const float* type_emb_row0, const float* gamma, const float* beta, float eps,
                        float* out, int ld_out, m2f_stream_t stream) {
    M2F_HIP(m2f_launch_embed_ln(input_ids, position_ids, word_emb, pos_emb, type_emb_row0, gamma, beta, eps, out, ld_out, T, d, g_sh,
                                static_cast<hipStream_t>(stream)));
    return 0;
}

int m2f_attention_long_fwd(int B, int S, int H, int hd, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                           const uint8_t* key_pad, float* out, int ldo, m2f_stream_t stream) {
    M2F_HIP(m2f_launch_attn_long_fwd(q, ldq, k, ldk, v, ldv, key_pad, out, ldo, B, S, H, hd, g_sh, static_cast<hipStream_t>(stream)));
    return 0;
}

int m2f_attention_long_fwd_bf16(int B, int S, int H, int hd, const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v,
                                int ldv, const uint8_t* key_pad, uint16_t* out16, float* out32, int ldo, m2f_stream_t stream) {
    return m2f_attention_long_fwd_bf16_out8(B, S, H, hd, q, ldq, k, ldk, v, ldv, key_pad, out16, out32, nullptr, 1.f, ldo, stream);
}

int m2f_attention_long_fwd_bf16_out8(int B, int S, int H, int hd, const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v,
                                     int ldv, const uint8_t* key_pad, uint16_t* out16, float* out32, uint8_t* out8, float out8_scale, int ldo,
                                     m2f_stream_t stream) {
    if (!q || !k || !v || !(out16 || out32 || out8)) return fail("m2f_attention_long_fwd_bf16: NULL operand / no output");
    if (hd < 8 || hd > 128 || (hd & 7) || ((ldq | ldk | ldv | ldo) & 7))
        return fail("m2f_attention_long_fwd_bf16: head dim and leading dimensions must be multiples of 8, head dim <= 128");
    if (out8 && ((hd & 15) || (ldo & 15))) return fail("m2f_attention_long_fwd_bf16: the e4m3 output needs head dim and ldo in multiples of 16");
    M2F_HIP(m2f_launch_attn_long_fwd_bf16(q, ldq, k, ldk, v, ldv, key_pad, out16, out32, out8, out8_scale, ldo, B, S, H, hd, static_cast<hipStream_t>(stream)));
    return 0;
}

int m2f_layernorm_bwd(int T, int d, const float* x, const float* gamma, const float* stats, const float* dy, const float* extra,
                      float* dx, float* partial, float* dgamma, float* dbeta, m2f_stream_t stream) {
    LnBatch lb;
    memset(&lb, 0, sizeof(lb));
    LnProblem& p = lb.pr[0];
    p.x = x; p.gamma = gamma; p.stats = const_cast<float*>(stats); p.dy = dy; p.extra = extra; p.dx = dx; p.partial = partial; p.d = d;
    lb.count = 1; lb.T = T; lb.eps = 0.f; lb.drop_scale = 1.f;
    M2F_HIP(m2f_launch_ln_bwd(lb, static_cast<hipStream_t>(stream)));
    LnReduceBatch rb;
    memset(&rb, 0, sizeof(rb));
    rb.it[0].partial = partial; rb.it[0].dgamma = dgamma; rb.it[0].dbeta = dbeta; rb.it[0].d = d; rb.it[0].nblk = m2f_ln_row_blocks(T);
    rb.count = 1;
    M2F_HIP(m2f_launch_ln_param_reduce(rb, static_cast<hipStream_t>(stream)));
    return 0;
}

int m2f_cross_entropy(int T, int C, const float* logits, const int64_t* labels, const float* class_w, float label_smoothing,
                      int normalise, float* loss_terms, float* dlogits, float* loss_out, m2f_stream_t stream) {
    CeArgs a;
    a.logits = logits; a.T = T; a.C = C; a.labels = labels; a.class_w = class_w; a.label_smoothing = label_smoothing;
    a.loss_terms = loss_terms; a.dlogits = dlogits;
    M2F_HIP(m2f_launch_ce(a, static_cast<hipStream_t>(stream)));
    M2F_HIP(m2f_launch_loss_finalize(loss_terms, T, C, dlogits, loss_out, normalise, static_cast<hipStream_t>(stream)));
    return 0;
}

}  // extern "C"
