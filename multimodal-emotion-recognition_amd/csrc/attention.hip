// Dialogue-level multi-head attention, forward and backward, for gfx950 (wave64).
//
// Replaces the attention inside nn.MultiheadAttention for both uses in the reference:
//   * encoder self-attention  (src/model.py:107,119 -> TransformerEncoderLayer._sa_block)
//   * FusionAttentionModule   (src/model.py:14: query = text, key = audio, value = text)
// with key_padding_mask semantics (-inf on padded keys before the softmax).
//
// One workgroup of four wavefronts owns one (dialogue, head): the sequence is the utterances of a dialogue
// (L <= 64), so the whole L x L problem fits one wave's registers.  Q/K/V (and dO, O in backward) tiles of
// the head are staged in LDS by all 256 threads (row stride = 2 mod 4 floats -> conflict-free MFMA fragment
// reads; every operand is in flight before the first one is committed, so the fetch costs one memory round
// trip), QK^T and PV run on the exact-fp32 MFMA v_mfma_f32_16x16x4_f32, the softmax runs in registers with
// wavefront shuffles.  S^T = K Q^T is computed so the probability tile is already laid out as the A operand of
// the PV product (accumulator as next operand, no LDS round trip).  Every wave evaluates the (tiny) score /
// dS tiles itself and the waves split the 16-column output tiles of O / dQ / dK / dV between them: the kernel
// is a chain of latencies (it sits in the dependent chain of the training step), so the work of one problem is
// spread over 4 SIMDs instead of being queued on one.  The backward pass evaluates dS in both orientations
// (cheap at these sizes) so dQ and dK/dV need no transposes and no atomics.
#include "common.h"
#include "ops.h"
#include <cstdlib>

// No floating-point contraction in this file: several kernels restate the same formulas in different surroundings (ring and
// register-staged GEMM epilogues, fp32 and bf16 attention forms, flat and shadow-writing Adam), and with -ffp-contract=fast (the
// HIP default) the compiler is free to fuse a*b+c in one of them and not in the other - a 1-ulp difference that would break the
// bit-for-bit comparisons the tests make between the forms.  These kernels are bound by memory or by MFMA, not by VALU multiplies.
#pragma clang fp contract(off)

namespace {

constexpr int NTHR = 256;          // 4 wavefronts per (dialogue, head)
constexpr int NWAVE = NTHR / 64;

// [Lp x W] zero-padded LDS copy of src rows [0, L) x cols [0, hd), generic form (any size / alignment).  Loads are
// UNCONDITIONAL (clamped address + select) and issued in batches before any LDS write (guarded loads compile to
// branch + s_waitcnt vmcnt(0) each).
__device__ __forceinline__ void load_slab(float* __restrict__ lds, int ld, int Lp, int W,
                                          const float* __restrict__ src, int ldg, int L, int hd, int tid) {
    const int total = Lp * W;
#pragma unroll 1
    for (int base = 0; base < total; base += NTHR * 4) {
        float x[4];
        int off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = base + tid + NTHR * u;
            const int r = e / W, c = e - r * W;
            const bool ok = e < total && r < L && c < hd;
            off[u] = e < total ? r * ld + c : -1;
            x[u] = src[ok ? (size_t)r * ldg + c : (size_t)0];
            if (!ok) x[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (off[u] >= 0) lds[off[u]] = x[u];
    }
}

// Fast form for slabs of at most NTHR*NV float4 whose rows are 16-byte aligned: the (row, column) of each of a thread's
// (up to) NV float4 is worked out ONCE (one integer division) and shared by every slab of the kernel (same L, hd, W),
// all slabs are issued before the first is committed.  NV = 2 * (Lp / 16) covers every head dim <= 128, so the generic
// (scalar, division-heavy) form below only serves unaligned operands; it is kept small on purpose.
template <int NV>
struct SlabGeom {
    int goff_rc[NV];     // r * 65536 + c   (r < 64, c < 256)
    int loff[NV];        // r * ld + c  (LDS float offset)
    bool inb[NV];        // element index < total (a slot of this thread exists)
    bool ok[NV];         // ... and lies inside [0, L) x [0, hd)
};
template <int NV>
__device__ __forceinline__ void slab_geom(SlabGeom<NV>& G, int L, int hd, int Lp, int W, int ld, int tid) {
    const int C4 = W >> 2, total = Lp * C4;
    int r = tid / C4, c4 = tid - r * C4;
    const int dr = NTHR / C4, dc = NTHR - dr * C4;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int e = tid + NTHR * u;
        const int c = c4 << 2;
        G.inb[u] = e < total;
        G.ok[u] = G.inb[u] && r < L && c < hd;
        G.goff_rc[u] = (r << 16) | c;
        G.loff[u] = r * ld + c;
        r += dr; c4 += dc;
        if (c4 >= C4) { c4 -= C4; ++r; }
    }
}
template <int NV>
__device__ __forceinline__ bool slab_fast_ok(const float* src, int ldg, int hd, int Lp, int W) {
    return ((hd & 3) == 0) && ((ldg & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (Lp * (W >> 2) <= NTHR * NV);
}
template <int NV>
struct SlabRegs { f32x4 x[NV]; uint32_t w0[NV], w1[NV]; };     // w0 / w1: the raw 4 bf16 of a chunk staged from a shadow
template <int NV>
__device__ __forceinline__ void slab_issue(SlabRegs<NV>& R, const SlabGeom<NV>& G, const float* __restrict__ src, int ldg) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int r = G.goff_rc[u] >> 16, c = G.goff_rc[u] & 0xFFFF;
        const uint32_t o = G.ok[u] ? (uint32_t)(r * ldg + c) * 4u : 0u;
        R.x[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(src) + (size_t)o);
    }
}
// bf16 mode: the same slab from the operand's bf16 SHADOW (same element index; the producer wrote both copies): half the bytes
// of the kernels' dominant cost - at C3 a merged encoder launch of the backward kernel read 37 MB of fp32 slabs.  The raw
// 8 bytes (4 bf16) wait in w0 / w1 until slab_value() widens them (exact: bf16 -> fp32 is a shift).
template <int NV>
__device__ __forceinline__ void slab_issue16(SlabRegs<NV>& R, const SlabGeom<NV>& G, const uint16_t* __restrict__ src, int ldg) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int r = G.goff_rc[u] >> 16, c = G.goff_rc[u] & 0xFFFF;
        const uint32_t o = G.ok[u] ? (uint32_t)(r * ldg + c) * 2u : 0u;
        const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(src) + (size_t)o);
        R.w0[u] = w.x; R.w1[u] = w.y;
    }
}
template <int NV>
__device__ __forceinline__ f32x4 slab_value(const SlabRegs<NV>& R, int u, bool from16) {
    if (!from16) return R.x[u];
    const uint32_t a = R.w0[u], b = R.w1[u];
    const f32x4 v = {__builtin_bit_cast(float, a << 16), __builtin_bit_cast(float, a & 0xFFFF0000u),
                     __builtin_bit_cast(float, b << 16), __builtin_bit_cast(float, b & 0xFFFF0000u)};
    return v;
}
// the bf16 shadow of `p` if this launch may stage from it (bf16 mode, shadowed buffer, 8-byte aligned rows), else null
__device__ __forceinline__ const uint16_t* slab_shadow(const AttnBatch& ab, const float* p, int ldg, int hd, int bit) {
    if (!(ab.bf16_math & bit)) return nullptr;
    const uint16_t* q = m2f_shadow_of(ab.sh, p);
    return (q && ((reinterpret_cast<uintptr_t>(q) & 7) == 0) && ((ldg & 3) == 0) && ((hd & 3) == 0)) ? q : nullptr;
}

template <int NV>
__device__ __forceinline__ void slab_commit(const SlabRegs<NV>& R, const SlabGeom<NV>& G, float* __restrict__ lds, bool from16 = false) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        if (G.inb[u]) {
            const bool ok = G.ok[u];
            const f32x4 v = slab_value(R, u, from16);
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2* d = reinterpret_cast<f32x2*>(lds + G.loff[u]);          // r*(W+2) + c is even: 8-byte aligned
            d[0] = f32x2{ok ? v[0] : 0.f, ok ? v[1] : 0.f};
            d[1] = f32x2{ok ? v[2] : 0.f, ok ? v[3] : 0.f};
        }
    }
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// bf16 mode: C[m][n] = sum_k A[m][k] B[n][k] over the W (padded head dim) columns of two fp32 LDS slab rows per lane, operands
// rounded to bf16 on the way into v_mfma_f32_16x16x32_bf16 (lane l15 = its row of A and of B, lane >> 4 = which 8 of the 32 k).
// Same C layout as the chain of mfma4's it replaces: 4 MFMAs of 16 cycles for head dim 128 instead of 32 of 32 cycles - with four
// workgroups per CU the exact-fp32 contractions were ~5 us of the backward kernel's 19.
__device__ __forceinline__ bf16x8 slab8_bf16(const float* p, bool in) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2* q = reinterpret_cast<const f32x2*>(p);                  // 8-byte aligned: slab strides and k offsets are even
    const f32x2 z = {0.f, 0.f};
    const f32x2 a = in ? q[0] : z, b = in ? q[1] : z, c = in ? q[2] : z, d = in ? q[3] : z;
    const bf16x8 r = {(__bf16)a[0], (__bf16)a[1], (__bf16)b[0], (__bf16)b[1], (__bf16)c[0], (__bf16)c[1], (__bf16)d[0], (__bf16)d[1]};
    return r;
}
__device__ __forceinline__ f32x4 dot_rows_bf16(const float* arow, const float* brow, int W, int lg) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < W; k0 += 32) {
        const int k = k0 + 8 * lg;
        const bool in = k < W;                                           // W is a multiple of 16: the last step may be half a step
        const int kc = in ? k : 0;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(slab8_bf16(arow + kc, in), slab8_bf16(brow + kc, in), acc, 0, 0, 0);
    }
    return acc;
}

template <int NT>
__global__ __launch_bounds__(NTHR) void m2f_attn_fwd_kernel(const AttnBatch ab) {
    static_assert(sizeof(AttnBatch) + 56 <= 192 + 512, "m2f_kernarg_warm ranges no longer cover AttnBatch + the hidden arguments");
    m2f_kernarg_warm<0, 8, 192>();                  // the descriptor block (648 B + hidden arguments) in one miss
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < M2F_ATTN_MAX_PROBLEMS; ++i)
        if ((int)blockIdx.x >= ab.bb[i]) pi = i;
    const AttnProblem& P = ab.pr[pi];
    const int H = P.H, hd = P.hd, LM = ab.L;                   // LM: the plan's utterances per dialogue (slab size, RNG index)
    const int bh = (int)blockIdx.x - P.block_begin;
    const int b = bh / H, h = bh - b * H;
    constexpr int Lp = 16 * NT;
    const int W = (hd + 15) & ~15, ld = W + 2;
    float* Qs = sm;
    float* Ks = Qs + Lp * ld;
    float* Vs = Ks + Lp * ld;
    // padded layout: dialogue b owns rows b*LM .. +LM-1, pads flagged in key_pad; packed layout: rows cu[b] .. cu[b+1]-1, all valid
    int L = LM;
    size_t tok0 = (size_t)b * LM;
    if (ab.cu) { const int c0 = ab.cu[b]; L = ab.cu[b + 1] - c0; tok0 = (size_t)c0; }
    const float* qg = P.q + tok0 * P.ldq + h * hd;
    const float* kg = P.k + tok0 * P.ldk + h * hd;
    const float* vg = P.v + tok0 * P.ldv + h * hd;
    const unsigned char kpad = ab.cu ? (unsigned char)0 : ab.key_pad[tok0 + (lane < L ? lane : 0)];
    constexpr int NV = 2 * NT;                                // float4 per thread and slab: covers W <= 128
    if (slab_fast_ok<NV>(qg, P.ldq, hd, Lp, W) && slab_fast_ok<NV>(kg, P.ldk, hd, Lp, W) && slab_fast_ok<NV>(vg, P.ldv, hd, Lp, W)) {
        SlabGeom<NV> G;
        slab_geom(G, L, hd, Lp, W, ld, tid);
        SlabRegs<NV> rq, rk, rv;                              // one round trip for the three operands
        const uint16_t* q16 = slab_shadow(ab, qg, P.ldq, hd, 2);
        const uint16_t* k16 = slab_shadow(ab, kg, P.ldk, hd, 4);
        const uint16_t* v16 = slab_shadow(ab, vg, P.ldv, hd, 8);
        if (q16) slab_issue16(rq, G, q16, P.ldq); else slab_issue(rq, G, qg, P.ldq);
        if (k16) slab_issue16(rk, G, k16, P.ldk); else slab_issue(rk, G, kg, P.ldk);
        if (v16) slab_issue16(rv, G, v16, P.ldv); else slab_issue(rv, G, vg, P.ldv);
        slab_commit(rq, G, Qs, q16 != nullptr);
        slab_commit(rk, G, Ks, k16 != nullptr);
        slab_commit(rv, G, Vs, v16 != nullptr);
    } else {
        load_slab(Qs, ld, Lp, W, qg, P.ldq, L, hd, tid);
        load_slab(Ks, ld, Lp, W, kg, P.ldk, L, hd, tid);
        load_slab(Vs, ld, Lp, W, vg, P.ldv, L, hd, tid);
    }
    const unsigned long long kvalid = __ballot(lane < L && kpad == 0);
    __syncthreads();

    const float scale = 1.0f / sqrtf((float)hd);
    const int l15 = lane & 15, lg = lane >> 4;
    const int ksteps = (hd + 3) >> 2;
    const uint32_t site = P.drop_site;
    uint32_t key = 0;
    if (site) key = m2f_site_key(ab.rng, site);
    float* probs = P.probs + (size_t)bh * Lp * Lp;
    uint16_t* out16 = m2f_shadow_of(ab.sh, P.out);
    const bool w32 = !(P.no_f32 && out16);                  // (no fp32 reader: the bf16 shadow is the result)

#pragma unroll 1
    for (int it = 0; it < NT; ++it) {
        f32x4 s[NT];
        const int i = 16 * it + l15;                        // this lane's query row
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            if (ab.bf16_math & 1) {
                s[jt] = dot_rows_bf16(Ks + (16 * jt + l15) * ld, Qs + i * ld, W, lg);
                continue;
            }
            // two interleaved accumulators: the 16x16x4 MFMA has a 40-cycle dependent latency at an 8..32-cycle issue
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* kp = Ks + (16 * jt + l15) * ld + lg;
            const float* qp = Qs + i * ld + lg;
            int ks = 0;
            for (; ks + 1 < ksteps; ks += 2) {
                acc0 = mfma4(kp[4 * ks], qp[4 * ks], acc0);
                acc1 = mfma4(kp[4 * ks + 4], qp[4 * ks + 4], acc1);
            }
            if (ks < ksteps) acc0 = mfma4(kp[4 * ks], qp[4 * ks], acc0);
            s[jt] = acc0 + acc1;                            // S[i][j = 16jt + 4lg + r]
        }
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * lg + r;
                const float v = ((kvalid >> j) & 1ull) ? s[jt][r] * scale : -INFINITY;
                s[jt][r] = v;
                m = fmaxf(m, v);
            }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[jt][r] - m);
                s[jt][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = (i < L) ? 1.0f / sum : 0.f;       // padded query rows of the tile: P = 0
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * lg + r;
                float p = s[jt][r] * inv;
                if (wv == 0) probs[(size_t)j * Lp + i] = p;  // P^T, pre-dropout (lanes: consecutive i); one wave writes it
                if (site) p = m2f_keep(key, (uint32_t)((bh * LM + i) * LM + j), ab.drop_thresh) ? p * ab.drop_scale : 0.f;
                s[jt][r] = p;
            }
        // O[i][c] = sum_j P[i][j] V[j][c]; the 16-column tiles are dealt to the four waves
        for (int ct = wv; ct < (W >> 4); ct += NWAVE) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const float* vp = Vs + (16 * jt + 4 * lg) * ld + 16 * ct + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) o = mfma4(s[jt][r], vp[r * ld], o);
            }
            const int c = 16 * ct + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int io = 16 * it + 4 * lg + r;
                if (io < L && c < hd) {
                    const size_t idx = (tok0 + io) * P.ldo + h * hd + c;
                    if (w32) P.out[idx] = o[r];
                    if (out16) out16[idx] = m2f_bf16_bits(o[r]);
                }
            }
        }
    }
    // packed layout: the token rows behind the last dialogue (cu[B] .. T-1) belong to nobody; a plan is re-used for batches of
    // other lengths, so they are written (zeros) by the last dialogue's workgroups - head h its own columns
    if (ab.cu && b == ab.B - 1) {
        for (int r = ab.cu[ab.B] + wv; r < ab.T; r += NWAVE)
            for (int c = lane; c < hd; c += 64) {
                const size_t idx = (size_t)r * P.ldo + h * hd + c;
                if (w32) P.out[idx] = 0.f;
                if (out16) out16[idx] = 0;
            }
    }
}

template <int NT>
__global__ __launch_bounds__(NTHR) void m2f_attn_bwd_kernel(const AttnBatch ab) {
    static_assert(sizeof(AttnBatch) + 56 <= 192 + 512, "m2f_kernarg_warm ranges no longer cover AttnBatch + the hidden arguments");
    m2f_kernarg_warm<0, 8, 192>();                  // the descriptor block (648 B + hidden arguments) in one miss
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < M2F_ATTN_MAX_PROBLEMS; ++i)
        if ((int)blockIdx.x >= ab.bb[i]) pi = i;
    const AttnProblem& P = ab.pr[pi];
    const int H = P.H, hd = P.hd, LM = ab.L;
    const int bh = (int)blockIdx.x - P.block_begin;
    const int b = bh / H, h = bh - b * H;
    constexpr int Lp = 16 * NT;
    const int W = (hd + 15) & ~15, ld = W + 2;
    float* Qs = sm;
    float* Ks = Qs + Lp * ld;
    float* Vs = Ks + Lp * ld;
    float* Gs = Vs + Lp * ld;          // dO
    float* Os = Gs + Lp * ld;          // O (only when ab.bwd_fast == 1: the one-round-trip path that keeps an O slab)
    float* delta = Os + (ab.bwd_fast == 1 ? Lp * ld : 0);  // [Lp]
    float* dpart = delta + Lp;         // [NV][NTHR] (ab.bwd_fast == 2: per-thread pieces of delta, summed per row in a fixed order)
    int L = LM;                                            // (packed layout: see the forward kernel)
    size_t tok0 = (size_t)b * LM;
    if (ab.cu) { const int c0 = ab.cu[b]; L = ab.cu[b + 1] - c0; tok0 = (size_t)c0; }
    const float* qg = P.q + tok0 * P.ldq + h * hd;
    const float* kg = P.k + tok0 * P.ldk + h * hd;
    const float* vg = P.v + tok0 * P.ldv + h * hd;
    const float* gg = P.dout + tok0 * P.lddo + h * hd;
    const float* og = P.out + tok0 * P.ldo + h * hd;
    const int l15 = lane & 15, lg = lane >> 4;
    const float* probs = P.probs + (size_t)bh * Lp * Lp;
    // NT == 1: the saved probabilities this lane needs in both orientations, fetched with everything else
    float px[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 py = {0.f, 0.f, 0.f, 0.f};
    constexpr int NV = 2 * NT;                                // float4 per thread and slab: covers W <= 128
    const bool fast = ab.bwd_fast && slab_fast_ok<NV>(qg, P.ldq, hd, Lp, W) && slab_fast_ok<NV>(kg, P.ldk, hd, Lp, W) &&
                      slab_fast_ok<NV>(vg, P.ldv, hd, Lp, W) && slab_fast_ok<NV>(gg, P.lddo, hd, Lp, W) && slab_fast_ok<NV>(og, P.ldo, hd, Lp, W);
    if (fast) {
        SlabGeom<NV> G;
        slab_geom(G, L, hd, Lp, W, ld, tid);
        SlabRegs<NV> rq, rk, rv, rg, ro;                      // one round trip for all five operands (+ the probabilities)
        const uint16_t* g16 = slab_shadow(ab, gg, P.lddo, hd, 16);
        const uint16_t* o16 = slab_shadow(ab, og, P.ldo, hd, 32);
        const uint16_t* v16 = slab_shadow(ab, vg, P.ldv, hd, 8);
        const uint16_t* k16 = slab_shadow(ab, kg, P.ldk, hd, 4);
        const uint16_t* q16 = slab_shadow(ab, qg, P.ldq, hd, 2);
        if (g16) slab_issue16(rg, G, g16, P.lddo); else slab_issue(rg, G, gg, P.lddo);
        if (o16) slab_issue16(ro, G, o16, P.ldo); else slab_issue(ro, G, og, P.ldo);
        if (v16) slab_issue16(rv, G, v16, P.ldv); else slab_issue(rv, G, vg, P.ldv);
        if (k16) slab_issue16(rk, G, k16, P.ldk); else slab_issue(rk, G, kg, P.ldk);
        if (q16) slab_issue16(rq, G, q16, P.ldq); else slab_issue(rq, G, qg, P.ldq);
        if constexpr (NT == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) px[r] = probs[(size_t)(4 * lg + r) * Lp + l15];
            py = *reinterpret_cast<const f32x4*>(probs + (size_t)l15 * Lp + 4 * lg);
        }
        slab_commit(rg, G, Gs, g16 != nullptr);
        if (ab.bwd_fast == 1) slab_commit(ro, G, Os, o16 != nullptr);
        else {
            // delta_i = sum_c dO[i][c] O[i][c] straight from the registers both operands arrived in: no O slab, so four
            // workgroups fit a CU's LDS (4 x 8.3 KB slabs at head dim 128) and the 1,024 (dialogue, head) problems of a merged
            // encoder launch at C3 run in ONE round instead of 768 + 256.  Thread t's u-th float4 is element 4 (t + NTHR u) of
            // the row-major [Lp][W] slab; row r sums its W / 4 pieces in index order (deterministic).
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const f32x4 a = slab_value(rg, u, g16 != nullptr), b = slab_value(ro, u, o16 != nullptr);
                dpart[u * NTHR + tid] = G.ok[u] ? (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]) : 0.f;
            }
        }
        slab_commit(rv, G, Vs, v16 != nullptr);
        slab_commit(rk, G, Ks, k16 != nullptr);
        slab_commit(rq, G, Qs, q16 != nullptr);
    } else {
        load_slab(Qs, ld, Lp, W, qg, P.ldq, L, hd, tid);
        load_slab(Ks, ld, Lp, W, kg, P.ldk, L, hd, tid);
        load_slab(Vs, ld, Lp, W, vg, P.ldv, L, hd, tid);
        load_slab(Gs, ld, Lp, W, gg, P.lddo, L, hd, tid);
        if constexpr (NT == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) px[r] = probs[(size_t)(4 * lg + r) * Lp + l15];
            py = *reinterpret_cast<const f32x4*>(probs + (size_t)l15 * Lp + 4 * lg);
        }
    }
    __syncthreads();
    // delta_i = sum_c dO[i][c] * O[i][c]  (= sum_j P[i][j] dP[i][j], also under dropout).  Four lanes per row, each
    // takes a quarter of the row (O from its LDS slab, or streamed from global memory on the slow path), shuffle-reduce.
    if (fast && ab.bwd_fast == 2) {
        if (tid < Lp) {
            const int C4 = W >> 2;
            float d = 0.f;
            for (int e = tid * C4; e < (tid + 1) * C4; ++e) d += dpart[e];     // e = t + NTHR u  ->  dpart[u * NTHR + t] = dpart[e]
            delta[tid] = tid < L ? d : 0.f;
        }
    } else
    for (int r0 = 0; r0 < Lp; r0 += NTHR / 4) {
        const int row = r0 + (tid >> 2), part = tid & 3;
        const bool rin = row < Lp, rok = row < L;
        const int rowc = rin ? row : 0;
        const float* o = fast ? Os + rowc * ld : og + (size_t)(rok ? row : 0) * P.ldo;
        const float* g = Gs + rowc * ld;
        float d = 0.f;
        for (int c = part; c < hd; c += 4) d += g[c] * o[c];
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        if (part == 0 && rin) delta[row] = rok ? d : 0.f;
    }
    __syncthreads();

    const float scale = 1.0f / sqrtf((float)hd);
    const int ksteps = (hd + 3) >> 2;
    const uint32_t site = P.drop_site;
    uint32_t key = 0;
    if (site) key = m2f_site_key(ab.rng, site);
    uint16_t* dq16 = m2f_shadow_of(ab.sh, P.dq);
    uint16_t* dk16 = m2f_shadow_of(ab.sh, P.dk);
    uint16_t* dv16 = m2f_shadow_of(ab.sh, P.dv);
    const bool w32 = !(P.no_f32 && dq16 && dk16 && dv16);

    // ---- orientation X: lane = query row i, registers = keys j  ->  dQ = dS K ----------------------
#pragma unroll 1
    for (int it = 0; it < NT; ++it) {
        f32x4 ds[NT];
        const int i = 16 * it + l15;
        const float dl = delta[i];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            if (ab.bf16_math & 1) {
                acc0 = dot_rows_bf16(Vs + (16 * jt + l15) * ld, Gs + i * ld, W, lg);
            } else {
            const float* vp = Vs + (16 * jt + l15) * ld + lg;
            const float* gp = Gs + i * ld + lg;
            int ks = 0;
            for (; ks + 1 < ksteps; ks += 2) {
                acc0 = mfma4(vp[4 * ks], gp[4 * ks], acc0);
                acc1 = mfma4(vp[4 * ks + 4], gp[4 * ks + 4], acc1);
            }
            if (ks < ksteps) acc0 = mfma4(vp[4 * ks], gp[4 * ks], acc0);
            }
            const f32x4 acc = acc0 + acc1;                  // dP[i][j]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * lg + r;
                const float p = (NT == 1) ? px[r] : probs[(size_t)j * Lp + i];
                float dp = acc[r];
                if (site) dp = m2f_keep(key, (uint32_t)((bh * LM + i) * LM + j), ab.drop_thresh) ? dp * ab.drop_scale : 0.f;
                ds[jt][r] = p * (dp - dl) * scale;
            }
        }
        for (int ct = wv; ct < (W >> 4); ct += NWAVE) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const float* kp = Ks + (16 * jt + 4 * lg) * ld + 16 * ct + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) o = mfma4(ds[jt][r], kp[r * ld], o);
            }
            const int c = 16 * ct + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int io = 16 * it + 4 * lg + r;
                if (io < L && c < hd) {
                    const size_t idx = (tok0 + io) * P.lddq + h * hd + c;
                    if (w32) P.dq[idx] = o[r];
                    if (dq16) dq16[idx] = m2f_bf16_bits(o[r]);
                }
            }
        }
    }

    // ---- orientation Y: lane = key j, registers = query rows i  ->  dK = dS^T Q, dV = Pd^T dO ------
#pragma unroll 1
    for (int jt = 0; jt < NT; ++jt) {
        f32x4 ds[NT], pd[NT];
        const int j = 16 * jt + l15;
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            if (ab.bf16_math & 1) {
                acc0 = dot_rows_bf16(Gs + (16 * it + l15) * ld, Vs + j * ld, W, lg);
            } else {
            const float* gp = Gs + (16 * it + l15) * ld + lg;
            const float* vp = Vs + j * ld + lg;
            int ks = 0;
            for (; ks + 1 < ksteps; ks += 2) {
                acc0 = mfma4(gp[4 * ks], vp[4 * ks], acc0);
                acc1 = mfma4(gp[4 * ks + 4], vp[4 * ks + 4], acc1);
            }
            if (ks < ksteps) acc0 = mfma4(gp[4 * ks], vp[4 * ks], acc0);
            }
            const f32x4 acc = acc0 + acc1;                  // dP[i][j]
            const f32x4 p4 = (NT == 1) ? py : *reinterpret_cast<const f32x4*>(probs + (size_t)j * Lp + 16 * it + 4 * lg);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + 4 * lg + r;
                float p = p4[r], dp = acc[r], pdv = p;
                if (site) {
                    const bool kp = m2f_keep(key, (uint32_t)((bh * LM + i) * LM + j), ab.drop_thresh);
                    dp = kp ? dp * ab.drop_scale : 0.f;
                    pdv = kp ? p * ab.drop_scale : 0.f;
                }
                ds[it][r] = p * (dp - delta[i]) * scale;
                pd[it][r] = pdv;
            }
        }
        for (int ct = wv; ct < (W >> 4); ct += NWAVE) {
            f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int it = 0; it < NT; ++it) {
                const float* qp = Qs + (16 * it + 4 * lg) * ld + 16 * ct + l15;
                const float* gp = Gs + (16 * it + 4 * lg) * ld + 16 * ct + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dk = mfma4(ds[it][r], qp[r * ld], dk);
                    dv = mfma4(pd[it][r], gp[r * ld], dv);
                }
            }
            const int c = 16 * ct + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jo = 16 * jt + 4 * lg + r;
                if (jo < L && c < hd) {
                    const size_t ik = (tok0 + jo) * P.lddk + h * hd + c, iv = (tok0 + jo) * P.lddv + h * hd + c;
                    if (w32) { P.dk[ik] = dk[r]; P.dv[iv] = dv[r]; }
                    if (dk16) dk16[ik] = m2f_bf16_bits(dk[r]);
                    if (dv16) dv16[iv] = m2f_bf16_bits(dv[r]);
                }
            }
        }
    }
    // packed layout: zero gradients for the token rows behind the last dialogue (see the forward kernel) - the input-gradient
    // and weight-gradient GEMMs read every row of these buffers
    if (ab.cu && b == ab.B - 1) {
        for (int r = ab.cu[ab.B] + wv; r < ab.T; r += NWAVE)
            for (int c = lane; c < hd; c += 64) {
                const size_t iq = (size_t)r * P.lddq + h * hd + c, ik = (size_t)r * P.lddk + h * hd + c, iv = (size_t)r * P.lddv + h * hd + c;
                if (w32) { P.dq[iq] = 0.f; P.dk[ik] = 0.f; P.dv[iv] = 0.f; }
                if (dq16) dq16[iq] = 0;
                if (dk16) dk16[ik] = 0;
                if (dv16) dv16[iv] = 0;
            }
    }
}

// ---- long-sequence forward (token-level self-attention of the in-loop text encoder) ------------------------
// One workgroup = 64 queries of one (sequence, head), wave w owns queries 16w..16w+15.  Keys / values stream through LDS in
// blocks of 64 with an online softmax (running max / sum per query, accumulators rescaled per block), so any S fits.
// Same MFMA orientation as the dialogue kernel: S^T = K Q^T puts, for query (lane & 15), four keys per 16-key tile in the
// accumulator, which is exactly the A operand of the P V product; the O accumulator holds query 4*(lane>>4)+r, so the
// per-query rescale factors are fetched across lanes with four shuffles per block.  Inference only (no probabilities kept).
__global__ __launch_bounds__(NTHR) void m2f_attn_long_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v, int ldq, int ldk, int ldv,
                                                                const uint8_t* __restrict__ key_pad, float* __restrict__ out,
                                                                int ldo, int S, int H, int hd, ShadowMap sh) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int b = (int)blockIdx.y / H, h = (int)blockIdx.y - b * H;
    const int q0 = (int)blockIdx.x * 64;
    const int W = (hd + 15) & ~15, ld = W + 2, CT = W >> 4;
    float* Qs = sm;
    float* Ks = Qs + 64 * ld;
    float* Vs = Ks + 64 * ld;
    const size_t tok0 = (size_t)b * S;
    const float* qg = q + (tok0 + q0) * ldq + h * hd;
    const int nq = S - q0 < 64 ? S - q0 : 64;
    constexpr int NV = 8;                                     // 64 rows x W <= 128
    const bool fast = slab_fast_ok<NV>(qg, ldq, hd, 64, W) && slab_fast_ok<NV>(k + tok0 * ldk + h * hd, ldk, hd, 64, W) &&
                      slab_fast_ok<NV>(v + tok0 * ldv + h * hd, ldv, hd, 64, W);
    SlabGeom<NV> G;
    if (fast) {
        slab_geom(G, nq, hd, 64, W, ld, tid);
        SlabRegs<NV> rq;
        slab_issue(rq, G, qg, ldq);
        slab_commit(rq, G, Qs);
    } else {
        load_slab(Qs, ld, 64, W, qg, ldq, nq, hd, tid);
    }
    const float scale = 1.0f / sqrtf((float)hd);
    const int ksteps = (hd + 3) >> 2;
    float m_run = -INFINITY, l_run = 0.f;                       // of query 16*wv + l15 (replicated over lg)
    f32x4 o[8];                                                 // O[query 16*wv + 4*lg + r][c = 16*ct + l15]
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) o[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int i = 16 * wv + l15;                                // this lane's query row (block-local)

    for (int kb = 0; kb < S; kb += 64) {
        const int nk = S - kb < 64 ? S - kb : 64;
        __syncthreads();                                        // previous block's K / V fully consumed (and Q committed)
        const float* kg = k + (tok0 + kb) * ldk + h * hd;
        const float* vg = v + (tok0 + kb) * ldv + h * hd;
        const unsigned char kp = key_pad ? key_pad[tok0 + kb + (lane < nk ? lane : 0)] : (unsigned char)0;
        if (fast) {
            SlabGeom<NV> Gk;
            slab_geom(Gk, nk, hd, 64, W, ld, tid);
            SlabRegs<NV> rk, rv;
            slab_issue(rk, Gk, kg, ldk);
            slab_issue(rv, Gk, vg, ldv);
            slab_commit(rk, Gk, Ks);
            slab_commit(rv, Gk, Vs);
        } else {
            load_slab(Ks, ld, 64, W, kg, ldk, nk, hd, tid);
            load_slab(Vs, ld, 64, W, vg, ldv, nk, hd, tid);
        }
        const unsigned long long kvalid = __ballot(lane < nk && kp == 0);
        __syncthreads();

        f32x4 s[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* kp_ = Ks + (16 * jt + l15) * ld + lg;
            const float* qp = Qs + i * ld + lg;
            int ks = 0;
            for (; ks + 1 < ksteps; ks += 2) {
                acc0 = mfma4(kp_[4 * ks], qp[4 * ks], acc0);
                acc1 = mfma4(kp_[4 * ks + 4], qp[4 * ks + 4], acc1);
            }
            if (ks < ksteps) acc0 = mfma4(kp_[4 * ks], qp[4 * ks], acc0);
            s[jt] = acc0 + acc1;                                // S[i][j = 16jt + 4lg + r]
        }
        float m_blk = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * lg + r;
                const float x = ((kvalid >> j) & 1ull) ? s[jt][r] * scale : -INFINITY;
                s[jt][r] = x;
                m_blk = fmaxf(m_blk, x);
            }
        m_blk = fmaxf(m_blk, __shfl_xor(m_blk, 16, 64));
        m_blk = fmaxf(m_blk, __shfl_xor(m_blk, 32, 64));
        const float m_new = fmaxf(m_run, m_blk);
        const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);     // exp(-inf) = 0 on the first live block
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = (m_new == -INFINITY) ? 0.f : __expf(s[jt][r] - m_new);
                s[jt][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        l_run = l_run * alpha + sum;
        m_run = m_new;
        // rescale the accumulators: row (4*lg + r) of this wave's 16 queries has its alpha in lanes with l15 == 4*lg + r
        float a4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a4[r] = __shfl(alpha, 4 * lg + r, 64);
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {                        // static indices keep o[] in registers
            if (ct >= CT) break;
            f32x4 acc = o[ct];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] *= a4[r];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const float* vp = Vs + (16 * jt + 4 * lg) * ld + 16 * ct + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = mfma4(s[jt][r], vp[r * ld], acc);
            }
            o[ct] = acc;
        }
    }
    float inv4[4];
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) inv4[r] = __shfl(inv, 4 * lg + r, 64);
    uint16_t* out16 = m2f_shadow_of(sh, out);
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
        if (ct >= CT) break;
        const int c = 16 * ct + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int io = q0 + 16 * wv + 4 * lg + r;
            if (io < S && c < hd) {
                const size_t idx = (tok0 + io) * ldo + h * hd + c;
                const float val = o[ct][r] * inv4[r];
                out[idx] = val;
                if (out16) out16[idx] = m2f_bf16_bits(val);
            }
        }
    }
}

// one-round-trip backward: every problem's slabs fit the register form (NV = 2 * Lp/16 float4) and the fifth (O) slab fits LDS
int attn_bwd_fast(int Lp, int maxW) {
    static const bool oslab = getenv("M2F_ATTN_BWD_OSLAB") && getenv("M2F_ATTN_BWD_OSLAB")[0] == '1';
    return (maxW <= 128 && ((size_t)5 * Lp * (maxW + 2) + Lp) * sizeof(float) <= 160 * 1024) ? (oslab ? 1 : 2) : 0;
}

template <bool BWD>
hipError_t launch(AttnBatch& ab, hipStream_t stream) {
    if (ab.count <= 0 || ab.count > M2F_ATTN_MAX_PROBLEMS || ab.L < 1 || ab.L > 64) return hipErrorInvalidValue;
    const int NT = (ab.L + 15) / 16, Lp = 16 * NT;
    int blocks = 0, maxW = 0;
    for (int i = 0; i < M2F_ATTN_MAX_PROBLEMS; ++i) ab.bb[i] = 0x7fffffff;
    for (int i = 0; i < ab.count; ++i) {
        AttnProblem& p = ab.pr[i];
        if (p.hd < 1 || p.hd > 256 || p.H < 1) return hipErrorInvalidValue;
        if (p.drop_site && !ab.rng) return hipErrorInvalidValue;
        p.block_begin = blocks;
        ab.bb[i] = blocks;
        blocks += ab.B * p.H;
        const int W = (p.hd + 15) & ~15;
        if (W > maxW) maxW = W;
    }
    // one-round-trip backward: every problem's slabs fit the register form (NV = 2 * Lp/16 float4) and the fifth (O) slab fits LDS
    // (2 = the default: O never enters LDS, see the kernel; 1 = with an O slab, the form the parked persistent kernels repeat -
    // M2F_ATTN_BWD_OSLAB=1 selects it so that their bit-for-bit tests still compare like with like)
    ab.bwd_fast = BWD ? attn_bwd_fast(Lp, maxW) : 0;
    const size_t lds = (size_t)(BWD ? (ab.bwd_fast == 1 ? 5 : 4) : 3) * Lp * (maxW + 2) * sizeof(float) +
                       (BWD ? (Lp + (ab.bwd_fast == 2 ? 2 * NT * NTHR : 0)) * sizeof(float) : 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
#define M2F_ATTN_CASE(N)                                                                                   \
    case N: {                                                                                              \
        auto kern = BWD ? m2f_attn_bwd_kernel<N> : m2f_attn_fwd_kernel<N>;                                 \
        if (lds > 64 * 1024) {                                                                             \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
            if (e != hipSuccess) return e;                                                                 \
        }                                                                                                  \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(NTHR), lds, stream, ab);                                 \
        break;                                                                                             \
    }
    switch (NT) {
        M2F_ATTN_CASE(1)
        M2F_ATTN_CASE(2)
        M2F_ATTN_CASE(3)
        M2F_ATTN_CASE(4)
        default: return hipErrorInvalidValue;
    }
#undef M2F_ATTN_CASE
    return hipGetLastError();
}

}  // namespace

// Host-side mirror of the kernels' staging decisions (slab_fast_ok, slab_shadow, attn_bwd_fast): the operands - bits 2 / 4 / 8 =
// Q / K / V, backward also 16 / 32 = dO / O - that problem `pi` of this launch reads ONLY through their bf16 shadows, for every
// (dialogue, head).  plan.hip::mark_unread_fp32 drops the fp32 copy of a buffer nobody reads as fp32 on the strength of this.
int m2f_attn_shadow_only_bits(const AttnBatch& ab, int pi, bool bwd) {
    if (pi < 0 || pi >= ab.count || ab.L < 1 || ab.L > 64) return 0;
    const AttnProblem& P = ab.pr[pi];
    const int NT = (ab.L + 15) / 16, Lp = 16 * NT, NV = 2 * NT, hd = P.hd, W = (hd + 15) & ~15;
    int maxW = 0;
    for (int i = 0; i < ab.count; ++i) maxW = std::max(maxW, (ab.pr[i].hd + 15) & ~15);
    auto fast_ok = [&](const float* src, int ldg) {
        return ((hd & 3) == 0) && ((ldg & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (Lp * (W >> 2) <= NTHR * NV);
    };
    auto shadowed = [&](const float* p, int ldg, int bit) {
        if (!(ab.bf16_math & bit) || !p || !ab.sh.shadow || p < ab.sh.ws_base || p >= ab.sh.ws_base + ab.sh.ws_floats) return false;
        const uint16_t* q = ab.sh.shadow + (p - ab.sh.ws_base);
        return ((reinterpret_cast<uintptr_t>(q) & 7) == 0) && ((ldg & 3) == 0) && ((hd & 3) == 0);
    };
    if (!(fast_ok(P.q, P.ldq) && fast_ok(P.k, P.ldk) && fast_ok(P.v, P.ldv))) return 0;
    if (bwd && !(attn_bwd_fast(Lp, maxW) && fast_ok(P.dout, P.lddo) && fast_ok(P.out, P.ldo))) return 0;
    int bits = (shadowed(P.q, P.ldq, 2) ? 2 : 0) | (shadowed(P.k, P.ldk, 4) ? 4 : 0) | (shadowed(P.v, P.ldv, 8) ? 8 : 0);
    if (bwd) bits |= (shadowed(P.dout, P.lddo, 16) ? 16 : 0) | (shadowed(P.out, P.ldo, 32) ? 32 : 0);
    return bits;
}

size_t m2f_attn_probs_elems(int B, int H, int L) {
    const size_t Lp = 16 * ((L + 15) / 16);
    return (size_t)B * H * Lp * Lp;
}
hipError_t m2f_launch_attn_long_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                    const uint8_t* key_pad, float* out, int ldo, int B, int S, int H, int hd, ShadowMap sh,
                                    hipStream_t stream) {
    if (B < 1 || S < 1 || H < 1 || hd < 1 || hd > 128) return hipErrorInvalidValue;
    const int W = (hd + 15) & ~15;
    const size_t lds = (size_t)3 * 64 * (W + 2) * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(m2f_attn_long_fwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(m2f_attn_long_fwd_kernel, dim3((S + 63) / 64, B * H), dim3(NTHR), lds, stream, q, k, v, ldq, ldk, ldv,
                       key_pad, out, ldo, S, H, hd, sh);
    return hipGetLastError();
}
hipError_t m2f_launch_attn_fwd(AttnBatch& ab, hipStream_t stream) { return launch<false>(ab, stream); }
hipError_t m2f_launch_attn_bwd(AttnBatch& ab, hipStream_t stream) { return launch<true>(ab, stream); }
