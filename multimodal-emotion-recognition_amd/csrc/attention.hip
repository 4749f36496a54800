// Dialogue-level multi-head attention, forward and backward, for gfx950 (wave64).
//
// Replaces the attention inside nn.MultiheadAttention for both uses in the reference:
//   * encoder self-attention  (src/model.py:107,119 -> TransformerEncoderLayer._sa_block)
//   * FusionAttentionModule   (src/model.py:14: query = text, key = audio, value = text)
// with key_padding_mask semantics (-inf on padded keys before the softmax).
//
// One wavefront owns one (dialogue, head): the sequence is the utterances of a dialogue (L <= 64), so
// the whole L x L problem fits one wave.  Q/K/V (and dO in backward) tiles of the head are staged in LDS
// (row stride = 2 mod 4 floats -> conflict-free MFMA fragment reads), QK^T and PV run on the exact-fp32
// MFMA v_mfma_f32_16x16x4_f32, the softmax runs in registers with wavefront shuffles.  S^T = K Q^T is
// computed so the probability tile is already laid out as the A operand of the PV product (accumulator
// as next operand, no LDS round trip).  The backward pass evaluates dS in both orientations (cheap at
// these sizes) so dQ and dK/dV need no transposes and no atomics.
#include "common.h"
#include "ops.h"

namespace {

// [Lp x W] zero-padded LDS copy of src rows [0, L) x cols [0, hd).  Loads are UNCONDITIONAL (clamped address +
// select) and issued in batches of up to 8 per lane before any LDS write, so a slab costs about one memory round
// trip instead of one per 16-byte piece (guarded loads compile to branch + s_waitcnt vmcnt(0) each).
__device__ __forceinline__ void load_slab(float* __restrict__ lds, int ld, int Lp, int W,
                                          const float* __restrict__ src, int ldg, int L, int hd,
                                          int lane) {
    const bool vec = ((hd & 3) == 0) && ((ldg & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    if (vec) {
        const int C4 = W >> 2, total = Lp * C4;
        for (int base = 0; base < total; base += 64 * 8) {
            f32x4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + lane + 64 * u;
                const int r = e / C4, c = (e - r * C4) << 2;
                const bool ok = e < total && r < L && c < hd;
                x[u] = *reinterpret_cast<const f32x4*>(src + (ok ? (size_t)r * ldg + c : (size_t)0));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + lane + 64 * u;
                const int r = e / C4, c = (e - r * C4) << 2;
                if (e < total) {
                    const bool ok = r < L && c < hd;
                    float* d = lds + r * ld + c;
                    d[0] = ok ? x[u][0] : 0.f; d[1] = ok ? x[u][1] : 0.f; d[2] = ok ? x[u][2] : 0.f; d[3] = ok ? x[u][3] : 0.f;
                }
            }
        }
    } else {
        const int total = Lp * W;
        for (int base = 0; base < total; base += 64 * 8) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + lane + 64 * u;
                const int r = e / W, c = e - r * W;
                const bool ok = e < total && r < L && c < hd;
                x[u] = src[ok ? (size_t)r * ldg + c : (size_t)0];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + lane + 64 * u;
                const int r = e / W, c = e - r * W;
                if (e < total) lds[r * ld + c] = (r < L && c < hd) ? x[u] : 0.f;
            }
        }
    }
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int NT>
__global__ __launch_bounds__(64) void m2f_attn_fwd_kernel(const AttnBatch ab) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < M2F_ATTN_MAX_PROBLEMS; ++i)
        if (i < ab.count && (int)blockIdx.x >= ab.pr[i].block_begin) pi = i;
    const AttnProblem& P = ab.pr[pi];
    const int H = P.H, hd = P.hd, L = ab.L;
    const int bh = (int)blockIdx.x - P.block_begin;
    const int b = bh / H, h = bh - b * H;
    constexpr int Lp = 16 * NT;
    const int W = (hd + 15) & ~15, ld = W + 2;
    float* Qs = sm;
    float* Ks = Qs + Lp * ld;
    float* Vs = Ks + Lp * ld;
    const size_t tok0 = (size_t)b * L;
    load_slab(Qs, ld, Lp, W, P.q + tok0 * P.ldq + h * hd, P.ldq, L, hd, lane);
    load_slab(Ks, ld, Lp, W, P.k + tok0 * P.ldk + h * hd, P.ldk, L, hd, lane);
    load_slab(Vs, ld, Lp, W, P.v + tok0 * P.ldv + h * hd, P.ldv, L, hd, lane);
    const unsigned long long kvalid = __ballot(lane < L && ab.key_pad[tok0 + (lane < L ? lane : 0)] == 0);
    __syncthreads();

    const float scale = 1.0f / sqrtf((float)hd);
    const int l15 = lane & 15, lg = lane >> 4;
    const int ksteps = (hd + 3) >> 2;
    const uint32_t site = P.drop_site;
    uint32_t key = 0;
    if (site) key = m2f_site_key(ab.rng, site);
    float* probs = P.probs + (size_t)bh * Lp * Lp;
    uint16_t* out16 = m2f_shadow_of(ab.sh, P.out);

#pragma unroll 1
    for (int it = 0; it < NT; ++it) {
        f32x4 s[NT];
        const int i = 16 * it + l15;                        // this lane's query row
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* kp = Ks + (16 * jt + l15) * ld + lg;
            const float* qp = Qs + i * ld + lg;
            for (int ks = 0; ks < ksteps; ++ks) acc = mfma4(kp[4 * ks], qp[4 * ks], acc);
            s[jt] = acc;                                    // S[i][j = 16jt + 4lg + r]
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
                probs[(size_t)j * Lp + i] = p;              // P^T, pre-dropout (lanes: consecutive i)
                if (site) p = m2f_keep(key, (uint32_t)((bh * L + i) * L + j), ab.drop_thresh) ? p * ab.drop_scale : 0.f;
                s[jt][r] = p;
            }
        // O[i][c] = sum_j P[i][j] V[j][c]
        for (int ct = 0; ct < (W >> 4); ++ct) {
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
                    P.out[idx] = o[r];
                    if (out16) out16[idx] = m2f_bf16_bits(o[r]);
                }
            }
        }
    }
}

template <int NT>
__global__ __launch_bounds__(64) void m2f_attn_bwd_kernel(const AttnBatch ab) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < M2F_ATTN_MAX_PROBLEMS; ++i)
        if (i < ab.count && (int)blockIdx.x >= ab.pr[i].block_begin) pi = i;
    const AttnProblem& P = ab.pr[pi];
    const int H = P.H, hd = P.hd, L = ab.L;
    const int bh = (int)blockIdx.x - P.block_begin;
    const int b = bh / H, h = bh - b * H;
    constexpr int Lp = 16 * NT;
    const int W = (hd + 15) & ~15, ld = W + 2;
    float* Qs = sm;
    float* Ks = Qs + Lp * ld;
    float* Vs = Ks + Lp * ld;
    float* Gs = Vs + Lp * ld;          // dO
    float* delta = Gs + Lp * ld;       // [Lp]
    const size_t tok0 = (size_t)b * L;
    load_slab(Qs, ld, Lp, W, P.q + tok0 * P.ldq + h * hd, P.ldq, L, hd, lane);
    load_slab(Ks, ld, Lp, W, P.k + tok0 * P.ldk + h * hd, P.ldk, L, hd, lane);
    load_slab(Vs, ld, Lp, W, P.v + tok0 * P.ldv + h * hd, P.ldv, L, hd, lane);
    load_slab(Gs, ld, Lp, W, P.dout + tok0 * P.lddo + h * hd, P.lddo, L, hd, lane);
    __syncthreads();
    // delta_i = sum_c dO[i][c] * O[i][c]  (= sum_j P[i][j] dP[i][j], also under dropout).  Four lanes per row, each
    // streams a quarter of the row of O from global memory (unconditional clamped loads), shuffle-reduce over the 4.
    for (int r0 = 0; r0 < Lp; r0 += 16) {
        const int row = r0 + (lane >> 2), part = lane & 3;
        const bool rok = row < L;
        const float* o = P.out + (tok0 + (rok ? row : 0)) * P.ldo + h * hd;
        const float* g = Gs + row * ld;
        float d = 0.f;
        for (int c = part; c < hd; c += 4) d += g[c] * o[c];
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        if (part == 0) delta[row] = rok ? d : 0.f;
    }
    __syncthreads();

    const float scale = 1.0f / sqrtf((float)hd);
    const int l15 = lane & 15, lg = lane >> 4;
    const int ksteps = (hd + 3) >> 2;
    const uint32_t site = P.drop_site;
    uint32_t key = 0;
    if (site) key = m2f_site_key(ab.rng, site);
    const float* probs = P.probs + (size_t)bh * Lp * Lp;
    uint16_t* dq16 = m2f_shadow_of(ab.sh, P.dq);
    uint16_t* dk16 = m2f_shadow_of(ab.sh, P.dk);
    uint16_t* dv16 = m2f_shadow_of(ab.sh, P.dv);

    // ---- orientation X: lane = query row i, registers = keys j  ->  dQ = dS K ----------------------
#pragma unroll 1
    for (int it = 0; it < NT; ++it) {
        f32x4 ds[NT];
        const int i = 16 * it + l15;
        const float dl = delta[i];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* vp = Vs + (16 * jt + l15) * ld + lg;
            const float* gp = Gs + i * ld + lg;
            for (int ks = 0; ks < ksteps; ++ks) acc = mfma4(vp[4 * ks], gp[4 * ks], acc);   // dP[i][j]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * lg + r;
                const float p = probs[(size_t)j * Lp + i];
                float dp = acc[r];
                if (site) dp = m2f_keep(key, (uint32_t)((bh * L + i) * L + j), ab.drop_thresh) ? dp * ab.drop_scale : 0.f;
                ds[jt][r] = p * (dp - dl) * scale;
            }
        }
        for (int ct = 0; ct < (W >> 4); ++ct) {
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
                    P.dq[idx] = o[r];
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
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* gp = Gs + (16 * it + l15) * ld + lg;
            const float* vp = Vs + j * ld + lg;
            for (int ks = 0; ks < ksteps; ++ks) acc = mfma4(gp[4 * ks], vp[4 * ks], acc);   // dP[i][j]
            const f32x4 p4 = *reinterpret_cast<const f32x4*>(probs + (size_t)j * Lp + 16 * it + 4 * lg);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + 4 * lg + r;
                float p = p4[r], dp = acc[r], pdv = p;
                if (site) {
                    const bool kp = m2f_keep(key, (uint32_t)((bh * L + i) * L + j), ab.drop_thresh);
                    dp = kp ? dp * ab.drop_scale : 0.f;
                    pdv = kp ? p * ab.drop_scale : 0.f;
                }
                ds[it][r] = p * (dp - delta[i]) * scale;
                pd[it][r] = pdv;
            }
        }
        for (int ct = 0; ct < (W >> 4); ++ct) {
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
                    P.dk[ik] = dk[r];
                    P.dv[iv] = dv[r];
                    if (dk16) dk16[ik] = m2f_bf16_bits(dk[r]);
                    if (dv16) dv16[iv] = m2f_bf16_bits(dv[r]);
                }
            }
        }
    }
}

template <bool BWD>
hipError_t launch(AttnBatch& ab, hipStream_t stream) {
    if (ab.count <= 0 || ab.count > M2F_ATTN_MAX_PROBLEMS || ab.L < 1 || ab.L > 64) return hipErrorInvalidValue;
    const int NT = (ab.L + 15) / 16, Lp = 16 * NT;
    int blocks = 0, maxW = 0;
    for (int i = 0; i < ab.count; ++i) {
        AttnProblem& p = ab.pr[i];
        if (p.hd < 1 || p.hd > 256 || p.H < 1) return hipErrorInvalidValue;
        if (p.drop_site && !ab.rng) return hipErrorInvalidValue;
        p.block_begin = blocks;
        blocks += ab.B * p.H;
        const int W = (p.hd + 15) & ~15;
        if (W > maxW) maxW = W;
    }
    const size_t lds = (size_t)(BWD ? 4 : 3) * Lp * (maxW + 2) * sizeof(float) + (BWD ? Lp * sizeof(float) : 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
#define M2F_ATTN_CASE(N)                                                                                   \
    case N: {                                                                                              \
        auto kern = BWD ? m2f_attn_bwd_kernel<N> : m2f_attn_fwd_kernel<N>;                                 \
        if (lds > 64 * 1024) {                                                                             \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
            if (e != hipSuccess) return e;                                                                 \
        }                                                                                                  \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, stream, ab);                                 \
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

size_t m2f_attn_probs_elems(int B, int H, int L) {
    const size_t Lp = 16 * ((L + 15) / 16);
    return (size_t)B * H * Lp * Lp;
}
hipError_t m2f_launch_attn_fwd(AttnBatch& ab, hipStream_t stream) { return launch<false>(ab, stream); }
hipError_t m2f_launch_attn_bwd(AttnBatch& ab, hipStream_t stream) { return launch<true>(ab, stream); }
