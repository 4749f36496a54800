// Row-wise (HBM-bound) kernels of the M2FNet step for gfx950: LayerNorm forward/backward with fused
// residual / dropout, the label-smoothed cross-entropy criterion (reference src/train.py:48-50), the
// dropout-mask re-application used in backward, and the fused Adam update (src/train.py:56).
// One wavefront per token row, rows cached in registers, 16-byte coalesced accesses, wave shuffles for
// the row reductions; column reductions (dgamma/dbeta) go through per-block partials (deterministic,
// no atomics, no memset).
#include "common.h"
#include "ops.h"
#include <algorithm>

// No floating-point contraction in this file: several kernels restate the same formulas in different surroundings (ring and
// register-staged GEMM epilogues, fp32 and bf16 attention forms, flat and shadow-writing Adam), and with -ffp-contract=fast (the
// HIP default) the compiler is free to fuse a*b+c in one of them and not in the other - a 1-ulp difference that would break the
// bit-for-bit comparisons the tests make between the forms.  These kernels are bound by memory or by MFMA, not by VALU multiplies.
#pragma clang fp contract(off)

namespace {

constexpr int LN_MAXV_LIMIT = 8;           // float4 chunks per lane  -> d <= 2048 (kernels templated on NV)
constexpr int LN_WAVES = 4;
constexpr int LN_ROWS_PER_WAVE = M2F_LN_ROWS_PER_BLOCK / LN_WAVES;

template <int NV> struct RowRegs { f32x4 v[NV]; };

// lane owns elements c = 4*(lane + 64*j) .. +3 ; scalar tail handling when d % 4 != 0 or unaligned.
template <int NV>
__device__ __forceinline__ void row_load(RowRegs<NV>& r, const float* __restrict__ p, int d, bool vec, int lane) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * (lane + 64 * j);
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (c < d) {
            if (vec) x = *reinterpret_cast<const f32x4*>(p + c);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c + e < d) x[e] = p[c + e];
            }
        }
        r.v[j] = x;
    }
}
template <int NV>
__device__ __forceinline__ void row_store(const RowRegs<NV>& r, float* __restrict__ p, int d, bool vec, int lane) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * (lane + 64 * j);
        if (c < d) {
            if (vec) *reinterpret_cast<f32x4*>(p + c) = r.v[j];
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c + e < d) p[c + e] = r.v[j][e];
            }
        }
    }
}
// bf16 shadow of a row (rows of shadowed buffers are 16-byte aligned in fp32, hence 8-byte aligned in bf16)
template <int NV>
__device__ __forceinline__ void row_store_bf16(const RowRegs<NV>& r, uint16_t* __restrict__ q, int d, int lane) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * (lane + 64 * j);
        if (c + 3 < d) {
            uint2 w;
            w.x = (uint32_t)m2f_bf16_bits(r.v[j][0]) | ((uint32_t)m2f_bf16_bits(r.v[j][1]) << 16);
            w.y = (uint32_t)m2f_bf16_bits(r.v[j][2]) | ((uint32_t)m2f_bf16_bits(r.v[j][3]) << 16);
            if ((reinterpret_cast<uintptr_t>(q + c) & 7) == 0) *reinterpret_cast<uint2*>(q + c) = w;
            else { q[c] = (uint16_t)w.x; q[c + 1] = (uint16_t)(w.x >> 16); q[c + 2] = (uint16_t)w.y; q[c + 3] = (uint16_t)(w.y >> 16); }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c + e < d) q[c + e] = m2f_bf16_bits(r.v[j][e]);
        }
    }
}
__device__ __forceinline__ bool is_vec(const void* p, int d) {
    return ((d & 3) == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
}

template <int NV>
__global__ __launch_bounds__(256) void m2f_ln_fwd_kernel(const LnBatch lb) {
    static_assert(sizeof(LnBatch) + 64 <= 136 + 512, "m2f_kernarg_warm ranges no longer cover LnBatch + the hidden arguments");
    m2f_kernarg_warm<0, 8, 136>();                  // the descriptor block (552 B + hidden arguments) in one miss
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < M2F_LN_MAX_PROBLEMS; ++i)
        if ((int)blockIdx.x >= lb.bb[i]) pi = i;
    const LnProblem& P = lb.pr[pi];
    const int d = P.d;
    const int ld = P.ld ? P.ld : d;
    const int blk = (int)blockIdx.x - P.block_begin;
    const bool vec = is_vec(P.x, d) && is_vec(P.out, d) && is_vec(P.gamma, d) && is_vec(P.beta, d) &&
                     (!P.res || is_vec(P.res, d)) && ((ld & 3) == 0);
    uint16_t* out16 = m2f_shadow_of(lb.sh, P.out);
    RowRegs<NV> g, be;
    row_load(g, P.gamma, d, vec, lane);
    row_load(be, P.beta, d, vec, lane);
    uint32_t key = 0;
    if (P.drop_site) key = m2f_site_key(lb.rng, P.drop_site);
    const float invd = 1.0f / (float)d;
    for (int rr = 0; rr < LN_ROWS_PER_WAVE; ++rr) {
        const int row = blk * M2F_LN_ROWS_PER_BLOCK + wave * LN_ROWS_PER_WAVE + rr;
        if (row >= lb.T) break;                             // wave-uniform
        RowRegs<NV> x;
        row_load(x, P.x + (size_t)row * ld, d, vec, lane);
        float mean, rstd;
        if (lb.pre_stats) {                                 // (diagnostic: statistics from memory, no reductions - block-uniform branch)
            mean = P.stats[2 * row]; rstd = P.stats[2 * row + 1];
        } else {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) s += (x.v[j][0] + x.v[j][1]) + (x.v[j][2] + x.v[j][3]);
            mean = m2f_wave_sum(s) * invd;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * (lane + 64 * j) + e;
                    const float t = (c < d) ? x.v[j][e] - mean : 0.f;
                    q += t * t;
                }
            rstd = 1.0f / sqrtf(m2f_wave_sum(q) * invd + lb.eps);
        }
        RowRegs<NV> res;
        if (P.res) row_load(res, P.res + (size_t)row * ld, d, vec, lane);
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float y = (x.v[j][e] - mean) * rstd * g.v[j][e] + be.v[j][e];
                if (P.res) y += res.v[j][e];
                if (P.drop_site) {
                    const int c = 4 * (lane + 64 * j) + e;
                    y = m2f_keep(key, (uint32_t)row * (uint32_t)d + (uint32_t)c, lb.drop_thresh) ? y * lb.drop_scale : 0.f;
                }
                x.v[j][e] = y;
            }
        row_store(x, P.out + (size_t)row * ld, d, vec, lane);
        if (out16) row_store_bf16(x, out16 + (size_t)row * ld, d, lane);
        if (lb.out8 && pi == 0) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int c = 4 * (lane + 64 * j);
                if (c + 3 < d)
                    *reinterpret_cast<uint32_t*>(lb.out8 + (size_t)row * d + c) =
                        m2f_fp8x4_bits(x.v[j][0] * lb.out8_scale, x.v[j][1] * lb.out8_scale, x.v[j][2] * lb.out8_scale, x.v[j][3] * lb.out8_scale);
            }
        }
        if (lane == 0) { P.stats[2 * row] = mean; P.stats[2 * row + 1] = rstd; }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void m2f_ln_bwd_kernel(const LnBatch lb) {
    static_assert(sizeof(LnBatch) + 64 <= 136 + 512, "m2f_kernarg_warm ranges no longer cover LnBatch + the hidden arguments");
    m2f_kernarg_warm<0, 8, 136>();                  // the descriptor block (552 B + hidden arguments) in one miss
    extern __shared__ __attribute__((aligned(16))) float red[];      // [LN_WAVES][2][dpad]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < M2F_LN_MAX_PROBLEMS; ++i)
        if ((int)blockIdx.x >= lb.bb[i]) pi = i;
    const LnProblem& P = lb.pr[pi];
    const int d = P.d;
    const int ld = P.ld ? P.ld : d;
    const int dpad = (d + 3) & ~3;
    const int blk = (int)blockIdx.x - P.block_begin;
    const bool vec = is_vec(P.x, d) && is_vec(P.dy, d) && is_vec(P.dx, d) && is_vec(P.gamma, d) &&
                     (!P.extra || is_vec(P.extra, d)) && (!P.dx_masked || is_vec(P.dx_masked, d)) && ((ld & 3) == 0);
    uint16_t* dx16 = (P.skip & 2) ? nullptr : m2f_shadow_of(lb.sh, P.dx);
    uint16_t* dxm16 = P.dx_masked ? m2f_shadow_of(lb.sh, P.dx_masked) : nullptr;
    const bool dxm32 = P.dx_masked && !((P.skip & 1) && dxm16);
    RowRegs<NV> g, dg, db;
    row_load(g, P.gamma, d, vec, lane);
#pragma unroll
    for (int j = 0; j < NV; ++j) { dg.v[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; db.v[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    uint32_t key = 0;
    if (P.drop_site2) key = m2f_site_key(lb.rng, P.drop_site2);
    const float invd = 1.0f / (float)d;
    for (int rr = 0; rr < LN_ROWS_PER_WAVE; ++rr) {
        const int row = blk * M2F_LN_ROWS_PER_BLOCK + wave * LN_ROWS_PER_WAVE + rr;
        if (row >= lb.T) break;
        RowRegs<NV> x, dy;
        row_load(x, P.x + (size_t)row * ld, d, vec, lane);
        row_load(dy, P.dy + (size_t)row * ld, d, vec, lane);
        RowRegs<NV> ex;                                         // fetched with x / dy: one memory round trip, not two
        if (P.extra) row_load(ex, P.extra + (size_t)row * ld, d, vec, lane);
        const float mean = P.stats[2 * row], rstd = P.stats[2 * row + 1];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * (lane + 64 * j) + e;
                const float xh = (c < d) ? (x.v[j][e] - mean) * rstd : 0.f;
                const float gy = dy.v[j][e] * g.v[j][e];
                x.v[j][e] = xh;
                s1 += gy;
                s2 += gy * xh;
                dg.v[j][e] += dy.v[j][e] * xh;
                db.v[j][e] += dy.v[j][e];
            }
        const float c1 = m2f_wave_sum(s1) * invd, c2 = m2f_wave_sum(s2) * invd;
        RowRegs<NV> msk;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dx = rstd * (dy.v[j][e] * g.v[j][e] - c1 - x.v[j][e] * c2);
                float dm = dx;
                if (P.drop_site2) {
                    const int c = 4 * (lane + 64 * j) + e;
                    dm = m2f_keep(key, (uint32_t)row * (uint32_t)d + (uint32_t)c, lb.drop_thresh) ? dx * lb.drop_scale : 0.f;
                }
                msk.v[j][e] = dm;
                dy.v[j][e] = P.extra ? dx + ex.v[j][e] : dx;
            }
        row_store(dy, P.dx + (size_t)row * ld, d, vec, lane);
        if (dx16) row_store_bf16(dy, dx16 + (size_t)row * ld, d, lane);
        if (dxm32) row_store(msk, P.dx_masked + (size_t)row * ld, d, vec, lane);
        if (dxm16) row_store_bf16(msk, dxm16 + (size_t)row * ld, d, lane);
    }
    // per-block partial dgamma / dbeta: waves -> LDS -> fixed-order sum (deterministic)
    float* mine = red + (size_t)wave * 2 * dpad;
    row_store(dg, mine, dpad, true, lane);
    row_store(db, mine + dpad, dpad, true, lane);
    __syncthreads();
    float* out = P.partial + (size_t)blk * 2 * d;
    for (int c = threadIdx.x; c < 2 * d; c += 256) {
        const int which = c >= d, cc = which ? c - d : c;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < LN_WAVES; ++w) s += red[(size_t)w * 2 * dpad + which * dpad + cc];
        out[c] = s;
    }
}

// one block = 64 columns of one LayerNorm; its 4 wavefronts each sum a quarter of the row-block partials, then a
// fixed-order LDS combine (deterministic).  Grid (ceil(max_d / 64), items).
__global__ __launch_bounds__(256) void m2f_ln_param_reduce_kernel(const LnReduceBatch rb) {
    __shared__ float part[4][2][64];
    const LnReduceItem& it = rb.it[blockIdx.y];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float sg = 0.f, sb = 0.f;
    if (c < it.d) {
        const int per = (it.nblk + 3) / 4;
        const int b0 = w * per, b1 = b0 + per < it.nblk ? b0 + per : it.nblk;
        // batches of 8 row blocks: 16 independent loads in flight per lane, summed in the fixed order b0, b0+1, ...
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float g[8], be[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                g[u] = it.partial[(size_t)(b + u) * 2 * it.d + c];
                be[u] = it.partial[(size_t)(b + u) * 2 * it.d + it.d + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { sg += g[u]; sb += be[u]; }
        }
        for (; b < b1; ++b) {
            sg += it.partial[(size_t)b * 2 * it.d + c];
            sb += it.partial[(size_t)b * 2 * it.d + it.d + c];
        }
    }
    part[w][0][lane] = sg;
    part[w][1][lane] = sb;
    __syncthreads();
    if (w == 0 && c < it.d) {
        it.dgamma[c] = (part[0][0][lane] + part[1][0][lane]) + (part[2][0][lane] + part[3][0][lane]);
        it.dbeta[c] = (part[0][1][lane] + part[1][1][lane]) + (part[2][1][lane] + part[3][1][lane]);
    }
}

// ---- criterion ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void m2f_ce_kernel(const CeArgs a) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= a.T) return;
    const int C = a.C;
    float z[16], w[16];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        z[c] = (c < C) ? a.logits[(size_t)t * C + c] : -INFINITY;
        w[c] = (c < C) ? (a.class_w ? a.class_w[c] : 1.f) : 0.f;
        m = fmaxf(m, z[c]);
    }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) se += (c < C) ? expf(z[c] - m) : 0.f;
    const float lse = m + logf(se);
    const int64_t y = a.labels[t];
    const bool valid = (y >= 0) && (y < C);
    float wy = 0.f, logpy = 0.f, W = 0.f, sm = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (c < C) {
            const float lp = z[c] - lse;
            W += w[c];
            sm -= w[c] * lp;
            if (valid && c == (int)y) { wy = w[c]; logpy = lp; }
        }
    }
    const float eps = a.label_smoothing;
    const float num = valid ? ((1.f - eps) * (-logpy) * wy + eps * sm / (float)C) : 0.f;
    a.loss_terms[2 * t] = num;
    a.loss_terms[2 * t + 1] = valid ? wy : 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (c < C) {
            const float p = expf(z[c] - lse);
            float g = 0.f;
            if (valid) g = (1.f - eps) * wy * (p - ((c == (int)y) ? 1.f : 0.f)) + (eps / (float)C) * (W * p - w[c]);
            a.dlogits[(size_t)t * C + c] = g;
        }
    }
}

__global__ __launch_bounds__(256) void m2f_loss_finalize_kernel(const float* __restrict__ terms, int T, int C,
                                                                float* __restrict__ dlogits, float* __restrict__ loss_out,
                                                                int normalise) {
    __shared__ float sn[4], sd[4];
    float n = 0.f, dd = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) { n += terms[2 * t]; dd += terms[2 * t + 1]; }
    n = m2f_wave_sum(n);
    dd = m2f_wave_sum(dd);
    if ((threadIdx.x & 63) == 0) { sn[threadIdx.x >> 6] = n; sd[threadIdx.x >> 6] = dd; }
    __syncthreads();
    const float num = (sn[0] + sn[1]) + (sn[2] + sn[3]);
    const float den = (sd[0] + sd[1]) + (sd[2] + sd[3]);
    if (threadIdx.x == 0) { loss_out[0] = num / den; loss_out[1] = den; loss_out[2] = num; }
    if (normalise) {
        const float inv = 1.0f / den;
        for (int e = threadIdx.x; e < T * C; e += 256) dlogits[e] *= inv;
    }
}

// (blockIdx.y = 1: the second buffer of the launch - the two modalities' post-projection gradients share one launch)
__global__ __launch_bounds__(256) void m2f_dropout_inplace_kernel(float* __restrict__ x0, float* __restrict__ x1, int T, int d, int ld,
                                                                  uint32_t site0, uint32_t site1, const uint32_t* __restrict__ rng,
                                                                  uint32_t thresh, float scale, ShadowMap sh) {
    float* __restrict__ x = blockIdx.y ? x1 : x0;
    const uint32_t key = m2f_site_key(rng, blockIdx.y ? site1 : site0);
    const size_t n = (size_t)T * d;
    uint16_t* x16 = m2f_shadow_of(sh, x);
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int r = (int)(e / d), c = (int)(e - (size_t)r * d);
        float* p = x + (size_t)r * ld + c;
        const float v = m2f_keep(key, (uint32_t)e, thresh) ? *p * scale : 0.f;
        *p = v;
        if (x16) x16[(size_t)r * ld + c] = m2f_bf16_bits(v);
    }
}

// fp32 -> bf16 copies of 2-D blocks (parameter matrices into their padded shadows, input staging buffers)
__global__ __launch_bounds__(256) void m2f_cast_kernel(const CastBatch cb) {
    const CastItem& it = cb.it[blockIdx.y];
    const size_t n4 = (size_t)it.rows * ((it.cols + 3) >> 2);
    const int c4n = (it.cols + 3) >> 2;
    const bool vec = ((it.cols & 3) == 0) && ((it.lds & 3) == 0) && ((it.ldd & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(it.src) & 15) == 0) && ((reinterpret_cast<uintptr_t>(it.dst) & 7) == 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / c4n), c = 4 * (int)(i - (size_t)r * c4n);
        const float* s = it.src + (size_t)r * it.lds + c;
        uint16_t* q = it.dst + (size_t)r * it.ldd + c;
        if (vec) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(s);
            uint2 w;
            w.x = (uint32_t)m2f_bf16_bits(x[0]) | ((uint32_t)m2f_bf16_bits(x[1]) << 16);
            w.y = (uint32_t)m2f_bf16_bits(x[2]) | ((uint32_t)m2f_bf16_bits(x[3]) << 16);
            *reinterpret_cast<uint2*>(q) = w;
        } else {
            for (int e = 0; e < 4; ++e) if (c + e < it.cols) q[e] = m2f_bf16_bits(s[e]);
        }
    }
}

// one wavefront per token slot: two row copies (16-byte pieces when aligned) + label / mask
__global__ __launch_bounds__(256) void m2f_gather_kernel(const GatherArgs a) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= a.T) return;
    const int row = a.rows[t];
    const bool pad = row < 0;
    for (int which = 0; which < 2; ++which) {
        const float* tab = which ? a.audio_table : a.text_table;
        float* out = which ? a.audio_out : a.text_out;
        const int d = which ? a.d_audio : a.d_text, ld = which ? a.ld_audio : a.ld_text;
        if (!tab || !out) continue;
        const float* src = tab + (size_t)(pad ? 0 : row) * d;
        float* dst = out + (size_t)t * ld;
        const bool vec = ((d & 3) == 0) && ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(tab) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
        if (vec) {
            for (int c = 4 * lane; c < d; c += 256) {
                f32x4 x = *reinterpret_cast<const f32x4*>(src + c);
                if (pad) x = (f32x4){0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(dst + c) = x;
            }
        } else {
            for (int c = lane; c < d; c += 64) dst[c] = pad ? 0.f : src[c];
        }
    }
    if (lane == 0) {
        if (a.key_pad) a.key_pad[t] = pad ? 1 : 0;
        if (a.labels) a.labels[t] = (pad || !a.label_table) ? -1 : a.label_table[row];
    }
}

__global__ void m2f_rng_advance_kernel(uint32_t* rng) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const uint32_t lo = rng[2] + 1u;
        rng[2] = lo;
        if (lo == 0u) rng[3] += 1u;
    }
}

// Adam update of four consecutive elements (registers in, registers out).  Every optimizer kernel of this file goes through this
// one function and its contractions are spelled out (the file is compiled with fp contraction off): left to the compiler
// (contract(fast)) the flat kernel and the shadow-writing kernel fused `g * gs + wd * p` differently once one of them became a
// template, and a data-parallel run (bucket-wise shadow-writing steps) drifted from the single-process run by an ulp per step.
__device__ __forceinline__ void adam4(f32x4& pp, const f32x4& gg, f32x4& mm, f32x4& vv, float gs, float lr_bc1, float beta1,
                                      float beta2, float eps, float wd, float inv_sqrt_bc2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gr = __builtin_fmaf(gg[e], gs, wd * pp[e]);     // coupled L2 (Adam, not AdamW)
        mm[e] = __builtin_fmaf(beta1, mm[e], (1.f - beta1) * gr);
        vv[e] = __builtin_fmaf(beta2, vv[e], ((1.f - beta2) * gr) * gr);
        const float denom = __builtin_fmaf(sqrtf(vv[e]), inv_sqrt_bc2, eps);
        pp[e] = __builtin_fmaf(-lr_bc1, mm[e] / denom, pp[e]);
    }
}

// G16: the gradient buffer holds bf16 (the data-parallel bf16 exchange), everything else stays fp32
template <bool G16>
__global__ __launch_bounds__(256) void m2f_adam_kernel(float* __restrict__ p, const void* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, int64_t n4, float lr_bc1, float beta1, float beta2,
                                                       float eps, float wd, float inv_sqrt_bc2, const float* __restrict__ gs_ptr) {
    // g, m, v are streamed once per step: nontemporal accesses keep them from displacing p (re-read by the bf16 cast that
    // opens the next forward) in the L2 / Infinity Cache; measured 0.554 -> 0.524 ms per C3 step for the optimizer part
    const float gs = gs_ptr ? 1.0f / *gs_ptr : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 pp = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p) + i);
        f32x4 gg;
        if constexpr (G16) {
            const uint2 raw = reinterpret_cast<const uint2*>(g)[i];
            gg[0] = __builtin_bit_cast(float, raw.x << 16); gg[1] = __builtin_bit_cast(float, raw.x & 0xFFFF0000u);
            gg[2] = __builtin_bit_cast(float, raw.y << 16); gg[3] = __builtin_bit_cast(float, raw.y & 0xFFFF0000u);
        } else {
            gg = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
        }
        f32x4 mm = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + i);
        f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + i);
        adam4(pp, gg, mm, vv, gs, lr_bc1, beta1, beta2, eps, wd, inv_sqrt_bc2);
        reinterpret_cast<f32x4*>(p)[i] = pp;
        __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(m) + i);
        __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + i);
    }
}

// four consecutive gradient elements from the fp32 buffer or (G16: the data-parallel bf16 exchange) from the reduced bf16 buffer
template <bool G16>
__device__ __forceinline__ f32x4 adam_grad4(const void* g, long long o) {
    if constexpr (G16) {
        const uint2 raw = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(g) + o);
        return (f32x4){__builtin_bit_cast(float, raw.x << 16), __builtin_bit_cast(float, raw.x & 0xFFFF0000u),
                       __builtin_bit_cast(float, raw.y << 16), __builtin_bit_cast(float, raw.y & 0xFFFF0000u)};
    } else {
        return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(static_cast<const float*>(g) + o));
    }
}
template <bool G16>
__device__ __forceinline__ float adam_grad1(const void* g, long long o) {
    if constexpr (G16) return __builtin_bit_cast(float, (uint32_t)static_cast<const uint16_t*>(g)[o] << 16);
    else return static_cast<const float*>(g)[o];
}

// see ops.h (AdamItem).  Persistent 1-D grid over tiles [tile_first, total_tiles) of items[0, n_items) (tile_begin[] holds absolute
// tile numbers); tile -> item by bisection of the prefix array (kept in LDS).
template <bool G16>
__device__ __forceinline__ void adam_shadow_body(float* __restrict__ p, const void* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, uint16_t* __restrict__ sh,
                                                 const AdamItem* __restrict__ items, const int* __restrict__ tile_begin,
                                                 int n_items, int tile_first, int total_tiles, float lr_bc1, float beta1,
                                                 float beta2, float eps, float wd, float inv_sqrt_bc2,
                                                 const float* __restrict__ gs_ptr) {
    __shared__ float tile[64][65];
    __shared__ int tb[M2F_ADAM_MAX_ITEMS + 1];
    const int tid = threadIdx.x;
    for (int i = tid; i <= n_items; i += 256) tb[i] = tile_begin[i];
    __syncthreads();
    const float gs = gs_ptr ? 1.0f / *gs_ptr : 1.0f;
    for (int t = tile_first + (int)blockIdx.x; t < total_tiles; t += gridDim.x) {
        int lo = 0, hi = n_items - 1;                               // last item whose first tile is <= t (block-uniform)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tb[mid] <= t) lo = mid; else hi = mid - 1;
        }
        const AdamItem it = items[lo];
        const int tl = t - tb[lo];
        if (it.rows == 0) {                                         // 1-D parameter: 4096 consecutive elements per tile
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long idx = (long long)tl * 4096 + q * 1024 + tid * 4;
                if (idx < it.cols) {
                    const long long o = it.off + idx;
                    f32x4 pp = *reinterpret_cast<const f32x4*>(p + o);
                    const f32x4 gg = adam_grad4<G16>(g, o);
                    f32x4 mm = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + o));
                    f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + o));
                    adam4(pp, gg, mm, vv, gs, lr_bc1, beta1, beta2, eps, wd, inv_sqrt_bc2);
                    *reinterpret_cast<f32x4*>(p + o) = pp;
                    __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(m + o));
                    __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v + o));
                }
            }
            continue;
        }
        const int rows = it.rows, cols = it.cols, ldd = (cols + 7) & ~7, ldt = (rows + 7) & ~7;
        const int r0 = (tl / it.tiles_c) << 6, c0 = (tl % it.tiles_c) << 6;
        const int lr = tid >> 4, c = 4 * (tid & 15), gc = c0 + c;
        const bool vec = (cols & 3) == 0;                           // then every 4-column group is whole and 16-byte aligned
        f32x4 pp[4], gg[4], mm[4], vv[4];
        bool in[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                               // all loads of the tile first, then the arithmetic
            const int gr = r0 + lr + 16 * i;
            in[i] = gr < rows && gc < cols;
            const long long o = it.off + (long long)(in[i] ? gr : 0) * cols + (in[i] ? gc : 0);
            if (vec) {
                pp[i] = *reinterpret_cast<const f32x4*>(p + o);
                gg[i] = adam_grad4<G16>(g, o);
                mm[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + o));
                vv[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + o));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long long oe = o + ((in[i] && gc + e < cols) ? e : 0);
                    pp[i][e] = p[oe]; gg[i][e] = adam_grad1<G16>(g, oe); mm[i][e] = m[oe]; vv[i][e] = v[oe];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lr + 16 * i, gr = r0 + r;
            adam4(pp[i], gg[i], mm[i], vv[i], gs, lr_bc1, beta1, beta2, eps, wd, inv_sqrt_bc2);
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[r][c + e] = (in[i] && gc + e < cols) ? pp[i][e] : 0.f;
            if (in[i]) {
                const long long o = it.off + (long long)gr * cols + gc;
                uint16_t* q = sh + it.soff + (long long)gr * ldd + gc;
                if (vec) {
                    *reinterpret_cast<f32x4*>(p + o) = pp[i];
                    __builtin_nontemporal_store(mm[i], reinterpret_cast<f32x4*>(m + o));
                    __builtin_nontemporal_store(vv[i], reinterpret_cast<f32x4*>(v + o));
                    uint2 w;
                    w.x = (uint32_t)m2f_bf16_bits(pp[i][0]) | ((uint32_t)m2f_bf16_bits(pp[i][1]) << 16);
                    w.y = (uint32_t)m2f_bf16_bits(pp[i][2]) | ((uint32_t)m2f_bf16_bits(pp[i][3]) << 16);
                    *reinterpret_cast<uint2*>(q) = w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (gc + e < cols) { p[o + e] = pp[i][e]; m[o + e] = mm[i][e]; v[o + e] = vv[i][e]; q[e] = m2f_bf16_bits(pp[i][e]); }
                }
            }
        }
        __syncthreads();
        {   // W^T shadow: 8 consecutive source rows of one column per lane = 16 bytes of a dst_t row (m2f_tile64)
            const int r8 = tid & 7;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cc = (tid >> 3) + 32 * j;
                const int gcol = c0 + cc, gr0 = r0 + 8 * r8;
                if (gcol < cols && gr0 < ldt) {
                    uint16_t h[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) h[k] = m2f_bf16_bits(tile[8 * r8 + k][cc]);
                    uint4 w;
                    w.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16); w.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
                    w.z = (uint32_t)h[4] | ((uint32_t)h[5] << 16); w.w = (uint32_t)h[6] | ((uint32_t)h[7] << 16);
                    *reinterpret_cast<uint4*>(sh + it.soff_t + (long long)gcol * ldt + gr0) = w;      // soff_t, ldt: multiples of 8
                }
            }
        }
        __syncthreads();
    }
}

template <bool G16>
__global__ __launch_bounds__(256) void m2f_adam_shadow_kernel(float* __restrict__ p, const void* __restrict__ g, float* __restrict__ m,
                                                              float* __restrict__ v, uint16_t* __restrict__ sh,
                                                              const AdamItem* __restrict__ items, const int* __restrict__ tile_begin,
                                                              int n_items, int tile_first, int total_tiles, float lr_bc1, float beta1,
                                                              float beta2, float eps, float wd, float inv_sqrt_bc2,
                                                              const float* __restrict__ gs_ptr) {
    adam_shadow_body<G16>(p, g, m, v, sh, items, tile_begin, n_items, tile_first, total_tiles, lr_bc1, beta1, beta2, eps, wd, inv_sqrt_bc2, gs_ptr);
}
// the same update with the step-dependent factors read from device memory (m2f_launch_adam_hyper): a node of the captured step graph
__global__ __launch_bounds__(256) void m2f_adam_shadow_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                                  float* __restrict__ v, uint16_t* __restrict__ sh,
                                                                  const AdamItem* __restrict__ items, const int* __restrict__ tile_begin,
                                                                  int n_items, int total_tiles, const float* __restrict__ hy,
                                                                  const float* __restrict__ gs_ptr) {
    adam_shadow_body<false>(p, g, m, v, sh, items, tile_begin, n_items, 0, total_tiles, hy[0], hy[1], hy[2], hy[3], hy[4], hy[5], gs_ptr);
}

template <bool BWD>
hipError_t ln_launch(LnBatch& lb, hipStream_t stream) {
    if (lb.count <= 0 || lb.count > M2F_LN_MAX_PROBLEMS || lb.T <= 0) return hipErrorInvalidValue;
    int blocks = 0, maxd = 0;
    for (int i = 0; i < M2F_LN_MAX_PROBLEMS; ++i) lb.bb[i] = 0x7fffffff;
    for (int i = 0; i < lb.count; ++i) {
        LnProblem& p = lb.pr[i];
        if (p.d < 1 || p.d > 256 * LN_MAXV_LIMIT) return hipErrorInvalidValue;
        if ((p.drop_site || p.drop_site2) && !lb.rng) return hipErrorInvalidValue;
        p.block_begin = blocks;
        lb.bb[i] = blocks;
        blocks += m2f_ln_row_blocks(lb.T);
        if (p.d > maxd) maxd = p.d;
    }
    const size_t lds = BWD ? (size_t)LN_WAVES * 2 * ((maxd + 3) & ~3) * sizeof(float) : 0;
    const int nv = m2f_cdiv(maxd, 256);
#define M2F_LN_CASE(N)                                                                                      \
    if (nv <= N) {                                                                                          \
        if (BWD) hipLaunchKernelGGL(m2f_ln_bwd_kernel<N>, dim3(blocks), dim3(256), lds, stream, lb);        \
        else hipLaunchKernelGGL(m2f_ln_fwd_kernel<N>, dim3(blocks), dim3(256), 0, stream, lb);              \
        return hipGetLastError();                                                                           \
    }
    M2F_LN_CASE(1)
    M2F_LN_CASE(2)
    M2F_LN_CASE(3)
    M2F_LN_CASE(4)
    M2F_LN_CASE(8)
#undef M2F_LN_CASE
    return hipGetLastError();
}

}  // namespace

hipError_t m2f_launch_ln_fwd(LnBatch& lb, hipStream_t stream) { return ln_launch<false>(lb, stream); }
hipError_t m2f_launch_ln_bwd(LnBatch& lb, hipStream_t stream) { return ln_launch<true>(lb, stream); }

hipError_t m2f_launch_ln_param_reduce(const LnReduceBatch& rb, hipStream_t stream) {
    if (rb.count <= 0) return hipSuccess;
    if (rb.count > M2F_LNRED_MAX_ITEMS) return hipErrorInvalidValue;
    int maxd = 0;
    for (int i = 0; i < rb.count; ++i) if (rb.it[i].d > maxd) maxd = rb.it[i].d;
    hipLaunchKernelGGL(m2f_ln_param_reduce_kernel, dim3(m2f_cdiv(maxd, 64), rb.count), dim3(256), 0, stream, rb);
    return hipGetLastError();
}

hipError_t m2f_launch_ce(const CeArgs& a, hipStream_t stream) {
    if (a.C < 1 || a.C > 16 || a.T < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(m2f_ce_kernel, dim3(m2f_cdiv(a.T, 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t m2f_launch_loss_finalize(const float* loss_terms, int T, int C, float* dlogits, float* loss_out,
                                    int normalise, hipStream_t stream) {
    hipLaunchKernelGGL(m2f_loss_finalize_kernel, dim3(1), dim3(256), 0, stream, loss_terms, T, C, dlogits, loss_out, normalise);
    return hipGetLastError();
}

hipError_t m2f_launch_dropout_inplace2(float* x, float* x2, int T, int d, int ld, uint32_t site, uint32_t site2, const uint32_t* rng,
                                       uint32_t thresh, float scale, ShadowMap sh, hipStream_t stream) {
    const size_t n = (size_t)T * d;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(m2f_dropout_inplace_kernel, dim3(blocks, x2 ? 2 : 1), dim3(256), 0, stream, x, x2, T, d, ld, site, site2, rng, thresh,
                       scale, sh);
    return hipGetLastError();
}
hipError_t m2f_launch_dropout_inplace(float* x, int T, int d, int ld, uint32_t site, const uint32_t* rng,
                                      uint32_t thresh, float scale, ShadowMap sh, hipStream_t stream) {
    return m2f_launch_dropout_inplace2(x, nullptr, T, d, ld, site, 0u, rng, thresh, scale, sh, stream);
}

// ---- 64x64 transposing tile: src fp32 [rows][ld] -> bf16 dst [rows][ldd] (optional) and bf16 dst_t [cols][ldt] ------------
// 16-byte loads (4 floats of a row per lane), 8-byte stores of the plain copy, and - through an LDS tile - 16-byte stores
// of the transposed copy (8 consecutive source rows of one column per lane), so all three streams move full 128-byte
// lines.  Source elements outside [rows) x [cols) read as zero, which also writes the zero pads of dst_t up to ldt.
// cs[4] accumulates this thread's four column sums of the (pre-relu) source.
template <bool PLAIN>
__device__ __forceinline__ void m2f_tile64(const float* __restrict__ src, int ld, int rows, int cols, int r0, int c0,
                                           uint16_t* __restrict__ dst, int ldd, uint16_t* __restrict__ dst_t, int ldt,
                                           bool relu, bool vec, float (&cs)[4], float (*tile)[65]) {
    const int tid = threadIdx.x;
    const int lr = tid >> 4, c = 4 * (tid & 15);
    const int gc = c0 + c;
    f32x4 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                       // unconditional clamped loads, select afterwards
        const int gr = r0 + lr + 16 * i;
        const int grc = gr < rows ? gr : rows - 1;
        if (vec && gc + 3 < cols) {
            x[i] = *reinterpret_cast<const f32x4*>(src + (size_t)grc * ld + gc);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[i][e] = src[(size_t)grc * ld + (gc + e < cols ? gc + e : cols - 1)];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = lr + 16 * i, gr = r0 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = (gr < rows && gc + e < cols) ? x[i][e] : 0.f;
            cs[e] += v;
            if (relu) v = fmaxf(v, 0.f);
            x[i][e] = v;
            tile[r][c + e] = v;
        }
        if (PLAIN && gr < rows) {
            uint16_t* q = dst + (size_t)gr * ldd + gc;
            if (vec && gc + 3 < cols && (ldd & 3) == 0) {
                uint2 w;
                w.x = (uint32_t)m2f_bf16_bits(x[i][0]) | ((uint32_t)m2f_bf16_bits(x[i][1]) << 16);
                w.y = (uint32_t)m2f_bf16_bits(x[i][2]) | ((uint32_t)m2f_bf16_bits(x[i][3]) << 16);
                *reinterpret_cast<uint2*>(q) = w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (gc + e < cols) q[e] = m2f_bf16_bits(x[i][e]);
            }
        }
    }
    __syncthreads();
    if (dst_t) {                                                        // block-uniform
        const int r8 = tid & 7;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cc = (tid >> 3) + 32 * j;                         // source column = row of dst_t
            const int gcol = c0 + cc, gr0 = r0 + 8 * r8;                // 8 consecutive source rows = 8 consecutive dst_t columns
            if (gcol < cols && gr0 < ldt) {
                uint16_t h[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) h[k] = m2f_bf16_bits(tile[8 * r8 + k][cc]);
                uint16_t* q = dst_t + (size_t)gcol * ldt + gr0;
                if ((ldt & 7) == 0 && ((reinterpret_cast<uintptr_t>(dst_t) & 15) == 0)) {
                    uint4 w;
                    w.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16); w.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
                    w.z = (uint32_t)h[4] | ((uint32_t)h[5] << 16); w.w = (uint32_t)h[6] | ((uint32_t)h[7] << 16);
                    *reinterpret_cast<uint4*>(q) = w;
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (gr0 + k < ldt) q[k] = h[k];
                }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ bool m2f_tile_vec_ok(const float* src, int ld) {
    return ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
}

// parameter matrix -> padded bf16 shadow + transposed padded bf16 shadow
__global__ __launch_bounds__(256) void m2f_cast_t_kernel(const CastBatch cb) {
    __shared__ float tile[64][65];
    const CastItem& it = cb.it[blockIdx.y];
    const int tiles_c = (it.cols + 63) >> 6, tiles_r = (it.rows + 63) >> 6;
    const bool vec = m2f_tile_vec_ok(it.src, it.lds);
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = blockIdx.x; t < tiles_c * tiles_r; t += gridDim.x) {
        const int r0 = (t / tiles_c) << 6, c0 = (t % tiles_c) << 6;
        m2f_tile64<true>(it.src, it.lds, it.rows, it.cols, r0, c0, it.dst, it.ldd, it.dst_t, it.ldd_t, false, vec, cs, tile);
    }
}

// [T, F] fp32 -> [F, ldt] bf16 (tokens contiguous), 64 features per workgroup, 64-token tiles through LDS: reads are
// 256-byte rows of the source, writes 128-byte rows of the destination.  (The 16-byte-load / 16-byte-store tile routine
// above was measured SLOWER here, 106 vs 90 us: this kernel already runs at 4.9 TB/s - 435 MB per step at C2 - and the
// simple 4-byte form keeps more loads in flight per lane.)
__global__ __launch_bounds__(256) void m2f_transpose_tokens_kernel(const TransBatch tb) {
    __shared__ float tile[64][65];
    __shared__ float part[4][64];
    const TransItem& it = tb.items[tb.block_item[blockIdx.x]];
    const int f0 = ((int)blockIdx.x - it.block_begin) * 64;
    const int T = tb.T;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;           // 64 x 4
    const bool fok = f0 + lx < it.F;
    // (the item's pointers are read from device memory, so these are FLAT accesses; casting them to the global address space
    // was measured SLOWER here - 99 vs 92 us - the per-load counted waits interleave badly with the LDS tile writes)
    const float* src = it.src + f0 + (fok ? lx : 0);
    const bool relu = it.relu != 0;
    float cs = 0.f;
    for (int t0 = 0; t0 < it.ldt; t0 += 64) {
        float x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {                                  // unconditional clamped loads, then select
            const int t = t0 + ly + 4 * i;
            x[i] = src[(size_t)(t < T ? t : T - 1) * it.ld];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = t0 + ly + 4 * i;
            float v = (fok && t < T) ? x[i] : 0.f;
            cs += v;
            if (relu) v = fmaxf(v, 0.f);
            tile[ly + 4 * i][lx] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = ly + 4 * i, t = t0 + lx;                      // lanes: consecutive tokens of one feature
            if (f0 + f < it.F && t < it.ldt) it.dst[(size_t)(f0 + f) * it.ldt + t] = m2f_bf16_bits(tile[lx][f]);
        }
        __syncthreads();
    }
    if (it.colsum) {                                                    // block-uniform
        part[ly][lx] = cs;
        __syncthreads();
        if (ly == 0 && fok) it.colsum[f0 + lx] = (part[0][lx] + part[1][lx]) + (part[2][lx] + part[3][lx]);
    }
}

hipError_t m2f_launch_transpose_tokens(const TransBatch& tb, hipStream_t stream) {
    if (tb.blocks <= 0) return hipSuccess;
    if (!tb.items || !tb.block_item || tb.T < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(m2f_transpose_tokens_kernel, dim3(tb.blocks), dim3(256), 0, stream, tb);
    return hipGetLastError();
}

hipError_t m2f_launch_cast(const CastBatch& cb, hipStream_t stream) {
    if (cb.count <= 0) return hipSuccess;
    if (cb.count > M2F_CAST_MAX_ITEMS) return hipErrorInvalidValue;
    bool transposed = true;
    for (int i = 0; i < cb.count; ++i) transposed = transposed && cb.it[i].dst_t != nullptr;
    if (transposed) {
        size_t mt = 0;
        for (int i = 0; i < cb.count; ++i) mt = std::max(mt, (size_t)((cb.it[i].rows + 63) / 64) * ((cb.it[i].cols + 63) / 64));
        const int bx = (int)std::min<size_t>(std::max<size_t>(mt, 1), 96);
        hipLaunchKernelGGL(m2f_cast_t_kernel, dim3(bx, cb.count), dim3(256), 0, stream, cb);
        return hipGetLastError();
    }
    size_t mx = 0;
    for (int i = 0; i < cb.count; ++i) mx = std::max(mx, (size_t)cb.it[i].rows * ((cb.it[i].cols + 3) / 4));
    int bx = (int)std::min<size_t>((mx + 255) / 256, 512);
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(m2f_cast_kernel, dim3(bx, cb.count), dim3(256), 0, stream, cb);
    return hipGetLastError();
}

// RoBERTa embeddings (transformers RobertaEmbeddings.forward, used by the reference at
// src/feature_extractors/text/model.py:16,19): LayerNorm(word[ids] + position[pos_ids] + token_type[0]), one wavefront per
// token, d <= 2048 (d % 4 == 0), values in registers.
__global__ __launch_bounds__(256) void m2f_embed_ln_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ pos_ids,
                                                           const float* __restrict__ word, const float* __restrict__ pos,
                                                           const float* __restrict__ type0, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float* __restrict__ out,
                                                           int ld, int T, int d, ShadowMap sh) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const float* w = word + (size_t)ids[t] * d;
    const float* p = pos + (size_t)pos_ids[t] * d;
    f32x4 x[8];
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 4 * (lane + 64 * j);
        if (c < d) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(w + c), b = *reinterpret_cast<const f32x4*>(p + c);
            const f32x4 ty = *reinterpret_cast<const f32x4*>(type0 + c);
            x[j] = a + b + ty;
            s1 += (x[j][0] + x[j][1]) + (x[j][2] + x[j][3]);
        } else {
            x[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = m2f_wave_sum(s1) / (float)d;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 4 * (lane + 64 * j);
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dv = x[j][e] - mean; s2 += dv * dv; }
        }
    }
    const float rstd = rsqrtf(m2f_wave_sum(s2) / (float)d + eps);
    float* o = out + (size_t)t * ld;
    uint16_t* o16 = m2f_shadow_of(sh, o);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 4 * (lane + 64 * j);
        if (c < d) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (x[j][e] - mean) * rstd * g[e] + be[e];
            *reinterpret_cast<f32x4*>(o + c) = y;
            if (o16) {
                uint2 h;
                h.x = (uint32_t)m2f_bf16_bits(y[0]) | ((uint32_t)m2f_bf16_bits(y[1]) << 16);
                h.y = (uint32_t)m2f_bf16_bits(y[2]) | ((uint32_t)m2f_bf16_bits(y[3]) << 16);
                *reinterpret_cast<uint2*>(o16 + c) = h;
            }
        }
    }
}

hipError_t m2f_launch_embed_ln(const int64_t* ids, const int64_t* pos_ids, const float* word, const float* pos, const float* type0,
                               const float* gamma, const float* beta, float eps, float* out, int ld, int T, int d, ShadowMap sh,
                               hipStream_t stream) {
    if (T < 1 || d < 4 || d > 2048 || (d & 3) || (ld & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(m2f_embed_ln_kernel, dim3(m2f_cdiv(T, 4)), dim3(256), 0, stream, ids, pos_ids, word, pos, type0, gamma, beta,
                       eps, out, ld, T, d, sh);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void m2f_quant_fp8_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst, int64_t n4, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 x = reinterpret_cast<const f32x4*>(src)[i];
        reinterpret_cast<uint32_t*>(dst)[i] = m2f_fp8x4_bits(x[0] * scale, x[1] * scale, x[2] * scale, x[3] * scale);
    }
}

hipError_t m2f_launch_quant_fp8(const float* src, uint8_t* dst, int64_t n, float scale, hipStream_t stream) {
    if (n <= 0 || (n & 3) || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 3)) return hipErrorInvalidValue;
    const int64_t n4 = n >> 2;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(m2f_quant_fp8_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, n4, scale);
    return hipGetLastError();
}

hipError_t m2f_launch_gather(const GatherArgs& a, hipStream_t stream) {
    if (a.T < 1 || !a.rows) return hipErrorInvalidValue;
    hipLaunchKernelGGL(m2f_gather_kernel, dim3(m2f_cdiv(a.T, 4)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t m2f_launch_rng_advance(uint32_t* rng, hipStream_t stream) {
    hipLaunchKernelGGL(m2f_rng_advance_kernel, dim3(1), dim3(64), 0, stream, rng);
    return hipGetLastError();
}

hipError_t m2f_launch_adam_shadowed(float* p, const void* g, int g_is_bf16, float* m, float* v, uint16_t* shadow, const AdamItem* items,
                                    const int* tile_begin, int n_items, int tile_first, int total_tiles, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, int step, const float* grad_scale_ptr,
                                    hipStream_t stream) {
    const int n_tiles = total_tiles - tile_first;
    if (n_items < 1 || n_items > M2F_ADAM_MAX_ITEMS || tile_first < 0 || n_tiles < 1 || !items || !tile_begin || !shadow) return hipErrorInvalidValue;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const int blocks = n_tiles < 256 * 8 ? n_tiles : 256 * 8;
    if (g_is_bf16)
        hipLaunchKernelGGL(m2f_adam_shadow_kernel<true>, dim3(blocks), dim3(256), 0, stream, p, g, m, v, shadow, items, tile_begin, n_items,
                           tile_first, total_tiles, (float)(lr / bc1), beta1, beta2, eps, weight_decay, (float)(1.0 / sqrt(bc2)), grad_scale_ptr);
    else
        hipLaunchKernelGGL(m2f_adam_shadow_kernel<false>, dim3(blocks), dim3(256), 0, stream, p, g, m, v, shadow, items, tile_begin, n_items,
                           tile_first, total_tiles, (float)(lr / bc1), beta1, beta2, eps, weight_decay, (float)(1.0 / sqrt(bc2)), grad_scale_ptr);
    return hipGetLastError();
}

// fp32 -> bf16 of the flat-buffer ranges of `items` (whole tensors incl. their pads: rows > 0 -> rows x cols elements, else `cols`): the
// gradients the weight-gradient table launch did NOT write as bf16 itself (biases, LayerNorm, the matrices outside the table) when the
// step leaves its gradients in a bf16 buffer (m2f_plan_grad_bf16); tiles of 4,096 elements, tile -> item by bisection
__global__ __launch_bounds__(256) void m2f_cast_items_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, const AdamItem* __restrict__ items,
                                                            const int* __restrict__ tile_begin, int n_items, int total_tiles) {
    __shared__ int tb[M2F_ADAM_MAX_ITEMS + 1];
    for (int i = threadIdx.x; i <= n_items; i += 256) tb[i] = tile_begin[i];
    __syncthreads();
    for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
        int lo = 0, hi = n_items - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tb[mid] <= t) lo = mid; else hi = mid - 1; }
        const AdamItem it = items[lo];
        const long long n = it.rows > 0 ? (long long)it.rows * it.cols : (long long)it.cols;
        const long long base = (long long)(t - tb[lo]) * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long idx = base + q * 1024 + threadIdx.x * 4;
            if (idx + 3 < n) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + it.off + idx);
                uint2 w;
                w.x = (uint32_t)m2f_bf16_bits(v[0]) | ((uint32_t)m2f_bf16_bits(v[1]) << 16);
                w.y = (uint32_t)m2f_bf16_bits(v[2]) | ((uint32_t)m2f_bf16_bits(v[3]) << 16);
                *reinterpret_cast<uint2*>(dst + it.off + idx) = w;
            } else {
                for (int e = 0; e < 4; ++e) if (idx + e < n) dst[it.off + idx + e] = m2f_bf16_bits(src[it.off + idx + e]);
            }
        }
    }
}
hipError_t m2f_launch_cast_items(const float* src, uint16_t* dst, const AdamItem* items, const int* tile_begin, int n_items, int total_tiles, hipStream_t stream) {
    if (n_items < 1 || n_items > M2F_ADAM_MAX_ITEMS || total_tiles < 1 || !items || !tile_begin || !src || !dst) return hipErrorInvalidValue;
    hipLaunchKernelGGL(m2f_cast_items_kernel, dim3(total_tiles < 2048 ? total_tiles : 2048), dim3(256), 0, stream, src, dst, items, tile_begin, n_items, total_tiles);
    return hipGetLastError();
}

// step-dependent factors of the update in device memory (a captured graph cannot take them as kernel arguments): one thread
__global__ void m2f_adam_hyper_kernel(float* __restrict__ h, float lr_bc1, float beta1, float beta2, float eps, float wd, float inv_sqrt_bc2) {
    h[0] = lr_bc1; h[1] = beta1; h[2] = beta2; h[3] = eps; h[4] = wd; h[5] = inv_sqrt_bc2; h[6] = 0.f; h[7] = 0.f;
}
hipError_t m2f_launch_adam_hyper(float* hyper_dev, float lr, float beta1, float beta2, float eps, float weight_decay, int step, hipStream_t stream) {
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);       // as m2f_launch_adam_shadowed computes them
    hipLaunchKernelGGL(m2f_adam_hyper_kernel, dim3(1), dim3(1), 0, stream, hyper_dev, (float)(lr / bc1), beta1, beta2, eps, weight_decay,
                       (float)(1.0 / sqrt(bc2)));
    return hipGetLastError();
}
// the shadow-writing kernel with those factors read from `hyper_dev` (m2f_adam_shadow_dev_kernel: same body, same arithmetic)
hipError_t m2f_launch_adam_shadowed_dev(float* p, const float* g, float* m, float* v, uint16_t* shadow, const AdamItem* items, const int* tile_begin,
                                        int n_items, int total_tiles, const float* hyper_dev, const float* grad_scale_ptr, hipStream_t stream) {
    if (n_items < 1 || n_items > M2F_ADAM_MAX_ITEMS || total_tiles < 1 || !items || !tile_begin || !shadow || !hyper_dev) return hipErrorInvalidValue;
    const int blocks = total_tiles < 256 * 8 ? total_tiles : 256 * 8;
    hipLaunchKernelGGL(m2f_adam_shadow_dev_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, shadow, items, tile_begin, n_items, total_tiles,
                       hyper_dev, grad_scale_ptr);
    return hipGetLastError();
}

hipError_t m2f_launch_adam(float* p, const void* g, int g_is_bf16, float* m, float* v, int64_t n, float lr, float beta1,
                           float beta2, float eps, float weight_decay, int step, const float* grad_scale_ptr,
                           hipStream_t stream) {
    if (n & 3) return hipErrorInvalidValue;      // flat buffers are padded to 64 floats per tensor
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const int64_t n4 = n >> 2;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (g_is_bf16)
        hipLaunchKernelGGL(m2f_adam_kernel<true>, dim3(blocks), dim3(256), 0, stream, p, g, m, v, n4, (float)(lr / bc1), beta1, beta2,
                           eps, weight_decay, (float)(1.0 / sqrt(bc2)), grad_scale_ptr);
    else
        hipLaunchKernelGGL(m2f_adam_kernel<false>, dim3(blocks), dim3(256), 0, stream, p, g, m, v, n4, (float)(lr / bc1), beta1, beta2,
                           eps, weight_decay, (float)(1.0 / sqrt(bc2)), grad_scale_ptr);
    return hipGetLastError();
}
