// Long-sequence self-attention forward on bf16 operands (round 4): the token-level attention of the in-loop text encoder
// (reference: transformers' RobertaSelfAttention behind src/feature_extractors/text/model.py:16-21) in its bf16 mode.
//
// attention.hip's m2f_attn_long_fwd_kernel reads Q / K / V as fp32 (12 bytes per element of the packed projection) and multiplies on the
// exact-fp32 MFMA; in the bf16 mode of the encoder nobody else reads those fp32 copies, so this kernel takes the bf16 result of the
// packed Q / K / V GEMM as it is (2 bytes per element), multiplies on v_mfma_f32_16x16x16_bf16 and writes the context rows as bf16 -
// the operand the output projection stages anyway.  Per layer at 32,768 tokens x 1,024: 0.27 GB instead of 1.0 GB.
//
// One workgroup = 64 queries of one (sequence, head); wave w owns queries 16w .. 16w+15.  Keys / values stream through LDS in blocks of
// 64 with an online softmax (running max / sum per query), so any S fits.  Orientation as in attention.hip: S^T = K Q^T leaves, for
// query (lane & 15), keys 16 jt + 4 (lane >> 4) + r of key tile jt in the accumulator - rounded to bf16 that IS the A operand
// (row = query, k = 4 (lane >> 4) + r) of the P V product; V stays row-major in LDS as it arrives and its B fragment (four consecutive
// keys of one column) comes from the transposing read ds_read_b64_tr_b16.  LDS rows are 2 W + 16 bytes (W = head dim rounded up to 16):
// the dword stride W / 2 + 4 is an odd multiple of 4, which puts the 16 rows of a fragment read on 16 disjoint groups of four banks.
#include "common.h"
#include "ops.h"

namespace {

constexpr int NTHR = 256;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int CT>
__global__ __launch_bounds__(NTHR) void m2f_attn_long_bf16_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                                 const uint16_t* __restrict__ v, int ldq, int ldk, int ldv,
                                                                 const uint8_t* __restrict__ key_pad, uint16_t* __restrict__ out16,
                                                                 float* __restrict__ out32, uint8_t* __restrict__ out8, float out8_scale, int ldo, int S,
                                                                 int H, int hd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int W = 16 * CT, LD = 2 * W + 16, CH = W / 8;      // LDS row bytes; 16-byte chunks per row
    char* Qs = smem;
    char* Ks = Qs + 64 * LD;
    char* Vs = Ks + 64 * LD;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int b = (int)blockIdx.x / H, h = (int)blockIdx.x - b * H;
    const int q0 = (int)blockIdx.y * 64;
    const size_t tok0 = (size_t)b * S;
    const int nq = S - q0 < 64 ? S - q0 : 64;

    // 64 rows x CH chunks of a [rows][hd] bf16 slab -> LDS, rows past `n` and columns past hd as zeros (a zero V row times a zero
    // probability must stay zero, not NaN)
    auto stage = [&](char* dst, const uint16_t* src, int ld, int n) {
#pragma unroll
        for (int e0 = 0; e0 < 64 * CH; e0 += NTHR) {
            const int e = e0 + tid;
            const int r = e / CH, c = e - r * CH;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (r < n && 8 * c < hd) val = *reinterpret_cast<const u32x4*>(src + (size_t)r * ld + 8 * c);
            *reinterpret_cast<u32x4*>(dst + r * LD + 16 * c) = val;
        }
    };
    stage(Qs, q + (tok0 + q0) * ldq + h * hd, ldq, nq);

    const float scale = 1.0f / sqrtf((float)hd);
    float m_run = -INFINITY, l_run = 0.f;                       // of query 16 wv + l15 (replicated over lg)
    f32x4 o[CT];                                                // O[query 16 wv + 4 lg + r][column 16 ct + l15]
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) o[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    s16x4 qf[CT];                                               // this lane's Q fragments: query 16 wv + l15, columns 16 ks + 4 lg + 0..3

    for (int kb = 0; kb < S; kb += 64) {
        const int nk = S - kb < 64 ? S - kb : 64;
        __syncthreads();                                        // previous block's K / V fully consumed
        stage(Ks, k + (tok0 + kb) * ldk + h * hd, ldk, nk);
        stage(Vs, v + (tok0 + kb) * ldv + h * hd, ldv, nk);
        const unsigned char kp = key_pad ? key_pad[tok0 + kb + (lane < nk ? lane : 0)] : (unsigned char)0;
        const unsigned long long kvalid = __ballot(lane < nk && kp == 0);
        __syncthreads();
        if (kb == 0) {
#pragma unroll
            for (int ks = 0; ks < CT; ++ks) qf[ks] = *reinterpret_cast<const s16x4*>(Qs + (16 * wv + l15) * LD + 32 * ks + 8 * lg);
        }
        f32x4 s[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < CT; ++ks) {
                const s16x4 kf = *reinterpret_cast<const s16x4*>(Ks + (16 * jt + l15) * LD + 32 * ks + 8 * lg);
                acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kf, qf[ks], acc, 0, 0, 0);
            }
            s[jt] = acc;                                        // S^T[key 16 jt + 4 lg + r][query l15]
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
        s16x4 pf[4];                                            // probabilities as the A operand: row = query l15, k = 4 lg + r of tile jt
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = (m_new == -INFINITY) ? 0.f : __expf(s[jt][r] - m_new);
                const uint16_t pb = m2f_bf16_bits(p);
                pf[jt][r] = (short)pb;
                sum += m2f_bf16_to_f32(pb);                     // the denominator sums what the product multiplies
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        l_run = l_run * alpha + sum;
        m_run = m_new;
        float a4[4];                                            // accumulator row 4 lg + r has its alpha in the lanes with l15 == 4 lg + r
#pragma unroll
        for (int r = 0; r < 4; ++r) a4[r] = __shfl(alpha, 4 * lg + r, 64);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            f32x4 acc = o[ct];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] *= a4[r];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                // block of keys 16 jt + 4 lg + 0..3 x columns 16 ct + 0..15: lane 4 q + p of the group addresses row q, columns 4 p .. 4 p + 3
                const char* vp = Vs + (16 * jt + 4 * lg + (l15 >> 2)) * LD + 32 * ct + 8 * (l15 & 3);
                const s16x4 vf = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(const_cast<char*>(vp)));
                acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pf[jt], vf, acc, 0, 0, 0);
            }
            o[ct] = acc;
        }
    }
    float inv4[4];
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) inv4[r] = __shfl(inv, 4 * lg + r, 64);
    // the wave's 16 x W result through ITS OWN rows of the Q slab (its Q fragments live in registers since the first block), so that the
    // rows leave as 16-byte stores: as bf16 (out16), and / or as OCP e4m3 bytes of value * out8_scale, saturating (out8: the operand the
    // fp8 output projection stages - no fp32 copy, no quantise pass)
    if (out16 || out32) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float val = o[ct][r] * inv4[r];
                if (out16) *reinterpret_cast<uint16_t*>(Qs + (16 * wv + 4 * lg + r) * LD + 2 * (16 * ct + l15)) = m2f_bf16_bits(val);
                if (out32) {
                    const int io = q0 + 16 * wv + 4 * lg + r, c = 16 * ct + l15;
                    if (io < S && c < hd) out32[(tok0 + io) * ldo + h * hd + c] = val;
                }
            }
        __syncthreads();
        if (out16) {
#pragma unroll
            for (int e0 = 0; e0 < 16 * CH; e0 += 64) {
                const int e = e0 + lane;
                const int r = e / CH, c = e - r * CH;
                if (e < 16 * CH && 16 * wv + r < nq && 8 * c < hd)
                    *reinterpret_cast<u32x4*>(out16 + (tok0 + q0 + 16 * wv + r) * ldo + h * hd + 8 * c) =
                        *reinterpret_cast<const u32x4*>(Qs + (16 * wv + r) * LD + 16 * c);
            }
        }
        __syncthreads();
    }
    if (out8) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float val = fminf(fmaxf(o[ct][r] * inv4[r] * out8_scale, -448.f), 448.f);
                const int w = __builtin_amdgcn_cvt_pk_fp8_f32(val, val, 0, false);
                *reinterpret_cast<uint8_t*>(Qs + (16 * wv + 4 * lg + r) * LD + (16 * ct + l15)) = (uint8_t)(w & 0xff);
            }
        __syncthreads();
        constexpr int CH8 = W / 16;                              // 16-byte chunks of a row of W bytes (hd % 16 == 0 on this path)
#pragma unroll
        for (int e0 = 0; e0 < 16 * CH8; e0 += 64) {
            const int e = e0 + lane;
            const int r = e / CH8, c = e - r * CH8;
            if (e < 16 * CH8 && 16 * wv + r < nq && 16 * c < hd)
                *reinterpret_cast<u32x4*>(out8 + (tok0 + q0 + 16 * wv + r) * ldo + h * hd + 16 * c) =
                    *reinterpret_cast<const u32x4*>(Qs + (16 * wv + r) * LD + 16 * c);
        }
    }
}

}  // namespace

// q / k / v: bf16 [B * S, ...] with leading dimensions ldq / ldk / ldv (elements), head h in columns h hd .. h hd + hd - 1; outputs, each
// nullable, at least one, all [B * S, ldo]: out16 (bf16), out32 (fp32), out8 (e4m3 of value * out8_scale, saturating; needs hd and ldo in
// units of 16).  hd, the leading dimensions and the base addresses in units of 8 elements.
hipError_t m2f_launch_attn_long_fwd_bf16(const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v, int ldv,
                                         const uint8_t* key_pad, uint16_t* out16, float* out32, uint8_t* out8, float out8_scale, int ldo,
                                         int B, int S, int H, int hd, hipStream_t stream) {
    if (B < 1 || S < 1 || H < 1 || hd < 8 || hd > 128 || (hd & 7) || !q || !k || !v || !(out16 || out32 || out8)) return hipErrorInvalidValue;
    if ((ldq | ldk | ldv | ldo) & 7) return hipErrorInvalidValue;
    if (out8 && ((hd & 15) || (ldo & 15) || (reinterpret_cast<uintptr_t>(out8) & 15))) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(out16) |
         reinterpret_cast<uintptr_t>(out32)) & 15)
        return hipErrorInvalidValue;
    if ((S + 63) / 64 > 65535 || (long long)B * H > 0x7fffffffLL) return hipErrorInvalidValue;
    const int CT = (hd + 15) / 16;
    const size_t lds = (size_t)3 * 64 * (2 * 16 * CT + 16);
    const dim3 grid((unsigned)(B * H), (unsigned)((S + 63) / 64));
#define M2F_ALB_CASE(N)                                                                                                          \
    case N:                                                                                                                      \
        hipLaunchKernelGGL(m2f_attn_long_bf16_kernel<N>, grid, dim3(NTHR), lds, stream, q, k, v, ldq, ldk, ldv, key_pad, out16,  \
                           out32, out8, out8_scale, ldo, S, H, hd);                                                               \
        break;
    switch (CT) {
        M2F_ALB_CASE(1) M2F_ALB_CASE(2) M2F_ALB_CASE(3) M2F_ALB_CASE(4) M2F_ALB_CASE(5) M2F_ALB_CASE(6) M2F_ALB_CASE(7) M2F_ALB_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef M2F_ALB_CASE
    return hipGetLastError();
}
