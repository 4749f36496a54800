// Skinny GEMMs of the classifier head (reference src/model.py:89-100: the last Linear has n_classes = 7 outputs).  As 64x64 MFMA
// tiles they are one column of tiles with 57 of 64 columns empty (logits, 12.6 us at C3) or a k-loop of 7 (the input gradient
// d h = d logits . W, 16.8 us): launch-latency-sized problems on 16 workgroups.  Here: plain FMA kernels over the whole chip.
//   NT, N <= 8:  C[t][n] = sum_k A[t][k] B[n][k] (+ bias[n]) (ReLU)            one wavefront per token row
//   NN, K <= 8:  C[t][j] = sum_c A[t][c] B[c][j], then the ReLU / dropout gate  one thread per four output columns
// Operand precision follows the MFMA kernels they replace: bf16 mode multiplies bf16-rounded operands (read from the bf16 shadows
// when EVERY operand of the launch has one - the rule plan.hip::mark_unread_fp32 mirrors - else rounded on the way in) and
// accumulates in fp32; fp32 mode is plain fp32.  Only the summation order differs from the MFMA forms.
#include "common.h"
#include "ops.h"
#include <cstring>
#include <cstdlib>

namespace {

__device__ __forceinline__ float sk_round(float x, bool r16) {
    return r16 ? __builtin_bit_cast(float, (uint32_t)m2f_bf16_bits(x) << 16) : x;
}
__device__ __forceinline__ float sk_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct SkinnyArgs {
    const float* a; const float* b; const uint16_t* a16; const uint16_t* b16;      // a16 / b16: both set or both null
    int lda, ldb, lda16, ldb16;
    float* c; int ldc; uint16_t* c16;
    const float* bias; const float* gate; int ldgate; float gscale;
    int M, N, K; int relu_out; int r16;                                          // r16: round fp32 operands to bf16 (bf16 mode without shadows)
    int vec;                                                                     // 16-byte accesses are legal (alignment, whole chunks)
};

// one wavefront per row; a lane takes 16-byte chunks of k (8 bf16 or 4 fp32 values) of A's row and of the N <= 8 rows of B
// (s.vec: every row 16-byte aligned and K a whole number of chunks - else one value at a time); N running sums per lane
__global__ __launch_bounds__(256) void m2f_skinny_nt_kernel(const SkinnyArgs s) {
    const int lane = threadIdx.x & 63;
    const int t = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= s.M) return;                                            // wave-uniform
    float acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = 0.f;
    const bool r16 = s.r16;
    if (s.vec && s.a16) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        for (int ch = lane; ch < (s.K >> 3); ch += 64) {
            const u32x4 aw = *reinterpret_cast<const u32x4*>(s.a16 + (size_t)t * s.lda16 + 8 * ch);
            float av[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { av[2 * e] = __builtin_bit_cast(float, aw[e] << 16); av[2 * e + 1] = __builtin_bit_cast(float, aw[e] & 0xFFFF0000u); }
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                if (n < s.N) {
                    const u32x4 bw = *reinterpret_cast<const u32x4*>(s.b16 + (size_t)n * s.ldb16 + 8 * ch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[n] += av[2 * e] * __builtin_bit_cast(float, bw[e] << 16);
                        acc[n] += av[2 * e + 1] * __builtin_bit_cast(float, bw[e] & 0xFFFF0000u);
                    }
                }
            }
        }
    } else if (s.vec) {
        for (int ch = lane; ch < (s.K >> 2); ch += 64) {
            const f32x4 aw = *reinterpret_cast<const f32x4*>(s.a + (size_t)t * s.lda + 4 * ch);
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                if (n < s.N) {
                    const f32x4 bw = *reinterpret_cast<const f32x4*>(s.b + (size_t)n * s.ldb + 4 * ch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[n] += sk_round(aw[e], r16) * sk_round(bw[e], r16);
                }
            }
        }
    } else {
        for (int k = lane; k < s.K; k += 64) {
            const float av = s.a16 ? __builtin_bit_cast(float, (uint32_t)s.a16[(size_t)t * s.lda16 + k] << 16) : sk_round(s.a[(size_t)t * s.lda + k], r16);
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                if (n < s.N) {
                    const float bv = s.b16 ? __builtin_bit_cast(float, (uint32_t)s.b16[(size_t)n * s.ldb16 + k] << 16) : sk_round(s.b[(size_t)n * s.ldb + k], r16);
                    acc[n] += av * bv;
                }
            }
        }
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = sk_wave_sum(acc[n]);
    if (lane < s.N) {
        float x = 0.f;
#pragma unroll
        for (int n = 0; n < 8; ++n) if (lane == n) x = acc[n];
        x += s.bias ? s.bias[lane] : 0.f;
        if (s.relu_out) x = fmaxf(x, 0.f);
        const size_t o = (size_t)t * s.ldc + lane;
        s.c[o] = x;
        if (s.c16) s.c16[o] = m2f_bf16_bits(x);
    }
}

// one thread per (row, four consecutive columns); K <= 8 terms in index order; s.vec: 16-byte accesses of B's rows, the gate and C
__global__ __launch_bounds__(256) void m2f_skinny_nn_kernel(const SkinnyArgs s) {
    const int n4 = (s.N + 3) >> 2;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)s.M * n4) return;
    const int t = (int)(id / n4), j = 4 * (int)(id - (size_t)t * n4);
    const bool r16 = s.r16;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    if (s.vec) {
        float av[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            av[c] = c < s.K ? (s.a16 ? __builtin_bit_cast(float, (uint32_t)s.a16[(size_t)t * s.lda16 + c] << 16) : sk_round(s.a[(size_t)t * s.lda + c], r16)) : 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c < s.K) {
                f32x4 bw;
                if (s.b16) {
                    const uint2 w = *reinterpret_cast<const uint2*>(s.b16 + (size_t)c * s.ldb16 + j);
                    bw = (f32x4){__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xFFFF0000u),
                                 __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xFFFF0000u)};
                } else {
                    bw = *reinterpret_cast<const f32x4*>(s.b + (size_t)c * s.ldb + j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bw[e] = sk_round(bw[e], r16);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] += av[c] * bw[e];
            }
        }
        f32x4 v = {x[0], x[1], x[2], x[3]};
        if (s.bias) { const f32x4 bb = *reinterpret_cast<const f32x4*>(s.bias + j); v += bb; }
        if (s.relu_out) { for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f); }
        if (s.gate) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(s.gate + (size_t)t * s.ldgate + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = g[e] > 0.f ? v[e] * s.gscale : 0.f;
        }
        const size_t o = (size_t)t * s.ldc + j;
        *reinterpret_cast<f32x4*>(s.c + o) = v;
        if (s.c16) {
            uint2 hh;
            hh.x = (uint32_t)m2f_bf16_bits(v[0]) | ((uint32_t)m2f_bf16_bits(v[1]) << 16);
            hh.y = (uint32_t)m2f_bf16_bits(v[2]) | ((uint32_t)m2f_bf16_bits(v[3]) << 16);
            *reinterpret_cast<uint2*>(s.c16 + o) = hh;
        }
        return;
    }
    for (int c = 0; c < s.K; ++c) {
        const float av = s.a16 ? __builtin_bit_cast(float, (uint32_t)s.a16[(size_t)t * s.lda16 + c] << 16) : sk_round(s.a[(size_t)t * s.lda + c], r16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (j + e < s.N) {
                const float bv = s.b16 ? __builtin_bit_cast(float, (uint32_t)s.b16[(size_t)c * s.ldb16 + j + e] << 16) : sk_round(s.b[(size_t)c * s.ldb + j + e], r16);
                x[e] += av * bv;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (j + e < s.N) {
            float v = x[e] + (s.bias ? s.bias[j + e] : 0.f);
            if (s.relu_out) v = fmaxf(v, 0.f);
            if (s.gate) v = s.gate[(size_t)t * s.ldgate + j + e] > 0.f ? v * s.gscale : 0.f;
            const size_t o = (size_t)t * s.ldc + j + e;
            s.c[o] = v;
            if (s.c16) s.c16[o] = m2f_bf16_bits(v);
        }
    }
}

}  // namespace

// 1: launched; 0: not a skinny problem (the caller goes on to the MFMA kernels); < 0: HIP error code (negated)
int m2f_launch_gemm_skinny(const GemmBatch& gb, int prec, int layout, hipStream_t stream) {
    static const int on = getenv("M2F_SKINNY") ? atoi(getenv("M2F_SKINNY")) : 1;
    if (!on || gb.count != 1 || (prec != M2F_PREC_BF16 && prec != M2F_PREC_F32)) return 0;
    const GemmProblem& p = gb.pr[0];
    const int K = p.a.k[0];
    if (p.a.k[1] || p.b.k[1] || p.res || p.drop_site || p.c8 || p.bias_grad || !p.c || K < 1 ||
        (p.flags & (GF_ACCUM | GF_RELU_A | GF_RELU_B | GF_GELU_OUT)))
        return 0;
    // (whatever M: a token row's result must not depend on how many rows the launch has - packed and padded plans agree bit for bit)
    const bool nt = layout == M2F_LAYOUT_NT && p.N <= 8 && !p.gate;
    const bool nn = layout == M2F_LAYOUT_NN && K <= 8;
    if (!nt && !nn) return 0;
    SkinnyArgs s = {};
    const bool use16 = prec == M2F_PREC_BF16 && p.a.q[0] && p.b.q[0] && m2f_gemm_stages_bf16(gb, layout);      // the launcher's own rule
    s.a = p.a.p[0]; s.b = p.b.p[0]; s.lda = p.a.ld[0]; s.ldb = p.b.ld[0];
    if (use16) { s.a16 = p.a.q[0]; s.b16 = p.b.q[0]; s.lda16 = p.a.ldq[0]; s.ldb16 = p.b.ldq[0]; }
    else if (!s.a || !s.b) return 0;
    s.r16 = prec == M2F_PREC_BF16 && !use16;
    s.c = p.c; s.ldc = p.ldc;
    {
        const ptrdiff_t i = gb.sh.shadow && p.c >= gb.sh.ws_base ? p.c - gb.sh.ws_base : -1;
        s.c16 = (i >= 0 && (size_t)i < gb.sh.ws_floats) ? gb.sh.shadow + i : nullptr;
    }
    s.bias = p.bias; s.gate = p.gate; s.ldgate = p.ldgate; s.gscale = p.gate_scale;
    s.M = p.M; s.N = p.N; s.K = K; s.relu_out = (p.flags & GF_RELU_OUT) ? 1 : 0;
    auto al = [](const void* q, int a) { return (reinterpret_cast<uintptr_t>(q) & (uintptr_t)(a - 1)) == 0; };
    if (nt) {       // rows of A and B in 16-byte chunks
        s.vec = use16 ? ((K & 7) == 0 && (s.lda16 & 7) == 0 && (s.ldb16 & 7) == 0 && al(s.a16, 16) && al(s.b16, 16))
                      : ((K & 3) == 0 && (s.lda & 3) == 0 && (s.ldb & 3) == 0 && al(s.a, 16) && al(s.b, 16));
    } else {        // four columns of B's rows, the gate, C (and their bf16 forms: 8 bytes)
        s.vec = (p.N & 3) == 0 && (s.ldc & 3) == 0 && al(s.c, 16) && (!s.c16 || al(s.c16, 8)) && (!s.bias || al(s.bias, 16)) &&
                (!s.gate || ((s.ldgate & 3) == 0 && al(s.gate, 16))) &&
                (use16 ? ((s.ldb16 & 3) == 0 && al(s.b16, 8)) : ((s.ldb & 3) == 0 && al(s.b, 16)));
    }
    if (nt) hipLaunchKernelGGL(m2f_skinny_nt_kernel, dim3((p.M + 3) / 4), dim3(256), 0, stream, s);
    else hipLaunchKernelGGL(m2f_skinny_nn_kernel, dim3((unsigned)(((size_t)p.M * ((p.N + 3) >> 2) + 255) / 256)), dim3(256), 0, stream, s);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 1 : -(int)e;
}
