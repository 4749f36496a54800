// Grouped MFMA GEMM for the M2FNet hot path (gfx950 / CDNA4, wave64).
//
//   C[M,N] = epilogue( sum_k A(m,k) * B(n,k) )      (see ops.h for layouts / epilogue order)
//
// Replaces every nn.Linear / in-projection / out-projection call the reference's model makes
// (reference src/model.py:14,18,111-113,123-125,143 and torch's TransformerEncoderLayer internals)
// in forward, input-gradient and weight-gradient form.
//
// Design (MI355X):
//   * one workgroup = 4 wavefronts (2x2), each wave owns (BM/2)x(BN/2) of the tile as 32x32 MFMA blocks;
//   * two arithmetic modes sharing one C/D layout:
//       F32  : v_mfma_f32_32x32x2_f32  (exact fp32, 157 TF peak)   - the 1e-3 parity mode
//       BF16 : v_mfma_f32_32x32x16_bf16 (fp32 accumulate, 2.5 PF)  - operands rounded to bf16 while
//              being staged into LDS; everything in HBM stays fp32 (weights are read once per use,
//              so no shadow copies / cast kernels are needed);
//   * operands are register-staged (coalesced 16-byte global loads issued one k-tile ahead of the
//     MFMAs that consume the previous tile) into a double-buffered LDS image laid out for
//     conflict-free fragment reads:
//       F32  : [k][row] floats, row stride BR+1 (ds_read_b32, lanes = consecutive rows)
//       BF16 : [row][k] bf16, row stride 144 B (ds_read_b128; any 16 consecutive rows cover the 64
//              banks exactly once)
//   * "grouped": one launch covers several independent problems (text+audio branch, q/k/v of a
//     fusion layer, split concat) so the 256 CUs see more workgroups per launch and the
//     launch-latency-bound chain gets shorter;
//   * the epilogue fuses bias, ReLU, dropout, residual add, ReLU-gate and accumulate; the wgrad form
//     also emits the bias gradient (column sums of dY) from the tiles it already streams.
#include "common.h"
#include "ops.h"

namespace {

template <int PREC, bool RC, int BR>
struct Stage {
    static constexpr int BK = (PREC == M2F_PREC_F32) ? 32 : 64;
    static constexpr int NT = BR / 32;                                   // staging tasks per thread
    static constexpr int FPT = (PREC == M2F_PREC_F32) ? 4 : 8;           // floats per task
    static constexpr int LDR = BR + 1;                                   // F32 row stride (floats)
    static constexpr int ROWB = BK * 2 + 16;                             // BF16 row stride (bytes)
    static constexpr int LDS_BYTES = (PREC == M2F_PREC_F32) ? BK * LDR * 4 : BR * ROWB;

    float v[NT][FPT];

    // Loads this thread's share of the [BR x BK] tile at (row0, kbase) of one operand segment.
    // colsum (RC only, wgrad bias gradient): per-thread running sums over k of the loaded values.
    __device__ __forceinline__ void load(const float* __restrict__ p, int ld, int rows, int row0,
                                         int kseg, int kbase, bool vec, bool relu, int tid,
                                         float* colsum) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int id = tid + 256 * t;
            if constexpr (PREC == M2F_PREC_F32 && !RC) {
                const int r = id >> 3, kc = id & 7;
                const int gr = row0 + r, gk = kbase + 4 * kc;
                if (vec) {
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
                    if (gr < rows && gk < kseg) x = *reinterpret_cast<const f32x4*>(p + (size_t)gr * ld + gk);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[t][e] = x[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[t][e] = (gr < rows && gk + e < kseg) ? p[(size_t)gr * ld + gk + e] : 0.f;
                }
            } else if constexpr (PREC == M2F_PREC_F32 && RC) {
                constexpr int CH = BR / 4;
                const int k = id / CH, rc = id % CH;
                const int gk = kbase + k, gr = row0 + 4 * rc;
                if (vec) {
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
                    if (gk < kseg && gr < rows) x = *reinterpret_cast<const f32x4*>(p + (size_t)gk * ld + gr);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[t][e] = x[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[t][e] = (gk < kseg && gr + e < rows) ? p[(size_t)gk * ld + gr + e] : 0.f;
                }
                if (colsum) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) colsum[e] += v[t][e];
                }
            } else if constexpr (PREC == M2F_PREC_BF16 && !RC) {
                const int r = id >> 3, kc = id & 7;
                const int gr = row0 + r, gk = kbase + 8 * kc;
                if (vec) {
                    f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
                    if (gr < rows) {
                        const float* q = p + (size_t)gr * ld + gk;
                        if (gk < kseg) x0 = *reinterpret_cast<const f32x4*>(q);
                        if (gk + 4 < kseg) x1 = *reinterpret_cast<const f32x4*>(q + 4);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[t][e] = x0[e]; v[t][4 + e] = x1[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        v[t][e] = (gr < rows && gk + e < kseg) ? p[(size_t)gr * ld + gk + e] : 0.f;
                }
            } else {   // BF16, RC: 8 consecutive k of one row; lanes run along the contiguous row dim
                const int kg = id / BR, r = id % BR;
                const int gr = row0 + r, gk = kbase + 8 * kg;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    v[t][e] = (gr < rows && gk + e < kseg) ? p[(size_t)(gk + e) * ld + gr] : 0.f;
                if (colsum) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) colsum[0] += v[t][e];
                }
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < FPT; ++e) v[t][e] = fmaxf(v[t][e], 0.f);
            }
        }
    }

    __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int id = tid + 256 * t;
            if constexpr (PREC == M2F_PREC_F32 && !RC) {
                const int r = id >> 3, kc = id & 7;
                float* f = reinterpret_cast<float*>(lds);
#pragma unroll
                for (int e = 0; e < 4; ++e) f[(4 * kc + e) * LDR + r] = v[t][e];
            } else if constexpr (PREC == M2F_PREC_F32 && RC) {
                constexpr int CH = BR / 4;
                const int k = id / CH, rc = id % CH;
                float* f = reinterpret_cast<float*>(lds);
#pragma unroll
                for (int e = 0; e < 4; ++e) f[k * LDR + 4 * rc + e] = v[t][e];
            } else {
                int r, c;
                if constexpr (!RC) { r = id >> 3; c = id & 7; } else { c = id / BR; r = id % BR; }
                bf16x8 h;
#pragma unroll
                for (int e = 0; e < 8; ++e) h[e] = (__bf16)v[t][e];
                *reinterpret_cast<bf16x8*>(lds + r * ROWB + c * 16) = h;
            }
        }
    }
};

template <int PREC, bool A_RC, bool B_RC, int BM, int BN>
__global__ __launch_bounds__(256) void m2f_gemm_kernel(const GemmBatch gb) {
    using SA = Stage<PREC, A_RC, BM>;
    using SB = Stage<PREC, B_RC, BN>;
    constexpr int BK = SA::BK;
    constexpr int MI = BM / 64, NI = BN / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    int pi = 0;
#pragma unroll
    for (int i = 1; i < M2F_GEMM_MAX_PROBLEMS; ++i)
        if (i < gb.count && (int)blockIdx.x >= gb.pr[i].tile_begin) pi = i;
    const GemmProblem& P = gb.pr[pi];

    const int M = P.M, N = P.N;
    const int tl = (int)blockIdx.x - P.tile_begin;
    const int m0 = (tl / P.tiles_n) * BM, n0 = (tl % P.tiles_n) * BN;
    const uint32_t flags = P.flags;
    const bool vecA = flags & GF_VEC_A, vecB = flags & GF_VEC_B;
    const bool reluA = flags & GF_RELU_A, reluB = flags & GF_RELU_B;
    const int k0 = P.a.k[0], k1 = P.a.k[1];
    const int nk0 = (k0 + BK - 1) / BK, nk = nk0 + (k1 + BK - 1) / BK;

    char* ldsA = smem;
    char* ldsB = smem + 2 * SA::LDS_BYTES;

    SA sa;
    SB sb;
    float colsum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool want_bg = A_RC && P.bias_grad != nullptr && n0 == 0;

    auto load_tile = [&](int kt) {
        const int seg = kt >= nk0 ? 1 : 0;
        const int kbase = (seg ? kt - nk0 : kt) * BK;
        sa.load(P.a.p[seg], P.a.ld[seg], M, m0, P.a.k[seg], kbase, vecA, reluA, tid, want_bg ? colsum : nullptr);
        sb.load(P.b.p[seg], P.b.ld[seg], N, n0, P.b.k[seg], kbase, vecB, reluB, tid, nullptr);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile(0);
    sa.store(ldsA, tid);
    sb.store(ldsB, tid);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) load_tile(kt + 1);                       // global loads in flight under the MFMAs

        const char* a_l = ldsA + cur * SA::LDS_BYTES;
        const char* b_l = ldsB + cur * SB::LDS_BYTES;
        if constexpr (PREC == M2F_PREC_F32) {
            const float* af = reinterpret_cast<const float*>(a_l) + wm * (BM / 2) + (lane & 31);
            const float* bf = reinterpret_cast<const float*>(b_l) + wn * (BN / 2) + (lane & 31);
#pragma unroll 4
            for (int ks = 0; ks < BK / 2; ++ks) {
                const int kk = 2 * ks + (lane >> 5);
                float a[MI], b[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = af[kk * SA::LDR + i * 32];
#pragma unroll
                for (int j = 0; j < NI; ++j) b[j] = bf[kk * SB::LDR + j * 32];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
            const char* ab = a_l + (wm * (BM / 2) + (lane & 31)) * SA::ROWB + (lane >> 5) * 16;
            const char* bb = b_l + (wn * (BN / 2) + (lane & 31)) * SB::ROWB + (lane >> 5) * 16;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 a[MI], b[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(ab + i * 32 * SA::ROWB + ks * 32);
#pragma unroll
                for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const bf16x8*>(bb + j * 32 * SB::ROWB + ks * 32);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }

        if (more) {
            sa.store(ldsA + (cur ^ 1) * SA::LDS_BYTES, tid);
            sb.store(ldsB + (cur ^ 1) * SB::LDS_BYTES, tid);
        }
        __syncthreads();
    }

    // ---- wgrad: bias gradient = column sums of the A operand (dY) over the reduction dim ----------
    if constexpr (A_RC) {
        if (want_bg) {      // block-uniform
            float* red = reinterpret_cast<float*>(smem);
            if constexpr (PREC == M2F_PREC_F32) {
                constexpr int CH = BM / 4, G = 256 / CH;
                const int g = tid / CH, rc = tid % CH;
#pragma unroll
                for (int e = 0; e < 4; ++e) red[g * BM + 4 * rc + e] = colsum[e];
                __syncthreads();
                if (tid < BM) {
                    float s = 0.f;
                    for (int q = 0; q < G; ++q) s += red[q * BM + tid];
                    if (m0 + tid < M) P.bias_grad[m0 + tid] = s;
                }
            } else {
                constexpr int G = 256 / BM;
                red[(tid / BM) * BM + (tid % BM)] = colsum[0];
                __syncthreads();
                if (tid < BM) {
                    float s = 0.f;
                    for (int q = 0; q < G; ++q) s += red[q * BM + tid];
                    if (m0 + tid < M) P.bias_grad[m0 + tid] = s;
                }
            }
        }
    }

    // ---- fused epilogue ---------------------------------------------------------------------------
    const float* bias = P.bias;
    const float* res = P.res;
    const float* gate = P.gate;
    float* C = P.c;
    const int ldc = P.ldc, ldres = P.ldres, ldgate = P.ldgate;
    const float gscale = P.gate_scale;
    const bool relu_out = flags & GF_RELU_OUT, accum = flags & GF_ACCUM;
    const uint32_t site = P.drop_site;
    uint32_t key = 0;
    if (site) key = m2f_site_key(gb.rng, site);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
            const float bv = (bias && col < N) ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < M && col < N) {
                    float v = acc[i][j][r] + bv;
                    if (relu_out) v = fmaxf(v, 0.f);
                    if (site) v = m2f_keep(key, (uint32_t)row * (uint32_t)N + (uint32_t)col, gb.drop_thresh) ? v * gb.drop_scale : 0.f;
                    if (res) v += res[(size_t)row * ldres + col];
                    if (gate) v = gate[(size_t)row * ldgate + col] > 0.f ? v * gscale : 0.f;
                    float* dst = C + (size_t)row * ldc + col;
                    if (accum) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

template <int PREC, bool A_RC, bool B_RC, int BM, int BN>
hipError_t launch_cfg(const GemmBatch& gb, int total_tiles, hipStream_t stream) {
    using SA = Stage<PREC, A_RC, BM>;
    using SB = Stage<PREC, B_RC, BN>;
    constexpr int lds = 2 * SA::LDS_BYTES + 2 * SB::LDS_BYTES;
    auto kern = m2f_gemm_kernel<PREC, A_RC, B_RC, BM, BN>;
    if (lds > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(total_tiles), dim3(256), lds, stream, gb);
    return hipGetLastError();
}

template <int PREC, bool A_RC, bool B_RC>
hipError_t launch_tile(GemmBatch& gb, int tile, hipStream_t stream) {
    auto count_tiles = [&](int bm, int bn) {
        int t = 0;
        for (int i = 0; i < gb.count; ++i) t += m2f_cdiv(gb.pr[i].M, bm) * m2f_cdiv(gb.pr[i].N, bn);
        return t;
    };
    if (tile == 0) tile = (count_tiles(128, 128) >= 512) ? 128 : 64;     // fill 256 CUs first
    const int bm = tile, bn = tile;
    int t = 0;
    for (int i = 0; i < gb.count; ++i) {
        gb.pr[i].tile_begin = t;
        gb.pr[i].tiles_n = m2f_cdiv(gb.pr[i].N, bn);
        t += m2f_cdiv(gb.pr[i].M, bm) * gb.pr[i].tiles_n;
    }
    if (t == 0) return hipSuccess;
    if (tile == 128) return launch_cfg<PREC, A_RC, B_RC, 128, 128>(gb, t, stream);
    return launch_cfg<PREC, A_RC, B_RC, 64, 64>(gb, t, stream);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool vec_ok(const GemmOperand& o, bool rc, int rows) {
    for (int s = 0; s < 2; ++s) {
        if (o.k[s] == 0) continue;
        if (!aligned16(o.p[s]) || (o.ld[s] & 3)) return false;
        if (!rc && (o.k[s] & 3)) return false;
    }
    if (rc && (rows & 3)) return false;
    return true;
}

}  // namespace

hipError_t m2f_launch_gemm(GemmBatch& gb, int prec, int layout, int tile, hipStream_t stream) {
    if (gb.count <= 0 || gb.count > M2F_GEMM_MAX_PROBLEMS) return hipErrorInvalidValue;
    const bool a_rc = layout == M2F_LAYOUT_TN;
    const bool b_rc = layout != M2F_LAYOUT_NT;
    for (int i = 0; i < gb.count; ++i) {
        GemmProblem& p = gb.pr[i];
        if (p.a.k[0] != p.b.k[0] || p.a.k[1] != p.b.k[1] || p.M <= 0 || p.N <= 0 || p.a.k[0] <= 0)
            return hipErrorInvalidValue;
        if (p.drop_site && !gb.rng) return hipErrorInvalidValue;
        p.flags &= ~(uint32_t)(GF_VEC_A | GF_VEC_B);
        if (vec_ok(p.a, a_rc, p.M)) p.flags |= GF_VEC_A;
        if (vec_ok(p.b, b_rc, p.N)) p.flags |= GF_VEC_B;
    }
    if (prec == M2F_PREC_F32) {
        if (layout == M2F_LAYOUT_NT) return launch_tile<M2F_PREC_F32, false, false>(gb, tile, stream);
        if (layout == M2F_LAYOUT_NN) return launch_tile<M2F_PREC_F32, false, true>(gb, tile, stream);
        return launch_tile<M2F_PREC_F32, true, true>(gb, tile, stream);
    }
    if (layout == M2F_LAYOUT_NT) return launch_tile<M2F_PREC_BF16, false, false>(gb, tile, stream);
    if (layout == M2F_LAYOUT_NN) return launch_tile<M2F_PREC_BF16, false, true>(gb, tile, stream);
    return launch_tile<M2F_PREC_BF16, true, true>(gb, tile, stream);
}
