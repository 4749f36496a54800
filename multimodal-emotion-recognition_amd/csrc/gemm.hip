// Grouped MFMA GEMM for the M2FNet hot path (gfx950 / CDNA4, wave64).
//
//   C[M,N] = epilogue( sum_k A(m,k) * B(n,k) )      (see ops.h for layouts / epilogue order)
//
// Replaces every nn.Linear / in-projection / out-projection call the reference's model makes
// (reference src/model.py:14,18,111-113,123-125,143 and torch's TransformerEncoderLayer internals)
// in forward, input-gradient and weight-gradient form.
//
// Design (MI355X) - two kernel families sharing one C/D layout and one fused epilogue:
//   * F32 mode (the 1e-3 parity mode) and the fallback for operands without bf16 shadows: `m2f_gemm_kernel`, fp32-source.
//     One workgroup = 4 wavefronts (2x2), each owning (BM/2)x(BN/2) of the tile as 32x32 blocks of
//     v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak) or, in BF16 mode, v_mfma_f32_32x32x16_bf16 with the operands
//     rounded to bf16 while being staged.  Register-staged operands (coalesced 16-byte loads one k-tile ahead), LDS images
//     laid out for conflict-free fragment reads ([k][row] floats with row stride BR+1 | [row][k] bf16 with 272-byte rows).
//   * BF16 mode proper: `m2f_gemm16_kernel` (+ `_dense_`, `_table_` variants), bf16-SOURCE: every operand also exists as a
//     bf16 shadow in HBM (half the bytes through the per-CU L1 fill path that bounds these GEMMs).  Workgroups of 8 waves
//     split into 4 producer waves (global -> register ring of D k-tiles -> LDS) and 4 consumer waves (LDS -> MFMA ->
//     epilogue); tiles 64x64 (chain launches), 128x128 (weight gradients, persistent walk over a device-resident problem
//     table) or 256x128 (text encoder); see the comment above gemm16_body.
//   * "grouped": one launch covers several independent problems (text+audio branch, q/k/v of a fusion layer, split
//     concat) so the 256 CUs see more workgroups per launch and the launch-latency-bound chain gets shorter;
//   * the epilogue fuses bias, ReLU / exact GELU, dropout, residual add, ReLU-gate and accumulate (32-bit offsets from
//     uniform bases, optional terms behind block-uniform branches) and writes the bf16 shadow of C; the row-contiguous
//     wgrad form also emits the bias gradient (column sums of dY) from the tiles it already streams.
#include "common.h"
#include "ops.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <cstring>

// No floating-point contraction in this file: several kernels restate the same formulas in different surroundings (ring and
// register-staged GEMM epilogues, fp32 and bf16 attention forms, flat and shadow-writing Adam), and with -ffp-contract=fast (the
// HIP default) the compiler is free to fuse a*b+c in one of them and not in the other - a 1-ulp difference that would break the
// bit-for-bit comparisons the tests make between the forms.  These kernels are bound by memory or by MFMA, not by VALU multiplies.
#pragma clang fp contract(off)

// Experiment knobs (register sets in flight).  NOTE: builds that spill VGPRs are NOT safe with the inline-asm staging
// loads (see Stage16KC::wait_loaded): D = 3 of the 128-VGPR chain build spills and faults.
#ifndef M2F_CHAIN_D
#define M2F_CHAIN_D 2         // register sets in flight of the 64x64 chain build (experiment knob)
#endif
#ifndef M2F_T256_D
#define M2F_T256_D 2          // register sets in flight of the 256x128 table build (experiment knob)
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------
// Operand staging: global (fp32) -> registers -> LDS image.
//   KC operand: element (row, k) at p[row*ld + k];  RC operand: element (row, k) at p[k*ld + row].
// The staging code is the instruction-issue bottleneck of these small-M GEMMs (measured: ~2 us per k-tile when
// every element carried its own bounds test and 64-bit address), so:
//   * per-thread element offsets are computed ONCE per workgroup (rows clamped into range, 32-bit), a k-tile only
//     adds a wave-uniform base;
//   * interior tiles (the common case, wave-uniform test) take a path with no masking at all; only edge tiles pay
//     for per-element predicates;
//   * every global load is unconditional (edge lanes read a clamped in-range address and are zeroed at store time):
//     hipcc turns a guarded load into a branch + s_waitcnt vmcnt(0) per load, which serialises the tile fetch
//     (cdna_hip_programming.md, ".s-level traps" (c));
//   * nothing consumes a loaded value before store(), so both operands' loads of a k-tile stay in flight while
//     earlier tiles are multiplied.
// ---------------------------------------------------------------------------------------------------------
template <int PREC, bool RC, int BR, int BK_, bool VEC>
struct Stage {
    static constexpr int BK = BK_;
    static constexpr bool F32 = (PREC == M2F_PREC_F32);
    // floats per task: F32 4 (one float4) | BF16-KC 8 (8 consecutive k) | BF16-RC 32 (8 k x 4 rows patch)
    static constexpr int FPT = F32 ? 4 : (RC ? 32 : 8);
    static constexpr int NT = BR * BK / FPT / 256;                       // staging tasks per thread
    static_assert(NT >= 1 && (BR * BK) % (FPT * 256) == 0, "tile must split evenly over 256 threads");
    static constexpr int KCH = BK / (F32 ? 4 : 8);                       // KC: chunks along k per row
    static constexpr int LDR = BR + 1;                                   // F32 LDS row stride (floats), image [k][row]
    static constexpr int ROWB = BK * 2 + 16;                             // BF16 LDS row stride (bytes), image [row][k]
    static constexpr int LDS_BYTES = F32 ? BK * LDR * 4 : BR * ROWB;

    float v[NT][FPT];
    int off[NT];             // element offset of task t at kbase = 0 (row clamped into range)
    int ld_;                 // leading dimension of the current segment
    int rows_, row0_, kseg_, kbase_;
    bool full_;              // tile in flight needs no masking (wave-uniform)
    int seg_ = -1;           // operand segment the offsets were set up for

    // task decomposition (t-th task of this thread)
    __device__ __forceinline__ static void task_rc(int id, int& r, int& k) {
        if constexpr (!RC) { r = id / KCH; k = (F32 ? 4 : 8) * (id % KCH); }           // KC: row r, k offset
        else if constexpr (F32) { constexpr int CH = BR / 4; k = id / CH; r = 4 * (id % CH); }   // float4 along rows
        else { constexpr int CH = BR / 4; k = 8 * (id / CH); r = 4 * (id % CH); }       // 8 k x 4 rows patch
    }

    __device__ __forceinline__ void setup(int ld, int rows, int row0, int tid) {
        ld_ = ld; rows_ = rows; row0_ = row0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int r, k;
            task_rc(tid + 256 * t, r, k);
            int gr = row0 + r;
            // clamp so that every (vector) load stays inside the operand; clamped lanes are masked at store time
            const int last = VEC && RC ? rows - 4 : rows - 1;
            gr = gr < last ? gr : (last > 0 ? last : 0);
            off[t] = RC ? k * ld + gr : gr * ld + k;
        }
    }

    template <bool FULL>
    __device__ __forceinline__ void issue_impl(const float* __restrict__ pt) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int r, k;
            task_rc(threadIdx.x + 256 * t, r, k);
            if constexpr (!RC) {
                constexpr int NV = FPT / 4;
#pragma unroll
                for (int h = 0; h < NV; ++h) {
                    if constexpr (VEC) {
                        int o = off[t] + 4 * h;
                        if constexpr (!FULL) o = (kbase_ + k + 4 * h < kseg_) ? o : off[t] - k - kbase_;   // row start (k = 0)
                        const f32x4 x = *reinterpret_cast<const f32x4*>(pt + o);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[t][4 * h + e] = x[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            int o = off[t] + 4 * h + e;
                            if constexpr (!FULL) o = (kbase_ + k + 4 * h + e < kseg_) ? o : off[t] - k - kbase_;
                            v[t][4 * h + e] = pt[o];
                        }
                    }
                }
            } else {
                constexpr int NK = F32 ? 1 : 8;                       // k rows of the patch
#pragma unroll
                for (int j = 0; j < NK; ++j) {
                    int o = off[t] + j * ld_;
                    if constexpr (!FULL) o = (kbase_ + k + j < kseg_) ? o : off[t] - (k + kbase_) * ld_;    // k = 0 row
                    if constexpr (VEC) {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(pt + o);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[t][4 * j + e] = x[e];
                    } else {
                        // scalar path: clamp each element's row separately (off[] holds the clamped first row)
                        int grc = row0_ + r;
                        grc = grc < rows_ - 1 ? grc : rows_ - 1;
                        const int room = rows_ - 1 - grc;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[t][4 * j + e] = pt[o + (e < room ? e : room)];
                    }
                }
            }
        }
    }

    // Issue the loads of the tile at (kbase) of segment pointer p.  kseg = 0 marks a tile past the end (all masked).
    __device__ __forceinline__ void issue(const float* __restrict__ p, int kseg, int kbase) {
        kseg_ = kseg; kbase_ = kbase;
        full_ = (row0_ + BR <= rows_) && (kbase + BK <= kseg);
        const float* pt = p + (RC ? (size_t)kbase * ld_ : (size_t)kbase);
        if (full_) issue_impl<true>(pt);
        else issue_impl<false>(pt);
    }

    __device__ __forceinline__ bool elem_ok(int r, int k) const { return row0_ + r < rows_ && kbase_ + k < kseg_; }

    // Mask (edge tiles only), optional ReLU, optional column sums (wgrad bias gradient), convert, write LDS.
    template <bool FULL>
    __device__ __forceinline__ void store_impl(char* lds, bool relu, bool do_cs, f32x4& cs) {
        if constexpr (!FULL) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                int r, k;
                task_rc(threadIdx.x + 256 * t, r, k);
#pragma unroll
                for (int e = 0; e < FPT; ++e) {
                    bool ok;
                    if constexpr (!RC) ok = elem_ok(r, k + e);
                    else ok = elem_ok(r + (e & 3), k + (e >> 2));
                    v[t][e] = ok ? v[t][e] : 0.f;
                }
            }
        }
        if (relu) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < FPT; ++e) v[t][e] = fmaxf(v[t][e], 0.f);
        }
        if constexpr (RC) {
            if (do_cs) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < FPT; ++e) cs[e & 3] += v[t][e];
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int r, k;
            task_rc(threadIdx.x + 256 * t, r, k);
            if constexpr (F32 && !RC) {
                float* f = reinterpret_cast<float*>(lds);
#pragma unroll
                for (int e = 0; e < 4; ++e) f[(k + e) * LDR + r] = v[t][e];
            } else if constexpr (F32 && RC) {
                float* f = reinterpret_cast<float*>(lds);
#pragma unroll
                for (int e = 0; e < 4; ++e) f[k * LDR + r + e] = v[t][e];
            } else if constexpr (!RC) {
                bf16x8 h;
#pragma unroll
                for (int e = 0; e < 8; ++e) h[e] = (__bf16)v[t][e];
                *reinterpret_cast<bf16x8*>(lds + r * ROWB + k * 2) = h;
            } else {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {                     // transpose the 8k x 4rows patch: one b128 per row
                    bf16x8 h;
#pragma unroll
                    for (int j = 0; j < 8; ++j) h[j] = (__bf16)v[t][4 * j + rr];
                    *reinterpret_cast<bf16x8*>(lds + (r + rr) * ROWB + k * 2) = h;
                }
            }
        }
    }
    __device__ __forceinline__ void store(char* lds, bool relu, bool do_cs, f32x4& cs) {
        if (full_) store_impl<true>(lds, relu, do_cs, cs);
        else store_impl<false>(lds, relu, do_cs, cs);
    }
};


// Fused epilogue shared by the fp32-source and bf16-source kernels:
//   +bias[n] -> relu -> dropout(site) -> +res[m,n] -> *(gate[m,n] > 0 ? gate_scale : 0) -> (C += | C =) [-> bf16 shadow of C]
// Side loads are unconditional (clamped) and issued before the arithmetic; stores are predicated.
// GELU is a COMPILE-TIME variant: carrying the erf expansion (and its constants) in every instantiation doubled the SGPR
// spills of the chain kernels (24 -> 52) and cost the M2FNet step 3.4 %; only the text encoder's GEMMs use it.
// (m2f_gelu: common.h)
template <int MI, int NI, int BM, int BN, bool GELU = false, bool SCALE = false, bool GELU_POLY = false>
__device__ __forceinline__ void gemm_epilogue(const GemmBatch& gb, const GemmProblem& P, f32x16 (&acc)[MI][NI], int m0, int n0,
                                              int lane, int wm, int wn) {
    const int M = P.M, N = P.N;
    const uint32_t flags = P.flags;
    const float* __restrict__ bias = P.bias;
    const char* __restrict__ res = reinterpret_cast<const char*>(P.res);
    const char* __restrict__ gate = reinterpret_cast<const char*>(P.gate);
    char* __restrict__ C = reinterpret_cast<char*>(P.c);
    char* __restrict__ C16 = reinterpret_cast<char*>(m2f_shadow_of(gb.sh, P.c));
    const int ldc = P.ldc, ldres = P.ldres, ldgate = P.ldgate;
    const float gscale = P.gate_scale;
    const bool relu_out = flags & GF_RELU_OUT, accum = flags & GF_ACCUM;
    const uint32_t site = P.drop_site;
    uint32_t key = 0;
    if (site) key = m2f_site_key(gb.rng, site);
    // Everything below is indexed with 32-bit element offsets from the (uniform) base pointers - the saddr form of
    // global_load / global_store - and every optional term sits behind ONE block-uniform branch per 32x32 block: the
    // per-element 64-bit multiplies and branches of the first version cost 2 us (64x64 tile) to 7 us (128x128) per
    // workgroup, more than the k-loop of these small-K problems.
    const bool interior = (m0 + BM <= M) && (n0 + BN <= N);         // block-uniform: no clamping / predicates needed
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
            const bool col_ok = col < N;
            const int colc = col_ok ? col : N - 1;
            const int row_base = m0 + wm * (BM / 2) + i * 32 + 4 * (lane >> 5);
            const float bv = bias ? bias[colc] : 0.f;
            // row of element r: row_base + (r & 3) + 8 * (r >> 2)
            uint32_t oc[16], orr[16], og[16];
            bool ok[16];
            if (interior) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    oc[r] = (uint32_t)((row_base + dr) * ldc + col);
                    ok[r] = true;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_base + (r & 3) + 8 * (r >> 2);
                    const int rowc = row < M ? row : M - 1;
                    oc[r] = (uint32_t)(rowc * ldc + colc);
                    ok[r] = row < M && col_ok;
                }
            }
            float rv[16], gv[16], cv[16];
            if (res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_base + (r & 3) + 8 * (r >> 2);
                    orr[r] = (uint32_t)((interior ? row : (row < M ? row : M - 1)) * ldres + colc);
                    rv[r] = *reinterpret_cast<const float*>(res + (size_t)(orr[r] * 4u));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = 0.f;
            }
            if (gate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_base + (r & 3) + 8 * (r >> 2);
                    og[r] = (uint32_t)((interior ? row : (row < M ? row : M - 1)) * ldgate + colc);
                    gv[r] = *reinterpret_cast<const float*>(gate + (size_t)(og[r] * 4u));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) gv[r] = 1.f;
            }
            if (accum) {
#pragma unroll
                for (int r = 0; r < 16; ++r) cv[r] = *reinterpret_cast<const float*>(C + (size_t)(oc[r] * 4u));
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) cv[r] = 0.f;
            }
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = (SCALE ? acc[i][j][r] * P.acc_scale : acc[i][j][r]) + bv;
                if (relu_out) x = fmaxf(x, 0.f);
                v[r] = x;
            }
            if constexpr (GELU) {
                if (flags & GF_GELU_OUT) {                      // block-uniform
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = m2f_gelu<GELU_POLY>(v[r]);
                }
            }
            if (site) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_base + (r & 3) + 8 * (r >> 2);
                    v[r] = m2f_keep(key, (uint32_t)row * (uint32_t)N + (uint32_t)col, gb.drop_thresh) ? v[r] * gb.drop_scale : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = v[r] + rv[r];
                if (gate) x = gv[r] > 0.f ? x * gscale : 0.f;
                v[r] = x + cv[r];
            }
            if constexpr (SCALE) {
                if (P.c8) {                                     // block-uniform: e4m3 result instead of fp32 C
                    const float s8 = P.c8_scale;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (ok[r]) P.c8[oc[r]] = (uint8_t)(m2f_fp8x4_bits(v[r] * s8, 0.f, 0.f, 0.f) & 0xFFu);
                    continue;
                }
            }
            if (interior) {
#pragma unroll
                for (int r = 0; r < 16; ++r) *reinterpret_cast<float*>(C + (size_t)(oc[r] * 4u)) = v[r];
                if (C16) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) *reinterpret_cast<uint16_t*>(C16 + (size_t)(oc[r] * 2u)) = m2f_bf16_bits(v[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (ok[r]) {
                        *reinterpret_cast<float*>(C + (size_t)(oc[r] * 4u)) = v[r];
                        if (C16) *reinterpret_cast<uint16_t*>(C16 + (size_t)(oc[r] * 2u)) = m2f_bf16_bits(v[r]);
                    }
            }
        }
    }
}

// XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8, each with its own
// L2), so consecutive blocks never share an L2.  Remap b -> pos so that every XCD walks a CONTIGUOUS range of the
// launch's tile list, and order tiles with the M index fastest: an XCD then owns whole column panels of B (the weight,
// cold in HBM), which are fetched into one L2 once instead of into all eight.  Bijective for any grid size; a different
// hardware placement only changes speed, never results.
__device__ __forceinline__ int xcd_remap(int b, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, x = b & 7, j = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// LDS-only workgroup barrier: wait for this wave's LDS traffic, then s_barrier.  __syncthreads() carries a
// workgroup-scope fence for which hipcc also drains vmcnt(0), i.e. the prefetched global loads of the next
// k-tiles - exactly the loads that must stay in flight across the barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int PREC, bool A_RC, bool B_RC, int BM, int BN, int BK, bool VEC, bool GELU = false>
__global__ __launch_bounds__(256) void m2f_gemm_kernel(const GemmBatch gb) {
    using SA = Stage<PREC, A_RC, BM, BK, VEC>;
    using SB = Stage<PREC, B_RC, BN, BK, VEC>;
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
    // split-K: `splitk` consecutive workgroups share one output tile, each reduces a slice of the k-tile list
    const int S = P.splitk;
    const int tb = (int)blockIdx.x - P.tile_begin;
    const int tl = tb / S, slice = tb - tl * S;
    const int m0 = (tl / P.tiles_n) * BM, n0 = (tl % P.tiles_n) * BN;
    const uint32_t flags = P.flags;
    const bool reluA = flags & GF_RELU_A, reluB = flags & GF_RELU_B;
    const int k0 = P.a.k[0], k1 = P.a.k[1];
    const int nk0 = (k0 + BK - 1) / BK, nk_all = nk0 + (k1 + BK - 1) / BK;
    const int kt_begin = (slice * nk_all) / S, nk = ((slice + 1) * nk_all) / S - kt_begin;   // this slice: nk k-tiles

    char* ldsA = smem;
    char* ldsB = smem + 2 * SA::LDS_BYTES;

    // Two register sets: the global loads of k-tile t+2 / t+3 are in flight while tile t is multiplied, so a
    // workgroup keeps ~2 tiles of HBM/L2 traffic outstanding (these GEMMs are latency-bound, M = B*L is small).
    SA sa0, sa1;
    SB sb0, sb1;
    f32x4 colsum = {0.f, 0.f, 0.f, 0.f}, cs_unused = {0.f, 0.f, 0.f, 0.f};
    const bool want_bg = A_RC && P.bias_grad != nullptr && n0 == 0;

    // UNCONDITIONAL: a k-tile index past the end issues the same number of loads (all masked to zero at store
    // time).  A branch around the prefetch would make the number of loads in flight path-dependent, and hipcc then
    // drains vmcnt to 0 at every store phase (seen in the .s) - no overlap left.
    auto load_tile = [&](SA& sa, SB& sb, int kt_raw) {
        const bool tv = kt_raw < nk;
        const int kt = tv ? kt_begin + kt_raw : 0;
        const int seg = kt >= nk0 ? 1 : 0;
        const int kbase = (seg ? kt - nk0 : kt) * BK;
        if (sa.seg_ != seg) {                                  // wave-uniform, at most twice per workgroup
            sa.setup(P.a.ld[seg], M, m0, tid); sa.seg_ = seg;
            sb.setup(P.b.ld[seg], N, n0, tid); sb.seg_ = seg;
        }
        sa.issue(P.a.p[seg], tv ? P.a.k[seg] : 0, kbase);
        sb.issue(P.b.p[seg], tv ? P.b.k[seg] : 0, kbase);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int cur) {
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
    };

    load_tile(sa0, sb0, 0);
    load_tile(sa1, sb1, 1);
    sa0.store(ldsA, reluA, want_bg, colsum);
    sb0.store(ldsB, reluB, false, cs_unused);
    lds_barrier();
    load_tile(sa0, sb0, 2);

    for (int kt = 0; kt < nk; kt += 2) {
        // even phase: multiply tile kt (buffer 0); stage tile kt+1 (register set 1) into buffer 1
        compute(0);
        // (stores are unconditional too: past the last tile they write the all-zero masked tile to an unused buffer)
        sa1.store(ldsA + SA::LDS_BYTES, reluA, want_bg, colsum);
        sb1.store(ldsB + SB::LDS_BYTES, reluB, false, cs_unused);
        lds_barrier();
        load_tile(sa1, sb1, kt + 3);
        if (kt + 1 >= nk) break;
        // odd phase: multiply tile kt+1 (buffer 1); stage tile kt+2 (register set 0) into buffer 0
        compute(1);
        sa0.store(ldsA, reluA, want_bg, colsum);
        sb0.store(ldsB, reluB, false, cs_unused);
        lds_barrier();
        load_tile(sa0, sb0, kt + 4);
    }

    // ---- wgrad: bias gradient = column sums of the A operand (dY) over the reduction dim ----------
    if constexpr (A_RC) {
        if (want_bg) {      // block-uniform
            float* red = reinterpret_cast<float*>(smem);
            constexpr int CH = BM / 4, G = 256 / CH;       // thread = (k-group g, 4 rows 4*rc..): same rows for all its tasks
            const int g = tid / CH, rc = tid % CH;
#pragma unroll
            for (int e = 0; e < 4; ++e) red[g * BM + 4 * rc + e] = colsum[e];
            __syncthreads();
            if (tid < BM) {
                float sum = 0.f;
                for (int q = 0; q < G; ++q) sum += red[q * BM + tid];
                if (m0 + tid < M) P.bias_grad[m0 + tid] = sum;
            }
        }
    }

    // ---- split-K: publish the partial tile; the LAST arriving slice sums all partials (fixed order -> bitwise
    // reproducible) and runs the epilogue.  Agent-scope release/acquire around a relaxed ticket, valid for any
    // placement of the slices over CUs / XCDs (cdna_hip_programming.md, "In-launch split-K reduction").
    if (S > 1) {
        static_assert(MI * NI * 16 * 256 == BM * BN, "slab layout");
        float* slab = gb.splitk_ws + (size_t)(P.slab_begin + tl * S) * (BM * BN);
        float* mine = slab + (size_t)slice * (BM * BN);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[((i * NI + j) * 16 + r) * 256 + tid] = acc[i][j][r];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(gb.splitk_cnt + P.cnt_begin + tl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = (t == (unsigned)(S - 1)) ? 1 : 0;
        }
        __syncthreads();
        if (*flag == 0) return;                              // block-uniform
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(gb.splitk_cnt + P.cnt_begin + tl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float sum = 0.f;
                    for (int q = 0; q < S; ++q) sum += slab[(size_t)q * (BM * BN) + ((i * NI + j) * 16 + r) * 256 + tid];
                    acc[i][j][r] = sum;
                }
    }

    gemm_epilogue<MI, NI, BM, BN, GELU>(gb, P, acc, m0, n0, lane, wm, wn);
}

// =========================================================================================================
// bf16-SOURCE kernel (bf16 mode): operands are read from the bf16 shadows (half the bytes through the per-CU load
// pipe, which is what bounds these GEMMs), 16 bytes = 8 elements per load, no conversion, D k-tiles in flight.
// Pad columns of every shadow are zero, so a logical width that is not a multiple of 8 needs no element masks.
// =========================================================================================================
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) {
    const uint32_t m = ((v >> 15) & 0x00010001u) * 0xFFFFu;      // 0xFFFF in every half whose sign bit is set
    return v & ~m;
}
__device__ __forceinline__ float bf16lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf16hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xFFFF0000u); }

// 16-byte load through an explicitly GLOBAL pointer (a native vector type: HIP's uint4 is a class whose copy goes through a
// generic reference and loses the address space again)
typedef unsigned int m2f_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16_global(const void* p) {
    const m2f_u32x4 r = *(const __attribute__((address_space(1))) m2f_u32x4*)p;
    return make_uint4(r.x, r.y, r.z, r.w);
}

template <int BR, int BK>
struct Stage16KC {                       // element (row, k) at q[row*ld + k]
    static constexpr int KCH = BK / 8, NT = BR * KCH / 256;
    static constexpr int ROWB = BK * 2 + 16, LDS_BYTES = BR * ROWB;
    static_assert(NT >= 1 && (BR * KCH) % 256 == 0, "tile split");
    m2f_u32x4 v[NT];
    uint32_t off[NT];                    // BYTE offset of chunk t from the row-panel base at kbase = 0 (row clamped)
    int rows_, row0_, kpad_, kbase_, tid_;
    bool full_;
    __device__ __forceinline__ void setup(int ld, int rows, int row0, int tid) {
        rows_ = rows; row0_ = row0; tid_ = tid;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int id = tid + 256 * t, r = id / KCH, kc = id % KCH;
            int gr = row0 + r;
            gr = gr < rows - 1 ? gr : rows - 1;
            off[t] = 2u * (uint32_t)(gr * ld + 8 * kc);
        }
    }
    // uniform 64-bit base + zero-extended 32-bit lane offset: the saddr form of global_load (no per-load 64-bit VALU add)
    // The address-space cast matters for the table kernel: its operand pointers are read from device memory, so the compiler
    // cannot prove them global and emits FLAT loads - which also count on lgkmcnt, share ordering with the ds_writes and
    // turn every counted wait of the ring into vmcnt(0) lgkmcnt(0).
    __device__ __forceinline__ static m2f_u32x4 ld16(const char* base, uint32_t o) {
        return *(const __attribute__((address_space(1))) m2f_u32x4*)(base + (size_t)o);
    }
    template <bool ASM = false>
    __device__ __forceinline__ void issue(const uint16_t* __restrict__ q, int kseg, int kbase) {
        kpad_ = (kseg + 7) & ~7; kbase_ = kbase;
        full_ = (row0_ + BR <= rows_) && (kbase + BK <= kpad_);
        const char* pt = reinterpret_cast<const char*>(q + kbase);
        // ONE straight-line path for interior and edge tiles (selects, no branch): with the loads of a set split over two
        // branches the waitcnt pass loses count at the merge and every wait of the ring becomes vmcnt(0)
        static_assert(256 % KCH == 0, "chunk column is the same for every t");
        const int kc = tid_ % KCH;
        const uint32_t back = (kbase + 8 * kc < kpad_) ? 0u : 16u * kc;     // past the padded width: this tile's first chunk
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            uint32_t o = off[t] - back;
            o = kseg == 0 ? 0u : o;               // dead tile past the end of the k-loop: every lane reads the same 16 bytes
            if constexpr (ASM) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[t]) : "v"(o), "s"(pt) : "memory");
            else v[t] = ld16(pt, o);
        }
    }
    // ASM issue: the compiler does not know these registers are in flight, so the caller waits by hand - `newer` = loads
    // issued after the last one of this set.  The registers are tied through the statement so that no use can be scheduled
    // above it.
    template <int NEWER>
    __device__ __forceinline__ void wait_loaded() {
        static_assert(NEWER >= 0 && NEWER <= 63 && NT <= 16, "vmcnt immediate / operand count");
        if constexpr (NT == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v[0]) : "n"(NEWER) : "memory");
        else if constexpr (NT == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(v[0]), "+v"(v[1]) : "n"(NEWER) : "memory");
        else if constexpr (NT == 4)
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "n"(NEWER) : "memory");
        else if constexpr (NT == 8)
            asm volatile("s_waitcnt vmcnt(%8)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]),
                         "+v"(v[7]) : "n"(NEWER) : "memory");
        else static_assert(NT == 1 || NT == 2 || NT == 4 || NT == 8, "chunks per thread");
    }
    // Branch-free variant for the table kernel (its operands never carry a ReLU flag): edge chunks are zeroed with selects,
    // so the registers of a set are only touched in straight-line code and the ring keeps counted vmcnt waits.
    __device__ __forceinline__ void store_select(char* lds) {
        const int kc = tid_ % KCH;
        const bool kok = full_ || (kbase_ + 8 * kc < kpad_);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int r = (tid_ + 256 * t) / KCH;
            const bool ok = kok && (full_ || row0_ + r < rows_);
            m2f_u32x4 x = v[t];
            x.x = ok ? x.x : 0u; x.y = ok ? x.y : 0u; x.z = ok ? x.z : 0u; x.w = ok ? x.w : 0u;
            *reinterpret_cast<m2f_u32x4*>(lds + r * ROWB + kc * 16) = x;
        }
    }
    __device__ __forceinline__ void store(char* lds, bool relu) {
        // the rare fix-ups (edge masks, relu of the operand) sit behind ONE block-uniform branch, so the common path
        // is eight back-to-back ds_write_b128 with progressive vmcnt waits and no control flow in between
        if (!full_ || relu) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int id = tid_ + 256 * t, r = id / KCH, kc = id % KCH;
                m2f_u32x4 x = v[t];
                const bool ok = full_ || ((row0_ + r < rows_) && (kbase_ + 8 * kc < kpad_));
                if (!ok) x = (m2f_u32x4){0u, 0u, 0u, 0u};
                if (relu) { x.x = relu_bf16x2(x.x); x.y = relu_bf16x2(x.y); x.z = relu_bf16x2(x.z); x.w = relu_bf16x2(x.w); }
                v[t] = x;
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int id = tid_ + 256 * t, r = id / KCH, kc = id % KCH;
            *reinterpret_cast<m2f_u32x4*>(lds + r * ROWB + kc * 16) = v[t];
        }
    }
};

template <int BR, int BK>
struct Stage16RC {   // element (row, k) at q[k*ld + row]; one 8(k) x 8(row) patch per active thread
    // LDS image is K-MAJOR: [k][row] bf16 with row stride ROWK = 2*BR + 64 bytes, written exactly as loaded (16-byte
    // vectors, lanes = consecutive row groups -> conflict-free) and consumed with ds_read_b64_tr_b16, the gfx950
    // transposing LDS read (4 k-rows x 16 rows per 16-lane group; with this stride the 4 k-rows of a group fall on
    // disjoint bank ranges).  No register transposes, no scattered stores.
    static constexpr int KP = BK / 8, RP = BR / 8, NPATCH = KP * RP;
    static constexpr int ROWK = 2 * BR + 16, LDS_BYTES = BK * ROWK;   // +16: 2-way conflicts on the tr reads, but 2 workgroups fit a CU
    uint4 v[8];
    int off, kp_, rp_;
    int ld_, rows_, row0_, kseg_, kbase_;
    bool full_, active_;
    __device__ __forceinline__ void setup(int ld, int rows, int row0, int pid) {
        static_assert(NPATCH == 128, "one patch per thread of a 128-thread half");
        ld_ = ld; rows_ = rows; row0_ = row0;
        active_ = pid >= 0 && pid < NPATCH;
        const int p = active_ ? pid : 0;
        kp_ = p / RP; rp_ = p % RP;
        const int rpad = (rows + 7) & ~7;
        int gr = row0 + 8 * rp_;
        gr = gr < rpad - 8 ? gr : rpad - 8;
        off = 8 * kp_ * ld + (gr > 0 ? gr : 0);
    }
    __device__ __forceinline__ void issue(const uint16_t* __restrict__ q, int kseg, int kbase) {
        kseg_ = kseg; kbase_ = kbase;
        full_ = (row0_ + BR <= rows_) && (kbase + BK <= kseg);
        if (!active_) return;                                   // wave-uniform (patches are dealt per 128-thread half)
        const uint16_t* pt = q + (size_t)kbase * ld_;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int o = off + j * ld_;
            if (!full_) o = (kbase + 8 * kp_ + j < kseg) ? o : off - 8 * kp_ * ld_ - kbase * ld_;   // else: k = 0 row
            if (kseg == 0) o = 0;                 // dead tile past the end of the k-loop: one broadcast line
            v[j] = ld16_global(pt + o);
        }
    }
    __device__ __forceinline__ void store(char* lds, bool relu, bool do_cs, float (&cs)[8]) {
        if (!active_) return;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
            if (!full_) {
                const bool kok = kbase_ + 8 * kp_ + j < kseg_;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int r = row0_ + 8 * rp_ + 2 * m;
                    uint32_t keep = 0u;
                    if (kok && r < rows_) keep |= 0x0000FFFFu;
                    if (kok && r + 1 < rows_) keep |= 0xFFFF0000u;
                    w[m] &= keep;
                }
            }
            if (relu) {
#pragma unroll
                for (int m = 0; m < 4; ++m) w[m] = relu_bf16x2(w[m]);
            }
            if (do_cs) {
#pragma unroll
                for (int m = 0; m < 4; ++m) { cs[2 * m] += bf16lo(w[m]); cs[2 * m + 1] += bf16hi(w[m]); }
            }
            *reinterpret_cast<uint4*>(lds + (8 * kp_ + j) * ROWK + 16 * rp_) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
};

typedef short m2f_s16x4 __attribute__((ext_vector_type(4)));
// MFMA 32x32x16 operand fragment (8 consecutive k of row `row`) out of a K-major LDS image via two transposing reads.
// Per 16-lane group g: lane 4q+p supplies the address of k-row (4t + q), rows 4p..4p+3; lane i receives row i.
__device__ __forceinline__ bf16x8 frag_from_kmajor(const char* img, int rowk, int row_base32, int k0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int kq = k0 + 8 * (g >> 1) + q;                       // + 4t
    const int col = row_base32 + 16 * (g & 1) + 4 * p;
    const char* a0 = img + kq * rowk + col * 2;
    const m2f_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (m2f_s16x4 __attribute__((address_space(3)))*)(const_cast<char*>(a0)));
    const m2f_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (m2f_s16x4 __attribute__((address_space(3)))*)(const_cast<char*>(a0 + 4 * rowk)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

// Workgroup = 8 waves in two ROLES (wave-specialised): waves 0-3 are CONSUMERS (LDS fragments -> MFMA -> epilogue,
// 2x2 over the tile), waves 4-7 are PRODUCERS (global -> registers, D k-tiles in flight -> LDS image).  With one
// workgroup per CU - all these small-M launches offer - a SIMD then holds one wave of each role, so the address
// arithmetic / ds_write stream of tile k+1 issues in the shadow of the ds_read + MFMA run of tile k instead of in front
// of it (measured before the split: 0.62 us per 128-wide k-tile, ~40 % of it instruction issue of one wave per SIMD,
// MFMA busy 13 %).  The two roles run separate loops with the same barrier count, so their register sets (the load ring
// vs accumulators + fragments) overlap in the allocation instead of adding up.
#ifdef M2F_EXP_TIMING
__device__ unsigned long long m2f_dbg[64];
#define M2F_TS(slot) do { if (blockIdx.x == 0 && (threadIdx.x & 255) == 0) m2f_dbg[(slot) + (threadIdx.x >= 256 ? 16 : 0)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define M2F_TS(slot) do {} while (0)
#endif
// FP8: the operands are OCP e4m3 bytes.  The producers are unchanged - a k-tile is BK byte PAIRS per row either way (the
// launcher passes k and ldq in byte pairs) - only the consumers differ: 2*BK/16 slices of v_mfma_f32_32x32x16_fp8_fp8 with
// 8-byte fragments, and the epilogue de-quantises the accumulator.
// Epilogue of the weight-gradient table kernel (plain stores: no bias / residual / gate / shadow).  The consumers multiply with
// the operands SWAPPED, so a lane holds C[row = lane & 31][8 * g + 4 * (lane >> 5) + 0..3] of its 32x32 block: four consecutive
// columns per register quad -> one 16-byte store instead of four 4-byte ones (16 instead of 64 store instructions per wave
// and 64x64 quadrant).
template <int MI, int NI, int BM, int BN>
__device__ __forceinline__ void table_epilogue_t(const GemmProblem& P, f32x16 (&acc)[MI][NI], int m0, int n0, int lane, int wm, int wn) {
    const int M = P.M, N = P.N, ldc = P.ldc;
    float* __restrict__ C = P.c;
    const bool vec = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) && (m0 + BM <= M) && (n0 + BN <= N);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int row = m0 + wm * (BM / 2) + i * 32 + (lane & 31);
            const int col0 = n0 + wn * (BN / 2) + j * 32 + 4 * (lane >> 5);
            if (vec) {                                              // block-uniform
                float* dst = C + (size_t)row * ldc + col0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(dst + 8 * g) = v;
                }
            } else if (row < M) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = col0 + 8 * (r >> 2) + (r & 3);
                    if (col < N) C[(size_t)row * ldc + col] = acc[i][j][r];
                }
            }
        }
    }
}

template <bool A_RC, bool B_RC, int BM, int BN, int BK, int D, bool TABLE, bool GELU = false, bool FP8 = false>
__device__ __forceinline__ void gemm16_body(const GemmBatch& gb) {
    static_assert(!(A_RC && !B_RC), "layouts: NT, NN, TN");
    static_assert(!FP8 || (!A_RC && !B_RC), "fp8: forward form only");
    constexpr bool TN = A_RC && B_RC;
    static_assert(!TN || BM == BN, "the TN stager shares one patch shape for both operands");
    using SAK = Stage16KC<BM, BK>;
    using SBK = Stage16KC<BN, BK>;
    using SBR = Stage16RC<BN, BK>;
    constexpr int ROWB = SAK::ROWB, ROWK = SBR::ROWK;
    constexpr int LDS_A = A_RC ? Stage16RC<BM, BK>::LDS_BYTES : SAK::LDS_BYTES;
    constexpr int LDS_B = B_RC ? SBR::LDS_BYTES : SBK::LDS_BYTES;
    constexpr int MI = BM / 64, NI = BN / 64;
    constexpr int U = (D % 2 == 0) ? D : 2 * D;                 // unroll so that set and LDS-buffer indices are static
    // NT form: the staging loads are issued from inline asm and waited for with hand-counted vmcnt (Stage16KC::wait_loaded).
    // hipcc's own waitcnt insertion answers the ring with vmcnt(0) as soon as the control flow around it is not trivial
    // (always in the table kernel: 3 sets, 6-fold unroll), which drains all sets in flight at every k-tile.
    constexpr bool ASM_LOADS = !A_RC && !B_RC;
    // table kernel: transposed accumulators + 16-byte stores (table_epilogue_t); its problems carry no epilogue terms
    constexpr bool TABLE_T = TABLE && !A_RC && !B_RC && !FP8 && !GELU;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const bool producer = threadIdx.x >= 256;                   // wave-uniform
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;      // role-local ids
    char* ldsA = smem;
    char* ldsB = smem + 2 * LDS_A;
    float* red = reinterpret_cast<float*>(smem + 2 * LDS_A + 2 * LDS_B);      // bias-grad partials: own region, so the
    constexpr int KP = BK / 8, RP = BM / 8;                                     // next tile's staging never overwrites them
    // PERSISTENT tile loop: a launch has at most chip-filling size; workgroup b walks tiles b, b + grid, ...  Both
    // roles see the same tile sequence and the same number of barriers per tile.  The producers of tile t+1 start as
    // soon as the last k-tile of t has been multiplied, i.e. they fetch under the consumers' epilogue of tile t.
    const int total_tiles = TABLE ? gb.total_tiles : (int)gridDim.x;
    M2F_TS(0);
  for (int bpos = xcd_remap((int)blockIdx.x, (int)gridDim.x); bpos < total_tiles; bpos += (int)gridDim.x) {
    int pi = 0;
    if constexpr (TABLE) pi = gb.tile_prob[bpos];
    else {
#pragma unroll
        for (int i = 1; i < M2F_GEMM_MAX_PROBLEMS; ++i)
            if (bpos >= gb.tb[i]) pi = i;                        // unused slots hold INT_MAX
    }
    const GemmProblem& P = TABLE ? gb.table[pi] : gb.pr[pi];     // epilogue terms (consumers, off the critical path)
    // descriptors the first loads depend on: from the compact header (grouped launches) or the device table
    GemmHot Hh;
    if constexpr (TABLE) {
        Hh.aq[0] = P.a.q[0]; Hh.aq[1] = P.a.q[1]; Hh.bq[0] = P.b.q[0]; Hh.bq[1] = P.b.q[1];
        Hh.M = P.M; Hh.N = P.N; Hh.k[0] = P.a.k[0]; Hh.k[1] = P.a.k[1];
        Hh.ldaq[0] = P.a.ldq[0]; Hh.ldaq[1] = P.a.ldq[1]; Hh.ldbq[0] = P.b.ldq[0]; Hh.ldbq[1] = P.b.ldq[1];
        Hh.flags = P.flags; Hh.tile_begin = P.tile_begin; Hh.has_bias_grad = P.bias_grad != nullptr;
    } else {
        Hh = gb.hot[pi];
    }
    const int M = Hh.M, N = Hh.N;
    const int tl = bpos - Hh.tile_begin;
    const int tiles_m = (M + BM - 1) / BM;
    const int m0 = (tl % tiles_m) * BM, n0 = (tl / tiles_m) * BN;       // M fastest: neighbours share the B panel
    const int nk0 = (Hh.k[0] + BK - 1) / BK, nk = nk0 + (Hh.k[1] + BK - 1) / BK;
    const bool has_bg = TN && Hh.has_bias_grad && n0 == 0;              // block-uniform

    if (producer) {
        // ================================ PRODUCER: global -> registers -> LDS ================================
        const uint32_t flags = Hh.flags;
        const bool reluA = flags & GF_RELU_A, reluB = flags & GF_RELU_B;
        struct Set {                   // one register set = this thread's share of one k-tile of both operands
            SAK ak;                    // NT, NN: A
            SBK bk;                    // NT: B
            SBR br;                    // NN: B (threads 0..127) | TN: A (threads 0..127) or B (threads 128..255)
        };
        Set sets[D];
        int cur_seg = -1;
        const bool second_half = tid >= 128;
        float colsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const bool want_bg = has_bg && !second_half;
        // operand descriptors in SGPRs for the whole k-loop: indexing P.a.q[seg] with a run-time seg makes the compiler
        // re-read the kernarg segment (s_load + lgkmcnt(0)) several times per k-tile, serialised in front of the loads
        const uint16_t* const aq0 = Hh.aq[0]; const uint16_t* const aq1 = Hh.aq[1];
        const uint16_t* const bq0 = Hh.bq[0]; const uint16_t* const bq1 = Hh.bq[1];
        const int ak0 = Hh.k[0], ak1 = Hh.k[1];
        const int ald0 = Hh.ldaq[0], ald1 = Hh.ldaq[1], bld0 = Hh.ldbq[0], bld1 = Hh.ldbq[1];
        auto setup_all = [&](int seg) {
            const int lda = seg ? ald1 : ald0, ldb = seg ? bld1 : bld0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                if constexpr (!A_RC) sets[i].ak.setup(lda, M, m0, tid);
                if constexpr (!B_RC) sets[i].bk.setup(ldb, N, n0, tid);
                if constexpr (!A_RC && B_RC) sets[i].br.setup(ldb, N, n0, tid);
                if constexpr (TN) {
                    if (second_half) sets[i].br.setup(ldb, N, n0, tid - 128);
                    else sets[i].br.setup(lda, M, m0, tid);
                }
            }
        };
        auto issue = [&](Set& st, int kt_raw) {                 // unconditional: past-the-end tiles are fully masked
            const bool tv = kt_raw < nk;
            const int kt = tv ? kt_raw : 0;
            const int seg = kt >= nk0 ? 1 : 0;
            const int kbase = (seg ? kt - nk0 : kt) * BK;
            if (seg != cur_seg) { setup_all(seg); cur_seg = seg; }   // wave-uniform, at most twice
            const int ks = tv ? (seg ? ak1 : ak0) : 0;
            const uint16_t* const qa = seg ? aq1 : aq0;
            const uint16_t* const qb = seg ? bq1 : bq0;
            if constexpr (ASM_LOADS) {
                st.ak.template issue<true>(qa, ks, kbase);
                st.bk.template issue<true>(qb, ks, kbase);
                return;
            }
            if constexpr (!A_RC) st.ak.issue(qa, ks, kbase);
            if constexpr (!B_RC) st.bk.issue(qb, ks, kbase);
            if constexpr (!A_RC && B_RC) st.br.issue(qb, ks, kbase);
            if constexpr (TN) st.br.issue(second_half ? qb : qa, ks, kbase);
        };
        auto store = [&](Set& st, int buf) {
            if constexpr (ASM_LOADS) {
                // every round issues one set and stores one: D - 1 whole sets are younger than the one stored now
                constexpr int PER_SET = SAK::NT + SBK::NT;
                st.ak.template wait_loaded<(D - 1) * PER_SET + SBK::NT>();
                st.bk.template wait_loaded<(D - 1) * PER_SET>();
            }
            if constexpr (TABLE && !A_RC && !B_RC) {
                st.ak.store_select(ldsA + buf * LDS_A);
                st.bk.store_select(ldsB + buf * LDS_B);
                return;
            }
            if constexpr (!A_RC) st.ak.store(ldsA + buf * LDS_A, reluA);
            if constexpr (!B_RC) st.bk.store(ldsB + buf * LDS_B, reluB);
            if constexpr (!A_RC && B_RC) { float dummy[8]; st.br.store(ldsB + buf * LDS_B, reluB, false, dummy); }
            if constexpr (TN) {
                if (second_half) { float dummy[8]; st.br.store(ldsB + buf * LDS_B, reluB, false, dummy); }
                else st.br.store(ldsA + buf * LDS_A, reluA, want_bg, colsum);
            }
        };

        M2F_TS(1);
#pragma unroll
        for (int i = 0; i < D; ++i) issue(sets[i], i);
        M2F_TS(2);
        store(sets[0], 0);
        M2F_TS(3);
        lds_barrier();                                          // (B0) tile 0 visible
        issue(sets[0], D);
        M2F_TS(4);
        // The loop is LEFT (not skipped through) when the k-tiles run out: a path that skips the rest of an unrolled round
        // and still reaches the loop header carries a different number of outstanding loads, and the waitcnt pass then
        // answers every wait at the header with vmcnt(0) - draining the ring once per round.
        for (int kt0 = 0;; kt0 += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kt = kt0 + u;
                if (kt >= nk) goto k_tiles_done;                 // block-uniform
                store(sets[(u + 1) % D], (u + 1) & 1);           // tile kt+1 (all-zero past the end) while tile kt is multiplied
                lds_barrier();                                   // (B1 per tile)
                issue(sets[(u + 1) % D], kt + 1 + D);
            }
        }
k_tiles_done:
        if constexpr (TN) {
            if (has_bg) {
                // threads 0..127 hold column sums of their 8 rows over their k-group: red[kp][row], then fixed-order sum
                if (!second_half) {
                    const int kp = tid / RP, rp = tid % RP;
#pragma unroll
                    for (int e = 0; e < 8; ++e) red[kp * BM + 8 * rp + e] = colsum[e];
                }
                lds_barrier();                                   // (B2)
            }
        }
        M2F_TS(5);
        continue;                                               // next tile (every issued load was consumed or is dead)
    }

    // ==================================== CONSUMER: LDS -> MFMA -> epilogue ====================================
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf) {
        const char* aimg = ldsA + buf * LDS_A;
        const char* bimg = ldsB + buf * LDS_B;
        const char* ab = aimg + (wm * (BM / 2) + (lane & 31)) * ROWB + (lane >> 5) * 16;      // row-major images
        const char* bb = bimg + (wn * (BN / 2) + (lane & 31)) * ROWB + (lane >> 5) * 16;
        if constexpr (FP8) {
            // row = 2*BK e4m3 values; slice ks covers 16 of them: lane (row, half) holds bytes [16*ks + 8*half, +8)
            constexpr int KS8 = BK / 8;
            const char* a8 = aimg + (wm * (BM / 2) + (lane & 31)) * ROWB + (lane >> 5) * 8;
            const char* b8 = bimg + (wn * (BN / 2) + (lane & 31)) * ROWB + (lane >> 5) * 8;
#pragma unroll
            for (int g0 = 0; g0 < KS8; g0 += 4) {
                long fa[4][MI], fb[4][NI];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) fa[kk][i] = *reinterpret_cast<const long*>(a8 + i * 32 * ROWB + (g0 + kk) * 16);
#pragma unroll
                    for (int j = 0; j < NI; ++j) fb[kk][j] = *reinterpret_cast<const long*>(b8 + j * 32 * ROWB + (g0 + kk) * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        // all fragments of the k-tile first (one exposed LDS latency per tile instead of one per 16-wide k-slice), then
        // the MFMA run back to back
        constexpr int KS = BK / 16;
        bf16x8 a[KS][MI], b[KS][NI];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if constexpr (A_RC) a[ks][i] = frag_from_kmajor(aimg, ROWK, wm * (BM / 2) + i * 32, 16 * ks, lane);
                else a[ks][i] = *reinterpret_cast<const bf16x8*>(ab + i * 32 * ROWB + ks * 32);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if constexpr (B_RC) b[ks][j] = frag_from_kmajor(bimg, ROWK, wn * (BN / 2) + j * 32, 16 * ks, lane);
                else b[ks][j] = *reinterpret_cast<const bf16x8*>(bb + j * 32 * ROWB + ks * 32);
            }
        }
        __builtin_amdgcn_sched_barrier(0);          // keep the reads ahead of the MFMA run (the scheduler sinks them otherwise)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    if constexpr (TABLE_T) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks][j], a[ks][i], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
    };

    M2F_TS(1);
    lds_barrier();                                              // (B0)
    M2F_TS(2);
    for (int kt0 = 0; kt0 < nk; kt0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kt0 + u < nk) {                                  // block-uniform
                compute(u & 1);
                lds_barrier();                                   // (B1 per tile)
            }
        }
    }
    if constexpr (TN) {
        if (has_bg) {
            lds_barrier();                                       // (B2) producers' partial sums are in LDS
            if (tid < BM) {
                float sum = 0.f;
                for (int q = 0; q < KP; ++q) sum += red[q * BM + tid];
                if (m0 + tid < M) P.bias_grad[m0 + tid] = sum;
            }
        }
    }
    M2F_TS(3);
    if constexpr (TABLE_T) table_epilogue_t<MI, NI, BM, BN>(P, acc, m0, n0, lane, wm, wn);
    else gemm_epilogue<MI, NI, BM, BN, GELU, FP8, true>(gb, P, acc, m0, n0, lane, wm, wn);
    M2F_TS(4);
  }
}

// Two register budgets of the same body.  "wide": up to 256 VGPRs, one workgroup (8 waves) per CU - the launches with
// at most one tile per CU, where a deep load ring is what matters.  "dense": at most 128 VGPRs (4 waves per SIMD), two
// workgroups per CU with a shallower ring each - the launches with more tiles than CUs.
template <bool A_RC, bool B_RC, int BM, int BN, int BK, int D, bool GELU = false, bool FP8 = false>
__global__ __launch_bounds__(512) void m2f_gemm16_kernel(const GemmBatch gb) {
    gemm16_body<A_RC, B_RC, BM, BN, BK, D, false, GELU, FP8>(gb);
}
template <bool A_RC, bool B_RC, int BM, int BN, int BK, int D>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void m2f_gemm16_dense_kernel(const GemmBatch gb) {
    gemm16_body<A_RC, B_RC, BM, BN, BK, D, false>(gb);
}
// TABLE form: problems in device memory, persistent walk over the tile list
template <bool A_RC, bool B_RC, int BM, int BN, int BK, int D>
__global__ __launch_bounds__(512) void m2f_gemm16_table_kernel(const GemmBatch gb) {
    gemm16_body<A_RC, B_RC, BM, BN, BK, D, true>(gb);
}
template <bool A_RC, bool B_RC, int BM, int BN, int BK, int D>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void m2f_gemm16_table_dense_kernel(const GemmBatch gb) {
    gemm16_body<A_RC, B_RC, BM, BN, BK, D, true>(gb);
}

template <bool A_RC, bool B_RC, int BM, int BN, int BK, int D, bool DENSE = false, bool GELU = false, bool FP8 = false>
hipError_t launch_cfg16(const GemmBatch& gb, int total_tiles, hipStream_t stream) {
    constexpr int lds = 2 * (A_RC ? Stage16RC<BM, BK>::LDS_BYTES : Stage16KC<BM, BK>::LDS_BYTES) +
                        2 * (B_RC ? Stage16RC<BN, BK>::LDS_BYTES : Stage16KC<BN, BK>::LDS_BYTES) + (BK / 8) * BM * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert(!DENSE || 2 * lds <= 160 * 1024, "two workgroups per CU");
    GemmBatch hb = gb;                                          // compact header for the kernel's critical path
    for (int i = 0; i < M2F_GEMM_MAX_PROBLEMS; ++i) {
        hb.tb[i] = i < gb.count ? gb.pr[i].tile_begin : 0x7fffffff;
        GemmHot& h = hb.hot[i];
        memset(&h, 0, sizeof(h));
        if (i >= gb.count) continue;
        const GemmProblem& p = gb.pr[i];
        h.aq[0] = p.a.q[0]; h.aq[1] = p.a.q[1]; h.bq[0] = p.b.q[0]; h.bq[1] = p.b.q[1];
        h.M = p.M; h.N = p.N; h.k[0] = p.a.k[0]; h.k[1] = p.a.k[1];
        h.ldaq[0] = p.a.ldq[0]; h.ldaq[1] = p.a.ldq[1]; h.ldbq[0] = p.b.ldq[0]; h.ldbq[1] = p.b.ldq[1];
        h.flags = p.flags; h.tile_begin = p.tile_begin; h.has_bias_grad = p.bias_grad != nullptr;
    }
    void (*kern)(const GemmBatch);
    static_assert(!(DENSE && GELU), "the GELU epilogue exists for the wide forward-form kernels only");
    if constexpr (DENSE) kern = m2f_gemm16_dense_kernel<A_RC, B_RC, BM, BN, BK, D>;
    else kern = m2f_gemm16_kernel<A_RC, B_RC, BM, BN, BK, D, GELU, FP8>;
    if (lds > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(total_tiles), dim3(512), lds, stream, hb);
    return hipGetLastError();
}

template <int PREC, bool A_RC, bool B_RC, int BM, int BN, int BK, bool VEC, bool GELU = false>
hipError_t launch_cfg(const GemmBatch& gb, int total_tiles, hipStream_t stream) {
    using SA = Stage<PREC, A_RC, BM, BK, VEC>;
    using SB = Stage<PREC, B_RC, BN, BK, VEC>;
    constexpr int lds = 2 * SA::LDS_BYTES + 2 * SB::LDS_BYTES;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = m2f_gemm_kernel<PREC, A_RC, B_RC, BM, BN, BK, VEC, GELU>;
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
    constexpr int BK64 = (PREC == M2F_PREC_F32) ? 64 : 128;
    constexpr int BK128 = (PREC == M2F_PREC_F32) ? 32 : 64;
    const int bk = tile == 128 ? BK128 : BK64;
    // split-K when the launch cannot occupy the chip: a CU pulls operands at a fixed ~25-45 GB/s (L1 miss queue x
    // latency), so small grids are spread over more CUs.  Only for the forward / dgrad forms on 64x64 tiles.
    int want_s = 1;
    const int plain = count_tiles(bm, bn);
    if (tile == 64 && !A_RC && gb.splitk_ws && gb.splitk_cnt && plain > 0 && plain < 224) {
        want_s = (352 + plain - 1) / plain;
        if (want_s > 4) want_s = 4;
    }
    int t = 0, slabs = 0, tiles_total = 0;
    for (int i = 0; i < gb.count; ++i) {
        GemmProblem& p = gb.pr[i];
        const int nk = m2f_cdiv(p.a.k[0], bk) + m2f_cdiv(p.a.k[1], bk);
        int sp = want_s;
        if (sp > nk / 2) sp = nk / 2;                 // at least two k-tiles per slice
        if (sp < 1) sp = 1;
        p.splitk = sp;
        p.tile_begin = t;
        p.slab_begin = slabs;                         // in slices (one [BM x BN] partial each)
        p.cnt_begin = tiles_total;
        p.tiles_n = m2f_cdiv(p.N, bn);
        const int tiles = m2f_cdiv(p.M, bm) * p.tiles_n;
        t += tiles * sp;
        slabs += tiles * sp;
        tiles_total += tiles;
    }
    if (want_s > 1 && (tiles_total > gb.splitk_max_tiles || slabs > 4 * gb.splitk_max_tiles)) {   // scratch too small: no split
        t = 0;
        for (int i = 0; i < gb.count; ++i) {
            GemmProblem& p = gb.pr[i];
            p.splitk = 1; p.tile_begin = t; p.slab_begin = 0; p.cnt_begin = 0;
            t += m2f_cdiv(p.M, bm) * p.tiles_n;
        }
    }
    if (t == 0) return hipSuccess;
    // 16-byte loads only when every operand of every problem of the launch allows them
    bool vec = true;
    for (int i = 0; i < gb.count; ++i)
        vec = vec && (gb.pr[i].flags & GF_VEC_A) && (gb.pr[i].flags & GF_VEC_B);
    if constexpr (!A_RC && !B_RC) {                             // GELU epilogue: forward form only (text encoder)
        bool gelu = false;
        for (int i = 0; i < gb.count; ++i) gelu = gelu || (gb.pr[i].flags & GF_GELU_OUT);
        if (gelu) {
            if (tile == 128)
                return vec ? launch_cfg<PREC, false, false, 128, 128, BK128, true, true>(gb, t, stream)
                           : launch_cfg<PREC, false, false, 128, 128, BK128, false, true>(gb, t, stream);
            return vec ? launch_cfg<PREC, false, false, 64, 64, BK64, true, true>(gb, t, stream)
                       : launch_cfg<PREC, false, false, 64, 64, BK64, false, true>(gb, t, stream);
        }
    }
    if (tile == 128)
        return vec ? launch_cfg<PREC, A_RC, B_RC, 128, 128, BK128, true>(gb, t, stream)
                   : launch_cfg<PREC, A_RC, B_RC, 128, 128, BK128, false>(gb, t, stream);
    return vec ? launch_cfg<PREC, A_RC, B_RC, 64, 64, BK64, true>(gb, t, stream)
               : launch_cfg<PREC, A_RC, B_RC, 64, 64, BK64, false>(gb, t, stream);
}

template <bool A_RC, bool B_RC>
hipError_t launch_tile16(GemmBatch& gb, int tile, hipStream_t stream) {
    auto count_tiles = [&](int bm, int bn) {
        int t = 0;
        for (int i = 0; i < gb.count; ++i) t += m2f_cdiv(gb.pr[i].M, bm) * m2f_cdiv(gb.pr[i].N, bn);
        return t;
    };
    const bool auto_tile = tile == 0;
    // 128x128 tiles halve the bytes pulled per output element; used once the launch still fills the chip with them
    // (wgrad launches: measured 3.25 -> 3.18 ms/step with the threshold at 256 instead of 512)
    if (tile == 0) tile = (count_tiles(128, 128) >= (A_RC ? 256 : 512)) ? 128 : 64;
    // very large forward-form launches (the text encoder: M = utterances x tokens): 256 (M) x 128 (N) tiles, 85 instead of
    // 64 FLOP per byte through the L1 fill path that bounds these kernels
    // (RoBERTa-large geometry, 512 utterances x 64 tokens: 59.3 -> 55.7 ms per forward)
    if (tile == 128 && !A_RC && !B_RC && count_tiles(256, 128) >= 1024) tile = 256;
    if constexpr (!A_RC && !B_RC) {
        // Forward-form launches run the ring form (gemm_ring.h, m2f_gemm16_ring_kernel): 128x128 tiles from ~one tile per CU
        // on, 128x64 tiles below that while those still cover most of the chip, 64x64 tiles for the rest down to 80 tiles;
        // what remains (a few tiles, e.g. the classifier) keeps the register-staged 64x64 build below.  Measured on MI355X,
        // fwd+bwd of the C3 step (B = 64): 3.21 ms without the ring form; 3.06 with the 128x128 form from 200 tiles (3.07 /
        // 3.07 / 3.16 at 100 / 150 / 260); 2.91 with the 128x64 form from 150 tiles on top (2.95 / 2.91 at 200 / 100); 2.89
        // with the 64x64 form from 80 tiles (no change down to 1).  C2 (B = 32): 1.957 -> 1.914 -> 1.853 (the 128x64 form
        // from 100 tiles: 2.014 - launches of ~100 tiles leave too many CUs idle).  M2F_RING=0 switches the form off,
        // M2F_RING_MIN / M2F_RING64_MIN / M2F_RING32_MIN move the thresholds.
        static const int ring = getenv("M2F_RING") ? atoi(getenv("M2F_RING")) : 1;
        static const int ring_min = getenv("M2F_RING_MIN") ? atoi(getenv("M2F_RING_MIN")) : 200;
        static const int ring64_min = getenv("M2F_RING64_MIN") ? atoi(getenv("M2F_RING64_MIN")) : 150;
        static const int ring32_min = getenv("M2F_RING32_MIN") ? atoi(getenv("M2F_RING32_MIN")) : 80;
        // Text-encoder-sized launches (M = utterances x tokens = 32,768): 256x128 ring tiles, whose k-loop runs at the MFMA rate
        // (1,050 cycles per 256x128x64 k-tile against 1,024) and whose fixed cost per tile is amortised by the persistent walk.
        // RoBERTa-large geometry 50.2 -> 43.4 ms per forward, base 17.4 -> 14.5 (threshold 512 tiles; 15.0 at 1,024; no
        // launch of the M2FNet step is that large).
        static const int ring256_min = getenv("M2F_RING256_MIN") ? atoi(getenv("M2F_RING256_MIN")) : 512;      // tiles of 256x128
        // Round 4: from one chip-filling round of 256 x 256 tiles on, the eight-phase form (gemm_p8.h: all eight waves load and multiply,
        // continuous prefetch stream across tiles): 32,768 x 1,024 x 1,024 752 TFLOP/s against ~460 for the 256x128 ring form.
        // M2F_P8=0 switches it off, M2F_P8_MIN moves the threshold (tiles of 256x256).
        static const int p8_on = getenv("M2F_P8") ? atoi(getenv("M2F_P8")) : 1;
        static const int p8_min = getenv("M2F_P8_MIN") ? atoi(getenv("M2F_P8_MIN")) : 256;
        // (... and only when those tiles fill whole rounds of the chip: a 256 x 256 tile is ~25 us of work, and 336 of them on 256 CUs -
        //  the merged QKV projections of the C3 geometry at B = 256 - ran 2 % of the STEP slower than three rounds of 256 x 128 ring tiles)
        {
            const int t256 = count_tiles(256, 256);
            const int rounds = (t256 + 255) / 256;
            if (ring && p8_on && auto_tile && t256 >= p8_min && t256 * 100 >= 85 * rounds * 256 && m2f_gemm_p8_ok(gb)) return m2f_p8_launch_kc(gb, stream);
        }
        if (ring && auto_tile && count_tiles(256, 128) >= ring256_min && m2f_gemm_ring256_ok(gb)) return m2f_launch_gemm_ring(gb, 256, 128, stream);
        // (what the 256x128 ring form cannot take - it has bias / ReLU / GELU / residual epilogues only - keeps the register-staged
        // 256x128 build from 1,024 such tiles on: RoBERTa-large geometry 52.3 vs 54.0 ms with 128x128 ring tiles)
        // launches of 257..511 tiles of 128x128 take two rounds on 256 CUs with the second one mostly empty (the merged audio +
        // text QKV in-projections at C3: 336 tiles); as 256x128 tiles they are one round (168 tiles).  M2F_RING256_2R=0 switches it off
        static const int ring256_2r = getenv("M2F_RING256_2R") ? atoi(getenv("M2F_RING256_2R")) : 1;
        if (ring && auto_tile && ring256_2r && tile != 256) {
            const int t128 = count_tiles(128, 128);
            if (t128 > 256 && t128 < 512 && count_tiles(256, 128) <= 256 && m2f_gemm_ring256_ok(gb) && m2f_gemm_ring_ok(gb))
                return m2f_launch_gemm_ring(gb, 256, 128, stream);
        }
        if (ring && auto_tile && tile != 256 && m2f_gemm_ring_ok(gb)) {
            if (count_tiles(128, 128) >= ring_min) return m2f_launch_gemm_ring(gb, 128, 128, stream);
            if (count_tiles(128, 64) >= ring64_min) return m2f_launch_gemm_ring(gb, 128, 64, stream);
            if (count_tiles(64, 64) >= ring32_min) return m2f_launch_gemm_ring(gb, 64, 64, stream);
        }
    }
    const int tile_m = tile, tile_n = tile == 256 ? 128 : tile;
    int t = 0;
    for (int i = 0; i < gb.count; ++i) {
        GemmProblem& p = gb.pr[i];
        p.splitk = 1; p.slab_begin = 0; p.cnt_begin = 0;
        p.tile_begin = t;
        p.tiles_n = m2f_cdiv(p.N, tile_n);
        t += m2f_cdiv(p.M, tile_m) * p.tiles_n;
    }
    if (t == 0) return hipSuccess;
    if constexpr (!A_RC && !B_RC) {
        bool gelu = false;
        for (int i = 0; i < gb.count; ++i) gelu = gelu || (gb.pr[i].flags & GF_GELU_OUT);
        if (gelu) {                                              // text encoder (forward form only): own instantiations
            if (tile == 256) return launch_cfg16<false, false, 256, 128, 64, 2, false, true>(gb, t, stream);
            if (tile == 128) return launch_cfg16<false, false, 128, 128, 64, 3, false, true>(gb, t, stream);
            return launch_cfg16<false, false, 64, 64, 128, 2, false, true>(gb, t, stream);
        }
        if (tile == 256) return launch_cfg16<false, false, 256, 128, 64, 2>(gb, t, stream);
    }
    if (tile == 128) return launch_cfg16<A_RC, B_RC, 128, 128, 64, 3>(gb, t, stream);
    // ONE 64x64 build for every launch size: ring depth 2 within 128 VGPRs (two workgroups per CU when the launch has more
    // tiles than CUs).  Deeper rings gain nothing for these launches (start-up and dispatch bound, not ring bound) and cost
    // code size: same-box A/B of the whole step 2.130 (depth 4 + separate 256-VGPR build) -> 2.107 ms, and again after the
    // ring kept its loads in flight: 2.34 / 2.38 (depth 3 / 4, 256-VGPR build for launches of
    // at most 256 tiles) vs 2.32 ms; the step alternates between ~8 kernels, so every kilobyte of code is instruction-cache
    // traffic at each launch.
#ifdef M2F_CHAIN_WIDE_D        // experiment: launches that give a CU one workgroup at most run a 256-VGPR build with a deeper ring
    if (t <= 256) return launch_cfg16<A_RC, B_RC, 64, 64, 128, M2F_CHAIN_WIDE_D, false>(gb, t, stream);
#endif
    return launch_cfg16<A_RC, B_RC, 64, 64, 128, M2F_CHAIN_D, true>(gb, t, stream);
}

// can every operand of every problem be staged from its bf16 shadow with 16-byte loads?
bool src16_ok(const GemmBatch& gb, bool a_rc, bool b_rc) {
    auto seg_ok = [](const GemmOperand& o, bool rc, int rows) {
        for (int s = 0; s < 2; ++s) {
            if (o.k[s] == 0) continue;
            if (!o.q[s] || (reinterpret_cast<uintptr_t>(o.q[s]) & 15) || (o.ldq[s] & 7)) return false;
            const int extent = rc ? rows : o.k[s];           // the contiguous dimension
            if ((extent & 7) && o.ldq[s] != ((extent + 7) & ~7)) return false;   // pads must be the buffer's own zero pads
            if (o.ldq[s] < extent) return false;
        }
        return true;
    };
    for (int i = 0; i < gb.count; ++i) {
        const GemmProblem& p = gb.pr[i];
        if (!seg_ok(p.a, a_rc, p.M) || !seg_ok(p.b, b_rc, p.N)) return false;
    }
    return true;
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

template <int BM, int BN, int BK, int D, bool DENSE>
hipError_t launch_table16(const GemmBatch& gb, hipStream_t stream) {
    constexpr int lds = 2 * Stage16KC<BM, BK>::LDS_BYTES + 2 * Stage16KC<BN, BK>::LDS_BYTES + (BK / 8) * BM * 4;
    static_assert(lds <= 160 * 1024 && (!DENSE || 2 * lds <= 160 * 1024), "LDS budget");
    void (*kern)(const GemmBatch);
    if constexpr (DENSE) kern = m2f_gemm16_table_dense_kernel<false, false, BM, BN, BK, D>;
    else kern = m2f_gemm16_table_kernel<false, false, BM, BN, BK, D>;
    if (lds > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
    }
    const int slots = 256 * (DENSE ? 2 : 1);                    // workgroups the chip holds at once
    const int grid = gb.total_tiles < slots ? gb.total_tiles : slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, gb);
    return hipGetLastError();
}

}  // namespace

int m2f_gemm_table_layout(std::vector<GemmProblem>& prs, int tile, std::vector<uint16_t>& tile_prob, bool operand_options) {
    // tile = 64: 64x64, 128: 128x128, 256: 256 (M) x 128 (N)
    const int tile_m = tile, tile_n = tile == 256 ? 128 : tile;
    tile_prob.clear();
    if (prs.size() > 65535) return -1;
    for (const GemmProblem& p : prs)
        if ((p.flags & (GF_RELU_OUT | GF_ACCUM | GF_GELU_OUT)) || (!operand_options && ((p.flags & (GF_RELU_A | GF_RELU_B)) || p.bias_grad)) ||
            p.bias || p.res || p.gate || p.drop_site || p.c8 || p.a.k[1] || p.b.k[1])
            return -1;       // the table kernel stages operands as they are (store_select) and stores plain results (table_epilogue_t)
    int t = 0;
    for (size_t i = 0; i < prs.size(); ++i) {
        GemmProblem& p = prs[i];
        p.splitk = 1; p.slab_begin = 0; p.cnt_begin = 0;
        p.tile_begin = t;
        p.tiles_n = m2f_cdiv(p.N, tile_n);
        const int n = m2f_cdiv(p.M, tile_m) * p.tiles_n;
        for (int j = 0; j < n; ++j) tile_prob.push_back((uint16_t)i);
        t += n;
    }
    return t;
}

int m2f_gemm_table_walk(const std::vector<GemmProblem>& prs_all, int walk, int n_wg, int tile_m, int tile_n, std::vector<uint32_t>& tile_rec, std::vector<int>& wg_begin,
                        const std::vector<int>* only) {
    const int TILE = tile_n;
    // `only` (nullable): walk just these problems (records still carry the index into prs_all = the device table)
    std::vector<size_t> sel;
    if (only) for (int i : *only) sel.push_back((size_t)i);
    else for (size_t i = 0; i < prs_all.size(); ++i) sel.push_back(i);
    struct View { const std::vector<GemmProblem>& a; const std::vector<size_t>& s; size_t size() const { return s.size(); }
                  const GemmProblem& operator[](size_t i) const { return a[s[i]]; } } prs{prs_all, sel};
    tile_rec.clear();
    wg_begin.assign((size_t)n_wg + 1, 0);
    if (n_wg < 1) return -1;
    auto rec = [&](size_t p, int mt, int nt) { return (uint32_t)sel[p] | ((uint32_t)mt << 16) | ((uint32_t)nt << 24); };
    if (prs_all.size() > 65535) return -1;
    std::vector<std::vector<uint32_t>> per_wg((size_t)n_wg);
    if (walk == 0) {
        // the order of m2f_gemm_table_layout (m fastest inside a problem), dealt as ring_xcd_remap deals it
        std::vector<uint32_t> all;
        for (size_t p = 0; p < prs.size(); ++p) {
            const int tm = m2f_cdiv(prs[p].M, tile_m), tn = m2f_cdiv(prs[p].N, TILE);
            if (tm > 255 || tn > 255) return -1;
            for (int nt = 0; nt < tn; ++nt)
                for (int mt = 0; mt < tm; ++mt) all.push_back(rec(p, mt, nt));
        }
        const int q = n_wg >> 3, r = n_wg & 7;
        for (int b = 0; b < n_wg; ++b) {
            const int x = b & 7, j = b >> 3;
            const int first = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
            for (size_t t = (size_t)first; t < all.size(); t += (size_t)n_wg) per_wg[(size_t)b].push_back(all[t]);
        }
    } else {
        // items = super-tiles of up to 8 (m) x 4 (n) tiles, problem-major, n-strip-major inside a problem (consecutive items
        // of a strip share their column panels); the item list is cut into eight contiguous ranges of about equal tile count
        struct Item { std::vector<uint32_t> tiles; };
        std::vector<Item> items;
        size_t total = 0;
        for (size_t p = 0; p < prs.size(); ++p) {
            const int tm = m2f_cdiv(prs[p].M, tile_m), tn = m2f_cdiv(prs[p].N, TILE);
            if (tm > 255 || tn > 255) return -1;
            const int sm = tm < 8 ? tm : 8;
            int sn = 32 / sm; if (sn < 1) sn = 1; if (sn > tn) sn = tn;
            for (int n0 = 0; n0 < tn; n0 += sn)
                for (int m0 = 0; m0 < tm; m0 += sm) {
                    Item it;
                    for (int nt = n0; nt < std::min(tn, n0 + sn); ++nt)
                        for (int mt = m0; mt < std::min(tm, m0 + sm); ++mt) it.tiles.push_back(rec(p, mt, nt));
                    total += it.tiles.size();
                    items.push_back(std::move(it));
                }
        }
        const int n_x = n_wg < 8 ? 1 : 8;
        std::vector<std::vector<uint32_t>> seq((size_t)n_x);
        size_t done = 0, i = 0;
        for (int x = 0; x < n_x; ++x) {
            const size_t target = total * (size_t)(x + 1) / (size_t)n_x;        // cumulative share after XCD x
            while (i < items.size() && (x == n_x - 1 || done + items[i].tiles.size() / 2 < target)) {
                seq[(size_t)x].insert(seq[(size_t)x].end(), items[i].tiles.begin(), items[i].tiles.end());
                done += items[i].tiles.size();
                ++i;
            }
        }
        // workgroups of XCD x: b = x, x + 8, ... ; the j-th of them takes elements j, j + per_x, ... of the XCD's sequence
        for (int x = 0; x < n_x; ++x) {
            std::vector<int> wgs;
            for (int b = x; b < n_wg; b += n_x) wgs.push_back(b);
            for (size_t t = 0; t < seq[(size_t)x].size(); ++t) per_wg[(size_t)wgs[t % wgs.size()]].push_back(seq[(size_t)x][t]);
        }
    }
    for (int b = 0; b < n_wg; ++b) {
        wg_begin[(size_t)b] = (int)tile_rec.size();
        tile_rec.insert(tile_rec.end(), per_wg[(size_t)b].begin(), per_wg[(size_t)b].end());
    }
    wg_begin[(size_t)n_wg] = (int)tile_rec.size();
    {   // every tile of every problem exactly once, whatever the order
        std::vector<uint32_t> got = tile_rec, want;
        for (size_t p = 0; p < prs.size(); ++p)
            for (int nt = 0; nt < m2f_cdiv(prs[p].N, TILE); ++nt)
                for (int mt = 0; mt < m2f_cdiv(prs[p].M, tile_m); ++mt) want.push_back(rec(p, mt, nt));
        std::sort(got.begin(), got.end());
        std::sort(want.begin(), want.end());
        if (got != want) return -1;
    }
    return (int)tile_rec.size();
}

#ifdef M2F_EXP_TIMING
extern "C" int m2f_dbg_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(m2f_dbg), sizeof(unsigned long long) * 64);
}
#endif

hipError_t m2f_launch_gemm_table(const GemmBatch& gb, hipStream_t stream) {
    if (!gb.table || !gb.tile_prob || gb.total_tiles <= 0) return hipErrorInvalidValue;
    if (gb.table_tile >= 129 && gb.table_tile <= 132) return m2f_launch_gemm_ring_table(gb, stream);                              // 128x128 tiles, ring form
    if (gb.table_tile == 256) return launch_table16<256, 128, 64, M2F_T256_D, false>(gb, stream);
    if (gb.table_tile == 128) return launch_table16<128, 128, 64, 3, false>(gb, stream);
    if (gb.table_tile == 64) return launch_table16<64, 64, 128, 2, true>(gb, stream);
    return hipErrorInvalidValue;
}

hipError_t m2f_ring_launch_256x128_fp8(GemmBatch& gb, hipStream_t stream);      // gemm_ring_256x128_fp8.hip

hipError_t m2f_launch_gemm_fp8(GemmBatch& gb, hipStream_t stream) {
    if (gb.count != 1) return hipErrorInvalidValue;
    GemmProblem& p = gb.pr[0];
    if (p.M <= 0 || p.N <= 0 || p.a.k[0] <= 0 || p.a.k[1] != 0 || p.a.k[0] != p.b.k[0] || (p.a.k[0] & 15) || !p.a.q[0] || !p.b.q[0] ||
        (p.a.ldq[0] & 15) || (p.b.ldq[0] & 15) || (reinterpret_cast<uintptr_t>(p.a.q[0]) & 15) || (reinterpret_cast<uintptr_t>(p.b.q[0]) & 15) ||
        p.gate || (p.flags & (GF_ACCUM | GF_RELU_A | GF_RELU_B)) || p.drop_site)
        return hipErrorInvalidValue;
    // the staging code moves 16-byte chunks of a row whatever they hold: hand it the rows in byte pairs
    p.a.k[0] >>= 1; p.b.k[0] >>= 1; p.a.ldq[0] >>= 1; p.b.ldq[0] >>= 1;
    const bool gelu = p.flags & GF_GELU_OUT;
    {   // round 4: 256x256 tiles on the eight-phase schedule (gemm_p8.h, EPI 4) from one chip-filling round on, when its tiles fill whole rounds
        // (M2F_P8=0 / M2F_P8_MIN as for the bf16 launches; k a multiple of 128)
        static const int p8_on = getenv("M2F_P8") ? atoi(getenv("M2F_P8")) : 1;
        static const int p8_min = getenv("M2F_P8_MIN") ? atoi(getenv("M2F_P8_MIN")) : 256;
        const int t256 = m2f_cdiv(p.M, 256) * m2f_cdiv(p.N, 256), rounds = m2f_cdiv(t256, 256);
        const bool small8 = (size_t)p.M * p.a.ldq[0] * 2 < 0x80000000ull && (size_t)p.N * p.b.ldq[0] * 2 < 0x80000000ull;
        if (p8_on && small8 && t256 >= p8_min && t256 * 100 >= 85 * rounds * 256 && !(p.a.k[0] & 63) && (!p.c8 || (p.ldc & 7) == 0))
            return m2f_p8_launch_kc_fp8(gb, stream);
    }
    {   // the ring form (gemm_ring_256x128_fp8.hip) from one chip-filling round of 256x128 tiles on; M2F_RING_FP8=0 keeps the register-staged build
        static const int ring8 = getenv("M2F_RING_FP8") ? atoi(getenv("M2F_RING_FP8")) : 1;
        const bool small = (size_t)p.M * p.a.ldq[0] * 2 < 0x80000000ull && (size_t)p.N * p.b.ldq[0] * 2 < 0x80000000ull;
        const bool whole = p.M % 256 == 0 && p.N % 128 == 0;       // (the ring epilogue writes e4m3 results of whole tiles only)
        if (ring8 && small && m2f_cdiv(p.M, 256) * m2f_cdiv(p.N, 128) >= 256 && !(p.flags & GF_RELU_OUT) && (!p.c8 || whole))
            return m2f_ring_launch_256x128_fp8(gb, stream);
    }
    int tile_m = 256, tile_n = 128;
    if (m2f_cdiv(p.M, 256) * m2f_cdiv(p.N, 128) < 1024) { tile_m = 128; tile_n = 128; }
    p.splitk = 1; p.slab_begin = 0; p.cnt_begin = 0; p.tile_begin = 0;
    p.tiles_n = m2f_cdiv(p.N, tile_n);
    const int t = m2f_cdiv(p.M, tile_m) * p.tiles_n;
    if (tile_m == 256)
        return gelu ? launch_cfg16<false, false, 256, 128, 64, 2, false, true, true>(gb, t, stream)
                    : launch_cfg16<false, false, 256, 128, 64, 2, false, false, true>(gb, t, stream);
    return gelu ? launch_cfg16<false, false, 128, 128, 64, 3, false, true, true>(gb, t, stream)
                : launch_cfg16<false, false, 128, 128, 64, 3, false, false, true>(gb, t, stream);
}

// Will m2f_launch_gemm stage this bf16-mode launch from the operands' bf16 shadows (true), or from their fp32 originals?  (Host-side
// mirror of the dispatch below, for plan.hip::mark_unread_fp32.)
bool m2f_gemm_stages_bf16(const GemmBatch& gb, int layout) {
    const bool a_rc = layout == M2F_LAYOUT_TN, b_rc = layout != M2F_LAYOUT_NT;
    if (layout == M2F_LAYOUT_NN) {
        bool all_t = true;
        for (int i = 0; i < gb.count; ++i)
            for (int sgm = 0; sgm < 2; ++sgm)
                if (gb.pr[i].b.k[sgm] && !gb.pr[i].b.qt[sgm]) all_t = false;
        if (all_t) {
            GemmBatch t = gb;
            for (int i = 0; i < t.count; ++i)
                for (int sgm = 0; sgm < 2; ++sgm) { t.pr[i].b.q[sgm] = t.pr[i].b.qt[sgm]; t.pr[i].b.ldq[sgm] = t.pr[i].b.ldqt[sgm]; }
            if (src16_ok(t, false, false)) return true;
        }
    }
    return src16_ok(gb, a_rc, b_rc);
}

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
    {   // the classifier head's [T, n_classes] problems: FMA kernels over the whole chip instead of one column of MFMA tiles (skinny.hip)
        const int sk = m2f_launch_gemm_skinny(gb, prec, layout, stream);
        if (sk == 1) return hipSuccess;
        if (sk < 0) return (hipError_t)(-sk);
    }
    if (prec == M2F_PREC_BF16 && layout == M2F_LAYOUT_NN) {
        // dgrad against a weight whose TRANSPOSED bf16 shadow exists runs as the k-contiguous (forward) form, the
        // fastest staging path: C = A * B  ==  A * (B^T)^T
        bool all_t = true;
        for (int i = 0; i < gb.count; ++i)
            for (int sgm = 0; sgm < 2; ++sgm)
                if (gb.pr[i].b.k[sgm] && !gb.pr[i].b.qt[sgm]) all_t = false;
        if (all_t) {
            GemmBatch t = gb;
            for (int i = 0; i < t.count; ++i)
                for (int sgm = 0; sgm < 2; ++sgm) { t.pr[i].b.q[sgm] = t.pr[i].b.qt[sgm]; t.pr[i].b.ldq[sgm] = t.pr[i].b.ldqt[sgm]; }
            if (src16_ok(t, false, false)) return launch_tile16<false, false>(t, tile, stream);
        }
    }
    if (prec == M2F_PREC_BF16 && src16_ok(gb, a_rc, b_rc)) {
        if (layout == M2F_LAYOUT_NT) return launch_tile16<false, false>(gb, tile, stream);
        if (layout == M2F_LAYOUT_NN) return launch_tile16<false, true>(gb, tile, stream);
        return launch_tile16<true, true>(gb, tile, stream);
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
