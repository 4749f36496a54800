// RING form of the k-contiguous bf16 GEMM: device code + launch templates.  Included by one translation unit per tile
// configuration (gemm_ring_*.hip) so that the configurations - each with 16 epilogue variants - compile in parallel.
#pragma once
#include "common.h"
#include "ops.h"
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <type_traits>

// (same setting as gemm.hip: the register-staged kernels repeat this arithmetic and must round identically)
#pragma clang fp contract(off)

extern long long m2f_g_ring_launches;         // host-side count of ring-form launches (gemm_ring.hip)

namespace {

#ifdef M2F_EXP_TIMING
__device__ unsigned long long m2f_ring_dbg[64];
#define M2F_TS(slot) do { if (blockIdx.x == 0 && (threadIdx.x & 255) == 0) m2f_ring_dbg[(slot) + (threadIdx.x >= 256 ? 16 : 0)] = __builtin_amdgcn_s_memtime(); } while (0)
// accumulating form (table kernel, one workgroup walks many tiles): slot += cycles; the host zeroes the buffer before the launch
#define M2F_NOW() __builtin_amdgcn_s_memtime()
// (sums are kept in registers and written once when the role ends: a memory update per k-tile would be what gets measured)
#define M2F_ACC_DECL unsigned long long m2f_acc_[4] = {0ull, 0ull, 0ull, 0ull}
#define M2F_ADD(slot, dt) do { m2f_acc_[(slot) - 8] += (unsigned long long)(dt); } while (0)
// (wave 1 of the role: as a consumer it owns column block 1 and never carries the bias-gradient sums)
#define M2F_ACC_FLUSH() do { if (blockIdx.x == 0 && (threadIdx.x & 255) == 64) { for (int q_ = 0; q_ < 4; ++q_) m2f_ring_dbg[8 + q_ + (threadIdx.x >= 256 ? 16 : 0)] += m2f_acc_[q_]; } } while (0)
#else
#define M2F_TS(slot) do {} while (0)
#define M2F_NOW() 0ull
#define M2F_ACC_DECL do {} while (0)
#define M2F_ADD(slot, dt) do { (void)sizeof(dt); } while (0)
#define M2F_ACC_FLUSH() do {} while (0)
#endif

typedef unsigned int ring_u32x4 __attribute__((ext_vector_type(4)));
typedef short ring_s16x4 __attribute__((ext_vector_type(4)));
typedef short ring_s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float ring_bf16lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float ring_bf16hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xFFFF0000u); }
__device__ __forceinline__ uint32_t ring_relu_bf16x2(uint32_t v) {
    // bf16 pairs as int16 pairs: sign bit set <=> negative int16, so max(., 0) zeroes exactly the halves the bit mask of
    // gemm.hip's relu_bf16x2 zeroes (same result bits) in ONE v_pk_max_i16 instead of four integer operations
    typedef short ring_s16x2 __attribute__((ext_vector_type(2)));
    const ring_s16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(ring_s16x2, v), z));
}
// sum of the two bf16 halves of v added to acc (v_dot2c_f32_bf16 against (1, 1)): the bias-gradient row sums
__device__ __forceinline__ float ring_bf16x2_sum(uint32_t v, float acc) {
    typedef __bf16 ring_bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ring_bf16x2, v), __builtin_bit_cast(ring_bf16x2, 0x3F803F80u), acc, false);
}
// XCD-aware block order (gemm.hip, xcd_remap): every XCD walks a contiguous range of the launch's tile list
__device__ __forceinline__ int ring_xcd_remap(int b, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, x = b & 7, j = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}
// LDS-only workgroup barrier (gemm.hip, lds_barrier): no vmcnt drain - the ring's loads stay in flight across it
__device__ __forceinline__ void ring_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Epilogue of the ring form.  The consumers multiplied with the operands SWAPPED (as for table_epilogue_t), so a lane holds, of
// its 32x32 block, row (lane & 31) and columns 8 g + 4 (lane >> 5) + 0..3 in registers 4 g .. 4 g + 3.  Stored from there, a
// wave instruction touches 32 different 128-byte lines (measured: 12.9k cycles per 128x128 tile; the row-per-register layout
// of gemm_epilogue with its 4-byte stores: 7.8k - both far above the tile's 8k cycles of MFMA work).  What the store path
// wants is whole lines per instruction, so each block makes one pass through a wave-private 4 KiB LDS image (4 ds_write_b128,
// 16-byte chunks XOR-swizzled by row, then 4 ds_read_b128 in row-major lane order: lane -> row (lane >> 3) + 8 p, columns
// 4 (lane & 7) + 0..3) and every term of the epilogue - bias, residual, gate, accumulate, C, the bf16 shadow - moves as
// 16-byte (8-byte for bf16) accesses of 8 lanes per 128-byte row segment.  Per element the operations and their order are
// those of gemm_epilogue, so both produce the same bits.  `ep` = this wave's 4 KiB of LDS (not part of the operand ring: the
// producers are already filling that with the next tile).
// (the descriptor fields are fetched - scalar loads from the kernel argument segment, the shadow map, the RNG state - BEFORE
// the k-loop: read after it they cost the tile 2k cycles of dependent load latency with the MFMA pipe idle)
struct RingEpi {
    int M, N, ldc, ldres, ldgate;
    const float* bias; const float* res; const float* gate; float* C; uint16_t* C16;
    float gscale; uint32_t site, key; bool relu_out, accum, vec, gelu, no32;
    float acc_scale, c8_scale; uint8_t* c8;      // fp8 launches (EPI 4): de-quantisation factor; e4m3 result instead of fp32 C
};
template <int BM, int BN>
__device__ __forceinline__ RingEpi ring_epilogue_args(const GemmBatch& gb, const GemmProblem& P, int m0, int n0) {
    RingEpi E;
    E.M = P.M; E.N = P.N; E.ldc = P.ldc; E.ldres = P.ldres; E.ldgate = P.ldgate;
    E.bias = P.bias; E.res = P.res; E.gate = P.gate; E.C = P.c;
    E.C16 = (P.flags & GF_NO_BF16) ? nullptr : reinterpret_cast<uint16_t*>(m2f_shadow_of(gb.sh, P.c));
    E.gscale = P.gate_scale;
    E.relu_out = P.flags & GF_RELU_OUT; E.accum = P.flags & GF_ACCUM; E.gelu = P.flags & GF_GELU_OUT;
    E.no32 = (P.flags & GF_NO_F32) && E.C16 && !E.accum;      // C has no fp32 reader: its bf16 shadow is the result
    E.acc_scale = P.acc_scale; E.c8_scale = P.c8_scale; E.c8 = P.c8;
    E.site = P.drop_site; E.key = 0;
    if (E.site) E.key = m2f_site_key(gb.rng, E.site);
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    E.vec = (m0 + BM <= E.M) && (n0 + BN <= E.N) && (E.c8 ? ((reinterpret_cast<uintptr_t>(E.c8) & 3) == 0) : al16(E.C)) && ((E.ldc & 3) == 0) && (!E.bias || al16(E.bias)) &&
            (!E.res || (al16(E.res) && (E.ldres & 3) == 0)) && (!E.gate || (al16(E.gate) && (E.ldgate & 3) == 0)) &&
            (!E.C16 || (reinterpret_cast<uintptr_t>(E.C16) & 7) == 0);                  // block-uniform
    return E;
}

// EPI selects the set of term combinations compiled in: 0 = all 16 (dropout site / gate / residual / accumulate), 1 = none
// (the weight-gradient table), 2 = {-, residual} x {-, GELU} (the in-loop text encoder's launches: 256x128 tiles hold 128
// accumulator registers, the full set would spill - a fifth variant already sends the accumulators to scratch), 3 = bias only,
// with or without the fp32 store (GF_NO_F32)
template <int MI, int NI, int BM, int BN, int EPI = 0>
__device__ __forceinline__ void ring_epilogue(const GemmBatch& gb, const RingEpi& E, f32x16 (&acc)[MI][NI], int m0, int n0,
                                              int lane, int wm, int wn, char* ep) {
    const int M = E.M, N = E.N;
    const float* __restrict__ bias = E.bias;
    const float* __restrict__ res = E.res;
    const float* __restrict__ gate = E.gate;
    float* __restrict__ C = E.C;
    uint16_t* __restrict__ C16 = E.C16;
    uint8_t* __restrict__ c8 = EPI == 4 ? E.c8 : nullptr;
    const float c8s = E.c8_scale;
    const int ldc = E.ldc, ldres = E.ldres, ldgate = E.ldgate;
    const float gscale = E.gscale;
    const bool relu_out = E.relu_out, accum = E.accum, vec = E.vec, w32 = !E.no32;
    const uint32_t site = E.site, key = E.key;
    // F = which optional terms this launch has (block-uniform): 1 dropout site, 2 gate, 4 residual, 8 accumulate.  As runtime
    // branches inside the per-element code they made the epilogue ~45 instructions per element (10k cycles per 128x128
    // tile, more than its k-loop) and every conditional load was followed by its own full wait; as a template parameter the
    // absent terms cost nothing - no instructions, no registers.
    auto element = [&](auto ftag, float a, float bv, float rv, float gv, float cv, int row, int col) {
        constexpr int F = decltype(ftag)::value;
        float x = (EPI == 4 ? a * E.acc_scale : a) + bv;
        x = relu_out ? fmaxf(x, 0.f) : x;
        if constexpr (F & 16) x = m2f_gelu<true>(x);                 // (the polynomial erf of the bf16 kernels, gemm.hip)
        if constexpr (F & 1) x = m2f_keep(key, (uint32_t)row * (uint32_t)N + (uint32_t)col, gb.drop_thresh) ? x * gb.drop_scale : 0.f;
        if constexpr (F & 4) x = x + rv; else x = x + 0.f;
        if constexpr (F & 2) x = gv > 0.f ? x * gscale : 0.f;
        if constexpr (F & 8) return x + cv; else return x + 0.f;
    };
    auto element_rt = [&](float a, float bv, float rv, float gv, float cv, int row, int col) {      // edge tiles
        float x = (EPI == 4 ? a * E.acc_scale : a) + bv;
        if (relu_out) x = fmaxf(x, 0.f);
        if ((EPI == 2 || EPI == 4) && E.gelu) x = m2f_gelu<true>(x);
        if (site) x = m2f_keep(key, (uint32_t)row * (uint32_t)N + (uint32_t)col, gb.drop_thresh) ? x * gb.drop_scale : 0.f;
        x = x + rv;
        if (gate) x = gv > 0.f ? x * gscale : 0.f;
        return x + cv;
    };
    const int wrow = lane & 31, wq = lane >> 5;                     // as accumulator owner: row, 16-byte chunk 2 g + wq
    const int rrow = lane >> 3, rq = lane & 7;                      // as row-major reader: row rrow + 8 p, chunk rq
    if (vec) {
        // Loads and stores share one in-order counter (vmcnt) on gfx9: a wait for freshly issued loads is also a wait for
        // every store in front of them, i.e. for a full write round trip per pass (that was 11k cycles per 128x128 tile).
        // So the optional terms of block b + 1 are requested BEFORE block b is stored, and by the time they are needed
        // only counted, younger operations are outstanding.
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        struct Terms { f32x4 r[4], g[4], c[4]; };
        f32x4 bv[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) bv[j] = bias ? *reinterpret_cast<const f32x4*>(bias + n0 + wn * (BN / 2) + j * 32 + 4 * rq) : z;
        auto blocks = [&](auto ftag) {
            constexpr int F = decltype(ftag)::value;
            auto load_terms = [&](int i, int j, Terms& T) {
                const int rb = m0 + wm * (BM / 2) + i * 32, col = n0 + wn * (BN / 2) + j * 32 + 4 * rq;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = rb + rrow + 8 * p;
                    if constexpr (F & 4) T.r[p] = *reinterpret_cast<const f32x4*>(res + (size_t)((uint32_t)(row * ldres + col)));
                    if constexpr (F & 2) T.g[p] = *reinterpret_cast<const f32x4*>(gate + (size_t)((uint32_t)(row * ldgate + col)));
                    if constexpr (F & 8) T.c[p] = *reinterpret_cast<const f32x4*>(C + (size_t)((uint32_t)(row * ldc + col)));
                }
            };
            // (two sets of terms in registers; the 256x128 tile with its 128 accumulator registers affords that for one term)
            constexpr int NT = ((F >> 1) & 1) + ((F >> 2) & 1) + ((F >> 3) & 1);
            constexpr bool AHEAD = MI * NI <= 4 || NT <= 1;
            Terms T[2];
            if constexpr (AHEAD) load_terms(0, 0, T[0]);
#pragma unroll
            for (int b = 0; b < MI * NI; ++b) {
                const int i = b / NI, j = b % NI;
                if constexpr (AHEAD) { if (b + 1 < MI * NI) load_terms((b + 1) / NI, (b + 1) % NI, T[(b + 1) & 1]); }
                else load_terms(i, j, T[b & 1]);
                if constexpr (NT != 0) __builtin_amdgcn_sched_barrier(0);            // keep them in front of this block's stores
                const int rb = m0 + wm * (BM / 2) + i * 32, col = n0 + wn * (BN / 2) + j * 32 + 4 * rq;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(ep + wrow * 128 + (((2 * g + wq) ^ (wrow & 7)) << 4)) = v;
                }
                asm volatile("" ::: "memory");                      // same wave, in-order LDS queue; pin the compiler's order too
                __builtin_amdgcn_wave_barrier();
                f32x4 a[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) a[p] = *reinterpret_cast<const f32x4*>(ep + (rrow + 8 * p) * 128 + ((rq ^ ((rrow + 8 * p) & 7)) << 4));
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                if (b == 0) { M2F_TS(5); }
                if (b == 1) { M2F_TS(7); }
                const Terms& t = T[b & 1];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = rb + rrow + 8 * p;
                    const uint32_t oc = (uint32_t)(row * ldc + col);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = element(ftag, a[p][e], bv[j][e], (F & 4) ? t.r[p][e] : 0.f, (F & 2) ? t.g[p][e] : 1.f, (F & 8) ? t.c[p][e] : 0.f, row, col + e);
                    if constexpr (EPI == 4) {                        // fp8 launches: e4m3(result * c8_scale) instead of fp32 C (block-uniform), no shadow
                        if (c8) *reinterpret_cast<uint32_t*>(c8 + (size_t)oc) = m2f_fp8x4_bits(v[0] * c8s, v[1] * c8s, v[2] * c8s, v[3] * c8s);
                        else if constexpr (!(F & 32)) *reinterpret_cast<f32x4*>(C + (size_t)oc) = v;      // (F & 32: bf16 shadow only, GF_NO_F32)
                    } else if constexpr (!(F & 32)) *reinterpret_cast<f32x4*>(C + (size_t)oc) = v;      // F & 32: C has no fp32 reader (GF_NO_F32)
                    if (C16) {
                        uint2 hh;
                        hh.x = (uint32_t)m2f_bf16_bits(v[0]) | ((uint32_t)m2f_bf16_bits(v[1]) << 16);
                        hh.y = (uint32_t)m2f_bf16_bits(v[2]) | ((uint32_t)m2f_bf16_bits(v[3]) << 16);
                        *reinterpret_cast<uint2*>(C16 + (size_t)oc) = hh;
                    }
                }
                if (b == 0) { M2F_TS(6); }
            }
        };
        // (bit 32 = no fp32 store; never together with accumulate, which reads C)
        const int fmask = EPI == 1 ? 0 : (site ? 1 : 0) | (gate ? 2 : 0) | (res ? 4 : 0) | (accum ? 8 : 0) | (E.gelu ? 16 : 0) | (w32 ? 0 : 32);
        if constexpr (EPI == 1) blocks(std::integral_constant<int, 0>{});
        else if constexpr (EPI == 2) {
            switch (fmask & ~32) {                                  // (the launcher admits nothing else: m2f_gemm_ring_ok)
                case 0: blocks(std::integral_constant<int, 0>{}); break;
                case 4: blocks(std::integral_constant<int, 4>{}); break;
                case 16: blocks(std::integral_constant<int, 16>{}); break;
                default: blocks(std::integral_constant<int, 20>{}); break;
            }
        } else if constexpr (EPI == 4) {                            // fp8 launches: {-, residual} x {-, GELU}, always de-quantising
            if (fmask == 32) blocks(std::integral_constant<int, 32>{});       // bias only, bf16 shadow only (the packed Q / K / V projection)
            else switch (fmask & ~32) {
                case 0: blocks(std::integral_constant<int, 0>{}); break;
                case 4: blocks(std::integral_constant<int, 4>{}); break;
                case 16: blocks(std::integral_constant<int, 16>{}); break;
                default: blocks(std::integral_constant<int, 20>{}); break;
            }
        } else if constexpr (EPI == 3) {                            // bias only, with or without the fp32 store (the merged QKV in-projections)
            if (fmask & 32) blocks(std::integral_constant<int, 32>{});
            else blocks(std::integral_constant<int, 0>{});
        } else switch (fmask & ~16) {
#define M2F_RING_EP(F) case F: blocks(std::integral_constant<int, F>{}); break;
            M2F_RING_EP(0) M2F_RING_EP(1) M2F_RING_EP(2) M2F_RING_EP(3) M2F_RING_EP(4) M2F_RING_EP(5) M2F_RING_EP(6) M2F_RING_EP(7)
            M2F_RING_EP(8) M2F_RING_EP(9) M2F_RING_EP(10) M2F_RING_EP(11) M2F_RING_EP(12) M2F_RING_EP(13) M2F_RING_EP(14) M2F_RING_EP(15)
            M2F_RING_EP(32) M2F_RING_EP(33) M2F_RING_EP(34) M2F_RING_EP(35) M2F_RING_EP(36) M2F_RING_EP(37) M2F_RING_EP(38) M2F_RING_EP(39)
#undef M2F_RING_EP
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int row = m0 + wm * (BM / 2) + i * 32 + wrow, col0 = n0 + wn * (BN / 2) + j * 32 + 4 * wq;
            if (row < M) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = col0 + 8 * (r >> 2) + (r & 3);
                    if (col < N) {
                        const uint32_t oc = (uint32_t)(row * ldc + col);
                        const float v = element_rt(acc[i][j][r], bias ? bias[col] : 0.f, res ? res[(size_t)((uint32_t)(row * ldres + col))] : 0.f,
                                                gate ? gate[(size_t)((uint32_t)(row * ldgate + col))] : 1.f, accum ? C[(size_t)oc] : 0.f, row, col);
                        C[(size_t)oc] = v;                                  // (edge tiles keep the fp32 store whatever GF_NO_F32 says; the fp8
                                                                            // launcher sends e4m3-result launches with edge tiles elsewhere)
                        if (C16) C16[(size_t)oc] = m2f_bf16_bits(v);
                    }
                }
            }
        }
    }
}

// =========================================================================================================
// RING form of the k-contiguous bf16 GEMM (forward / input-gradient launches of a step whose tile count fills the chip).
//
// What bounds gemm16_body on these launches is neither L2 nor HBM (profiles/r02: 290-cycle average L1->L2 latency, 84 % L2
// hits, yet 37 GB/s per CU) but its own pipeline: per k-tile the producers move every operand byte global -> VGPR -> LDS
// (ds_write_b128: 8 LDS-array cycles per KiB, twice what the fragment reads of the same bytes cost), the consumers read ALL
// fragments of the k-tile, wait, then run the MFMAs - LDS and MFMA never overlap inside a wave and there is one consumer
// wave per SIMD.  This form changes both ends:
//   * staging is LDS-direct (buffer_load_dwordx4 ... lds): no destination registers, no ds_write pass, S k-tiles of
//     BM x 64 + BN x 64 in a ring (S - 1 in flight per workgroup); out-of-range rows / k-chunks are range-checked away by
//     the buffer descriptor and land as zeros, so edge tiles need no masks;
//   * the LDS image is the lane-linear one LDS-DMA produces - 128-byte rows, 16-byte chunk c of row r stored at position
//     c ^ ((r >> 1) & 7) (the source address is permuted, not the destination) - which makes every ds_read_b128 fragment
//     read conflict-free (MI355X_MICROARCH.md, LDS table: the four 16-lane groups of a b128 read);
//   * consumers own (BM/2) x (BN/2) per wave (64x64 or 128x64: 1 KiB or 0.75 KiB of fragments per MFMA instead of 2) and
//     prefetch the fragments of k-slice s+1 while the MFMAs of slice s run;
//   * one workgroup barrier per k-tile; the producers run S - 1 k-tiles ahead ACROSS output tiles, so the next tile's
//     first operands arrive under the epilogue of the current one (persistent tile loop, grid <= number of CUs).
// Same accumulation order per output element as gemm16_body (k ascending, one MFMA chain) and the same epilogue operations per
// element (ring_epilogue), so the results are bit-identical to the register-staged builds (tests/test_gemm_ring_gpu.py).
// Tile configurations: 128x128 / 128x64 / 64x64 (4-5 slots) for the M2FNet step by launch size, 256x128 (3 slots) for
// text-encoder-sized launches, and the table forms of the weight-gradient launch (TABLE; RC = row-major operands).
// =========================================================================================================
typedef __amdgpu_buffer_rsrc_t m2f_rsrc_t;
template <int BM, int BN, int S>
struct RingCfg {
    static constexpr int BK = 64, ROW = BK * 2;
    static constexpr int A_BYTES = BM * ROW, B_BYTES = BN * ROW, SLOT = A_BYTES + B_BYTES, LDS = S * SLOT;
    static constexpr int A_INSTR = A_BYTES / 1024 / 4, B_INSTR = B_BYTES / 1024 / 4;     // per producer wave and k-tile (1 KiB = 8 rows each)
    static constexpr int PER_WAVE = A_INSTR + B_INSTR;
    static constexpr int LDS_ALL = LDS + 4 * 4096;               // + one 32x32 fp32 epilogue image per consumer wave
    static_assert(LDS_ALL <= 160 * 1024 && (S - 2) * PER_WAVE <= 63, "LDS budget / vmcnt range");
};

template <int N>
__device__ __forceinline__ void ring_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ int ring_problem_of(const GemmBatch& gb, int bpos) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < M2F_GEMM_MAX_PROBLEMS; ++i)
        if (bpos >= gb.tb[i]) pi = i;
    return pi;
}

// The fields both roles need of the problem a tile belongs to.  Grouped launches: the compact header in the kernel arguments.
// TABLE form (the weight-gradient launch, ~100 problems): tile -> problem through gb.tile_prob, the problem from the device
// table; the index is made provably uniform so that these are scalar loads.
struct RingDesc {
    const uint16_t* aq[2]; const uint16_t* bq[2];
    int M, N, k[2], ldaq[2], ldbq[2], pi, m0, n0;
    uint32_t flags;
    float* bias_grad;           // TABLE + RC form only (the weight-gradient launch sums the bias gradients itself)
};
template <bool TABLE, int BM, int BN>
__device__ __forceinline__ RingDesc ring_desc(const GemmBatch& gb, int bpos) {
    RingDesc D;
    if constexpr (TABLE) {
        const uint32_t rec = (uint32_t)__builtin_amdgcn_readfirstlane((int)gb.tile_rec[bpos]);
        D.pi = (int)(rec & 0xFFFFu); D.m0 = (int)((rec >> 16) & 0xFFu) * BM; D.n0 = (int)(rec >> 24) * BN;
        const GemmProblem& P = gb.table[D.pi];
        D.aq[0] = P.a.q[0]; D.aq[1] = P.a.q[0]; D.bq[0] = P.b.q[0]; D.bq[1] = P.b.q[0];
        D.M = P.M; D.N = P.N; D.k[0] = P.a.k[0]; D.k[1] = 0;
        D.ldaq[0] = P.a.ldq[0]; D.ldaq[1] = P.a.ldq[0]; D.ldbq[0] = P.b.ldq[0]; D.ldbq[1] = P.b.ldq[0];
        D.flags = P.flags; D.bias_grad = P.bias_grad;      // (k-contiguous tables carry neither)
    } else {
        D.pi = ring_problem_of(gb, bpos);
        const GemmHot& H = gb.hot[D.pi];
        D.aq[0] = H.aq[0]; D.aq[1] = H.aq[1]; D.bq[0] = H.bq[0]; D.bq[1] = H.bq[1];
        D.M = H.M; D.N = H.N; D.k[0] = H.k[0]; D.k[1] = H.k[1];
        D.ldaq[0] = H.ldaq[0]; D.ldaq[1] = H.ldaq[1]; D.ldbq[0] = H.ldbq[0]; D.ldbq[1] = H.ldbq[1];
        D.flags = H.flags; D.bias_grad = nullptr;
        const int tl = bpos - H.tile_begin, tiles_m = (H.M + BM - 1) / BM;
        D.m0 = (tl % tiles_m) * BM; D.n0 = (tl / tiles_m) * BN;
    }
    return D;
}

// (the two roles are functions of their own: with the producer's lambdas inside the __global__ template hipcc emitted no host
// stub for the kernel - no diagnostic, an undefined symbol at load time)
template <int BM, int BN, int S, bool TABLE, bool RC>
__device__ __forceinline__ void ring_producer(const GemmBatch& gb, char* smem, int wave, int lane, int first, int grid, int total_tiles) {
    using C = RingCfg<BM, BN, S>;
    constexpr int BK = C::BK;
    typedef __attribute__((address_space(3))) void lds_void;
    // ================================ PRODUCER: global -> LDS ring (no registers) ================================
    // the k-loops are bound by how fast the LDS-DMA instructions get issued: the producer waves go ahead of the MFMA waves at the
    // SIMDs' issue arbiters (levels 1 / 2 / 3 alike: C3 fwd+bwd -11 .. -18 us in two same-box A/Bs, profiles/r03_dev_producer_priority_ab.txt)
    __builtin_amdgcn_s_setprio(2);
    auto rsrc_of = [](const uint16_t* q, int rows, int ld) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(q);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(rows * ld * 2), 0x00020000);
    };
    constexpr unsigned OOB = 0x80000000u;
    // lane -> (row within the 8-row piece, chunk position); the chunk stored at this position is pos ^ ((row >> 1) & 7)
    const int lrow = lane >> 3, pos = lane & 7;
    const int wv = __builtin_amdgcn_readfirstlane(wave);       // provably uniform: LDS destinations go through M0
    // issue cursor: k-tile ikt of output tile ibpos goes to ring slot islot.  Per (tile, segment) the byte offset of
    // every piece's lane at k = 0 is kept in registers, a k-tile adds one uniform term: one v_add per load in the loop
    int ibpos = first, ikt = 0, islot = 0, issued = 0;
    int im0 = 0, in0 = 0, ink0 = 0, ink = 0, kp0 = 0, kp1 = 0, ald0 = 0, ald1 = 0, bld0 = 0, bld1 = 0;
    unsigned offA[C::A_INSTR], offB[C::B_INSTR];
    int kpad = 0, kcur = 0;                                     // current segment: padded length, k of the next k-tile
    unsigned stepA = 0, stepB = 0;                              // RC form: bytes per k-tile step (BK rows)
    m2f_rsrc_t ra0, ra1, rb0, rb1, ra, rb;
    bool idone = ibpos >= total_tiles;
    M2F_TS(0);
    auto set_segment = [&](int seg) {
        const int lda = seg ? ald1 : ald0, ldb = seg ? bld1 : bld0;
        if (seg) { ra = ra1; rb = rb1; } else { ra = ra0; rb = rb0; }
        kpad = seg ? kp1 : kp0; kcur = 0;
        if constexpr (RC) {
            // RC operands ([k][row] in memory, e.g. the [token][feature] activations of the weight-gradient launch): the LDS
            // image is k-major, 256-byte rows of 128 features; a 1 KiB piece = 4 k-rows, lane -> (k-row lane >> 4, chunk
            // position lane & 15), the 16-byte chunk stored at a position is pos ^ (4 * (k-row & 3)) - which spreads the four
            // k-rows of a transposing fragment read (ds_read_b64_tr_b16) over four disjoint bank ranges
            // (a 256-wide operand is two such images of 128 features behind each other: piece p belongs to image p / 16)
            static_assert(!RC || ((BM == 128 || BM == 256) && BN == 128), "RC form: 256-byte tile rows");
            const int kr = lane >> 4, p16 = lane & 15;
#if defined(M2F_RING_EXP_ADDR)      // timing experiment (results are wrong): 8 k-rows x 128 bytes per piece instead of 4 x 256
            const int kr8 = lane >> 3, c8 = lane & 7;
#pragma unroll
            for (int j = 0; j < C::A_INSTR; ++j) {
                const int pc = wv * C::A_INSTR + j;
                offA[j] = (unsigned)(((pc & 7) * 8 + kr8) * lda + im0 + 64 * (pc >> 3)) * 2u + 16u * (unsigned)(c8 ^ (kr8 & 7));
            }
#pragma unroll
            for (int j = 0; j < C::B_INSTR; ++j) {
                const int pc = wv * C::B_INSTR + j;
                offB[j] = (unsigned)(((pc & 7) * 8 + kr8) * ldb + in0 + 64 * (pc >> 3)) * 2u + 16u * (unsigned)(c8 ^ (kr8 & 7));
            }
            (void)kr; (void)p16;
#else
#pragma unroll
            for (int j = 0; j < C::A_INSTR; ++j) {
                const int pc = wv * C::A_INSTR + j;
                offA[j] = (unsigned)(((pc & 15) * 4 + kr) * lda + im0 + 128 * (pc >> 4)) * 2u + 16u * (unsigned)(p16 ^ (4 * kr));
            }
#pragma unroll
            for (int j = 0; j < C::B_INSTR; ++j)
                offB[j] = (unsigned)(((wv * C::B_INSTR + j) * 4 + kr) * ldb + in0) * 2u + 16u * (unsigned)(p16 ^ (4 * kr));
#endif
            stepA = (unsigned)(BK * lda) * 2u; stepB = (unsigned)(BK * ldb) * 2u;
            return;
        }
#pragma unroll
        for (int j = 0; j < C::A_INSTR; ++j) {
            const int r = (wv * C::A_INSTR + j) * 8 + lrow;             // row within the tile
            offA[j] = (unsigned)((im0 + r) * lda + 8 * (pos ^ ((r >> 1) & 7))) * 2u;
        }
#pragma unroll
        for (int j = 0; j < C::B_INSTR; ++j) {
            const int r = (wv * C::B_INSTR + j) * 8 + lrow;
            offB[j] = (unsigned)((in0 + r) * ldb + 8 * (pos ^ ((r >> 1) & 7))) * 2u;
        }
    };
    auto load_desc = [&]() {
        const RingDesc H = ring_desc<TABLE, BM, BN>(gb, ibpos);
        im0 = H.m0; in0 = H.n0;
        ink0 = (H.k[0] + BK - 1) / BK; ink = ink0 + (H.k[1] + BK - 1) / BK;
        kp0 = (H.k[0] + 7) & ~7; kp1 = (H.k[1] + 7) & ~7;
        ald0 = H.ldaq[0]; ald1 = H.ldaq[1]; bld0 = H.ldbq[0]; bld1 = H.ldbq[1];
        const bool two = H.k[1] > 0;
        // (RC form: the rows of the buffer are the k index - loads past the reduction length are range-checked to zero)
        ra0 = rsrc_of(H.aq[0], RC ? H.k[0] : H.M, ald0); rb0 = rsrc_of(H.bq[0], RC ? H.k[0] : H.N, bld0);
        ra1 = rsrc_of(two ? H.aq[1] : H.aq[0], H.M, two ? ald1 : ald0);
        rb1 = rsrc_of(two ? H.bq[1] : H.bq[0], H.N, two ? bld1 : bld0);
        set_segment(0);
    };
    if (!idone) load_desc();
    auto issue_next = [&]() {
        if (ikt == ink0 && ikt > 0) set_segment(1);
        char* dstA = smem + islot * C::SLOT + wv * (C::A_INSTR * 1024);
        char* dstB = smem + islot * C::SLOT + C::A_BYTES + wv * (C::B_INSTR * 1024);
        if constexpr (RC) {
            const unsigned ka = (unsigned)(kcur / BK) * stepA, kbb = (unsigned)(kcur / BK) * stepB;
#if defined(M2F_RING_EXP) && M2F_RING_EXP == 1      // experiment: no loads (consumer floor)
            if (false)
#endif
            {
#pragma unroll
            for (int j = 0; j < C::A_INSTR; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(dstA + j * 1024), 16, offA[j] + ka, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < C::B_INSTR; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(dstB + j * 1024), 16, offB[j] + kbb, 0, 0, 0);
            }
        } else {
        const unsigned kb = (unsigned)kcur * 2u;
#if defined(M2F_RING_EXP) && M2F_RING_EXP == 1      // experiment: no loads (consumer floor)
        if (false) {
#else
        if (kcur + BK <= kpad) {
#endif                                // whole k-tile inside the (padded) reduction length
#pragma unroll
            for (int j = 0; j < C::A_INSTR; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(dstA + j * 1024), 16, offA[j] + kb, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < C::B_INSTR; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(dstB + j * 1024), 16, offB[j] + kb, 0, 0, 0);
#if defined(M2F_RING_EXP) && M2F_RING_EXP == 1
        } else if (false) {
#else
        } else {                                                // tail k-tile: chunks past the padded length land as zeros
#endif
#pragma unroll
            for (int j = 0; j < C::A_INSTR; ++j) {
                const int r = (wv * C::A_INSTR + j) * 8 + lrow;
                const bool in = kcur + 8 * (pos ^ ((r >> 1) & 7)) < kpad;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(dstA + j * 1024), 16, in ? offA[j] + kb : OOB, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < C::B_INSTR; ++j) {
                const int r = (wv * C::B_INSTR + j) * 8 + lrow;
                const bool in = kcur + 8 * (pos ^ ((r >> 1) & 7)) < kpad;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(dstB + j * 1024), 16, in ? offB[j] + kb : OOB, 0, 0, 0);
            }
        }
        }
        kcur += BK;
        ++issued;
        islot = islot + 1 == S ? 0 : islot + 1;
        if (++ikt == ink) {
            ikt = 0; ibpos += grid;
            idone = ibpos >= total_tiles;
            if (!idone) load_desc();
        }
    };
    // k-tile g (counted over this workgroup's whole tile sequence) has landed once at most the pieces issued after it
    // are outstanding: loads of one wave complete in issue order
    auto wait_landed = [&](int g) {
        const int younger = issued - (g + 1);
        if (younger <= 0) ring_wait_vm<0>();
        else if (younger == 1) ring_wait_vm<C::PER_WAVE>();
        else if (S < 4 || younger == 2) ring_wait_vm<(S > 3 ? 2 : 1) * C::PER_WAVE>();
        else ring_wait_vm<(S > 4 ? 3 : 1) * C::PER_WAVE>();
    };
    static_assert(S >= 3 && S <= 5, "ring depth");
#pragma unroll 1
    M2F_TS(1);
    for (int t = 0; t < S - 1; ++t)
        if (!idone) issue_next();
    M2F_TS(2);
    wait_landed(0);
    M2F_TS(3);
    ring_lds_barrier();                                              // (#0) k-tile 0 is in the ring
    M2F_TS(4);
    int g = 0;
    M2F_ACC_DECL;
#pragma unroll 1
    for (int bpos = first; bpos < total_tiles; bpos += grid) {
        const RingDesc H = ring_desc<TABLE, BM, BN>(gb, bpos);
        const int nk = (H.k[0] + BK - 1) / BK + (H.k[1] + BK - 1) / BK;
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt, ++g) {
            // the consumers multiply k-tile g; the slot of k-tile g - 1 was released at the last barrier
            const unsigned long long tp0 = M2F_NOW();
            if (!idone) issue_next();
            const unsigned long long tp1 = M2F_NOW();
            if (issued > g + 1) wait_landed(g + 1);
            const unsigned long long tp2 = M2F_NOW();
            ring_lds_barrier();                                      // (#g+1)
            M2F_ADD(8, tp1 - tp0); M2F_ADD(9, tp2 - tp1); M2F_ADD(10, M2F_NOW() - tp2); M2F_ADD(11, 1);
        }
    }
    M2F_TS(5);
    M2F_ACC_FLUSH();
}

template <int BM, int BN, int S, bool TABLE, bool RC, int EPI>
__device__ __forceinline__ void ring_consumer(const GemmBatch& gb, char* smem, int wave, int lane, int first, int grid, int total_tiles) {
    using C = RingCfg<BM, BN, S>;
    constexpr int MI = BM / 64, NI = BN / 64, BK = C::BK;
    // ==================================== CONSUMER: LDS -> MFMA -> epilogue ====================================
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, x = ((lane & 31) >> 1) & 7;
    int fo[BK / 16];                                            // swizzled chunk offset of k-slice ks for this lane
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) fo[ks] = ((2 * ks + h) ^ x) << 4;
    const int arow = (wm * (BM / 2) + (lane & 31)) * C::ROW, brow = C::A_BYTES + (wn * (BN / 2) + (lane & 31)) * C::ROW;
    // RC form (k-major image, 256-byte rows): per 16-lane group g, lane 4 q + p supplies the address of k-row
    // 8 (g >> 1) + q (+ 4 for the second read), features 16 (g & 1) + 4 p .. + 3 of the 32-feature block; lane i of the
    // group receives feature i.  Chunk index of those 8 bytes = (block base + 16 (g & 1)) / 8 + (p >> 1), XOR 4 q (the image's swizzle).
    int rc_row = 0, rc_a[MI], rc_b[NI];
    {
        const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
        rc_row = (8 * (g >> 1) + q) * 256 + 8 * (pp & 1);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int r0 = wm * (BM / 2) + i * 32;              // first tile row of the block: image r0 / 128, feature r0 % 128 of it
            rc_a[i] = (r0 >> 7) * (BK * 256) + (((((r0 & 127) + 16 * (g & 1)) / 8 + (pp >> 1)) ^ (4 * q)) << 4);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) rc_b[j] = (((wn * (BN / 2) + j * 32 + 16 * (g & 1)) / 8 + (pp >> 1)) ^ (4 * q)) << 4;
    }
    int slot = 0;
    M2F_TS(0);
    M2F_ACC_DECL;
#pragma unroll 1
    for (int bpos = first; bpos < total_tiles; bpos += grid) {
        const unsigned long long tc0 = M2F_NOW();
        const RingDesc H = ring_desc<TABLE, BM, BN>(gb, bpos);
        const GemmProblem& P = TABLE ? gb.table[H.pi] : gb.pr[H.pi];
        const int m0 = H.m0, n0 = H.n0;
        const int nk = (H.k[0] + BK - 1) / BK + (H.k[1] + BK - 1) / BK;
        const bool reluA = H.flags & GF_RELU_A, reluB = RC && (H.flags & GF_RELU_B);
        const bool bgrad = RC && H.bias_grad && n0 == 0 && wn == 0;      // wave-uniform: this wave sums its rows of A over k
        float bsum[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) bsum[i] = 0.f;
        f32x16 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // the operand options (block-uniform: ReLU on A / on B, bias-gradient sums) select a copy of the loop: a branch inside
        // would cut the k-tile into basic blocks and keep the scheduler from interleaving fragment reads and MFMAs
        auto kloop = [&](auto ra_tag, auto rb_tag, auto bg_tag) {
            constexpr bool RELU_A = decltype(ra_tag)::value, RELU_B = decltype(rb_tag)::value, BGRAD = decltype(bg_tag)::value;
            auto relu8 = [](bf16x8 f) {
                ring_u32x4 w = __builtin_bit_cast(ring_u32x4, f);
                w.x = ring_relu_bf16x2(w.x); w.y = ring_relu_bf16x2(w.y); w.z = ring_relu_bf16x2(w.z); w.w = ring_relu_bf16x2(w.w);
                return __builtin_bit_cast(bf16x8, w);
            };
#pragma unroll 1
            for (int kt = 0; kt < nk; ++kt) {
                ring_lds_barrier();                                      // (#g) k-tile g has landed
                const char* ab = smem + slot * C::SLOT + arow;
                const char* bb = smem + slot * C::SLOT + brow;
                if constexpr (EPI == 4) {
                    // FP8 (OCP e4m3) operands: the producers moved the same 128-byte rows (the launcher hands over k and the leading
                    // dimensions in byte PAIRS), a row now holds 128 k-values.  v_mfma_scale_f32_32x32x64_f8f6f4 (scale operands 0 =
                    // unscaled; 2x the bf16 rate): lane (row l & 31, half l >> 5) supplies k = 64 s + 32 (l >> 5) + 0..31 of its row,
                    // i.e. 16-byte chunks 4 s + 2 h and 4 s + 2 h + 1 of the swizzled image (tools/mfma_probe/f8f6f4_probe.hip).
                    typedef int ring_v8i __attribute__((ext_vector_type(8)));
                    const int h2 = (lane >> 5) * 2;
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const int o0 = ((4 * st + h2) ^ x) << 4, o1 = ((4 * st + h2 + 1) ^ x) << 4;
                        ring_v8i f8a[MI], f8b[NI];
#pragma unroll
                        for (int i = 0; i < MI; ++i) {
                            const ring_u32x4 lo = *reinterpret_cast<const ring_u32x4*>(ab + i * 32 * C::ROW + o0);
                            const ring_u32x4 hi = *reinterpret_cast<const ring_u32x4*>(ab + i * 32 * C::ROW + o1);
                            f8a[i] = (ring_v8i){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
                        }
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            const ring_u32x4 lo = *reinterpret_cast<const ring_u32x4*>(bb + j * 32 * C::ROW + o0);
                            const ring_u32x4 hi = *reinterpret_cast<const ring_u32x4*>(bb + j * 32 * C::ROW + o1);
                            f8b[j] = (ring_v8i){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
                        }
#pragma unroll
                        for (int i = 0; i < MI; ++i)
#pragma unroll
                            for (int j = 0; j < NI; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(f8b[j], f8a[i], acc[i][j], 0, 0, 0, 0, 0, 0);   // operands swapped, as below
                    }
                    slot = slot + 1 == S ? 0 : slot + 1;
                    continue;
                }
                constexpr int KS = BK / 16;
                bf16x8 fa[2][MI], fb[2][NI];
                auto frags = [&](int ks, int buf) {
                    if constexpr (RC) {
                        // MFMA operand (8 consecutive k of one row) out of the k-major image: two transposing reads of 4 k-rows
                        const char* img = smem + slot * C::SLOT + ks * 16 * 256 + rc_row;
#pragma unroll
                        for (int i = 0; i < MI; ++i) {
                            const char* q = img + rc_a[i];
                            const ring_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ring_s16x4 __attribute__((address_space(3)))*)(const_cast<char*>(q)));
                            const ring_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ring_s16x4 __attribute__((address_space(3)))*)(const_cast<char*>(q + 4 * 256)));
                            const ring_s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            fa[buf][i] = __builtin_bit_cast(bf16x8, r);
                        }
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            const char* q = img + C::A_BYTES + rc_b[j];
                            const ring_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ring_s16x4 __attribute__((address_space(3)))*)(const_cast<char*>(q)));
                            const ring_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ring_s16x4 __attribute__((address_space(3)))*)(const_cast<char*>(q + 4 * 256)));
                            const ring_s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            fb[buf][j] = __builtin_bit_cast(bf16x8, r);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < MI; ++i) fa[buf][i] = *reinterpret_cast<const bf16x8*>(ab + i * 32 * C::ROW + fo[ks]);
#pragma unroll
                        for (int j = 0; j < NI; ++j) fb[buf][j] = *reinterpret_cast<const bf16x8*>(bb + j * 32 * C::ROW + fo[ks]);
                    }
                    if constexpr (BGRAD) {                          // bias gradient = sum over k of A's rows (before any ReLU on A)
#pragma unroll
                        for (int i = 0; i < MI; ++i) {
                            const ring_u32x4 w = __builtin_bit_cast(ring_u32x4, fa[buf][i]);
                            bsum[i] = ring_bf16x2_sum(w.w, ring_bf16x2_sum(w.z, ring_bf16x2_sum(w.y, ring_bf16x2_sum(w.x, bsum[i]))));
                        }
                    }
                    if constexpr (RELU_A) {                         // e.g. relu(cat(x, text)) of the fusion layer's Linear
#pragma unroll
                        for (int i = 0; i < MI; ++i) fa[buf][i] = relu8(fa[buf][i]);
                    }
                    if constexpr (RELU_B) {
#pragma unroll
                        for (int j = 0; j < NI; ++j) fb[buf][j] = relu8(fb[buf][j]);
                    }
                };
#if defined(M2F_RING_EXP) && M2F_RING_EXP == 2      // experiment: no fragment reads / MFMAs (producer floor)
                if (false)
#endif
                {
                frags(0, 0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + 1 < KS) frags(ks + 1, (ks + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);              // the next slice's reads go out BEFORE this slice's MFMAs
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks & 1][j], fa[ks & 1][i], acc[i][j], 0, 0, 0);   // operands swapped: ring_epilogue
                }
                }
                slot = slot + 1 == S ? 0 : slot + 1;
            }
        };
        const RingEpi E = ring_epilogue_args<BM, BN>(gb, P, m0, n0);
        M2F_TS(1);
        const unsigned long long tc1 = M2F_NOW();
        {
            const std::true_type T1{}; const std::false_type F0{};
            if constexpr (RC) {
                const int sel = (reluA ? 1 : 0) | (reluB ? 2 : 0) | (bgrad ? 4 : 0);
                switch (sel) {
                    case 0: kloop(F0, F0, F0); break;  case 1: kloop(T1, F0, F0); break;
                    case 2: kloop(F0, T1, F0); break;  case 3: kloop(T1, T1, F0); break;
                    case 4: kloop(F0, F0, T1); break;  case 5: kloop(T1, F0, T1); break;
                    case 6: kloop(F0, T1, T1); break;  default: kloop(T1, T1, T1); break;
                }
            } else {
                if (reluA) kloop(T1, F0, F0);
                else kloop(F0, F0, F0);
            }
        }
        if constexpr (RC) {
            if (bgrad) {                                            // lanes l and l + 32 hold the two k-halves of row l
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const float tot = bsum[i] + __shfl_xor(bsum[i], 32);
                    const int m = m0 + wm * (BM / 2) + i * 32 + (lane & 31);
                    if (lane < 32 && m < H.M) H.bias_grad[m] = tot;
                }
            }
        }
        M2F_TS(3);
        const unsigned long long tc2 = M2F_NOW();
        ring_epilogue<MI, NI, BM, BN, EPI>(gb, E, acc, m0, n0, lane, wm, wn, smem + C::LDS + wave * 4096);
        M2F_TS(4);
        M2F_ADD(8, tc1 - tc0); M2F_ADD(9, tc2 - tc1); M2F_ADD(10, M2F_NOW() - tc2); M2F_ADD(11, 1);
    }
    ring_lds_barrier();                                                  // matches the producers' last barrier
    M2F_ACC_FLUSH();
}

template <int BM, int BN, int S, bool TABLE, bool RC = false, int EPI = (TABLE ? 1 : 0)>
__global__ __launch_bounds__(512) void m2f_gemm16_ring_kernel(const GemmBatch gb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;      // role-local ids
    // everything both roles read before the first operand load - tb[], hot[] (bytes 0..735), count .. the hidden launch
    // geometry (2900..3100) - and, for grouped launches, pr[0] / pr[1] (736..1280), which the epilogue reads
    // (the second range must reach from `count` over the end of the struct into the hidden launch geometry behind it)
    static_assert(offsetof(GemmBatch, count) >= 2880 - 256 && sizeof(GemmBatch) + 64 <= 2880 - 256 + 512 && offsetof(GemmBatch, pr) + 2 * sizeof(GemmProblem) <= 24 * 64,
                  "m2f_kernarg_warm ranges no longer cover GemmBatch: recompute them");
    if constexpr (TABLE) m2f_kernarg_warm<0, 8, 2880 - 256>();
    else m2f_kernarg_warm<0, 24, 2880 - 256>();
    // grouped launches: workgroup b walks tiles remap(b), + grid, ... of the launch's tile list; TABLE form: its own list
    // (gb.tile_rec[gb.wg_begin[b] .. gb.wg_begin[b + 1]), built by m2f_gemm_table_walk)
    const int grid = TABLE ? 1 : (int)gridDim.x;
    const int total_tiles = TABLE ? __builtin_amdgcn_readfirstlane(gb.wg_begin[blockIdx.x + 1]) : gb.total_tiles;
    const int first = TABLE ? __builtin_amdgcn_readfirstlane(gb.wg_begin[blockIdx.x]) : ring_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    if (threadIdx.x >= 256) ring_producer<BM, BN, S, TABLE, RC>(gb, smem, wave, lane, first, grid, total_tiles);      // wave-uniform
    else ring_consumer<BM, BN, S, TABLE, RC, EPI>(gb, smem, wave, lane, first, grid, total_tiles);
}

template <int BM, int BN, int S, bool TABLE, bool RC = false, int EPI = (TABLE ? 1 : 0)>
hipError_t launch_ring_grid(const GemmBatch& hb, int t, hipStream_t stream) {
    using C = RingCfg<BM, BN, S>;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    auto kern = m2f_gemm16_ring_kernel<BM, BN, S, TABLE, RC, EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_ALL);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    ++m2f_g_ring_launches;
    if constexpr (TABLE) {
        if (!hb.tile_rec || !hb.wg_begin || hb.wg_count < 1) return hipErrorInvalidValue;
        hipLaunchKernelGGL(kern, dim3(hb.wg_count), dim3(512), C::LDS_ALL, stream, hb);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(t < n_cu ? t : n_cu), dim3(512), C::LDS_ALL, stream, hb);
    return hipGetLastError();
}


template <int BM, int BN, int S, int EPI = 0>
hipError_t launch_ring16(GemmBatch& gb, hipStream_t stream) {
    int t = 0;
    for (int i = 0; i < gb.count; ++i) {
        GemmProblem& p = gb.pr[i];
        p.splitk = 1; p.slab_begin = 0; p.cnt_begin = 0;
        p.tile_begin = t;
        p.tiles_n = m2f_cdiv(p.N, BN);
        t += m2f_cdiv(p.M, BM) * p.tiles_n;
    }
    if (t == 0) return hipSuccess;
    GemmBatch hb = gb;
    for (int i = 0; i < M2F_GEMM_MAX_PROBLEMS; ++i) {
        hb.tb[i] = i < gb.count ? gb.pr[i].tile_begin : 0x7fffffff;
        GemmHot& h = hb.hot[i];
        memset(&h, 0, sizeof(h));
        if (i >= gb.count) continue;
        const GemmProblem& p = gb.pr[i];
        h.aq[0] = p.a.q[0]; h.aq[1] = p.a.q[1]; h.bq[0] = p.b.q[0]; h.bq[1] = p.b.q[1];
        h.M = p.M; h.N = p.N; h.k[0] = p.a.k[0]; h.k[1] = p.a.k[1];
        h.ldaq[0] = p.a.ldq[0]; h.ldaq[1] = p.a.ldq[1]; h.ldbq[0] = p.b.ldq[0]; h.ldbq[1] = p.b.ldq[1];
        h.flags = p.flags; h.tile_begin = p.tile_begin; h.has_bias_grad = 0;
    }
    hb.total_tiles = t;
    return launch_ring_grid<BM, BN, S, false, false, EPI>(hb, t, stream);
}

}  // namespace
