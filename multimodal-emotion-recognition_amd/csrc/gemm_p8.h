// 256x256 bf16 GEMM tiles on the EIGHT-PHASE schedule (cdna_hip_programming.md section 5, "The 256^2 8-phase template"): device code +
// launch templates.  Round 4.  Replaces the 4-producer / 4-consumer ring form (gemm_ring.h) where a launch is large enough that the
// main loop, not the launch's fixed cost, sets its time: the weight-gradient table launch (row-major RC operands) and the in-loop text
// encoder's launches (k-contiguous KC operands, M = utterances x tokens).
//
// Why another form.  In the ring form four waves only move bytes and four multiply, one MFMA wave per SIMD: its k-loops run at
// 1,530-2,130 cycles per 256x128x64 k-tile for 1,024 cycles of MFMA, i.e. 31-33 % of the bf16 peak at
// best.  Here all eight waves load AND multiply:
//   * tile 256 x 256 x 64, wave (wr, wc) = (w >> 2, w & 3) owns rows {wr 64 .. +63} u {128 + wr 64 .. +63} x columns {wc 32 .. +31} u
//     {128 + wc 32 .. +31}: one 64 x 32 quadrant out of each (A half, B half) pair - 128 accumulator registers;
//   * LDS = 2 buffers x (A0, A1, B0, B1) half-tiles of 16 KiB = 128 KiB, filled by LDS-DMA (buffer_load ... lds, no staging registers),
//     every wave moving 1/8 of each half-tile (two 1 KiB pieces);
//   * a k-tile is FOUR phases = the four quadrants (A0 B0, A0 B1, A1 B1, A1 B0); a phase = {fragment reads of the half that is new
//     (12 / 4 / 8 / 0 ds_reads) + ONE half-tile of prefetch (2 LDS-DMA pieces per wave)} - barrier - {16 MFMAs 16x16x32} - barrier.
//     The waves of the lower half (wr = 1: the SIMD partners of wr = 0) run ONE BARRIER BEHIND, so on every SIMD one wave multiplies
//     while its partner reads and stages;
//   * the prefetch stream runs three half-tiles ahead of the single counted wait per k-tile (s_waitcnt vmcnt(6) in phase 4) and is
//     CONTINUOUS across the output tiles of a workgroup's list: while the last k-tiles of one tile are multiplied the first of the
//     next are already landing, and the epilogue is nothing but the stores (a lane holds four consecutive columns: 16-byte stores,
//     no LDS pass), drained under the next tile's k-loop.
// Hazards (placement, not luck - cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait that retires it"):
//   RAW  every wave waits vmcnt(6) in the load segment of phase 4; all of k-tile t + 1 was issued before the three youngest half-
//        tiles, so after the barrier that ends the LOWER half's phase-4 load segment every wave's pieces of t + 1 have landed; the
//        first reads of t + 1 come after that barrier for both halves.
//   WAR  B0 is read in phase 1 (retired by lgkmcnt(8) BEFORE the phase's first barrier) and re-staged in phase 2; A0 is read in phase
//        1 and re-staged in phase 3; B1 read in 2, re-staged in 4; A1 read in 3, re-staged in phase 1 of the next k-tile - always
//        two barriers after the reads of BOTH halves have been waited for.
// Per wave all pieces of one LDS region are written by the same wave and LDS-DMA completes in issue order, so re-staging never
// overtakes an older (range-checked, zero-filling) piece of the same region.
#pragma once
#include "gemm_ring.h"

namespace {

#define P8_ADAM_T_BYTES 2304                          // per wave: the W^T image of the Adam epilogue, 16 x 136 bytes (+ pad)
struct P8Cfg {
    static constexpr int BM = 256, BN = 256, BK = 64;
    static constexpr int HALF = 128 * BK * 2;          // one operand half-tile: 128 rows (KC) or features (RC) x 64 k, 16 KiB
    static constexpr int BUF = 4 * HALF;               // A0 | A1 | B0 | B1
    static constexpr int LDS = 2 * BUF;                // 128 KiB
};

// max(f, floor) on bf16 pairs read as int16 pairs (one v_pk_max_i16 per dword): floor = 0 is ReLU (gemm_ring.h, ring_relu_bf16x2),
// floor = 0x8000 (-32768) leaves every value as it is - the choice is a block-uniform OPERAND, not a branch
__device__ __forceinline__ bf16x8 p8_floor8(bf16x8 f, uint32_t floor2) {
    typedef short p8_s16x2 __attribute__((ext_vector_type(2)));
    ring_u32x4 w = __builtin_bit_cast(ring_u32x4, f);
    const p8_s16x2 fl = __builtin_bit_cast(p8_s16x2, floor2);
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(p8_s16x2, (uint32_t)w[e]), fl));
    return __builtin_bit_cast(bf16x8, w);
}

// One 1 KiB LDS-DMA piece (64 lanes x 16 bytes to LDS address `lds` + 16 lane) issued from INLINE ASM: with the builtin hipcc knows the
// instruction writes LDS and answers every later ds_read of the wave with s_waitcnt vmcnt(0) - the prefetch stream would be drained
// four times per k-tile.  Hidden in asm, the pieces are ordered against the fragment reads by the kernel's own counted vmcnt + barriers
// (header comment); M0 is written in the statement that uses it (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void p8_dma16(const ring_u32x4& rsrc, unsigned lds, unsigned voff) {
    // (one wait state between the write of M0 and the LDS-DMA instruction that reads it: s_nop 0; s_nop 4 cost 3-6 % at K >= 4,096)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds), "v"(voff), "s"(rsrc) : "memory");
}
// raw buffer descriptor (stride 0, range-checked against `bytes`), all words provably uniform
__device__ __forceinline__ ring_u32x4 p8_rsrc(const void* q, unsigned bytes) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(q);
    ring_u32x4 r;
    r.x = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    r.y = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32)) & 0xFFFFu;
    r.z = (unsigned)__builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000u;
    return r;
}
// What both the prefetch stream and the multiplying side need of the tile at position `bpos` of this workgroup's sequence - read with
// SCALAR loads (constant address space): a vector load here would sit in the vmcnt queue between the stream's pieces.
struct P8Desc {
    const uint16_t* aq; const uint16_t* bq;
    int M, N, K, lda, ldb, pi, m0, n0;
    uint32_t flags;
};
template <bool TABLE>
__device__ __forceinline__ P8Desc p8_desc(const GemmBatch& gb, int bpos) {
    P8Desc D;
    if constexpr (TABLE) {
        typedef const __attribute__((address_space(4))) uint32_t* c_u32p;
        typedef const __attribute__((address_space(4))) GemmProblem* c_probp;
        const uint32_t rec = *((c_u32p)(gb.tile_rec) + bpos);
        D.pi = (int)(rec & 0xFFFFu); D.m0 = (int)((rec >> 16) & 0xFFu) * P8Cfg::BM; D.n0 = (int)(rec >> 24) * P8Cfg::BN;
        c_probp P = (c_probp)(gb.table) + D.pi;
        D.aq = P->a.q[0]; D.bq = P->b.q[0]; D.M = P->M; D.N = P->N; D.K = P->a.k[0]; D.lda = P->a.ldq[0]; D.ldb = P->b.ldq[0];
        D.flags = P->flags;
    } else {
        D.pi = ring_problem_of(gb, bpos);
        const GemmHot& H = gb.hot[D.pi];
        D.aq = H.aq[0]; D.bq = H.bq[0]; D.M = H.M; D.N = H.N; D.K = H.k[0]; D.lda = H.ldaq[0]; D.ldb = H.ldbq[0]; D.flags = H.flags;
        // tile order inside a problem: bands of 8 tile rows, m fastest inside a band - 32 consecutive tiles (what one XCD's 32 workgroups
        // hold at a time: ring_xcd_remap) are 8 row panels x 4 column panels, 12 operand panels through that L2 instead of the 33 of a
        // plain m-fastest order
        const int tl = bpos - H.tile_begin, tiles_m = (H.M + P8Cfg::BM - 1) / P8Cfg::BM, tiles_n = (H.N + P8Cfg::BN - 1) / P8Cfg::BN;
        const int band = tl / (8 * tiles_n), rem = tl - band * 8 * tiles_n, rows = tiles_m - 8 * band < 8 ? tiles_m - 8 * band : 8;
        D.m0 = (8 * band + rem % rows) * P8Cfg::BM; D.n0 = (rem / rows) * P8Cfg::BN;
    }
    return D;
}

#ifdef P8_TIMING
// diagnostic build (make p8timing; tools/p8_timing.py): accumulated s_memtime differences of waves 0 and 4 of workgroup 0, per phase:
// [wave half][phase 1..4][load segment + barrier + fragment wait, MFMA segment, closing barrier] + k-tile count + epilogue cycles
__device__ unsigned long long m2f_p8_dbg[64];
#define P8_STAMP(v) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); } while (0)
#define P8_ACC(p) do { tacc[p][0] += ts1 - ts0; tacc[p][1] += ts2 - ts1; tacc[p][2] += ts3 - ts2; if ((p) == 3) ++tk; } while (0)
#else
#define P8_STAMP(v) do { (void)sizeof(v); } while (0)
#define P8_ACC(p) do {} while (0)
#endif

// rows of 16 lanes [x0 x1 x2 x3], [y0 y1 y2 y3]  ->  [x0 x2 y0 y2], [x1 x3 y1 y3]   (v_permlane32_swap, then v_permlane16_swap)
__device__ __forceinline__ void p8_swap2(uint32_t& x, uint32_t& y) {
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    auto q = __builtin_amdgcn_permlane16_swap(r[0], r[1], false, false);
    x = q[0]; y = q[1];
}

// torch.optim.Adam's update of four consecutive elements: the arithmetic of rowops.hip::adam4, contractions spelled out the same way
// (every optimizer kernel must produce the same bits: tests/test_shared_shadows_gpu.py compares the trajectories)
__device__ __forceinline__ void p8_adam4(f32x4& pp, const f32x4& gg, f32x4& mm, f32x4& vv, float gs, float lr_bc1, float beta1,
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

// EPI: 3 = the weight-gradient table with the OPTIMIZER in its epilogue: the accumulators are the gradient; the lane that holds four
//          consecutive elements of dW loads p, m, v of those elements, applies Adam, stores p, m, v and both bf16 parameter shadows
//          (W 8 bytes; W^T four 2-byte stores, 16 consecutive rows per 16 lanes) - dW never reaches memory (-8 bytes per parameter
//          and step, and the HBM-bound optimizer streams while other workgroups multiply)
// EPI: 1 = plain fp32 result (the weight-gradient table: + bias-gradient row sums, ReLU on either operand's fragments),
//      2 = text-encoder launches: bias, ReLU / GELU, residual, fp32 result unless GF_NO_F32, bf16 shadow
template <bool RC, bool TABLE, int EPI>
__global__ __launch_bounds__(512) void m2f_gemm_p8_kernel(const GemmBatch gb) {
    using C = P8Cfg;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds0 = (unsigned)(size_t)(lds_char*)smem;       // LDS byte address of the operand buffers
    static_assert(offsetof(GemmBatch, count) >= 2880 - 256 && sizeof(GemmBatch) + 64 <= 2880 - 256 + 512, "kernarg warm-up ranges");
    if constexpr (TABLE) m2f_kernarg_warm<0, 8, 2880 - 256>();
    else m2f_kernarg_warm<0, 24, 2880 - 256>();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = wave >> 2, wc = wave & 3;
    const int lr = lane & 15, g = lane >> 4;
    const int grid = TABLE ? 1 : (int)gridDim.x;
    typedef const __attribute__((address_space(4))) int* c_i32p;
    const int total = TABLE ? *((c_i32p)(gb.wg_begin) + blockIdx.x + 1) : gb.total_tiles;
    const int first = TABLE ? *((c_i32p)(gb.wg_begin) + blockIdx.x) : ring_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    if (first >= total) return;                                   // (uniform over the workgroup)

    // ------------------------------------------------ prefetch stream ------------------------------------------------
    int s_idx = first, s_kleft = 0;
    // EPI 3 (optimizer in the epilogue): the stream ENDS with its output tile - the epilogue needs the operand buffers as its own
    // staging ring - so pieces issued past the tile's last k-tile (the schedule keeps issuing them: the counted waits count them) are
    // range-checked to zeros AND redirected into a 2 KiB dump area behind the operand buffers (the wave's W^T image, unused until then)
    constexpr bool PER_TILE = EPI == 3;
    constexpr bool FWD = EPI == 2 || EPI == 4;          // forward-form epilogue (bias / activation / residual); 4: OCP e4m3 operands (see quad)
    unsigned s_dump = 0;                                           // 0 = normal destinations
    unsigned s_kA = 0, s_kB = 0, s_stepA = 0, s_stepB = 0, s_halfA = 0, s_halfB = 0, vA[2], vB[2];
    ring_u32x4 ra, rb;
    auto cursor_close = [&]() {
        ra = p8_rsrc(nullptr, 0); rb = ra; s_kleft = 0x7fffffff;
        vA[0] = vA[1] = vB[0] = vB[1] = 0; s_kA = s_kB = s_stepA = s_stepB = s_halfA = s_halfB = 0;
        s_dump = lds0 + (unsigned)(C::LDS + wave * P8_ADAM_T_BYTES);
    };
    auto cursor_open = [&]() {
        s_dump = 0;
        if (s_idx >= total) {                                      // past the end of the list: every piece is range-checked to zeros
            ra = p8_rsrc(nullptr, 0); rb = ra; s_kleft = 0x7fffffff;
            vA[0] = vA[1] = vB[0] = vB[1] = 0; s_kA = s_kB = s_stepA = s_stepB = s_halfA = s_halfB = 0;
            return;
        }
        const P8Desc D = p8_desc<TABLE>(gb, s_idx);
        const int K = D.K, lda = D.lda, ldb = D.ldb;
        s_kleft = (K + C::BK - 1) / C::BK;
        s_kA = s_kB = 0;
        if constexpr (RC) {
            // k-major image: 64 k-rows of 128 features (256 bytes); a 1 KiB piece = 4 k-rows; the 16-byte chunk stored at position p
            // of k-row q is source chunk p ^ (4 (q & 3) | 2 ((q >> 3) & 1)): the eight (k-row, lane-group) combinations of one half of a
            // transposing fragment read (ds_read_b64_tr_b16 for the 16x16x32 operand: k-rows 8 g + 0..3 of groups g, g + 1) fall on
            // eight disjoint 32-byte bank ranges
            const int kr = lane >> 4, p16 = lane & 15;
            {   // (the wave's second piece = 4 k-rows further down, same swizzle: q & 3 and (q >> 3) & 1 do not change)
                const int q = 8 * wave + kr;
                const int sw = p16 ^ (4 * (q & 3) | 2 * ((q >> 3) & 1));
                vA[0] = (unsigned)(q * lda + D.m0 + 8 * sw) * 2u; vA[1] = (unsigned)(4 * lda) * 2u;       // [1]: uniform
                vB[0] = (unsigned)(q * ldb + D.n0 + 8 * sw) * 2u; vB[1] = (unsigned)(4 * ldb) * 2u;
            }
            s_stepA = (unsigned)(C::BK * lda) * 2u; s_stepB = (unsigned)(C::BK * ldb) * 2u;
            s_halfA = 256u; s_halfB = 256u;
            ra = p8_rsrc(D.aq, (unsigned)K * (unsigned)lda * 2u); rb = p8_rsrc(D.bq, (unsigned)K * (unsigned)ldb * 2u);
        } else {
            // row-major image of gemm_ring.h: 128-byte rows, chunk c of row r at position c ^ ((r >> 1) & 7); a piece = 8 rows
            const int lrow = lane >> 3, pos = lane & 7;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = (2 * wave + j) * 8 + lrow;
                const int sw = pos ^ ((r >> 1) & 7);
                vA[j] = (unsigned)((D.m0 + r) * lda + 8 * sw) * 2u;
                vB[j] = (unsigned)((D.n0 + r) * ldb + 8 * sw) * 2u;
            }
            s_stepA = s_stepB = (unsigned)C::BK * 2u;
            s_halfA = (unsigned)(128 * lda) * 2u; s_halfB = (unsigned)(128 * ldb) * 2u;
            ra = p8_rsrc(D.aq, (unsigned)D.M * (unsigned)lda * 2u); rb = p8_rsrc(D.bq, (unsigned)D.N * (unsigned)ldb * 2u);
        }
    };
    // one half-tile: kind 0 = A0, 1 = A1, 2 = B0, 3 = B1 of the cursor's k-tile into buffer X; A1 closes the k-tile
    auto stage = [&](auto kind_tag, int x) {
        constexpr int KIND = decltype(kind_tag)::value;
        const unsigned dst = (PER_TILE && s_dump) ? s_dump : lds0 + (unsigned)(x * C::BUF + KIND * C::HALF + wave * 2048);
        if constexpr (KIND < 2) {
            const unsigned add = s_kA + (KIND == 1 ? s_halfA : 0u);
            p8_dma16(ra, dst, vA[0] + add);
            p8_dma16(ra, dst + 1024u, RC ? vA[0] + (add + vA[1]) : vA[1] + add);
        } else {
            const unsigned add = s_kB + (KIND == 3 ? s_halfB : 0u);
            p8_dma16(rb, dst, vB[0] + add);
            p8_dma16(rb, dst + 1024u, RC ? vB[0] + (add + vB[1]) : vB[1] + add);
        }
    };
    // A1 closes the cursor's k-tile.  The step to the next one - with, at the end of an output tile, the scalar loads and address arithmetic
    // of the next tile's descriptor - is taken at the head of phase 1, where only the accumulators are live (inside phase 4, with every
    // fragment register live as well, the branch sent 130 registers to scratch)
    auto cursor_advance = [&]() {
        s_kA += s_stepA; s_kB += s_stepB;
        if (--s_kleft == 0) {
            if constexpr (PER_TILE) cursor_close();
            else { s_idx += grid; cursor_open(); }
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    // ------------------------------------------------ fragment addresses ------------------------------------------------
    // KC: lane -> row lr of a 16-row tile, 16-byte chunk 4 s + g of its 128-byte row (k-step s), stored at chunk ^ ((row >> 1) & 7)
    // RC: lane 4 qq + pp of group g -> k-row 32 s + 8 g + qq (+ 4: second read), features f0 + 4 pp .. + 3; lane i receives feature f0 + i
    int offA[4], offB[2];          // KC: [0], [1] = k-step 0 / 1 (row tile i / j adds 2048); RC: per row tile (k-step adds 8192)
    if constexpr (RC) {
        const int qq = lr >> 2, pp = lr & 3, sw = 4 * qq | 2 * (g & 1);
        const int base = (8 * g + qq) * 256 + 8 * (pp & 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) offA[i] = base + (((8 * wr + 2 * i + (pp >> 1)) ^ sw) << 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) offB[j] = base + (((4 * wc + 2 * j + (pp >> 1)) ^ sw) << 4);
    } else {
        const int x = (lr >> 1) & 7;
        const int f0 = (g ^ x) << 4, f1 = ((4 + g) ^ x) << 4;
        offA[0] = (wr * 64 + lr) * 128 + f0; offA[1] = (wr * 64 + lr) * 128 + f1; offA[2] = offA[3] = 0;
        offB[0] = (wc * 32 + lr) * 128 + f0; offB[1] = (wc * 32 + lr) * 128 + f1;
    }
    auto frag = [&](const char* half, int off_row_tile, int s, int kc_off0, int kc_off1) -> bf16x8 {
        if constexpr (RC) {
            const char* q = half + off_row_tile + s * 8192;
            const ring_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ring_s16x4 __attribute__((address_space(3)))*)(const_cast<char*>(q)));
            const ring_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ring_s16x4 __attribute__((address_space(3)))*)(const_cast<char*>(q + 1024)));
            const ring_s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8, r);
        } else {
            return *reinterpret_cast<const bf16x8*>(half + (s ? kc_off1 : kc_off0) + off_row_tile);
        }
    };
    bf16x8 af[4][2], bf0[2][2], bf1[2][2];
    // (the buffer offset is a run-time value on purpose: as a constant every fragment address of either buffer is loop-invariant, gets
    //  hoisted out of the k-loop and the kernel spills hundreds of registers - whose reloads would share the hand-counted vmcnt)
    auto read_a = [&](int xo, int ah) {
        const char* half = smem + xo + ah * C::HALF;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) af[i][s] = RC ? frag(half, offA[i], s, 0, 0) : frag(half, i * 2048, s, offA[0], offA[1]);
    };
    auto read_b = [&](int xo, int bh, bf16x8 (&dst)[2][2]) {
        const char* half = smem + xo + (2 + bh) * C::HALF;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) dst[j][s] = RC ? frag(half, offB[j], s, 0, 0) : frag(half, j * 2048, s, offB[0], offB[1]);
    };

    // ------------------------------------------------ prologue ------------------------------------------------
    // Skew.  Every tile of a launch takes the same time, so workgroups that start together reach their epilogues together: 256 CUs x
    // 256 KiB of result in one burst (measured: 25k of a tile's 90k cycles at K = 1,024 went into the stores and the wait for them -
    // the chip's HBM write rate - with the MFMA pipes idle; profiles/r04_p8_phase_totals_v1.txt).  Workgroups whose list is
    // shorter than the longest (`gb.p8_max_tiles`) can start late for free: they wait q / 4 of a tile time, q = (workgroup / 8) mod 4,
    // so that at most about a quarter of the chip stores at any time.  gb.p8_skew = estimated cycles per k-tile (0 = off);
    // bit 30 set: skew every workgroup, slack or not.
    if (gb.p8_skew) {
        const int mine = TABLE ? total - first : (total - first + grid - 1) / grid;
        // (bit 29: skew whole XCDs against each other - workgroups b, b + 8, ... share an L2 and the operand panels in it, and stay in step)
        const int q = (gb.p8_skew & 0x20000000) ? (((int)blockIdx.x & 7) >> 1) : (((int)blockIdx.x >> 3) & 3);
        if (q && (mine < gb.p8_max_tiles || (gb.p8_skew & 0x40000000))) {
            const P8Desc D0 = p8_desc<TABLE>(gb, first);
            const unsigned long long wait = (unsigned long long)((D0.K + C::BK - 1) / C::BK) * (unsigned)(gb.p8_skew & 0xFFFFF) * (unsigned)q / 4u;
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
        }
    }
    auto prologue = [&]() {
        cursor_open();
        stage(I2{}, 0); stage(I0{}, 0); stage(I3{}, 0); stage(I1{}, 0); cursor_advance();           // k-tile 0 -> buffer 0
        stage(I2{}, 1); stage(I0{}, 1); stage(I3{}, 1); stage(I1{}, 1);                              // k-tile 1 -> buffer 1 (advance: head of phase 1)
        ring_wait_vm<8>();
        __builtin_amdgcn_s_barrier();                                                 // k-tile 0 is in LDS, for every wave
        if (wr == 1) __builtin_amdgcn_s_barrier();                                    // the lower half runs one barrier behind
    };
    if constexpr (!PER_TILE) prologue();

    // ------------------------------------------------ tiles ------------------------------------------------
    int par = 0;                                                                       // buffer of the next k-tile to multiply
#ifdef P8_TIMING
    unsigned long long tacc[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, tk = 0, tep = 0;
    unsigned long long tb0 = 0, tb1 = 0, tb2 = 0, tb3 = 0, tb4 = 0, tb5 = 0, ntile = 0;      // whole k-tiles by their index inside an output tile
#endif
    int ep_relax = 0;                                  // vector-memory instructions (stores) the LAST epilogue left behind it: 0, 16, 32, 48; anything else: 57 or more
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
    (void)ts0; (void)ts1; (void)ts2; (void)ts3;
#pragma unroll 1
    for (int bpos = first; bpos < total; bpos += grid) {
        if constexpr (PER_TILE) { s_idx = bpos; par = 0; prologue(); }
        const P8Desc H = p8_desc<TABLE>(gb, bpos);
        // (the table lives in device memory: its fields come through the constant address space = scalar loads; a plain reference is
        //  read with VECTOR loads, whose wait would drain the prefetch stream once per tile)
        typedef const __attribute__((address_space(4))) GemmProblem* c_probp;
        const auto& P = *(TABLE ? (c_probp)(gb.table) + H.pi : (c_probp)(const GemmProblem*)&gb.pr[H.pi]);
        const int m0 = H.m0, n0 = H.n0, nk = (H.K + C::BK - 1) / C::BK;
        const bool reluA = false, reluB = H.flags & GF_RELU_B;
        float* bias_grad = (EPI == 1 || EPI == 3) ? P.bias_grad : nullptr;
        const bool bgrad = (EPI == 1 || EPI == 3) && RC && bias_grad && n0 == 0 && wc == 0;      // wave-uniform
        float bsum[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) bsum[a][i] = 0.f;
        f32x4 acc[2][2][4][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // epilogue fields before the k-loop (scalar loads from a cold line cost the tile its MFMA pipe when read after it)
        float* __restrict__ Cp = P.c;
        const int ldc = P.ldc, Mm = H.M, Nn = H.N;
        const float* __restrict__ bias = FWD ? P.bias : nullptr;
        const float* __restrict__ res = FWD ? P.res : nullptr;
        const int ldres = P.ldres;
        uint16_t* __restrict__ C16 = FWD && !(P.flags & GF_NO_BF16) ? reinterpret_cast<uint16_t*>(m2f_shadow_of(gb.sh, P.c)) : nullptr;
        const bool relu_out = P.flags & GF_RELU_OUT, gelu = P.flags & GF_GELU_OUT, no32 = FWD && (P.flags & GF_NO_F32) && C16;
        // EPI 4: the accumulator is de-quantised first (1 / (scale_a scale_b)); c8 set: the result leaves as e4m3(x * c8_scale) INSTEAD of fp32 / bf16
        const float acc_scale = EPI == 4 ? P.acc_scale : 1.f, c8_scale = EPI == 4 ? P.c8_scale : 1.f;
        uint8_t* __restrict__ C8 = EPI == 4 ? P.c8 : nullptr;
        // EPI 1, `res` set: the weight gradient leaves as bf16 INSTEAD of fp32, at the same element index of a bf16 gradient buffer (the
        // data-parallel bf16 exchange sends that buffer as it is: no fp32 dW round trip, no rounding pass; m2f_plan_grad_bf16)
        uint16_t* __restrict__ G16 = EPI == 1 ? reinterpret_cast<uint16_t*>(const_cast<float*>(P.res)) : nullptr;

        auto quad = [&](auto a_tag, auto b_tag, const bf16x8 (&bb)[2][2]) {
            constexpr int AH = decltype(a_tag)::value, BH = decltype(b_tag)::value;
            if constexpr (EPI == 4) {
                // OCP e4m3 operands: the same bytes through the same LDS image - a 128-byte row holds 128 k-values, the lane's two 16-byte reads of a
                // row ARE the 32-byte operand of v_mfma_scale_f32_16x16x128_f8f6f4 (scale operands 0 = unscaled; 2x the bf16 rate).  Which 32 of the
                // 128 k-values a lane group supplies does not matter as long as A and B agree - and both come from the same chunk positions.
                typedef int p8_v8i __attribute__((ext_vector_type(8)));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const ring_u32x4 a0 = __builtin_bit_cast(ring_u32x4, af[i][0]), a1 = __builtin_bit_cast(ring_u32x4, af[i][1]);
                        const ring_u32x4 b0 = __builtin_bit_cast(ring_u32x4, bb[j][0]), b1 = __builtin_bit_cast(ring_u32x4, bb[j][1]);
                        const p8_v8i fa = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
                        const p8_v8i fb = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
                        acc[AH][BH][i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa, acc[AH][BH][i][j], 0, 0, 0, 0, 0, 0);
                    }
            } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)         // operands swapped: the lane then holds 4 consecutive COLUMNS of row lr (16-byte stores)
                        acc[AH][BH][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[j][s], af[i][s], acc[AH][BH][i][j], 0, 0, 0);
            }
        };
        auto bias_sums = [&](int ah) {                  // bias gradient = sum over k of A's rows (before any ReLU on A)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const ring_u32x4 w = __builtin_bit_cast(ring_u32x4, af[i][s]);
                    bsum[ah][i] = ring_bf16x2_sum(w.w, ring_bf16x2_sum(w.z, ring_bf16x2_sum(w.y, ring_bf16x2_sum(w.x, bsum[ah][i]))));
                }
        };
        const uint32_t floorA = reluA ? 0u : 0x80008000u, floorB = reluB ? 0u : 0x80008000u;
        auto relu_a = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) af[i][s] = p8_floor8(af[i][s], floorA);
        };
        auto relu_b = [&](bf16x8 (&bb)[2][2]) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s) bb[j][s] = p8_floor8(bb[j][s], floorB);
        };
        // one k-tile out of buffer x: four phases.  OPT: 0 = plain, 1 = ReLU floor on the fragments + bias-gradient sums (a copy of the
        // LOOP, chosen per output tile: branches inside a phase cut it into basic blocks and sent 200 registers to scratch)
        auto ktile = [&](int x, auto opt_tag, int relax) {
            constexpr int OPT = decltype(opt_tag)::value;      // bit 0: bias-gradient sums, bit 1: ReLU floor on A, bit 2: on B
            const int xo = x * C::BUF;
            // ---- phase 1: B part 0 (4 reads), A part 0 (8 reads); no prefetch piece (the phase with the most reads) ----
            P8_STAMP(ts0);
            cursor_advance();                             // (the A1 piece issued in the previous phase 4 closed a k-tile)
            __builtin_amdgcn_sched_barrier(0);
            read_b(xo, 0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            read_a(xo, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // the four B reads, issued first, are back: B0 may be re-staged next phase
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            P8_STAMP(ts1);
            if constexpr (OPT & 1) bias_sums(0);
            if constexpr (OPT & 2) relu_a();
            if constexpr (OPT & 4) relu_b(bf0);
            __builtin_amdgcn_s_setprio(1);
            quad(I0{}, I0{}, bf0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            P8_STAMP(ts2);
            __builtin_amdgcn_s_barrier();
            P8_STAMP(ts3);
            P8_ACC(0);
            // ---- phase 2: B part 1 (4 reads); stage B0 of k-tile + 2 (this buffer) ----
            // (measured in round 4: these four reads issued under phase 1's MFMAs instead - same results, row-major 4,096^3 1,137 -> 1,058 TFLOP/s, k-contiguous
            //  8,192^3 1,255 -> 1,228: transposing reads between MFMAs cost more than they hide)
            P8_STAMP(ts0);
            read_b(xo, 1, bf1);
            stage(I2{}, x);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            P8_STAMP(ts1);
            if constexpr (OPT & 4) relu_b(bf1);
            __builtin_amdgcn_s_setprio(1);
            quad(I0{}, I1{}, bf1);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            P8_STAMP(ts2);
            __builtin_amdgcn_s_barrier();
            P8_STAMP(ts3);
            P8_ACC(1);
            // ---- phase 3: A part 1 (8 reads); stage A0 of k-tile + 2 ----
            P8_STAMP(ts0);
            read_a(xo, 1);
            stage(I0{}, x);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            P8_STAMP(ts1);
            if constexpr (OPT & 1) bias_sums(1);
            if constexpr (OPT & 2) relu_a();
            __builtin_amdgcn_s_setprio(1);
            quad(I1{}, I1{}, bf1);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            P8_STAMP(ts2);
            __builtin_amdgcn_s_barrier();
            P8_STAMP(ts3);
            P8_ACC(2);
            // ---- phase 4: no reads; stage B1 of k-tile + 2; the ONE counted wait: everything but the three youngest half-tiles ----
            P8_STAMP(ts0);
            stage(I3{}, x);
            // everything but the three youngest half-tiles has landed - and, in the first k-tile after an epilogue, but the epilogue's stores,
            // which were issued between A1 of the next k-tile (below) and those three: loads and stores share ONE in-order counter, and
            // waiting for 256 KiB of stores here would stall the MFMA pipe for thousands of cycles
            if (relax == 0) ring_wait_vm<6>();
            else if (relax == 16) ring_wait_vm<22>();
            else if (relax == 32) ring_wait_vm<38>();
            else if (relax == 48) ring_wait_vm<54>();
            else ring_wait_vm<63>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            P8_STAMP(ts1);
            stage(I1{}, x);                               // A1 of k-tile + 2 (this buffer: its A1 half was last read in phase 3) - closes the cursor's k-tile
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            quad(I1{}, I0{}, bf0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            P8_STAMP(ts2);
            __builtin_amdgcn_s_barrier();
            P8_STAMP(ts3);
            P8_ACC(3);
        };
        // (a copy of the LOOP per option set the weight-gradient table uses - plan.hip sets GF_RELU_B only: the FAM layer's
        //  relu(cat(x, text)) operand - ReLU on A has no copy: the launchers refuse it)
#ifdef P8_TIMING
#define P8_LOOP(MASK) do { _Pragma("unroll 1") for (int kt = 0; kt < nk; ++kt) { unsigned long long tq0, tq1; P8_STAMP(tq0);                    \
        ktile(par, std::integral_constant<int, MASK>{}, kt == 0 ? ep_relax : 0); par ^= 1; P8_STAMP(tq1);                                       \
        const unsigned long long dq = tq1 - tq0; tb0 += kt == 0 ? dq : 0; tb1 += kt == 1 ? dq : 0; tb2 += kt == 2 ? dq : 0; tb3 += kt == 3 ? dq : 0; \
        tb4 += (kt >= 4 && kt < 8) ? dq : 0; tb5 += kt >= 8 ? dq : 0; } } while (0)
#else
#define P8_LOOP(MASK) do { _Pragma("unroll 1") for (int kt = 0; kt < nk; ++kt) { ktile(par, std::integral_constant<int, MASK>{}, kt == 0 ? ep_relax : 0); par ^= 1; } } while (0)
#endif
        const int optm = (EPI == 1 || EPI == 3) ? (bgrad ? 1 : 0) | (reluA ? 2 : 0) | (reluB ? 4 : 0) : 0;
        if (optm == 0) P8_LOOP(0);
        else if (optm == 1) P8_LOOP(1);
        else if (optm == 4) P8_LOOP(4);
        else if (optm == 5) P8_LOOP(5);
        else P8_LOOP(5);                                  // (ReLU on A is not this form's: the launchers refuse it)
#undef P8_LOOP

        P8_STAMP(ts0);
        // ---------------------------------------- epilogue: stores only ----------------------------------------
        if (bgrad) {                                    // the four lane groups hold four k-slices of row lr
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = bsum[a][i];
                    t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
                    const int m = m0 + a * 128 + wr * 64 + i * 16 + lr;
                    if (g == 0 && m < Mm) bias_grad[m] = t;
                }
        }
        if constexpr (EPI == 3) {
            typedef const __attribute__((address_space(4))) M2FAdamFuse* c_adamp;
            typedef const __attribute__((address_space(4))) float* c_f32p;
            const auto& AD = *(c_adamp)(gb.adam);
            const ptrdiff_t doff = Cp - AD.g_base;                                     // this problem's first element in the flat buffers
            float* __restrict__ Pp = AD.p + doff; float* __restrict__ Mp = AD.m + doff; float* __restrict__ Vp = AD.v + doff;
            uint16_t* __restrict__ shw = reinterpret_cast<uint16_t*>(const_cast<float*>(P.res));
            uint16_t* __restrict__ shwt = reinterpret_cast<uint16_t*>(const_cast<float*>(P.gate));
            const int ldd = P.ldres, ldt = P.ldgate;
            c_f32p hy = (c_f32p)(AD.hyper);
            const float lr_bc1 = hy[0], beta1 = hy[1], beta2 = hy[2], eps = hy[3], wd = hy[4], inv_sqrt_bc2 = hy[5];
            const float gs = AD.gs_ptr ? 1.0f / *((c_f32p)(AD.gs_ptr)) : 1.0f;
            const bool avec = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(Pp) & 15) == 0) && ((reinterpret_cast<uintptr_t>(shw) & 7) == 0) && ((ldd & 3) == 0);
            const bool wt8 = ((reinterpret_cast<uintptr_t>(shwt) & 7) == 0) && ((ldt & 3) == 0), wt16 = ((reinterpret_cast<uintptr_t>(shwt) & 15) == 0) && ((ldt & 7) == 0);
            const bool awhole = avec && wt8 && m0 + C::BM <= Mm && n0 + C::BN <= Nn;
            ep_relax = 0;
            // the stream of this tile has ended (PER_TILE): its last pieces - zeros into the dump area - must have landed, and every wave
            // must be past its last fragment read, before the operand buffers become the epilogue's staging ring.  The upper half waits one
            // extra barrier here (the lower half's last closing barrier): from now to the next tile's prologue both halves run aligned.
            ring_wait_vm<0>();
            if (wr == 0) __builtin_amdgcn_s_barrier();
            if (awhole) {
                // 32 blocks of 16 rows x 16 columns per wave.  p, m, v of a block arrive by LDS-DMA (no staging registers) in a wave-private
                // ring of FIVE blocks (3 KiB each: 15 of the wave's 16 KiB of the operand buffers), four blocks ahead of the block being
                // updated: ~128 KiB in flight per CU.  With eight waves per CU and registers for two blocks in flight the epilogue moved
                // 3.4 TB/s (15 GB/s per CU = 48 KiB in flight / 3 us of loaded latency) and the fused launch was SLOWER than table launch +
                // optimizer kernel (0.95 vs 0.83 ms at C3).  Lane -> its own 16 bytes of every piece (source = its four elements).
                // W^T shadow: the four row blocks of a column block go through a wave-private LDS image [16 columns][64 rows] (136-byte
                // rows) and leave as whole 128-byte rows of W^T.
                char* tw = smem + C::LDS + wave * P8_ADAM_T_BYTES;
                const unsigned ring0 = lds0 + (unsigned)(wave * 16384);
                const char* ringp = smem + wave * 16384 + lane * 16;
                const ring_u32x4 rp = p8_rsrc(Pp, 0x7FFFFF00u), rm = p8_rsrc(Mp, 0x7FFFFF00u), rv = p8_rsrc(Vp, 0x7FFFFF00u);
                auto blk = [&](int n, int& a, int& b, int& j, int& i) { a = n >> 4; b = (n >> 3) & 1; j = (n >> 2) & 1; i = n & 3; };
                auto issue = [&](int n) {
                    int a, b, j, i; blk(n, a, b, j, i);
                    const int row = m0 + a * 128 + wr * 64 + i * 16 + lr, col = n0 + b * 128 + wc * 32 + j * 16 + 4 * g;
                    const unsigned vo = (unsigned)(row * ldc + col) * 4u;
                    const unsigned slot = ring0 + (unsigned)((n % 5) * 3072);
                    p8_dma16(rp, slot, vo); p8_dma16(rm, slot + 1024u, vo); p8_dma16(rv, slot + 2048u, vo);
                };
#pragma unroll
                for (int n = 0; n < 5; ++n) issue(n);
#pragma unroll
                for (int n = 0; n < 32; ++n) {
                    // block n has landed once at most the operations issued AFTER its three pieces are outstanding: the pieces of the blocks
                    // ahead (3 each) and the stores of the blocks behind (at least 4 each: p, m, v, W shadow; the W^T stores only add)
                    constexpr int S4 = 4;
                    const int ahead = 31 - n < 4 ? 31 - n : 4;
                    const int allowed = n < 5 ? 3 * (4 - n) + (S4 + 3) * n : S4 * 4 + 3 * ahead;
                    switch (allowed) {                         // (compile-time after unrolling: one s_waitcnt)
                        case 12: ring_wait_vm<12>(); break; case 16: ring_wait_vm<16>(); break; case 19: ring_wait_vm<19>(); break;
                        case 20: ring_wait_vm<20>(); break; case 22: ring_wait_vm<22>(); break; case 24: ring_wait_vm<24>(); break;
                        case 25: ring_wait_vm<25>(); break; case 28: ring_wait_vm<28>(); break; default: ring_wait_vm<0>(); break;
                    }
                    int a, b, j, i; blk(n, a, b, j, i);
                    const int row = m0 + a * 128 + wr * 64 + i * 16 + lr, col = n0 + b * 128 + wc * 32 + j * 16 + 4 * g;
                    const size_t oc = (size_t)((uint32_t)(row * ldc + col));
                    const char* sl = ringp + (n % 5) * 3072;
                    f32x4 pp = *reinterpret_cast<const f32x4*>(sl), mm = *reinterpret_cast<const f32x4*>(sl + 1024), vv = *reinterpret_cast<const f32x4*>(sl + 2048);
                    p8_adam4(pp, acc[a][b][i][j], mm, vv, gs, lr_bc1, beta1, beta2, eps, wd, inv_sqrt_bc2);
                    *reinterpret_cast<f32x4*>(Pp + oc) = pp;
                    __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(Mp + oc));
                    __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(Vp + oc));
                    const uint16_t h0 = m2f_bf16_bits(pp[0]), h1 = m2f_bf16_bits(pp[1]), h2 = m2f_bf16_bits(pp[2]), h3 = m2f_bf16_bits(pp[3]);
                    uint2 w;
                    w.x = (uint32_t)h0 | ((uint32_t)h1 << 16); w.y = (uint32_t)h2 | ((uint32_t)h3 << 16);
                    *reinterpret_cast<uint2*>(shw + (size_t)((uint32_t)(row * ldd + col))) = w;
                    uint16_t* tq = reinterpret_cast<uint16_t*>(tw + (4 * g) * 136) + i * 16 + lr;      // image [column 4 g + e][row 16 i + lr]
                    tq[0] = h0; tq[68] = h1; tq[136] = h2; tq[204] = h3;
                    // (the slot's three reads are back - their values were just used - before the slot is handed to block n + 5)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (n + 5 < 32) issue(n + 5);
                    if (i == 3) {                          // a column block's four row blocks are in the image: 16 rows of W^T x 128 bytes
                        __builtin_amdgcn_wave_barrier();
                        const int tc = lane >> 2, part = lane & 3;      // lane -> column tc, rows 16 part .. 16 part + 15 (32 bytes)
                        const uint2* src = reinterpret_cast<const uint2*>(tw + tc * 136 + part * 32);
                        const uint2 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        const int colT = n0 + b * 128 + wc * 32 + j * 16 + tc, rowT = m0 + a * 128 + wr * 64 + 16 * part;
                        uint16_t* dst = shwt + (size_t)((uint32_t)(colT * ldt + rowT));
                        if (wt16) {
                            *reinterpret_cast<uint4*>(dst) = make_uint4(r0.x, r0.y, r1.x, r1.y);
                            *reinterpret_cast<uint4*>(dst + 8) = make_uint4(r2.x, r2.y, r3.x, r3.y);
                        } else {
                            reinterpret_cast<uint2*>(dst)[0] = r0; reinterpret_cast<uint2*>(dst)[1] = r1;
                            reinterpret_cast<uint2*>(dst)[2] = r2; reinterpret_cast<uint2*>(dst)[3] = r3;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int row = m0 + a * 128 + wr * 64 + i * 16 + lr, col = n0 + b * 128 + wc * 32 + j * 16 + 4 * g;
                                f32x4 pp, mm, vv;
                                bool in[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    in[e] = row < Mm && col + e < Nn;
                                    const size_t oc = in[e] ? (size_t)((uint32_t)(row * ldc + col + e)) : 0;
                                    pp[e] = Pp[oc]; mm[e] = Mp[oc]; vv[e] = Vp[oc];
                                }
                                p8_adam4(pp, acc[a][b][i][j], mm, vv, gs, lr_bc1, beta1, beta2, eps, wd, inv_sqrt_bc2);
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (in[e]) {
                                        const size_t oc = (size_t)((uint32_t)(row * ldc + col + e));
                                        Pp[oc] = pp[e]; Mp[oc] = mm[e]; Vp[oc] = vv[e];
                                        const uint16_t h = m2f_bf16_bits(pp[e]);
                                        shw[(size_t)((uint32_t)(row * ldd + col + e))] = h;
                                        shwt[(size_t)((uint32_t)((col + e) * ldt + row))] = h;
                                    }
                            }
            }
            // every wave is done with its staging ring (and its dump / image area) before the next tile's prologue refills the buffers
            ring_wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            continue;                                      // (next output tile)
        }
        // (16-byte stores: fp32 at columns 4 g, bf16 at columns 8 g of a row)
        const bool vec = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cp) & 15) == 0) &&
                         (!G16 || (((ldc & 7) == 0) && (reinterpret_cast<uintptr_t>(G16) & 15) == 0)) &&
                         (!res || (((ldres & 3) == 0) && (reinterpret_cast<uintptr_t>(res) & 15) == 0)) &&
                         (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0) &&
                         (!C16 || (((ldc & 7) == 0) && (reinterpret_cast<uintptr_t>(C16) & 15) == 0));
        bool whole = vec && m0 + C::BM <= Mm && n0 + C::BN <= Nn;                     // block-uniform
        if constexpr (EPI == 4) whole = whole && (!C8 || (((ldc & 7) == 0) && (reinterpret_cast<uintptr_t>(C8) & 7) == 0));       // (8-byte e4m3 stores)
        // one 16-row x 16-column block of the wave: lane -> row lr, columns 4 g .. 4 g + 3
        auto element = [&](float a, float bv, float rv) {                              // the element order of ring_epilogue
            if constexpr (!FWD) return a;
            float x = (EPI == 4 ? a * acc_scale : a) + bv;
            x = relu_out ? fmaxf(x, 0.f) : x;
            if (gelu) x = m2f_gelu<true>(x);
            return x + rv;
        };
        ep_relax = 0;
        if (whole) {
            // Store SHAPE decides how fast a CU gets its tile out (tools/store_probe: every CU writing 256 x 256 tiles, cycles per tile):
            //   fp32, 16 rows x 64 bytes per instruction (a lane's four columns of one 16-column block)   33.9k   3.8 TB/s
            //   fp32,  8 rows x 128 bytes                                                               23.0k   5.2 TB/s
            //   bf16, 16 rows x 32 bytes (8 bytes per lane)                                              16.7k   4.1 TB/s
            //   bf16, 16 rows x 64 bytes (16 bytes per lane)                                              9.1k   5.7 TB/s
            // so the two 16-column blocks (j = 0, 1) of a row leave TOGETHER:
            //   fp32: lanes lr >= 8 trade registers with lanes lr - 8 (DPP row_ror:8 under a bank mask) - instruction 0 carries rows 0..7 of the
            //         16-row block with all 32 columns (j = 0 from the lanes lr < 8, j = 1 from the lanes lr >= 8), instruction 1 rows 8..15;
            //   bf16: v_permlane32_swap + v_permlane16_swap of the packed pairs leave lane group q with columns 8 q .. 8 q + 7 of row lr.
            // The residual (EPI 2) is added in place BEFORE the first store, in batches of eight loads: loads and stores share one in-order
            // counter, and a load issued behind stores waits for them (24k cycles per tile with the loads between the stores, 12k without).
            const bool has16 = ((FWD && C16) || (EPI == 1 && G16)) && !C8;
            const bool has32 = !(no32 || (EPI == 1 && G16) || C8);
            const int n_st = (has32 ? 32 : 0) + (has16 ? 16 : 0) + (C8 ? 16 : 0);
            ep_relax = n_st;
            if constexpr (FWD) {
                if (bias || relu_out || gelu || EPI == 4) {
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int col = n0 + b * 128 + wc * 32 + j * 16 + 4 * g;
                            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                            if (bias) bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
                            for (int a = 0; a < 2; ++a)
#pragma unroll
                                for (int i = 0; i < 4; ++i)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) acc[a][b][i][j][e] = element(acc[a][b][i][j][e], bv[e], 0.f);
                        }
                }
                if (res) {
                    // two batches of sixteen loads (the fragment registers are free here): two memory round trips per tile
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        f32x4 rv[2][4][2];
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    const int row = m0 + a * 128 + wr * 64 + i * 16 + lr, col = n0 + b * 128 + wc * 32 + j * 16 + 4 * g;
                                    rv[b][i][j] = *reinterpret_cast<const f32x4*>(res + (size_t)((uint32_t)(row * ldres + col)));
                                }
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int j = 0; j < 2; ++j) acc[a][b][i][j] += rv[b][i][j];
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);                   // (no store may move above the last residual load)
            uint16_t* __restrict__ O16 = EPI == 1 ? G16 : C16;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 x0 = acc[a][b][i][0], x1 = acc[a][b][i][1];
                        const int rowb = m0 + a * 128 + wr * 64 + i * 16, colb = n0 + b * 128 + wc * 32;
                        if (has32) {
                            f32x4 s0, s1;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                // lanes lr >= 8 (banks 2, 3 of every row of 16 lanes) take x1 of lane lr - 8; lanes lr < 8 take x0 of lane lr + 8
                                // (inline asm: through __builtin_amdgcn_update_dpp on vector elements hipcc 7.2 delivered element 0 four times)
                                float d0 = x0[e], d1 = x1[e];
                                asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(d0) : "v"(x1[e]));
                                asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0x3" : "+v"(d1) : "v"(x0[e]));
                                s0[e] = d0; s1[e] = d1;
                            }
                            const size_t o0 = (size_t)((uint32_t)((rowb + (lr & 7)) * ldc + colb + 16 * (lr >> 3) + 4 * g));
                            *reinterpret_cast<f32x4*>(Cp + o0) = s0;
                            *reinterpret_cast<f32x4*>(Cp + o0 + (size_t)((uint32_t)(8 * ldc))) = s1;
                        }
                        if (has16) {
                            uint32_t a0 = (uint32_t)m2f_bf16_bits(x0[0]) | ((uint32_t)m2f_bf16_bits(x0[1]) << 16);
                            uint32_t a1 = (uint32_t)m2f_bf16_bits(x0[2]) | ((uint32_t)m2f_bf16_bits(x0[3]) << 16);
                            uint32_t b0 = (uint32_t)m2f_bf16_bits(x1[0]) | ((uint32_t)m2f_bf16_bits(x1[1]) << 16);
                            uint32_t b1 = (uint32_t)m2f_bf16_bits(x1[2]) | ((uint32_t)m2f_bf16_bits(x1[3]) << 16);
                            // rows of 16 lanes [A0 A1 A2 A3], [B0 B1 B2 B3] -> (32-swap) [A0 A1 B0 B1], [A2 A3 B2 B3] -> (16-swap) [A0 A2 B0 B2], [A1 A3 B1 B3]
                            p8_swap2(a0, b0); p8_swap2(a1, b1);
                            ring_u32x4 w = {a0, a1, b0, b1};                    // lane group q: columns 8 q .. 8 q + 3 | 8 q + 4 .. 8 q + 7 of row lr
                            *reinterpret_cast<ring_u32x4*>(O16 + (size_t)((uint32_t)((rowb + lr) * ldc + colb + 8 * g))) = w;
                        }
                        if constexpr (EPI == 4) {
                            if (C8) {                                           // e4m3 result: 8 bytes per lane after the same exchange
                                uint32_t a0 = m2f_fp8x4_bits(x0[0] * c8_scale, x0[1] * c8_scale, x0[2] * c8_scale, x0[3] * c8_scale);
                                uint32_t b0 = m2f_fp8x4_bits(x1[0] * c8_scale, x1[1] * c8_scale, x1[2] * c8_scale, x1[3] * c8_scale);
                                p8_swap2(a0, b0);
                                uint2 w8; w8.x = a0; w8.y = b0;
                                *reinterpret_cast<uint2*>(C8 + (size_t)((uint32_t)((rowb + lr) * ldc + colb + 8 * g))) = w8;
                            }
                        }
                    }
        } else {
            // edge tiles / unaligned results: element by element, masked (rare: 300-wide audio features, the [7, d] classifier weight)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = m0 + a * 128 + wr * 64 + i * 16 + lr, col = n0 + b * 128 + wc * 32 + j * 16 + 4 * g;
                            const f32x4 v = acc[a][b][i][j];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const bool in = row < Mm && col + e < Nn;
                                const size_t oc = (size_t)((uint32_t)(row * ldc + col + e));
                                float bvv = 0.f, rvv = 0.f;
                                if constexpr (FWD) {
                                    if (bias && in) bvv = bias[col + e];
                                    if (res && in) rvv = res[(size_t)((uint32_t)(row * ldres + col + e))];
                                }
                                const float x = element(v[e], bvv, rvv);
                                if (in && EPI == 1 && G16) G16[oc] = m2f_bf16_bits(x);
                                else if (in && C8) C8[oc] = (uint8_t)(m2f_fp8x4_bits(x * c8_scale, 0.f, 0.f, 0.f) & 0xffu);
                                else if (in) {
                                    Cp[oc] = x;                         // (edge tiles keep the fp32 store whatever GF_NO_F32 says, as the ring form does)
                                    if constexpr (FWD) { if (C16) C16[oc] = m2f_bf16_bits(x); }
                                }
                            }
                        }
        }
#ifdef P8_TIMING
        P8_STAMP(ts1); tep += ts1 - ts0; ++ntile;
#endif
    }
#ifdef P8_TIMING
    if (blockIdx.x == 0 && (wave & 3) == 0 && lane == 0) {
        for (int p = 0; p < 4; ++p) for (int q = 0; q < 3; ++q) m2f_p8_dbg[wr * 16 + p * 3 + q] += tacc[p][q];
        m2f_p8_dbg[wr * 16 + 12] += tk; m2f_p8_dbg[wr * 16 + 13] += tep;
        m2f_p8_dbg[32 + wr * 16 + 0] += tb0; m2f_p8_dbg[32 + wr * 16 + 1] += tb1; m2f_p8_dbg[32 + wr * 16 + 2] += tb2; m2f_p8_dbg[32 + wr * 16 + 3] += tb3;
        m2f_p8_dbg[32 + wr * 16 + 4] += tb4; m2f_p8_dbg[32 + wr * 16 + 5] += tb5; m2f_p8_dbg[32 + wr * 16 + 6] += ntile;
    }
#endif
    // the stream ran past the end of the list: its last (zero-filling) pieces must not outlive the workgroup's LDS allocation
    ring_wait_vm<0>();
    if constexpr (!PER_TILE) { if (wr == 0) __builtin_amdgcn_s_barrier(); }           // the upper half catches up with the lower one
}

template <bool RC, bool TABLE, int EPI>
hipError_t launch_p8_grid(const GemmBatch& hb, int tiles, hipStream_t stream) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    auto kern = m2f_gemm_p8_kernel<RC, TABLE, EPI>;
    constexpr int lds_bytes = P8Cfg::LDS + (EPI == 3 ? 8 * P8_ADAM_T_BYTES : 0);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    ++m2f_g_ring_launches;
    if constexpr (TABLE) {
        if (!hb.tile_rec || !hb.wg_begin || hb.wg_count < 1) return hipErrorInvalidValue;
        hipLaunchKernelGGL(kern, dim3(hb.wg_count), dim3(512), lds_bytes, stream, hb);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(tiles < n_cu ? tiles : n_cu), dim3(512), lds_bytes, stream, hb);
    return hipGetLastError();
}

// grouped launch (problems in the kernel arguments): tile order m fastest inside a problem, workgroups walk remap(b), + grid, ...
template <bool RC, int EPI>
hipError_t launch_p8_grouped(GemmBatch& gb, hipStream_t stream) {
    int t = 0;
    for (int i = 0; i < gb.count; ++i) {
        GemmProblem& p = gb.pr[i];
        p.splitk = 1; p.slab_begin = 0; p.cnt_begin = 0;
        p.tile_begin = t;
        p.tiles_n = m2f_cdiv(p.N, P8Cfg::BN);
        t += m2f_cdiv(p.M, P8Cfg::BM) * p.tiles_n;
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
    {
        int dev = 0, n_cu = 256; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
        const int g = t < n_cu ? t : n_cu;
        hb.p8_max_tiles = (t + g - 1) / g;
    }
    return launch_p8_grid<RC, false, EPI>(hb, t, stream);
}

}  // namespace
