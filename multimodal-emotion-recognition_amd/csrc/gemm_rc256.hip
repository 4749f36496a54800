// Weight-gradient table launch, 256 x 256 tiles (M2F_TABLE_TILE=132): dW = dY^T X with both operands ROW-MAJOR bf16 activation
// shadows ([token][feature]: the reduction index is the row), every 2-D parameter gradient of the step in ONE launch.
//
// Why another form: the 128x128 / 256x128 ring forms of this launch are bound by the bytes the 256 workgroups pull through the
// L2s together (profiles/r03: 26.0 M / 19.5 M L1->L2 requests of 128 B per launch, duration proportional to them; the producers
// sit in their load-issue loop 75 % of the time, the fragments' consumers wait for the barrier).  A 256x256 tile moves half the
// operand bytes per FLOP of a 128x128 one.  It needs 128 accumulator registers per lane with EIGHT waves (each 128 x 64), i.e.
// the whole register file - so there are no producer waves here: every wave issues an eighth of the next k-tile's LDS-DMA loads
// right after the barrier, then multiplies the current one (two waves per SIMD cover each other's LDS latency), and waits for
// its own loads at the end of the k-tile, 2,048 MFMA cycles after it issued them.  Two LDS slots of 64 KiB (double buffer), one
// barrier per k-tile, the next tile's descriptor is fetched a whole tile ahead.
//   LDS image of a 256-wide operand = two of gemm_ring.h's row-major images behind each other (64 k-rows x 256 bytes = 128
//   features each, 16-byte chunk at position p of k-row q holds source chunk p ^ 4 (q & 3)); fragments by ds_read_b64_tr_b16.
// Same products, same k order, same epilogue (ring_epilogue, plain-store form) as the other table forms: the results agree with
// them to fp32 summation noise of the bias-gradient sums only (tests/test_model_gpu.py, table-tile variants).
#include "gemm_ring.h"

namespace {

constexpr int RC256_SLOT = 64 * 1024, RC256_IMG = 64 * 256, RC256_LDS = 2 * RC256_SLOT + 8 * 4096;

struct Rc256Desc {
    const uint16_t* aq; const uint16_t* bq;
    int M, N, K, lda, ldb, pi, m0, n0;
    uint32_t flags;
    float* bias_grad;
};
__device__ __forceinline__ Rc256Desc rc256_desc(const GemmBatch& gb, int bpos) {
    Rc256Desc D;
    const uint32_t rec = (uint32_t)__builtin_amdgcn_readfirstlane((int)gb.tile_rec[bpos]);
    D.pi = (int)(rec & 0xFFFFu); D.m0 = (int)((rec >> 16) & 0xFFu) * 256; D.n0 = (int)(rec >> 24) * 256;
    const GemmProblem& P = gb.table[D.pi];
    D.aq = P.a.q[0]; D.bq = P.b.q[0]; D.M = P.M; D.N = P.N; D.K = P.a.k[0]; D.lda = P.a.ldq[0]; D.ldb = P.b.ldq[0];
    D.flags = P.flags; D.bias_grad = P.bias_grad;
    return D;
}

__global__ __launch_bounds__(512) void m2f_gemm16_rc256_kernel(const GemmBatch gb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int BK = 64, MI = 4, NI = 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));       // provably uniform: LDS destinations go through M0
    const int wm = wave >> 2, wn = wave & 3;
    const int first = __builtin_amdgcn_readfirstlane(gb.wg_begin[blockIdx.x]);
    const int last = __builtin_amdgcn_readfirstlane(gb.wg_begin[blockIdx.x + 1]);
    if (first >= last) return;

    auto rsrc_of = [](const uint16_t* q, int rows, int ld) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(q);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(rows * ld * 2), 0x00020000);
    };
    // ---- load side: this wave's 8 pieces (1 KiB = 4 k-rows x 256 bytes each) of every k-tile; waves 0-3 stage A, 4-7 stage B ----
    const bool loadsB = wave >= 4;
    const int kr = lane >> 4, p16 = lane & 15;
    unsigned off[8], kstep = 0;
    m2f_rsrc_t rs;
    auto set_loads = [&](const Rc256Desc& D) {
        const int ld = loadsB ? D.ldb : D.lda, x0 = loadsB ? D.n0 : D.m0;
        rs = rsrc_of(loadsB ? D.bq : D.aq, D.K, ld);            // rows of the buffer = tokens: loads past the last token land as zeros
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pc = ((wave & 3) * 8 + j);                // piece of this operand: image pc >> 4, k-rows 4 (pc & 15) ..
            off[j] = (unsigned)(((pc & 15) * 4 + kr) * ld + x0 + 128 * (pc >> 4)) * 2u + 16u * (unsigned)(p16 ^ (4 * kr));
        }
        kstep = (unsigned)(BK * ld) * 2u;
    };
    // (Issuing the eight pieces in one burst keeps the wave in its issue loop for 577-1485 cycles per k-tile - sixty-four LDS-DMA
    //  instructions arrive at the CU's texture addresser together.  Spreading them over the four k-slices was tried: the longer
    //  live ranges spill 29 VGPRs and the k-tile got slower, 5.5k cycles against 5.2k.)
    auto issue = [&](int kt, int slot) {
        char* dst = smem + slot * RC256_SLOT + wave * 8192;     // A images at [0, 32 KiB), B images behind them
        const unsigned kb = (unsigned)kt * kstep;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + j * 1024), 16, off[j] + kb, 0, 0, 0);
    };
    // ---- multiply side: fragment addresses (gemm_ring.h, RC form) ----
    int rc_row, rc_a[MI], rc_b[NI];
    {
        const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
        rc_row = (8 * (g >> 1) + q) * 256 + 8 * (pp & 1);
#pragma unroll
        for (int i = 0; i < MI; ++i) rc_a[i] = wm * RC256_IMG + ((((i * 32 + 16 * (g & 1)) / 8 + (pp >> 1)) ^ (4 * q)) << 4);
#pragma unroll
        for (int j = 0; j < NI; ++j)
            rc_b[j] = 2 * RC256_IMG + (wn >> 1) * RC256_IMG + (((((wn & 1) * 64 + j * 32 + 16 * (g & 1)) / 8 + (pp >> 1)) ^ (4 * q)) << 4);
    }

    Rc256Desc cur = rc256_desc(gb, first);
    set_loads(cur);
    issue(0, 0);
    int slot = 0;
    M2F_ACC_DECL;
    unsigned long long t_ep = 0, n_kt = 0;
#pragma unroll 1
    for (int bpos = first; bpos < last; ++bpos) {
        // the NEXT tile's descriptor: scalar loads that have the whole tile to arrive
        const bool has_next = bpos + 1 < last;
        const Rc256Desc nxt = rc256_desc(gb, has_next ? bpos + 1 : bpos);
        const GemmProblem& P = gb.table[cur.pi];
        const int m0 = cur.m0, n0 = cur.n0, nk = (cur.K + BK - 1) / BK;
        const bool reluB = cur.flags & GF_RELU_B;
        const bool bgrad = cur.bias_grad && n0 == 0 && wn == 0;          // wave-uniform: this wave sums its rows of A over k
        const RingEpi E = ring_epilogue_args<256, 256>(gb, P, m0, n0);
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
        auto kloop = [&](auto rb_tag, auto bg_tag) {
            constexpr bool RELU_B = decltype(rb_tag)::value, BGRAD = decltype(bg_tag)::value;
#pragma unroll 1
            for (int kt = 0; kt < nk; ++kt) {
                const unsigned long long ts0 = M2F_NOW();
                ring_wait_vm<0>();                                   // this wave's pieces of k-tile kt have landed ...
                const unsigned long long ts1 = M2F_NOW();
                ring_lds_barrier();                                  // ... everybody's have, and everybody is done reading the other slot
                const unsigned long long ts2 = M2F_NOW();
                if (kt + 1 < nk) issue(kt + 1, slot ^ 1);
                else if (has_next) { set_loads(nxt); issue(0, slot ^ 1); }
                const unsigned long long ts3 = M2F_NOW();
                M2F_ADD(8, ts1 - ts0); M2F_ADD(9, ts2 - ts1); M2F_ADD(10, ts3 - ts2);
                const char* img0 = smem + slot * RC256_SLOT + rc_row;
                bf16x8 fa[MI], fb[NI];
                // (one set of fragments: the SIMD's other wave multiplies while this one waits for its reads - a second set would
                //  not fit beside 128 accumulator registers)
#pragma unroll
                for (int ks = 0; ks < BK / 16; ++ks) {
                    const char* img = img0 + ks * 16 * 256;
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const char* q = img + rc_a[i];
                        const ring_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ring_s16x4 __attribute__((address_space(3)))*)(const_cast<char*>(q)));
                        const ring_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ring_s16x4 __attribute__((address_space(3)))*)(const_cast<char*>(q + 4 * 256)));
                        const ring_s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        fa[i] = __builtin_bit_cast(bf16x8, r);
                    }
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const char* q = img + rc_b[j];
                        const ring_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ring_s16x4 __attribute__((address_space(3)))*)(const_cast<char*>(q)));
                        const ring_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ring_s16x4 __attribute__((address_space(3)))*)(const_cast<char*>(q + 4 * 256)));
                        const ring_s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        fb[j] = __builtin_bit_cast(bf16x8, r);
                    }
                    if constexpr (BGRAD) {
#pragma unroll
                        for (int i = 0; i < MI; ++i) {
                            const ring_u32x4 w = __builtin_bit_cast(ring_u32x4, fa[i]);
                            bsum[i] = ring_bf16x2_sum(w.w, ring_bf16x2_sum(w.z, ring_bf16x2_sum(w.y, ring_bf16x2_sum(w.x, bsum[i]))));
                        }
                    }
                    if constexpr (RELU_B) {
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            ring_u32x4 w = __builtin_bit_cast(ring_u32x4, fb[j]);
                            w.x = ring_relu_bf16x2(w.x); w.y = ring_relu_bf16x2(w.y); w.z = ring_relu_bf16x2(w.z); w.w = ring_relu_bf16x2(w.w);
                            fb[j] = __builtin_bit_cast(bf16x8, w);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // operands swapped: ring_epilogue
                }
                slot ^= 1;
                M2F_ADD(11, M2F_NOW() - ts3);
            }
        };
        {
            const std::true_type T1{}; const std::false_type F0{};
            if (bgrad) { if (reluB) kloop(T1, T1); else kloop(F0, T1); }
            else { if (reluB) kloop(T1, F0); else kloop(F0, F0); }
        }
        if (bgrad) {                                                // lanes l and l + 32 hold the two k-halves of row l
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const float tot = bsum[i] + __shfl_xor(bsum[i], 32);
                const int m = m0 + wm * 128 + i * 32 + (lane & 31);
                if (lane < 32 && m < cur.M) cur.bias_grad[m] = tot;
            }
        }
        // the first k-tile of the next tile is in flight while this one is stored.  (BN = 128 in the epilogue's template: its
        // column origin is n0 + wn * (BN / 2) = n0 + 64 wn, which is this wave's with wn in 0..3)
        const unsigned long long te0 = M2F_NOW();
        ring_epilogue<MI, NI, 256, 128, 1>(gb, E, acc, m0, n0, lane, wm, wn, smem + 2 * RC256_SLOT + wave * 4096);
        t_ep += M2F_NOW() - te0; n_kt += (unsigned long long)nk;
        cur = nxt;
    }
    ring_wait_vm<0>();
    M2F_ACC_FLUSH();
#ifdef M2F_EXP_TIMING
    if (blockIdx.x == 0 && (threadIdx.x & 255) == 64) { m2f_ring_dbg[12 + (threadIdx.x >= 256 ? 16 : 0)] += t_ep; m2f_ring_dbg[13 + (threadIdx.x >= 256 ? 16 : 0)] += n_kt; }
#else
    (void)t_ep; (void)n_kt;
#endif
}

}  // namespace

hipError_t m2f_ring_launch_table_rc_256x256(const GemmBatch& gb, hipStream_t stream) {
    static bool attr_set = false;
    auto kern = m2f_gemm16_rc256_kernel;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, RC256_LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (!gb.tile_rec || !gb.wg_begin || gb.wg_count < 1 || !gb.table) return hipErrorInvalidValue;
    ++m2f_g_ring_launches;
    hipLaunchKernelGGL(kern, dim3(gb.wg_count), dim3(512), RC256_LDS, stream, gb);
    return hipGetLastError();
}
#ifdef M2F_EXP_TIMING
// diagnostic build only: phase totals (cycles) of waves 1 and 5 of workgroup 0: [8] waiting for the own loads, [9] at the barrier,
// [10] issuing, [11] fragments + MFMAs, [12] epilogues, [13] k-tiles
extern "C" int m2f_rc256_dbg_read(unsigned long long* out, int reset) {
    int r = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(m2f_ring_dbg), sizeof(unsigned long long) * 64);
    if (reset) {
        unsigned long long z[64] = {0};
        r |= (int)hipMemcpyToSymbol(HIP_SYMBOL(m2f_ring_dbg), z, sizeof(z));
    }
    return r;
}
#endif
