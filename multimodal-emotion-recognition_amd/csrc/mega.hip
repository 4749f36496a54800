// Strip-dataflow persistent kernel of the M2FNet step for gfx950 (interface, protocol and rationale: mega.h).
//
// One workgroup = 8 wavefronts, one per CU (the kernel declares the whole 160 KB of LDS).  Item bodies:
//   * GEMM tile 64x64, k-tiles of 128, bf16 operands (k-contiguous "NT" form; the input-gradient GEMMs run against the W^T
//     shadows): waves 4-7 PRODUCE (global -> registers -> LDS, two k-tiles in flight, inline-asm loads with hand-counted
//     vmcnt; the weight tiles are requested BEFORE the dependency wait, the activation tiles - sc1 loads - after it),
//     waves 0-3 CONSUME (LDS fragments -> v_mfma_f32_32x32x16_bf16, same k order as m2f_gemm16_kernel, so results are
//     bit-identical to the launch-list path) and run the fused epilogue through an LDS transpose: every global access of
//     the epilogue is a 16-byte write-through (sc1) access of whole row segments.
//   * attention forward / backward: the bodies of m2f_attn_fwd_kernel / m2f_attn_bwd_kernel (attention.hip) with the two
//     halves of the workgroup working on two (dialogue, head) problems at once.
//   * LayerNorm forward / backward: the bodies of m2f_ln_fwd_kernel / m2f_ln_bwd_kernel (rowops.hip), 8 rows per item.
//   * in-place dropout (backward of the post-projection dropout, reference src/model.py:113,125).
// Reference arithmetic: src/model.py:13-20,102-145 and torch's TransformerEncoderLayer (see gemm.hip / attention.hip /
// rowops.hip, whose kernels these bodies restate operation for operation).
#include "common.h"
#include "mega.h"

// No floating-point contraction in this file: the launch-list kernels and the persistent kernel (mega.hip) restate the same
// formulas in different surroundings, and with -ffp-contract=fast (the HIP default) the compiler is free to fuse a*b+c in one
// of them and not in the other - a 1-ulp difference that would hide real hand-off bugs from the bit-for-bit comparison of
// the two paths (tests/test_mega_gpu.py).  These kernels are bound by memory or by MFMA, not by VALU multiplies.
#pragma clang fp contract(off)

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int MG_BM = 64, MG_BK = 128;                  // tile 64 x 64, k-tiles of 128
// A staged operand tile is a LINEAR LDS image [64 rows][16 chunks of 16 bytes] (what an LDS-direct load writes: one
// wave-instruction = 1 KB = 4 rows) with the chunks of row r XOR-swizzled: chunk c sits at position c ^ (r & 15).  The
// swizzle is applied to the per-lane SOURCE address and undone by the fragment reads (cdna_hip_programming.md rule 21);
// a 16-lane group of ds_read_b128 then touches 16 different positions = all 64 banks: conflict-free.
constexpr int MG_TILE = MG_BM * MG_BK * 2;              // 16 KB per operand tile
constexpr int MG_NBUF = 4;                              // ring: one tile being multiplied, three landed or in flight
constexpr int LDS_MISC_BYTES = 256;
constexpr int LDS_EP_STRIDE = 36;                          // floats per row of a consumer wave's 32x32 epilogue image
constexpr int LDS_EP_BYTES = 4 * 32 * LDS_EP_STRIDE * 4;
constexpr int LDS_MISC_OFF = M2F_MEGA_LDS - LDS_MISC_BYTES;
constexpr int LDS_EP_OFF = LDS_MISC_OFF - LDS_EP_BYTES;    // [0, LDS_EP_OFF): GEMM operand rings | attention slabs | LN partials
static_assert(LDS_EP_OFF == M2F_MEGA_LDS_WORK, "mega.h");
static_assert(2 * MG_NBUF * MG_TILE <= LDS_EP_OFF, "LDS budget");
enum { MISC_ABORT = 0, MISC_EPOCH = 1, MISC_ARRIVE4 = 2, MISC_ARRIVE8 = 3, MISC_TICKET = 4 /* and 5: ticket of the item after this one, by parity */ };

// give-up codes in status[0]
enum { MEGA_OK = 0, MEGA_TIMEOUT = 1 };
constexpr unsigned long long MEGA_SPIN_LIMIT = 50000000ull;     // s_memrealtime ticks (100 MHz): 0.5 s

// ---------------------------------------------------------------------------------------------------------
// write-through (sc1) global accesses through buffer descriptors: the only way activations move between items
// ---------------------------------------------------------------------------------------------------------
constexpr int AUX_SC1 = 16;
__device__ __forceinline__ rsrc_t mk_rsrc(const void* p) {       // p must be wave-uniform; make that provable (T20)
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ f32x4 ld4(rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX_SC1));
}
__device__ __forceinline__ float ld1(rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, AUX_SC1));
}
__device__ __forceinline__ void st4(rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, AUX_SC1);
}
__device__ __forceinline__ void st1(rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, AUX_SC1);
}
__device__ __forceinline__ void st_u4(rsrc_t r, unsigned off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX_SC1); }
__device__ __forceinline__ void st_u2(rsrc_t r, unsigned off, u32x2 v) { __builtin_amdgcn_raw_buffer_store_b64(v, r, off, 0, AUX_SC1); }
__device__ __forceinline__ void st_h(rsrc_t r, unsigned off, uint16_t v) { __builtin_amdgcn_raw_buffer_store_b16(v, r, off, 0, AUX_SC1); }
__device__ __forceinline__ unsigned pack_bf16(float a, float b) { return (unsigned)m2f_bf16_bits(a) | ((unsigned)m2f_bf16_bits(b) << 16); }

__device__ __forceinline__ void lds_barrier() {       // LDS-only workgroup barrier: never drains the loads in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned lds_load_u32(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store_u32(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#ifdef M2F_MEGA_PROF
// diagnostic build: ticks of s_memrealtime (100 MHz) per item phase, summed per kind (never in the shipped library)
#define PROF_NOW() __builtin_amdgcn_s_memrealtime()
__device__ __forceinline__ void prof_add(const MegaArgs& a, int kind, int field, unsigned long long v) {
    if (a.prof) atomicAdd(a.prof + kind * 8 + field, v);
}
#else
#define PROF_NOW() 0ull
__device__ __forceinline__ void prof_add(const MegaArgs&, int, int, unsigned long long) {}
#endif

// ---------------------------------------------------------------------------------------------------------
// dependency protocol
// ---------------------------------------------------------------------------------------------------------
// ONE lane: wait until every strip of the item has been completed by all earlier ops.  false = give up (the status word says why).
__device__ __forceinline__ bool mega_poll(const MegaArgs& a, const MegaItem& it, int idx) {
    for (int s = 0; s < (int)it.nstrips; ++s) {
        const unsigned need = a.need[(size_t)it.op * a.n_strips + it.s0 + s];
        if (need == 0) continue;
        const unsigned* p = a.progress + (size_t)(it.s0 + s) * 32;
        unsigned have = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (have >= need) continue;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        while (true) {
            __builtin_amdgcn_s_sleep(1);
            have = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (have >= need) break;
            if ((++spins & 31u) == 0u) {
                if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != MEGA_OK) return false;
                if (__builtin_amdgcn_s_memrealtime() - t0 > MEGA_SPIN_LIMIT) {
                    a.status[1] = (unsigned)idx; a.status[2] = (unsigned)(it.s0 + s); a.status[3] = have;
                    __hip_atomic_store(a.status, (unsigned)MEGA_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return false;
                }
            }
        }
    }
    return true;
}
// Every storing wave: drain its stores, arrive on an LDS counter; the last arriver publishes the item on its strips.
__device__ __forceinline__ void mega_arrive(const MegaArgs& a, const MegaItem& it, unsigned* cnt, unsigned mask, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (((old + 1u) & mask) == 0u) {
            for (int s = 0; s < (int)it.nstrips; ++s)
                __hip_atomic_fetch_add(a.progress + (size_t)(it.s0 + s) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// Ticket of the workgroup's NEXT item: drawn by wave 4 / lane 0 at the start of the current item (the atomic's latency hides
// behind the item), published in LDS before the first barrier of the current item, read by every wave at the next loop top.
__device__ __forceinline__ unsigned mega_draw(const MegaArgs& a, int xcc) {
    return __hip_atomic_fetch_add(a.queue + 32 * xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// All-wave items: wave 4 draws the next ticket and polls, everybody meets at a barrier.  false = the launch has given up
// (uniform for the workgroup).
__device__ __forceinline__ bool mega_wait_all(const MegaArgs& a, const MegaItem& it, int idx, unsigned* misc, int wave, int lane,
                                              int xcc, unsigned nth) {
    if (wave == 4 && lane == 0) {
        const unsigned tn = mega_draw(a, xcc);
        if (lds_load_u32(misc + MISC_ABORT) == 0u && !mega_poll(a, it, idx)) lds_store_u32(misc + MISC_ABORT, 1u);
        lds_store_u32(misc + MISC_TICKET + ((nth + 1u) & 1u), tn);
    }
    __syncthreads();
    return lds_load_u32(misc + MISC_ABORT) == 0u;
}

// ---------------------------------------------------------------------------------------------------------
// GEMM tile
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned relu_bf16x2(unsigned v) {
    const unsigned m = ((v >> 15) & 0x00010001u) * 0xFFFFu;
    return v & ~m;
}

// epilogue of one element: +bias -> relu -> dropout(site) -> +res -> ReLU gate -> (+= C); same order as gemm_epilogue (gemm.hip)
__device__ __forceinline__ float ep_value(float acc, float bias, float res, float gate, float cprev, bool relu_out, bool has_gate,
                                          float gscale, unsigned site, unsigned key, unsigned thresh, float dscale, unsigned flat) {
    float x = acc + bias;
    if (relu_out) x = fmaxf(x, 0.f);
    if (site) x = m2f_keep(key, flat, thresh) ? x * dscale : 0.f;
    x = x + res;
    if (has_gate) x = gate > 0.f ? x * gscale : 0.f;
    return x + cprev;
}

// The consumers' view of an item whose epilogue stores have been issued but not yet published
struct Pending { int s0, nstrips; bool on; };
__device__ __forceinline__ void mega_flush(const MegaArgs& a, Pending& pd, unsigned* misc, int lane) {
    if (!pd.on) return;                                          // uniform for the consumer waves
    MegaItem it; it.s0 = (uint16_t)pd.s0; it.nstrips = (uint8_t)pd.nstrips;
    mega_arrive(a, it, misc + MISC_ARRIVE4, 3u, lane);
    pd.on = false;
}

// Consumers (waves 0-3).  returns false when the launch has given up (decided by wave 4 BEFORE it published the epoch
// word and arrived at barrier B0: every wave reads the abort word behind B0, so the whole workgroup leaves at the same point).
//
// Publishing an item needs every storing wave to drain its write-through stores (s_waitcnt vmcnt(0): ~3 us of memory
// round trip).  The drain is DEFERRED while the workgroup has work: the stores of item i are left in flight, the waves go
// on to item i+1 and publish i after i+1's k-loop, when the drain is free.  While the workgroup is stalled - the
// producers' poll for i+1 has not come through - the pending item is published at once, which also covers the case that
// i+1 depends on i itself (same strip, next op): no deadlock, and a latency-bound chain is not slowed down.
__device__ __forceinline__ bool mega_gemm_consumer(const MegaArgs& a, const MegaItem& it, const GemmProblem& P, char* smem,
                                                   unsigned* misc, int wave, int lane, unsigned seq, Pending& pd) {
    const int m0 = it.a, n0 = it.b;
    const int M = P.M, N = P.N;
    const int nk = (P.a.k[0] + MG_BK - 1) / MG_BK + (P.a.k[1] + MG_BK - 1) / MG_BK;
    const int wm = wave >> 1, wn = wave & 1;
    const char* ldsA = smem;
    const char* ldsB = smem + MG_NBUF * MG_TILE;
    const bool reluA = P.flags & GF_RELU_A;
    const unsigned long long t_top = PROF_NOW();
    // ---- wait until the producers' dependency poll for this item is through; publish the pending item meanwhile -----
    while (lds_load_u32(misc + MISC_EPOCH) < seq) {
        if (pd.on) mega_flush(a, pd, misc, lane);
        else __builtin_amdgcn_s_sleep(1);
    }
    // ---- epilogue operands (residual / ReLU gate / accumulate): requested now, consumed after the k-loop -----------
    const unsigned flags = P.flags;
    const bool relu_out = flags & GF_RELU_OUT, accum = flags & GF_ACCUM;
    const float* __restrict__ bias = P.bias;
    const bool has_res = P.res != nullptr, has_gate = P.gate != nullptr;
    const int ldc = P.ldc, ldres = P.ldres, ldgate = P.ldgate;
    const float gscale = P.gate_scale;
    const unsigned site = P.drop_site;
    uint16_t* c16p = (ldc & 7) ? nullptr : m2f_shadow_of(a.sh, P.c);
    const rsrc_t rc = mk_rsrc(P.c);
    const rsrc_t rres = mk_rsrc(has_res ? P.res : P.c);
    const rsrc_t rgate = mk_rsrc(has_gate ? P.gate : P.c);
    const rsrc_t rc16 = mk_rsrc(c16p ? (const void*)c16p : (const void*)P.c);
    const bool al = ((reinterpret_cast<uintptr_t>(P.c) & 15) == 0) && ((ldc & 3) == 0) &&
                    (!has_res || (((reinterpret_cast<uintptr_t>(P.res) & 15) == 0) && ((ldres & 3) == 0))) &&
                    (!has_gate || (((reinterpret_cast<uintptr_t>(P.gate) & 15) == 0) && ((ldgate & 3) == 0))) &&
                    (!c16p || ((reinterpret_cast<uintptr_t>(c16p) & 15) == 0));
    const int c8 = 8 * (lane & 3);
    const int col = n0 + wn * 32 + c8;
    const bool vec_lane = al && col + 8 <= N;
    f32x4 pre[2][6];                                                 // [pass][res lo, res hi, gate lo, gate hi, C lo, C hi]
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = m0 + wm * 32 + (lane >> 2) + 16 * p;
#pragma unroll
        for (int q = 0; q < 6; ++q) pre[p][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (vec_lane && row < M) {
            const unsigned oc = (unsigned)(row * ldc + col), ores = (unsigned)(row * ldres + col), og = (unsigned)(row * ldgate + col);
            if (has_res) { pre[p][0] = ld4(rres, ores * 4u); pre[p][1] = ld4(rres, ores * 4u + 16u); }
            if (has_gate) { pre[p][2] = ld4(rgate, og * 4u); pre[p][3] = ld4(rgate, og * 4u + 16u); }
            if (accum) { pre[p][4] = ld4(rc, oc * 4u); pre[p][5] = ld4(rc, oc * 4u + 16u); }
        }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    lds_barrier();                                                  // (B0) tile 0 has landed
    if (lds_load_u32(misc + MISC_ABORT) != 0u) return false;
    const unsigned long long t_b0 = PROF_NOW();
    // fragment of k-slice ks: row (lane & 31) of the wave's 32-row block, chunk 2 ks + (lane >> 5), at its swizzled position
    const int x = lane & 15, h = lane >> 5;
    const int arow = (wm * 32 + (lane & 31)) * (MG_BK * 2), brow = (wn * 32 + (lane & 31)) * (MG_BK * 2);
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const char* ab = ldsA + slot * MG_TILE + arow;
        const char* bb = ldsB + slot * MG_TILE + brow;
        constexpr int KS = MG_BK / 16;
        bf16x8 fa[KS], fb[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int o = ((2 * ks + h) ^ x) << 4;
            fa[ks] = *reinterpret_cast<const bf16x8*>(ab + o);
            fb[ks] = *reinterpret_cast<const bf16x8*>(bb + o);
        }
        if (reluA) {                                                // block-uniform: relu(cat(x, text)) of the fusion layer's Linear
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                u32x4 w = __builtin_bit_cast(u32x4, fa[ks]);
                w.x = relu_bf16x2(w.x); w.y = relu_bf16x2(w.y); w.z = relu_bf16x2(w.z); w.w = relu_bf16x2(w.w);
                fa[ks] = __builtin_bit_cast(bf16x8, w);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], fb[ks], acc, 0, 0, 0);
        lds_barrier();                                              // (B1 per k-tile)
        slot = slot == MG_NBUF - 1 ? 0 : slot + 1;
    }
    const unsigned long long t_k = PROF_NOW();
    mega_flush(a, pd, misc, lane);                                  // the previous item's stores have long landed

    // ---- epilogue: accumulators -> this wave's LDS image -> row-major 8-column groups -> 16-byte sc1 accesses ----------
    float* ep = reinterpret_cast<float*>(smem + LDS_EP_OFF) + wave * 32 * LDS_EP_STRIDE;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        ep[row * LDS_EP_STRIDE + (lane & 31)] = acc[r];
    }
    // same wave, in-order LDS queue: the writes land before the reads below; the compiler, however, sees float stores and
    // f32x4 loads (no common type) and would hoist the loads - pin the order
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    unsigned key = 0;
    if (site) key = m2f_site_key(a.rng, site);
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (bias && col + e < N) ? bias[col + e] : 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int rl = (lane >> 2) + 16 * p;
        const int row = m0 + wm * 32 + rl;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(ep + rl * LDS_EP_STRIDE + c8);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(ep + rl * LDS_EP_STRIDE + c8 + 4);
        const float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        if (row >= M || col >= N) continue;
        const unsigned oc = (unsigned)(row * ldc + col), ores = (unsigned)(row * ldres + col), og = (unsigned)(row * ldgate + col);
        float out[8];
        if (vec_lane) {
            const float rv[8] = {pre[p][0][0], pre[p][0][1], pre[p][0][2], pre[p][0][3], pre[p][1][0], pre[p][1][1], pre[p][1][2], pre[p][1][3]};
            const float gv[8] = {pre[p][2][0], pre[p][2][1], pre[p][2][2], pre[p][2][3], pre[p][3][0], pre[p][3][1], pre[p][3][2], pre[p][3][3]};
            const float cv[8] = {pre[p][4][0], pre[p][4][1], pre[p][4][2], pre[p][4][3], pre[p][5][0], pre[p][5][1], pre[p][5][2], pre[p][5][3]};
#pragma unroll
            for (int e = 0; e < 8; ++e)
                out[e] = ep_value(xv[e], bv[e], rv[e], gv[e], cv[e], relu_out, has_gate, gscale, site, key, a.drop_thresh, a.drop_scale,
                                  (unsigned)row * (unsigned)N + (unsigned)(col + e));
            st4(rc, oc * 4u, (f32x4){out[0], out[1], out[2], out[3]});
            st4(rc, oc * 4u + 16u, (f32x4){out[4], out[5], out[6], out[7]});
            if (c16p) st_u4(rc16, oc * 2u, (u32x4){pack_bf16(out[0], out[1]), pack_bf16(out[2], out[3]), pack_bf16(out[4], out[5]), pack_bf16(out[6], out[7])});
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (col + e < N) {
                    const float rv = has_res ? ld1(rres, (ores + e) * 4u) : 0.f;
                    const float gv = has_gate ? ld1(rgate, (og + e) * 4u) : 0.f;
                    const float cv = accum ? ld1(rc, (oc + e) * 4u) : 0.f;
                    const float o = ep_value(xv[e], bv[e], rv, gv, cv, relu_out, has_gate, gscale, site, key, a.drop_thresh, a.drop_scale,
                                             (unsigned)row * (unsigned)N + (unsigned)(col + e));
                    st1(rc, (oc + e) * 4u, o);
                    if (c16p) st_h(rc16, (oc + e) * 2u, m2f_bf16_bits(o));
                }
            }
        }
    }
    pd.s0 = it.s0; pd.nstrips = it.nstrips; pd.on = true;           // published later (mega_flush)
#ifdef M2F_MEGA_PROF
    if (wave == 0 && lane == 0) {
        const unsigned long long t_e = PROF_NOW();
        prof_add(a, MK_GEMM, 0, 1); prof_add(a, MK_GEMM, 1, t_e - t_top); prof_add(a, MK_GEMM, 2, t_b0 - t_top);
        prof_add(a, MK_GEMM, 3, t_k - t_b0); prof_add(a, MK_GEMM, 4, t_e - t_k); prof_add(a, MK_GEMM, 6, (unsigned long long)nk);
    }
#endif
    return true;
}

// Producers (waves 4-7): LDS-direct loads (buffer_load_dwordx4 ... lds: no destination registers, nothing for the register
// allocator to move while a load is in flight).  Per k-tile and operand a wave issues 4 instructions of 1 KB (4 rows each).
// Out-of-range rows and chunks past the padded reduction length are range-checked away by the buffer descriptor (they
// land as zeros), so edge tiles need no masking pass.  The weight tiles of the first two k-tiles are requested BEFORE the
// dependency wait (they do not depend on the predecessor), the activation tiles - sc1 loads - right after it.
__device__ __forceinline__ bool mega_gemm_producer(const MegaArgs& a, const MegaItem& it, int idx, const GemmProblem& P, char* smem,
                                                   unsigned* misc, int wave, int lane, unsigned seq, int xcc, unsigned nth) {
    typedef __attribute__((address_space(3))) void lds_void;
    unsigned tnext = 0;
    if (wave == 4 && lane == 0) tnext = mega_draw(a, xcc);          // the next item's ticket; lands while this item streams
    const int pw = wave - 4;
    const int m0 = it.a, n0 = it.b;
    const int M = P.M, N = P.N;
    const int ak0 = P.a.k[0], ak1 = P.a.k[1];
    const int ald0 = P.a.ldq[0], ald1 = P.a.ldq[1], bld0 = P.b.ldq[0], bld1 = P.b.ldq[1];
    const int nk0 = (ak0 + MG_BK - 1) / MG_BK, nk = nk0 + (ak1 + MG_BK - 1) / MG_BK;
    auto rsrc_of = [](const uint16_t* q, int rows, int ld) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(q);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(rows * ld * 2), 0x00020000);
    };
    const rsrc_t ra0 = rsrc_of(P.a.q[0], M, ald0), rb0 = rsrc_of(P.b.q[0], N, bld0);
    const rsrc_t ra1 = rsrc_of(ak1 ? P.a.q[1] : P.a.q[0], M, ak1 ? ald1 : ald0), rb1 = rsrc_of(ak1 ? P.b.q[1] : P.b.q[0], N, ak1 ? bld1 : bld0);
    char* ldsA = smem + pw * 4096;                               // this wave's 16 rows of a tile
    char* ldsB = smem + MG_NBUF * MG_TILE + pw * 4096;
    // lane -> (row within the 4-row piece, chunk position); instruction j covers tile rows 16 pw + 4 j .. + 3
    const int lrow = lane >> 4, pos = lane & 15;
    constexpr unsigned OOB = 0x80000000u;
    auto issue_op = [&](bool is_a, int kt, int slot) {
        const int seg = kt >= nk0 ? 1 : 0, kbase = (seg ? kt - nk0 : kt) * MG_BK;
        const int kpad = ((seg ? ak1 : ak0) + 7) & ~7;
        const int ld = is_a ? (seg ? ald1 : ald0) : (seg ? bld1 : bld0);
        const int row0 = (is_a ? m0 : n0) + 16 * pw + lrow;
        char* dst = (is_a ? ldsA : ldsB) + slot * MG_TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = pos ^ (4 * j + lrow);                      // chunk whose home is this lane's position in row 4 j + lrow (mod 16)
            const int k = kbase + 8 * c;
            const unsigned voff = k < kpad ? (unsigned)((row0 + 4 * j) * ld + k) * 2u : OOB;
            if (is_a) {
                if (seg) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra1, (lds_void*)(dst + j * 1024), 16, voff, 0, 0, AUX_SC1);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(ra0, (lds_void*)(dst + j * 1024), 16, voff, 0, 0, AUX_SC1);
            } else {
                if (seg) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb1, (lds_void*)(dst + j * 1024), 16, voff, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb0, (lds_void*)(dst + j * 1024), 16, voff, 0, 0, 0);
            }
        }
    };
    auto wait_vm = [](int n) {                                       // s_waitcnt takes an immediate
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        }
    };
    // weights of the first three k-tiles: before the dependency wait
    issue_op(false, 0, 0);
    if (nk > 1) issue_op(false, 1, 1);
    if (nk > 2) issue_op(false, 2, 2);
    // dependency: wave 4 polls the strip counters, the other producer waves wait for its LDS word
    if (wave == 4) {
        if (lane == 0) {
            const unsigned long long t_p = PROF_NOW();
            if (lds_load_u32(misc + MISC_ABORT) == 0u && !mega_poll(a, it, idx)) lds_store_u32(misc + MISC_ABORT, 1u);
            lds_store_u32(misc + MISC_TICKET + ((nth + 1u) & 1u), tnext);     // visible to every wave behind barrier B0
            lds_store_u32(misc + MISC_EPOCH, seq);
            prof_add(a, MK_GEMM, 5, PROF_NOW() - t_p);
        }
    } else {
        while (lds_load_u32(misc + MISC_EPOCH) < seq) __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const bool aborted = lds_load_u32(misc + MISC_ABORT) != 0u;      // written before the epoch word
    if (aborted) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();                                               // (B0): the consumers leave behind it as well
        return false;
    }
    issue_op(true, 0, 0);
    if (nk > 1) issue_op(true, 1, 1);
    if (nk > 2) issue_op(true, 2, 2);
    // loads complete in issue order: tile 0 is whole once only the activation pieces of tiles 1 and 2 (4 each) are outstanding
    wait_vm((nk > 1 ? 4 : 0) + (nk > 2 ? 4 : 0));
    lds_barrier();                                                   // (B0)
    int slot3 = 3;                                                   // ring slot of tile kt + 3
    for (int kt = 0; kt < nk; ++kt) {
        // the consumers multiply tile kt; slot (kt + 3) % 4 held tile kt - 1, whose reads finished before the last barrier
        if (kt + 3 < nk) { issue_op(false, kt + 3, slot3); issue_op(true, kt + 3, slot3); }
        // tile kt + 1 has landed once only what was issued after its last piece is outstanding
        int younger = (kt + 3 < nk ? 8 : 0);
        if (kt == 0) younger += (nk > 2 ? 4 : 0);                    // prologue order: ... A1 A2
        else younger += (kt + 2 < nk ? 8 : 0);
        wait_vm(kt + 1 < nk ? younger : 0);
        lds_barrier();                                               // (B1 per k-tile)
        slot3 = slot3 == MG_NBUF - 1 ? 0 : slot3 + 1;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// attention (bodies of attention.hip's kernels; one 256-thread half of the workgroup per (dialogue, head))
// ---------------------------------------------------------------------------------------------------------
constexpr int ANTHR = 256, ANWAVE = 4;

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// generic zero-padded slab copy (any alignment): scalar sc1 loads
__device__ __forceinline__ void load_slab(float* __restrict__ lds, int ld, int Lp, int W, rsrc_t src, int ldg, int L, int hd, int tid) {
    const int total = Lp * W;
#pragma unroll 1
    for (int base = 0; base < total; base += ANTHR * 4) {
        float x[4];
        int off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = base + tid + ANTHR * u;
            const int r = e / W, c = e - r * W;
            const bool ok = e < total && r < L && c < hd;
            off[u] = e < total ? r * ld + c : -1;
            x[u] = ld1(src, ok ? (unsigned)(r * ldg + c) * 4u : 0u);
            if (!ok) x[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (off[u] >= 0) lds[off[u]] = x[u];
    }
}
template <int NV>
struct SlabGeom { int goff_rc[NV]; int loff[NV]; bool inb[NV]; bool ok[NV]; };
template <int NV>
__device__ __forceinline__ void slab_geom(SlabGeom<NV>& G, int L, int hd, int Lp, int W, int ld, int tid) {
    const int C4 = W >> 2, total = Lp * C4;
    int r = tid / C4, c4 = tid - r * C4;
    const int dr = ANTHR / C4, dc = ANTHR - dr * C4;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int e = tid + ANTHR * u;
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
    return ((hd & 3) == 0) && ((ldg & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (Lp * (W >> 2) <= ANTHR * NV);
}
template <int NV> struct SlabRegs { f32x4 x[NV]; };
template <int NV>
__device__ __forceinline__ void slab_issue(SlabRegs<NV>& R, const SlabGeom<NV>& G, rsrc_t src, int ldg) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int r = G.goff_rc[u] >> 16, c = G.goff_rc[u] & 0xFFFF;
        const unsigned o = G.ok[u] ? (unsigned)(r * ldg + c) * 4u : 0u;
        R.x[u] = ld4(src, o);
    }
}
template <int NV>
__device__ __forceinline__ void slab_commit(const SlabRegs<NV>& R, const SlabGeom<NV>& G, float* __restrict__ lds) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        if (G.inb[u]) {
            const bool ok = G.ok[u];
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2* d = reinterpret_cast<f32x2*>(lds + G.loff[u]);
            d[0] = f32x2{ok ? R.x[u][0] : 0.f, ok ? R.x[u][1] : 0.f};
            d[1] = f32x2{ok ? R.x[u][2] : 0.f, ok ? R.x[u][3] : 0.f};
        }
    }
}

// `active` is uniform per half; an inactive half only keeps the barrier count
template <int NT>
__device__ __forceinline__ void mega_attn_fwd(const MegaArgs& a, const AttnProblem& P, int bh, bool active, float* sm, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    const int H = P.H, hd = P.hd, L = a.L;
    const int b = bh / H, h = bh - b * H;
    constexpr int Lp = 16 * NT;
    const int W = (hd + 15) & ~15, ld = W + 2;
    float* Qs = sm;
    float* Ks = Qs + Lp * ld;
    float* Vs = Ks + Lp * ld;
    const size_t tok0 = (size_t)b * L;
    const float* qg = P.q + tok0 * P.ldq + h * hd;
    const float* kg = P.k + tok0 * P.ldk + h * hd;
    const float* vg = P.v + tok0 * P.ldv + h * hd;
    unsigned char kpad = 0;
    if (active) {
        kpad = a.key_pad[tok0 + (lane < L ? lane : 0)];
        const rsrc_t rq = mk_rsrc(qg), rk = mk_rsrc(kg), rv = mk_rsrc(vg);
        constexpr int NV = 2 * NT;
        if (slab_fast_ok<NV>(qg, P.ldq, hd, Lp, W) && slab_fast_ok<NV>(kg, P.ldk, hd, Lp, W) && slab_fast_ok<NV>(vg, P.ldv, hd, Lp, W)) {
            SlabGeom<NV> G;
            slab_geom(G, L, hd, Lp, W, ld, tid);
            SlabRegs<NV> xq, xk, xv;
            slab_issue(xq, G, rq, P.ldq);
            slab_issue(xk, G, rk, P.ldk);
            slab_issue(xv, G, rv, P.ldv);
            slab_commit(xq, G, Qs);
            slab_commit(xk, G, Ks);
            slab_commit(xv, G, Vs);
        } else {
            load_slab(Qs, ld, Lp, W, rq, P.ldq, L, hd, tid);
            load_slab(Ks, ld, Lp, W, rk, P.ldk, L, hd, tid);
            load_slab(Vs, ld, Lp, W, rv, P.ldv, L, hd, tid);
        }
    }
    const unsigned long long kvalid = __ballot(lane < L && kpad == 0);
    __syncthreads();
    if (!active) return;

    const float scale = 1.0f / sqrtf((float)hd);
    const int l15 = lane & 15, lg = lane >> 4;
    const int ksteps = (hd + 3) >> 2;
    const unsigned site = P.drop_site;
    unsigned key = 0;
    if (site) key = m2f_site_key(a.rng, site);
    float* probs = P.probs + (size_t)bh * Lp * Lp;
    uint16_t* out16 = m2f_shadow_of(a.sh, P.out);
    const rsrc_t ro = mk_rsrc(P.out);
    const rsrc_t ro16 = mk_rsrc(out16 ? (const void*)out16 : (const void*)P.out);

#pragma unroll 1
    for (int it = 0; it < NT; ++it) {
        f32x4 s[NT];
        const int i = 16 * it + l15;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* kp = Ks + (16 * jt + l15) * ld + lg;
            const float* qp = Qs + i * ld + lg;
            int ks = 0;
            for (; ks + 1 < ksteps; ks += 2) {
                acc0 = mfma4(kp[4 * ks], qp[4 * ks], acc0);
                acc1 = mfma4(kp[4 * ks + 4], qp[4 * ks + 4], acc1);
            }
            if (ks < ksteps) acc0 = mfma4(kp[4 * ks], qp[4 * ks], acc0);
            s[jt] = acc0 + acc1;
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
        const float inv = (i < L) ? 1.0f / sum : 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * lg + r;
                float p = s[jt][r] * inv;
                if (wv == 0) probs[(size_t)j * Lp + i] = p;          // read by the backward launch only: plain store
                if (site) p = m2f_keep(key, (unsigned)((bh * L + i) * L + j), a.drop_thresh) ? p * a.drop_scale : 0.f;
                s[jt][r] = p;
            }
        for (int ct = wv; ct < (W >> 4); ct += ANWAVE) {
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
                    const unsigned idx = (unsigned)((tok0 + io) * P.ldo + h * hd + c);
                    st1(ro, idx * 4u, o[r]);
                    if (out16) st_h(ro16, idx * 2u, m2f_bf16_bits(o[r]));
                }
            }
        }
    }
}

template <int NT>
__device__ __forceinline__ void mega_attn_bwd(const MegaArgs& a, const AttnProblem& P, int bh, bool active, float* sm, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    const int H = P.H, hd = P.hd, L = a.L;
    const int b = bh / H, h = bh - b * H;
    constexpr int Lp = 16 * NT;
    const int W = (hd + 15) & ~15, ld = W + 2;
    const bool bwd_fast = a.attn_bwd_fast != 0;
    float* Qs = sm;
    float* Ks = Qs + Lp * ld;
    float* Vs = Ks + Lp * ld;
    float* Gs = Vs + Lp * ld;
    float* Os = Gs + Lp * ld;
    float* delta = Os + (bwd_fast ? Lp * ld : 0);
    const size_t tok0 = (size_t)b * L;
    const float* qg = P.q + tok0 * P.ldq + h * hd;
    const float* kg = P.k + tok0 * P.ldk + h * hd;
    const float* vg = P.v + tok0 * P.ldv + h * hd;
    const float* gg = P.dout + tok0 * P.lddo + h * hd;
    const float* og = P.out + tok0 * P.ldo + h * hd;
    const int l15 = lane & 15, lg = lane >> 4;
    const float* probs = P.probs + (size_t)bh * Lp * Lp;
    float px[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 py = {0.f, 0.f, 0.f, 0.f};
    constexpr int NV = 2 * NT;
    const bool fast = bwd_fast && slab_fast_ok<NV>(qg, P.ldq, hd, Lp, W) && slab_fast_ok<NV>(kg, P.ldk, hd, Lp, W) &&
                      slab_fast_ok<NV>(vg, P.ldv, hd, Lp, W) && slab_fast_ok<NV>(gg, P.lddo, hd, Lp, W) && slab_fast_ok<NV>(og, P.ldo, hd, Lp, W);
    const rsrc_t rog = mk_rsrc(og);
    if (active) {
        const rsrc_t rq = mk_rsrc(qg), rk = mk_rsrc(kg), rv = mk_rsrc(vg), rg = mk_rsrc(gg);
        if (fast) {
            SlabGeom<NV> G;
            slab_geom(G, L, hd, Lp, W, ld, tid);
            SlabRegs<NV> xq, xk, xv, xg, xo;
            slab_issue(xg, G, rg, P.lddo);
            slab_issue(xo, G, rog, P.ldo);
            slab_issue(xv, G, rv, P.ldv);
            slab_issue(xk, G, rk, P.ldk);
            slab_issue(xq, G, rq, P.ldq);
            if constexpr (NT == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) px[r] = probs[(size_t)(4 * lg + r) * Lp + l15];
                py = *reinterpret_cast<const f32x4*>(probs + (size_t)l15 * Lp + 4 * lg);
            }
            slab_commit(xg, G, Gs);
            slab_commit(xo, G, Os);
            slab_commit(xv, G, Vs);
            slab_commit(xk, G, Ks);
            slab_commit(xq, G, Qs);
        } else {
            load_slab(Qs, ld, Lp, W, rq, P.ldq, L, hd, tid);
            load_slab(Ks, ld, Lp, W, rk, P.ldk, L, hd, tid);
            load_slab(Vs, ld, Lp, W, rv, P.ldv, L, hd, tid);
            load_slab(Gs, ld, Lp, W, rg, P.lddo, L, hd, tid);
            if constexpr (NT == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) px[r] = probs[(size_t)(4 * lg + r) * Lp + l15];
                py = *reinterpret_cast<const f32x4*>(probs + (size_t)l15 * Lp + 4 * lg);
            }
        }
    }
    __syncthreads();
    if (active) {
        for (int r0 = 0; r0 < Lp; r0 += ANTHR / 4) {
            const int row = r0 + (tid >> 2), part = tid & 3;
            const bool rin = row < Lp, rok = row < L;
            const int rowc = rin ? row : 0;
            const float* g = Gs + rowc * ld;
            float d = 0.f;
            if (fast) {
                const float* o = Os + rowc * ld;
                for (int c = part; c < hd; c += 4) d += g[c] * o[c];
            } else {
                const unsigned ob = (unsigned)((rok ? row : 0) * P.ldo) * 4u;
                for (int c = part; c < hd; c += 4) d += g[c] * ld1(rog, ob + 4u * c);
            }
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            if (part == 0 && rin) delta[row] = rok ? d : 0.f;
        }
    }
    __syncthreads();
    if (!active) return;

    const float scale = 1.0f / sqrtf((float)hd);
    const int ksteps = (hd + 3) >> 2;
    const unsigned site = P.drop_site;
    unsigned key = 0;
    if (site) key = m2f_site_key(a.rng, site);
    uint16_t* dq16 = m2f_shadow_of(a.sh, P.dq);
    uint16_t* dk16 = m2f_shadow_of(a.sh, P.dk);
    uint16_t* dv16 = m2f_shadow_of(a.sh, P.dv);
    const rsrc_t rdq = mk_rsrc(P.dq), rdk = mk_rsrc(P.dk), rdv = mk_rsrc(P.dv);
    const rsrc_t rdq16 = mk_rsrc(dq16 ? (const void*)dq16 : (const void*)P.dq);
    const rsrc_t rdk16 = mk_rsrc(dk16 ? (const void*)dk16 : (const void*)P.dk);
    const rsrc_t rdv16 = mk_rsrc(dv16 ? (const void*)dv16 : (const void*)P.dv);

    // ---- orientation X: lane = query row i, registers = keys j  ->  dQ = dS K ----------------------
#pragma unroll 1
    for (int it = 0; it < NT; ++it) {
        f32x4 ds[NT];
        const int i = 16 * it + l15;
        const float dl = delta[i];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* vp = Vs + (16 * jt + l15) * ld + lg;
            const float* gp = Gs + i * ld + lg;
            int ks = 0;
            for (; ks + 1 < ksteps; ks += 2) {
                acc0 = mfma4(vp[4 * ks], gp[4 * ks], acc0);
                acc1 = mfma4(vp[4 * ks + 4], gp[4 * ks + 4], acc1);
            }
            if (ks < ksteps) acc0 = mfma4(vp[4 * ks], gp[4 * ks], acc0);
            const f32x4 acc = acc0 + acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * lg + r;
                const float p = (NT == 1) ? px[r] : probs[(size_t)j * Lp + i];
                float dp = acc[r];
                if (site) dp = m2f_keep(key, (unsigned)((bh * L + i) * L + j), a.drop_thresh) ? dp * a.drop_scale : 0.f;
                ds[jt][r] = p * (dp - dl) * scale;
            }
        }
        for (int ct = wv; ct < (W >> 4); ct += ANWAVE) {
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
                    const unsigned idx = (unsigned)((tok0 + io) * P.lddq + h * hd + c);
                    st1(rdq, idx * 4u, o[r]);
                    if (dq16) st_h(rdq16, idx * 2u, m2f_bf16_bits(o[r]));
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
            const float* gp = Gs + (16 * it + l15) * ld + lg;
            const float* vp = Vs + j * ld + lg;
            int ks = 0;
            for (; ks + 1 < ksteps; ks += 2) {
                acc0 = mfma4(gp[4 * ks], vp[4 * ks], acc0);
                acc1 = mfma4(gp[4 * ks + 4], vp[4 * ks + 4], acc1);
            }
            if (ks < ksteps) acc0 = mfma4(gp[4 * ks], vp[4 * ks], acc0);
            const f32x4 acc = acc0 + acc1;
            const f32x4 p4 = (NT == 1) ? py : *reinterpret_cast<const f32x4*>(probs + (size_t)j * Lp + 16 * it + 4 * lg);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + 4 * lg + r;
                float p = p4[r], dp = acc[r], pdv = p;
                if (site) {
                    const bool kp = m2f_keep(key, (unsigned)((bh * L + i) * L + j), a.drop_thresh);
                    dp = kp ? dp * a.drop_scale : 0.f;
                    pdv = kp ? p * a.drop_scale : 0.f;
                }
                ds[it][r] = p * (dp - delta[i]) * scale;
                pd[it][r] = pdv;
            }
        }
        for (int ct = wv; ct < (W >> 4); ct += ANWAVE) {
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
                    const unsigned ik = (unsigned)((tok0 + jo) * P.lddk + h * hd + c), iv = (unsigned)((tok0 + jo) * P.lddv + h * hd + c);
                    st1(rdk, ik * 4u, dk[r]);
                    st1(rdv, iv * 4u, dv[r]);
                    if (dk16) st_h(rdk16, ik * 2u, m2f_bf16_bits(dk[r]));
                    if (dv16) st_h(rdv16, iv * 2u, m2f_bf16_bits(dv[r]));
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm (bodies of rowops.hip's kernels): one wave per row, a 4-wave half per 4-row block
// ---------------------------------------------------------------------------------------------------------
template <int NV> struct RowRegs { f32x4 v[NV]; };
template <int NV>
__device__ __forceinline__ void prow_load(RowRegs<NV>& r, const float* __restrict__ p, int d, bool vec, int lane) {   // parameters: plain loads
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
__device__ __forceinline__ void row_load(RowRegs<NV>& r, rsrc_t src, unsigned row_off /*floats*/, int d, bool vec, int lane) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * (lane + 64 * j);
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (c < d) {
            if (vec) x = ld4(src, (row_off + c) * 4u);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c + e < d) x[e] = ld1(src, (row_off + c + e) * 4u);
            }
        }
        r.v[j] = x;
    }
}
template <int NV>
__device__ __forceinline__ void row_store(const RowRegs<NV>& r, rsrc_t dst, unsigned row_off, int d, bool vec, int lane) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * (lane + 64 * j);
        if (c < d) {
            if (vec) st4(dst, (row_off + c) * 4u, r.v[j]);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c + e < d) st1(dst, (row_off + c + e) * 4u, r.v[j][e]);
            }
        }
    }
}
template <int NV>
__device__ __forceinline__ void row_store_bf16(const RowRegs<NV>& r, rsrc_t dst, unsigned row_off, int d, bool al8, int lane) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * (lane + 64 * j);
        if (c + 3 < d && al8) {
            st_u2(dst, (row_off + c) * 2u, (u32x2){pack_bf16(r.v[j][0], r.v[j][1]), pack_bf16(r.v[j][2], r.v[j][3])});
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c + e < d) st_h(dst, (row_off + c + e) * 2u, m2f_bf16_bits(r.v[j][e]));
        }
    }
}
template <int NV>
__device__ __forceinline__ void lds_row_store(const RowRegs<NV>& r, float* __restrict__ p, int d, int lane) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * (lane + 64 * j);
        if (c < d) *reinterpret_cast<f32x4*>(p + c) = r.v[j];
    }
}
__device__ __forceinline__ bool is_vec(const void* p, int d) { return ((d & 3) == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0); }

// wave w of the workgroup: row = (blk0 + (w >> 2)) * 4 + (w & 3)
template <int NV>
__device__ __forceinline__ void mega_ln_fwd(const MegaArgs& a, const LnProblem& P, int blk0, int nblk, int wave, int lane) {
    const int half = wave >> 2;
    const int row = (blk0 + half) * M2F_LN_ROWS_PER_BLOCK + (wave & 3);
    if (half >= nblk || row >= a.T) return;                       // wave-uniform
    const int d = P.d;
    const int ld = P.ld ? P.ld : d;
    const bool vec = is_vec(P.x, d) && is_vec(P.out, d) && is_vec(P.gamma, d) && is_vec(P.beta, d) &&
                     (!P.res || is_vec(P.res, d)) && ((ld & 3) == 0);
    uint16_t* out16 = m2f_shadow_of(a.sh, P.out);
    const rsrc_t rx = mk_rsrc(P.x), rres = mk_rsrc(P.res ? P.res : P.x), rout = mk_rsrc(P.out);
    const rsrc_t rout16 = mk_rsrc(out16 ? (const void*)out16 : (const void*)P.out);
    RowRegs<NV> g, be;
    prow_load(g, P.gamma, d, vec, lane);
    prow_load(be, P.beta, d, vec, lane);
    unsigned key = 0;
    if (P.drop_site) key = m2f_site_key(a.rng, P.drop_site);
    const float invd = 1.0f / (float)d;
    const unsigned ro = (unsigned)row * (unsigned)ld;
    RowRegs<NV> x;
    row_load(x, rx, ro, d, vec, lane);
    RowRegs<NV> res;
    if (P.res) row_load(res, rres, ro, d, vec, lane);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s += (x.v[j][0] + x.v[j][1]) + (x.v[j][2] + x.v[j][3]);
    const float mean = m2f_wave_sum(s) * invd;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (lane + 64 * j) + e;
            const float t = (c < d) ? x.v[j][e] - mean : 0.f;
            q += t * t;
        }
    const float rstd = 1.0f / sqrtf(m2f_wave_sum(q) * invd + a.ln_eps);
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float y = (x.v[j][e] - mean) * rstd * g.v[j][e] + be.v[j][e];
            if (P.res) y += res.v[j][e];
            if (P.drop_site) {
                const int c = 4 * (lane + 64 * j) + e;
                y = m2f_keep(key, (unsigned)row * (unsigned)d + (unsigned)c, a.drop_thresh) ? y * a.drop_scale : 0.f;
            }
            x.v[j][e] = y;
        }
    row_store(x, rout, ro, d, vec, lane);
    if (out16) row_store_bf16(x, rout16, ro, d, ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(out16) & 7) == 0), lane);
    if (lane == 0) { P.stats[2 * row] = mean; P.stats[2 * row + 1] = rstd; }     // read by the backward launch: plain
}

template <int NV>
__device__ __forceinline__ void mega_ln_bwd(const MegaArgs& a, const LnProblem& P, int blk0, int nblk, float* lds, int wave, int lane) {
    const int half = wave >> 2, w4 = wave & 3, tid = w4 * 64 + lane;
    const int blk = blk0 + half;
    const int row = blk * M2F_LN_ROWS_PER_BLOCK + w4;
    const bool half_on = half < nblk;                             // uniform per half
    const bool row_on = half_on && row < a.T;
    const int d = P.d;
    const int ld = P.ld ? P.ld : d;
    const int dpad = (d + 3) & ~3;
    float* red = lds + (size_t)half * 4 * 2 * dpad;               // [4 waves][2][dpad]
    const bool vec = is_vec(P.x, d) && is_vec(P.dy, d) && is_vec(P.dx, d) && is_vec(P.gamma, d) &&
                     (!P.extra || is_vec(P.extra, d)) && (!P.dx_masked || is_vec(P.dx_masked, d)) && ((ld & 3) == 0);
    RowRegs<NV> dg, db;
#pragma unroll
    for (int j = 0; j < NV; ++j) { dg.v[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; db.v[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    if (row_on) {
        uint16_t* dx16 = m2f_shadow_of(a.sh, P.dx);
        uint16_t* dxm16 = P.dx_masked ? m2f_shadow_of(a.sh, P.dx_masked) : nullptr;
        const rsrc_t rx = mk_rsrc(P.x), rdy = mk_rsrc(P.dy), rex = mk_rsrc(P.extra ? P.extra : P.dy);
        const rsrc_t rdx = mk_rsrc(P.dx), rdxm = mk_rsrc(P.dx_masked ? P.dx_masked : P.dx);
        const rsrc_t rdx16 = mk_rsrc(dx16 ? (const void*)dx16 : (const void*)P.dx);
        const rsrc_t rdxm16 = mk_rsrc(dxm16 ? (const void*)dxm16 : (const void*)P.dx);
        RowRegs<NV> g;
        prow_load(g, P.gamma, d, vec, lane);
        unsigned key = 0;
        if (P.drop_site2) key = m2f_site_key(a.rng, P.drop_site2);
        const float invd = 1.0f / (float)d;
        const unsigned ro = (unsigned)row * (unsigned)ld;
        RowRegs<NV> x, dy, ex;
        row_load(x, rx, ro, d, vec, lane);
        row_load(dy, rdy, ro, d, vec, lane);
        if (P.extra) row_load(ex, rex, ro, d, vec, lane);
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
                    dm = m2f_keep(key, (unsigned)row * (unsigned)d + (unsigned)c, a.drop_thresh) ? dx * a.drop_scale : 0.f;
                }
                msk.v[j][e] = dm;
                dy.v[j][e] = P.extra ? dx + ex.v[j][e] : dx;
            }
        const bool al8 = (ld & 3) == 0;
        row_store(dy, rdx, ro, d, vec, lane);
        if (dx16) row_store_bf16(dy, rdx16, ro, d, al8 && ((reinterpret_cast<uintptr_t>(dx16) & 7) == 0), lane);
        if (P.dx_masked) row_store(msk, rdxm, ro, d, vec, lane);
        if (dxm16) row_store_bf16(msk, rdxm16, ro, d, al8 && ((reinterpret_cast<uintptr_t>(dxm16) & 7) == 0), lane);
    }
    // per-block partial dgamma / dbeta: waves -> LDS -> fixed-order sum (a wave whose row lies past T contributes zeros,
    // exactly as in m2f_ln_bwd_kernel)
    if (half_on) {
        float* mine = red + (size_t)w4 * 2 * dpad;
        lds_row_store(dg, mine, dpad, lane);
        lds_row_store(db, mine + dpad, dpad, lane);
    }
    __syncthreads();
    if (half_on && blk * M2F_LN_ROWS_PER_BLOCK < a.T) {
        float* out = P.partial + (size_t)blk * 2 * d;                           // read by the reduce launch: plain stores
        for (int c = tid; c < 2 * d; c += 256) {
            const int which = c >= d, cc = which ? c - d : c;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += red[(size_t)w * 2 * dpad + which * dpad + cc];
            out[c] = s;
        }
    }
}

__device__ __forceinline__ void mega_dropout(const MegaArgs& a, const MegaDrop& D, int row0, int nrows) {
    const unsigned key = m2f_site_key(a.rng, D.site);
    uint16_t* x16 = m2f_shadow_of(a.sh, D.x);
    const rsrc_t rx = mk_rsrc(D.x), rx16 = mk_rsrc(x16 ? (const void*)x16 : (const void*)D.x);
    const int r1 = row0 + nrows < D.T ? row0 + nrows : D.T;
    const bool vec = ((D.d & 3) == 0) && ((D.ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(D.x) & 15) == 0) &&
                     (!x16 || ((reinterpret_cast<uintptr_t>(x16) & 7) == 0));
    if (vec) {
        // four float4 per lane in flight (all loads of a batch before the first store): a 16 x 768 item is 6 such batches
        const int c4n = D.d >> 2, n4 = (r1 - row0) * c4n;
        for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * M2F_MEGA_THREADS) {
            f32x4 v[4]; unsigned o[4]; unsigned flat[4]; bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * M2F_MEGA_THREADS;
                ok[u] = i < n4;
                const int r = row0 + (ok[u] ? i / c4n : 0), c = 4 * (ok[u] ? i % c4n : 0);
                o[u] = (unsigned)(r * D.ld + c); flat[u] = (unsigned)(r * D.d + c);
                v[u] = ld4(rx, o[u] * 4u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!ok[u]) continue;
                f32x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = m2f_keep(key, flat[u] + e, a.drop_thresh) ? v[u][e] * a.drop_scale : 0.f;
                st4(rx, o[u] * 4u, w);
                if (x16) st_u2(rx16, o[u] * 2u, (u32x2){pack_bf16(w[0], w[1]), pack_bf16(w[2], w[3])});
            }
        }
        return;
    }
    const int n = (r1 - row0) * D.d;
    for (int i = threadIdx.x; i < n; i += M2F_MEGA_THREADS) {
        const int r = row0 + i / D.d, c = i % D.d;
        const unsigned o = (unsigned)(r * D.ld + c);
        const float v = m2f_keep(key, (unsigned)(r * D.d + c), a.drop_thresh) ? ld1(rx, o * 4u) * a.drop_scale : 0.f;
        st1(rx, o * 4u, v);
        if (x16) st_h(rx16, o * 2u, m2f_bf16_bits(v));
    }
}

template <int NT>
__global__ __launch_bounds__(M2F_MEGA_THREADS) void m2f_mega_kernel(const MegaArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* misc = reinterpret_cast<unsigned*>(smem + LDS_MISC_OFF);
    if (threadIdx.x < 16) misc[threadIdx.x] = 0u;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    unsigned seq = 0;                                            // GEMM items this workgroup has started
    Pending pd; pd.s0 = 0; pd.nstrips = 0; pd.on = false;        // consumer waves: an item stored but not yet published
    constexpr int Lp = 16 * NT;
    const unsigned long long t_kernel = PROF_NOW();
    // this workgroup's queue = its XCD's (HW_REG_XCC_ID; which workgroups share an L2 is all that matters)
    const int xcc = __builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 7u));
    const int q0 = a.qoff[xcc];
    const unsigned qn = (unsigned)(a.qoff[xcc + 1] - q0);
    if (wave == 4 && lane == 0) lds_store_u32(misc + MISC_TICKET, mega_draw(a, xcc));
    __syncthreads();
    for (unsigned nth = 0;; ++nth) {
        const unsigned ticket = lds_load_u32(misc + MISC_TICKET + (nth & 1u));
        if (ticket >= qn) break;                                 // uniform: every wave reads the same word
        const int idx = q0 + (int)ticket;
        const MegaItem it = a.items[idx];
        if (it.kind == MK_GEMM) {
            const GemmProblem& P = a.gemm[it.prob];
            ++seq;
            const bool ok = wave >= 4 ? mega_gemm_producer(a, it, idx, P, smem, misc, wave, lane, seq, xcc, nth)
                                      : mega_gemm_consumer(a, it, P, smem, misc, wave, lane, seq, pd);
            if (!ok) return;
            continue;
        }
        if (wave < 4) mega_flush(a, pd, misc, lane);             // this item's poll may be waiting for it
        const unsigned long long t_top = PROF_NOW();
        if (!mega_wait_all(a, it, idx, misc, wave, lane, xcc, nth)) return;
        const unsigned long long t_w = PROF_NOW();
        const int half = wave >> 2, tid = threadIdx.x & 255;
        switch (it.kind) {
            case MK_ATTN_FWD: {
                const AttnProblem& P = a.attn[it.prob];
                const int W = a.attn_w, hf = 3 * Lp * (W + 2);
                const bool active = half < it.b && half < a.attn_halves_fwd;
                mega_attn_fwd<NT>(a, P, it.a + (active ? half : 0), active, reinterpret_cast<float*>(smem) + (size_t)half * hf, tid);
                break;
            }
            case MK_ATTN_BWD: {
                const AttnProblem& P = a.attn[it.prob];
                const int W = a.attn_w, hf = (a.attn_bwd_fast ? 5 : 4) * Lp * (W + 2) + Lp;
                const bool active = half < it.b && half < a.attn_halves_bwd;
                mega_attn_bwd<NT>(a, P, it.a + (active ? half : 0), active, reinterpret_cast<float*>(smem) + (size_t)(active ? half : 0) * hf, tid);
                break;
            }
            case MK_LN_FWD: {
                const LnProblem& P = a.ln[it.prob];
                const int nv = (P.d + 255) / 256;
                if (nv <= 1) mega_ln_fwd<1>(a, P, it.a, it.b, wave, lane);
                else if (nv <= 2) mega_ln_fwd<2>(a, P, it.a, it.b, wave, lane);
                else if (nv <= 3) mega_ln_fwd<3>(a, P, it.a, it.b, wave, lane);
                else mega_ln_fwd<4>(a, P, it.a, it.b, wave, lane);          // d <= 1024 (checked when the plan is built)
                break;
            }
            case MK_LN_BWD: {
                const LnProblem& P = a.ln[it.prob];
                const int nv = (P.d + 255) / 256;
                float* lds = reinterpret_cast<float*>(smem);
                if (nv <= 1) mega_ln_bwd<1>(a, P, it.a, it.b, lds, wave, lane);
                else if (nv <= 2) mega_ln_bwd<2>(a, P, it.a, it.b, lds, wave, lane);
                else if (nv <= 3) mega_ln_bwd<3>(a, P, it.a, it.b, lds, wave, lane);
                else mega_ln_bwd<4>(a, P, it.a, it.b, lds, wave, lane);
                break;
            }
            case MK_DROPOUT: mega_dropout(a, a.drop[it.prob], it.a, it.b); break;
            default: break;
        }
        mega_arrive(a, it, misc + MISC_ARRIVE8, 7u, lane);
        // the next item's LDS writes must not overtake this item's LDS reads by slower waves
        __syncthreads();
#ifdef M2F_MEGA_PROF
        if (wave == 0 && lane == 0) {
            const unsigned long long t_e = PROF_NOW();
            prof_add(a, it.kind, 0, 1); prof_add(a, it.kind, 1, t_e - t_top); prof_add(a, it.kind, 2, t_w - t_top); prof_add(a, it.kind, 3, t_e - t_w);
        }
#endif
    }
    if (wave < 4) mega_flush(a, pd, misc, lane);
#ifdef M2F_MEGA_PROF
    if (wave == 0 && lane == 0) { prof_add(a, 0, 0, 1); prof_add(a, 0, 1, PROF_NOW() - t_kernel); }
    if (wave == 0 && lane == 0 && a.prof) atomicMax(a.prof + 2, PROF_NOW() - t_kernel);
#endif
}

}  // namespace

// Re-arms the ticket and strip counters of a run.  A kernel, not a hipMemsetAsync: inside a captured graph the memset node
// was observed not to take effect on replay when a second engine's graph ran in between (the persistent kernel then found
// its queues exhausted and returned at once, leaving the previous step's outputs) - a kernel node has no such problem.
__global__ void m2f_mega_rearm_kernel(unsigned* p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}

hipError_t m2f_launch_mega(const MegaArgs& a, int nt, int grid, hipStream_t stream) {
    // NT = 3, 4 (dialogues of more than 32 utterances) are not instantiated: the five-slab attention backward spills
    // VGPRs at 256, and a spill is not allowed next to the inline-asm staging loads (check_spills.py); such plans keep
    // the launch list.
    if (!a.items || !a.queue || a.qoff[8] <= 0 || grid < 1 || nt < 1 || nt > M2F_MEGA_MAX_NT) return hipErrorInvalidValue;
    void (*kern)(const MegaArgs) = nt == 1 ? m2f_mega_kernel<1> : m2f_mega_kernel<2>;
    static bool attr_set[M2F_MEGA_MAX_NT + 1] = {false, false, false};
    if (!attr_set[nt]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, M2F_MEGA_LDS);
        if (e != hipSuccess) return e;
        attr_set[nt] = true;
    }
    hipLaunchKernelGGL(m2f_mega_rearm_kernel, dim3(1), dim3(256), 0, stream, a.queue, (8 + a.n_strips) * 32);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(M2F_MEGA_THREADS), M2F_MEGA_LDS, stream, a);
    return hipGetLastError();
}
