// Shared device helpers for the M2FNet gfx950 kernels (wave64, MFMA, LDS).
// gfx950 only: no CUDA shims, no alternate back-ends.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Kernel-argument warm-up.  The kernels take their launch descriptors (GemmBatch, AttnBatch, LnBatch: 0.5-3 KB) BY VALUE; the
// kernarg segment of a launch is cold in the scalar cache and in L2, and hipcc issues its s_loads lazily - a grouped GEMM
// prologue walked four DEPENDENT misses (hidden grid size -> total_tiles -> tb[] -> hot[pi]) before its first operand load.
// One wave-wide batch of s_load_dword, one per 64-byte line the kernel is going to read, turns them into one miss + hits.
// (The wait sits INSIDE the asm: the compiler does not track loads issued by inline asm, and the scratch register is dead
// the moment the block ends.)
#define M2F_KW1(op, i) "s_load_dword %0, %1, " op "+(" #i ")*64\n"
#define M2F_KW8(op, b) M2F_KW1(op, b+0) M2F_KW1(op, b+1) M2F_KW1(op, b+2) M2F_KW1(op, b+3) M2F_KW1(op, b+4) M2F_KW1(op, b+5) M2F_KW1(op, b+6) M2F_KW1(op, b+7)
// LINES (8, 16 or 24) consecutive 64-byte lines from byte offset B on, plus eight more from B2 on (B2 < 0: none; the hidden
// arguments behind a large struct); every load writes the same scratch register.
template <int B, int LINES = 8, int B2 = -1>
__device__ __forceinline__ void m2f_kernarg_warm() {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(M2F_NO_KERNARG_WARM)       // (the switch exists for A/B builds)
    static_assert(B % 4 == 0 && (B2 < 0 || B2 % 4 == 0) && (LINES == 8 || LINES == 16 || LINES == 24), "dword offsets, whole groups of eight lines");
    const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    constexpr int C = B2 < 0 ? B : B2;          // (no second range: its loads repeat the first lines - hits)
    uint32_t t;
    if constexpr (LINES == 8) asm volatile(M2F_KW8("%2", 0) M2F_KW8("%3", 0) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(kp), "n"(B), "n"(C) : "memory");
    else if constexpr (LINES == 16) asm volatile(M2F_KW8("%2", 0) M2F_KW8("%2", 8) M2F_KW8("%3", 0) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(kp), "n"(B), "n"(C) : "memory");
    else asm volatile(M2F_KW8("%2", 0) M2F_KW8("%2", 8) M2F_KW8("%2", 16) M2F_KW8("%3", 0) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(kp), "n"(B), "n"(C) : "memory");
#endif
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define M2F_WAVE 64

// ---- counter-based dropout RNG -----------------------------------------------------------------
// The state lives in device memory (4 x u32: seed_lo, seed_hi, step_lo, step_hi) so a captured
// hipGraph replays with a fresh stream of masks each step (a 1-thread kernel bumps `step`).
// keep(site, idx) is a pure function of (state, site, idx): the backward pass regenerates the
// forward mask instead of storing it.
__device__ __forceinline__ uint32_t m2f_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t m2f_site_key(const uint32_t* __restrict__ rng, uint32_t site) {
    uint32_t k = m2f_mix32(rng[0] ^ 0x9e3779b9U);
    k = m2f_mix32(k ^ rng[1]);
    k = m2f_mix32(k + rng[2] * 0x85ebca6bU);
    k = m2f_mix32(k ^ (rng[3] + site * 0xc2b2ae35U));
    return k;
}
__device__ __forceinline__ bool m2f_keep(uint32_t key, uint32_t idx, uint32_t thresh) {
    uint32_t h = m2f_mix32(idx * 0x9E3779B1U + key);
    h = m2f_mix32(h ^ (key >> 7) ^ 0x68e31da4U);
    return h >= thresh;       // P(keep) = 1 - thresh / 2^32
}

__device__ __forceinline__ float m2f_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float m2f_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int m2f_cdiv(int a, int b) { return (a + b - 1) / b; }

// fp32 -> bf16 bits, round to nearest even (same rounding the GEMM staging applies; NaN stays NaN)
__device__ __forceinline__ uint16_t m2f_bf16_bits(float x) {
    const __bf16 h = (__bf16)x;
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float m2f_bf16_to_f32(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }

// fp32 -> OCP e4m3 byte, saturating at +-448 (the value range the fp8 GEMM operands use)
__device__ __forceinline__ uint32_t m2f_fp8x4_bits(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);     // bytes 0, 1
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);      // bytes 2, 3
    return (uint32_t)w;
}

#ifdef __HIPCC__
// erf for the GELU epilogues.  POLY (bf16 / fp8 kernels): odd degree-13 minimax polynomial on |z| <= 3, max error 4.3e-4
// (far inside bf16 operand rounding), 7 FMAs, no quarter-rate instructions; otherwise Abramowitz-Stegun 7.1.26 (1.5e-7) on
// the hardware exp / rcp (libm's erff is several times slower still).
template <bool POLY>
__device__ __forceinline__ float m2f_gelu(float x) {
    const float z = x * 0.70710678118654752f;
    float e;
    if constexpr (POLY) {
        const float zc = fminf(fmaxf(z, -3.0f), 3.0f), t = zc * zc;
        float p = 3.4737140595098026e-06f;
        p = p * t - 0.0001298444258281961f;
        p = p * t + 0.0020486447028815746f;
        p = p * t - 0.018010087311267853f;
        p = p * t + 0.098881796002388f;
        p = p * t - 0.3658691942691803f;
        p = p * t + 1.1261212825775146f;
        e = fminf(fmaxf(p * zc, -1.0f), 1.0f);
    } else {
        const float az = fabsf(z);
        const float tt = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);
        const float poly = tt * (0.254829592f + tt * (-0.284496736f + tt * (1.421413741f + tt * (-1.453152027f + tt * 1.061405429f))));
        e = copysignf(1.0f - poly * __expf(-az * az), z);
    }
    return 0.5f * x * (1.0f + e);
}
#endif
