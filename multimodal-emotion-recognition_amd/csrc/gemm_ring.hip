// RING form of the k-contiguous bf16 GEMM (gemm_ring.h): host-side dispatch over the tile configurations, each of which is
// its own translation unit (gemm_ring_128x128.hip, gemm_ring_128x64.hip, gemm_ring_64x64.hip, gemm_ring_256x128.hip, gemm_ring_table.hip).
#include "common.h"
#include "ops.h"

long long m2f_g_ring_launches = 0;

extern "C" long long m2f_gemm_ring_launches(void) { return m2f_g_ring_launches; }

hipError_t m2f_ring_launch_128x128(GemmBatch& gb, hipStream_t stream);
hipError_t m2f_ring_launch_128x64(GemmBatch& gb, hipStream_t stream);
hipError_t m2f_ring_launch_64x64(GemmBatch& gb, hipStream_t stream);
hipError_t m2f_ring_launch_256x128(GemmBatch& gb, hipStream_t stream);
hipError_t m2f_ring_launch_table_128x128(const GemmBatch& gb, hipStream_t stream);
hipError_t m2f_ring_launch_table_rc_128x128(const GemmBatch& gb, hipStream_t stream);
hipError_t m2f_ring_launch_table_rc_256x128(const GemmBatch& gb, hipStream_t stream);
hipError_t m2f_p8_launch_table_rc(const GemmBatch& gb, hipStream_t stream);          // gemm_p8.hip: 256 x 256 tiles, eight-phase schedule

// can this forward-form launch run as the ring form?  (no GELU / FP8 / B-side ReLU epilogue variants there; 32-bit byte
// offsets into every operand)
bool m2f_gemm_ring_ok(const GemmBatch& gb) {
    for (int i = 0; i < gb.count; ++i) {
        const GemmProblem& p = gb.pr[i];
        if ((p.flags & (GF_GELU_OUT | GF_RELU_B)) || p.c8 || p.bias_grad) return false;
        for (int sgm = 0; sgm < 2; ++sgm) {
            if ((size_t)p.M * p.a.ldq[sgm] * 2 >= 0x80000000ull || (size_t)p.N * p.b.ldq[sgm] * 2 >= 0x80000000ull) return false;
        }
    }
    return true;
}

// ... and as the 256x128 ring form, whose epilogue knows bias, ReLU, GELU and a residual only?
bool m2f_gemm_ring256_ok(const GemmBatch& gb) {
    for (int i = 0; i < gb.count; ++i) {
        const GemmProblem& p = gb.pr[i];
        if ((p.flags & (GF_RELU_B | GF_ACCUM)) || p.c8 || p.bias_grad || p.gate || p.drop_site) return false;
        for (int sgm = 0; sgm < 2; ++sgm) {
            if ((size_t)p.M * p.a.ldq[sgm] * 2 >= 0x80000000ull || (size_t)p.N * p.b.ldq[sgm] * 2 >= 0x80000000ull) return false;
        }
    }
    return true;
}

hipError_t m2f_launch_gemm_ring(GemmBatch& gb, int bm, int bn, hipStream_t stream) {
    if (bm == 128 && bn == 128) return m2f_ring_launch_128x128(gb, stream);
    if (bm == 128 && bn == 64) return m2f_ring_launch_128x64(gb, stream);
    if (bm == 64 && bn == 64) return m2f_ring_launch_64x64(gb, stream);
    if (bm == 256 && bn == 128) return m2f_ring_launch_256x128(gb, stream);          // needs m2f_gemm_ring256_ok
    return hipErrorInvalidValue;
}

hipError_t m2f_launch_gemm_ring_table(const GemmBatch& gb, hipStream_t stream) {
    if (gb.table_tile == 132) return m2f_p8_launch_table_rc(gb, stream);
    if (gb.table_tile == 131) return m2f_ring_launch_table_rc_256x128(gb, stream);
    return gb.table_tile == 130 ? m2f_ring_launch_table_rc_128x128(gb, stream) : m2f_ring_launch_table_128x128(gb, stream);
}
