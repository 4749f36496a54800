// ring form of the weight-gradient table launch, 128x128 tiles (M2F_TABLE_TILE=129)
#include "gemm_ring.h"
hipError_t m2f_ring_launch_table_128x128(const GemmBatch& gb, hipStream_t stream) {
    return launch_ring_grid<128, 128, 4, true>(gb, gb.total_tiles, stream);
}
// the same launch with ROW-MAJOR operands ([token][feature] activation shadows, M2F_TABLE_TILE=130): no token-transposed
// copies; the kernel also sums the bias gradients (problem.bias_grad)
hipError_t m2f_ring_launch_table_rc_128x128(const GemmBatch& gb, hipStream_t stream) {
    return launch_ring_grid<128, 128, 4, true, true>(gb, gb.total_tiles, stream);
}
// row-major operands, 256 (M) x 128 (N) tiles, 3 ring slots (M2F_TABLE_TILE=131): a quarter fewer operand bytes and LDS-DMA
// instructions per FLOP than 128x128 - the launch is bound by how fast a CU's texture addresser takes those instructions
hipError_t m2f_ring_launch_table_rc_256x128(const GemmBatch& gb, hipStream_t stream) {
    return launch_ring_grid<256, 128, 3, true, true>(gb, gb.total_tiles, stream);
}
#ifdef M2F_EXP_TIMING
// diagnostic build only (make ttiming): phase totals of workgroup 0 of the table launches, see tools/table_timing.py
extern "C" int m2f_ring_table_dbg_read(unsigned long long* out, int reset) {
    int r = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(m2f_ring_dbg), sizeof(unsigned long long) * 64);
    if (reset) {
        unsigned long long z[64] = {0};
        r |= (int)hipMemcpyToSymbol(HIP_SYMBOL(m2f_ring_dbg), z, sizeof(z));
    }
    return r;
}
#endif
