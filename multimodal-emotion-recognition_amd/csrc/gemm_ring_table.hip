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
