// ring form of the weight-gradient table launch, 128x128 tiles (M2F_TABLE_TILE=129)
#include "gemm_ring.h"
hipError_t m2f_ring_launch_table_128x128(const GemmBatch& gb, hipStream_t stream) {
    return launch_ring_grid<128, 128, 4, true>(gb, gb.total_tiles, stream);
}
