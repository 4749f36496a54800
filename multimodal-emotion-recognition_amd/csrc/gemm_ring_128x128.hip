// ring form, 128x128 tiles, 4 ring slots (grouped launches)
#include "gemm_ring.h"
hipError_t m2f_ring_launch_128x128(GemmBatch& gb, hipStream_t stream) { return launch_ring16<128, 128, 4>(gb, stream); }
#ifdef M2F_EXP_TIMING
extern "C" int m2f_ring_dbg_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(m2f_ring_dbg), sizeof(unsigned long long) * 64);
}
#endif
