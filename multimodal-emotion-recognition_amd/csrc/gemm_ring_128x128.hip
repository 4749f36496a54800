// ring form, 128x128 tiles, 4 ring slots (grouped launches)
#include "gemm_ring.h"
#ifndef M2F_TT_BM        // (timing builds may put another tile configuration behind this entry point: make timing TT="-DM2F_TT_BM=64 ...")
#define M2F_TT_BM 128
#define M2F_TT_BN 128
#define M2F_TT_S 4
#endif
hipError_t m2f_ring_launch_128x128(GemmBatch& gb, hipStream_t stream) { return launch_ring16<M2F_TT_BM, M2F_TT_BN, M2F_TT_S>(gb, stream); }
#ifdef M2F_EXP_TIMING
extern "C" int m2f_ring_dbg_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(m2f_ring_dbg), sizeof(unsigned long long) * 64);
}
#endif
