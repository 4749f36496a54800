// ring form, 256x128 tiles, 3 ring slots: launches of 1,024 and more such tiles (the in-loop text encoder at M = 32,768);
// epilogue variants {-, residual} x {-, GELU} only
#include "gemm_ring.h"
hipError_t m2f_ring_launch_256x128_plain(GemmBatch& gb, hipStream_t stream);      // gemm_ring_256x128_plain.hip
hipError_t m2f_ring_launch_256x128(GemmBatch& gb, hipStream_t stream) {
    // bias-only launches (M2FNet's merged QKV in-projections) have a kernel of their own, which also knows GF_NO_F32
    bool plain = true;
    for (int i = 0; i < gb.count; ++i) plain = plain && !gb.pr[i].res && !(gb.pr[i].flags & GF_GELU_OUT);
    return plain ? m2f_ring_launch_256x128_plain(gb, stream) : launch_ring16<256, 128, 3, 2>(gb, stream);
}
