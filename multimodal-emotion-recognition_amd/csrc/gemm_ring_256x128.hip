// ring form, 256x128 tiles, 3 ring slots: launches of 1,024 and more such tiles (the in-loop text encoder at M = 32,768);
// epilogue variants {-, residual} x {-, GELU} only
#include "gemm_ring.h"
hipError_t m2f_ring_launch_256x128(GemmBatch& gb, hipStream_t stream) { return launch_ring16<256, 128, 3, 2>(gb, stream); }
