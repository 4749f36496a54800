// ring form, 64x64 tiles, 5 ring slots (grouped launches too small for the 128-row tiles)
#include "gemm_ring.h"
hipError_t m2f_ring_launch_64x64(GemmBatch& gb, hipStream_t stream) { return launch_ring16<64, 64, 5>(gb, stream); }
