// ring form, 128x64 tiles, 4 ring slots (grouped launches with fewer than ~one 128x128 tile per CU)
#include "gemm_ring.h"
hipError_t m2f_ring_launch_128x64(GemmBatch& gb, hipStream_t stream) { return launch_ring16<128, 64, 4>(gb, stream); }
