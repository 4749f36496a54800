// ring form, 256x128 tiles, 3 ring slots, epilogue = bias only, with or without the fp32 store (GF_NO_F32): M2FNet's merged QKV
// in-projections.  (A translation unit of its own: two kernels sharing one ring_producer instantiation do not compile for the host.)
#include "gemm_ring.h"
hipError_t m2f_ring_launch_256x128_plain(GemmBatch& gb, hipStream_t stream) { return launch_ring16<256, 128, 3, 3>(gb, stream); }
