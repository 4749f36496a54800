// ring form, 256x128 tiles, 3 ring slots, OCP e4m3 operands on v_mfma_scale_f32_32x32x64_f8f6f4 (the in-loop text encoder's fp8 mode):
// same producers and LDS image as the bf16 form - a 128-byte row holds 128 k-values instead of 64 - de-quantising epilogue
// {-, residual} x {-, GELU}, fp32 or e4m3 result.  (A translation unit of its own, like every ring kernel.)
#include "gemm_ring.h"
hipError_t m2f_ring_launch_256x128_fp8(GemmBatch& gb, hipStream_t stream) { return launch_ring16<256, 128, 3, 4>(gb, stream); }
