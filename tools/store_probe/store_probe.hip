// How fast does a CU get a 256 x 256 result tile out, by store shape?  (round 4; tools/store_probe: hipcc --offload-arch=gfx950 -O3)
// 256 workgroups x 512 threads; every workgroup writes `reps` tiles of 256 x 256 elements (fp32 or bf16) of a [M, N] matrix, each wave 64 rows
// x 32 columns per (quadrant) as the eight-phase GEMM's epilogue does, in one of these shapes per store instruction:
//   0: fp32, 16 rows x 64 bytes   (lane = row lr, columns 4 g .. 4 g + 3 of a 16-column block: what gemm_p8.h does)
//   1: fp32,  8 rows x 128 bytes  (lane = row lane >> 3, 16-byte chunk lane & 7 of the wave's 32 columns)
//   2: bf16, 16 rows x 32 bytes   (8 bytes per lane: gemm_p8.h's bf16 stores)
//   3: bf16, 16 rows x 64 bytes   (16 bytes per lane: lane = row lane >> 2 ... (lane & 3) chunks of the wave's 32 columns)
//   4: fp32,  4 rows x 256 bytes  (a wave owning 64 columns: lane = row lane >> 4, chunk lane & 15)
// prints cycles per tile (s_memtime of wave 0 of workgroup 0, stores issued .. vmcnt(0)) and the launch's GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(char* __restrict__ out, int ldc_elems, int tiles_n, int reps, unsigned long long* dbg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
    const int lr = lane & 15, g = lane >> 4;
    unsigned long long t_all = 0;
    for (int r = 0; r < reps; ++r) {
        const int tile = ((int)blockIdx.x + r * (int)gridDim.x) % (tiles_n * 128);      // (M = 32,768 rows = 128 tile rows: stay inside the buffer)
        const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const f32x4 v = {(float)lane, (float)r, 1.f, 2.f};
        const u32x2 h = {(unsigned)lane, (unsigned)r};
        const u32x4 h4 = {(unsigned)lane, (unsigned)r, 3u, 4u};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int rowq = m0 + a * 128 + wr * 64, colq = n0 + b * 128 + wc * 32;
                if constexpr (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            *reinterpret_cast<f32x4*>(out + ((size_t)(rowq + i * 16 + lr) * ldc_elems + colq + j * 16 + 4 * g) * 4) = v;
                } else if constexpr (MODE == 1) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        *reinterpret_cast<f32x4*>(out + ((size_t)(rowq + i * 8 + (lane >> 3)) * ldc_elems + colq + 4 * (lane & 7)) * 4) = v;
                } else if constexpr (MODE == 2) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            *reinterpret_cast<u32x2*>(out + ((size_t)(rowq + i * 16 + lr) * ldc_elems + colq + j * 16 + 4 * g) * 2) = h;
                } else if constexpr (MODE == 3) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *reinterpret_cast<u32x4*>(out + ((size_t)(rowq + i * 16 + (lane >> 2)) * ldc_elems + colq + 8 * (lane & 3)) * 2) = h4;
                } else {
                    // (wave = 32 rows x 64 columns of the quadrant instead: same bytes per wave)
                    const int rowq2 = m0 + a * 128 + (wave >> 1) * 32, colq2 = n0 + b * 128 + (wave & 1) * 64;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        *reinterpret_cast<f32x4*>(out + ((size_t)(rowq2 + i * 4 + (lane >> 4)) * ldc_elems + colq2 + 4 * (lane & 15)) * 4) = v;
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        t_all += __builtin_amdgcn_s_memtime() - t0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) dbg[0] = t_all;
}

int main(int argc, char** argv) {
    const int M = 32768, N = argc > 1 ? atoi(argv[1]) : 4096, reps = 8;
    char* out; unsigned long long* dbg;
    hipMalloc(&out, (size_t)M * N * 4); hipMalloc(&dbg, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles_n = N / 256;
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9f; unsigned long long cyc = 0;
        for (int it = 0; it < 5; ++it) {
            hipEventRecord(e0);
            switch (mode) {
                case 0: hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, out, N, tiles_n, reps, dbg); break;
                case 1: hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, out, N, tiles_n, reps, dbg); break;
                case 2: hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 0, 0, out, N, tiles_n, reps, dbg); break;
                case 3: hipLaunchKernelGGL(probe<3>, dim3(256), dim3(512), 0, 0, out, N, tiles_n, reps, dbg); break;
                default: hipLaunchKernelGGL(probe<4>, dim3(256), dim3(512), 0, 0, out, N, tiles_n, reps, dbg); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) { best = ms; hipMemcpy(&cyc, dbg, 8, hipMemcpyDeviceToHost); }
        }
        const double bytes = 256.0 * reps * 256 * 256 * ((mode == 2 || mode == 3) ? 2 : 4);
        printf("N=%d mode %d: %.1f us per launch, %.0f GB/s, %.0f cycles per tile (store issue .. vmcnt(0) + barrier, workgroup 0)\n", N, mode, best * 1e3,
               bytes / (best * 1e-3) / 1e9, (double)cyc / reps);
    }
    return 0;
}
