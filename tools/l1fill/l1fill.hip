// L1-fill microbenchmark (gfx950): how many bytes per second can ONE CU pull from L2 / Infinity Cache when every CU
// pulls at once, (a) with ordinary 16-byte vector loads into registers, (b) with 16-byte LDS-direct loads
// (global_load_lds_dwordx4), which need no destination registers and no ds_write.  The GEMM kernels of this repo sit on
// this rate; the number decides the tile shape of the next GEMM kernel.
// build: hipcc -O3 --offload-arch=gfx950 l1fill.hip -o l1fill ; run: ./l1fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every wave walks its own 1 KB-per-instruction stream: wave w of workgroup b starts at a different offset of the buffer,
// UNROLL independent instructions in flight, ITERS rounds
template <int UNROLL>
__global__ __launch_bounds__(512) void vgpr_kernel(const uint4* __restrict__ buf, size_t n16, int iters, uint4* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t pos = ((size_t)blockIdx.x * 8 + wave) * 64 * UNROLL * 37 % n16;
    uint4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            size_t i = pos + (size_t)u * 64 + lane;
            if (i >= n16) i -= n16;
            v[u] = buf[i];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        pos += (size_t)64 * UNROLL * 4099;                          // a far jump: no reuse inside L1
        pos %= n16;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[threadIdx.x] = acc;     // never true: keeps the loads alive
}

template <int UNROLL>
__global__ __launch_bounds__(512) void lds_kernel(const uint4* __restrict__ buf, size_t n16, int iters, uint4* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t pos = ((size_t)blockIdx.x * 8 + wave) * 64 * UNROLL * 37 % n16;
    char* mine = smem + (size_t)wave * UNROLL * 1024;               // 1 KB per instruction per wave
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            size_t i = pos + (size_t)u * 64 + lane;
            if (i >= n16) i -= n16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(buf + i),
                                             (__attribute__((address_space(3))) void*)(mine + u * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pos += (size_t)64 * UNROLL * 4099;
        pos %= n16;
    }
    __syncthreads();
    const uint4 v = reinterpret_cast<const uint4*>(smem)[threadIdx.x];
    if (v.x == 0x12345678u && v.y == 0x9abcdef0u) sink[threadIdx.x] = v;
}

template <typename K>
static double run(K kern, int grid, int lds, const uint4* buf, size_t n16, int iters, uint4* sink) {
    hipEvent_t a, b;
    if (lds > 64 * 1024) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, buf, n16, iters, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a, 0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, buf, n16, iters, sink);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1e-3;
}

int main() {
    const size_t footprints[] = {2u << 20, 16u << 20, 128u << 20, 1024u << 20};
    uint4* sink;
    CHECK(hipMalloc(&sink, 512 * sizeof(uint4)));
    printf("%-10s %-6s %-5s %10s %12s %12s\n", "footprint", "path", "CUs", "unroll", "TB/s total", "GB/s per CU");
    for (size_t fp : footprints) {
        uint4* buf;
        CHECK(hipMalloc(&buf, fp));
        CHECK(hipMemset(buf, 1, fp));
        const size_t n16 = fp / 16;
        for (int grid : {64, 256}) {
            const int iters = 400;
#define ROW(NAME, KERN, U, LDS)                                                                             \
            {                                                                                               \
                const double s = run(KERN<U>, grid, LDS, buf, n16, iters, sink);                            \
                const double bytes = (double)grid * 8 * iters * U * 1024.0;                                 \
                printf("%-10zu %-6s %-5d %10d %12.2f %12.1f\n", fp >> 20, NAME, grid, U, bytes / s / 1e12, bytes / s / grid / 1e9); \
            }
            ROW("vgpr", vgpr_kernel, 4, 0)
            ROW("vgpr", vgpr_kernel, 8, 0)
            ROW("vgpr", vgpr_kernel, 16, 0)
            ROW("lds", lds_kernel, 4, 8 * 4 * 1024)
            ROW("lds", lds_kernel, 8, 8 * 8 * 1024)
            ROW("lds", lds_kernel, 16, 8 * 16 * 1024)
        }
        CHECK(hipFree(buf));
    }
    return 0;
}
