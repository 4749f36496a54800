// Launch-floor microbenchmark (MI355X): what does ONE dependent kernel boundary cost on this runtime, eager and inside a
// captured hipGraph, as a function of the kernarg size and of where the launch descriptors live?
//   hipcc --offload-arch=gfx950 -O3 -o launchfloor launchfloor.hip && ./launchfloor
// Run twice: plain, and with HIP_FORCE_DEV_KERNARG=1 in the environment.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Small { float* p; int n; };
struct Big { float* p; int n; int pad[760]; int last; };          // ~3 KB, like GemmBatch
struct Desc { float* p; int n; int pad[61]; int last; };          // 256-byte descriptor in device memory

__global__ __launch_bounds__(256) void k_small(Small a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[0] += 1.f; }
__global__ __launch_bounds__(256) void k_big(Big a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[a.last] += 1.f; }
__global__ __launch_bounds__(256) void k_dev(const Desc* d, int i) { if (threadIdx.x == 0 && blockIdx.x == 0) d[i].p[d[i].last] += 1.f; }
// every workgroup reads the big kernarg's tail (what the GEMM's tile -> problem search does)
__global__ __launch_bounds__(256) void k_big_all(Big a) { if (threadIdx.x == 0) atomicAdd(a.p + 1 + a.last, 1.f); }

template <typename F>
static void run(const char* name, int n, int grid, hipStream_t s, F launch) {
    // eager
    for (int i = 0; i < 20; ++i) launch(i % n);
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) launch(i);
    CK(hipStreamSynchronize(s));
    const double eager = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    // graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const int reps = 10;
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (n * reps);
    printf("%-34s grid %4d: eager %6.2f us/kernel   graph %6.2f us/kernel\n", name, grid, eager, graph);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
    const char* env = getenv("HIP_FORCE_DEV_KERNARG");
    printf("HIP_FORCE_DEV_KERNARG=%s\n", env ? env : "(unset)");
    hipStream_t s; CK(hipStreamCreate(&s));
    float* buf; CK(hipMalloc(&buf, 1 << 20)); CK(hipMemset(buf, 0, 1 << 20));
    const int n = 200;
    std::vector<Desc> hd(n);
    for (int i = 0; i < n; ++i) { memset(&hd[i], 0, sizeof(Desc)); hd[i].p = buf; hd[i].last = 0; }
    Desc* dd; CK(hipMalloc(&dd, n * sizeof(Desc))); CK(hipMemcpy(dd, hd.data(), n * sizeof(Desc), hipMemcpyHostToDevice));
    for (int grid : {1, 256, 1024}) {
        Small sm{buf, 0};
        run("small kernarg (16 B)", n, grid, s, [&](int) { hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, s, sm); });
        Big bg; memset(&bg, 0, sizeof(bg)); bg.p = buf; bg.last = 0;
        run("big kernarg (3 KB), 1 reader", n, grid, s, [&](int) { hipLaunchKernelGGL(k_big, dim3(grid), dim3(256), 0, s, bg); });
        run("big kernarg (3 KB), all WGs read", n, grid, s, [&](int) { hipLaunchKernelGGL(k_big_all, dim3(grid), dim3(256), 0, s, bg); });
        run("device-resident descriptor", n, grid, s, [&](int i) { hipLaunchKernelGGL(k_dev, dim3(grid), dim3(256), 0, s, (const Desc*)dd, i); });
    }
    return 0;
}
