// What do the cross-lane moves of gemm_p8.h's epilogue deliver?  One wave; x0 = 1000 + lane, x1 = 2000 + lane.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    const int x0 = 1000 + lane, x1 = 2000 + lane;
    out[lane] = __builtin_amdgcn_update_dpp(x0, x1, 0x128, 0xF, 0xC, false);
    out[64 + lane] = __builtin_amdgcn_update_dpp(x1, x0, 0x128, 0xF, 0x3, false);
    unsigned a = (unsigned)x0, b = (unsigned)x1;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[128 + lane] = (int)r[0]; out[192 + lane] = (int)r[1];
    auto q = __builtin_amdgcn_permlane16_swap(r[0], r[1], false, false);
    out[256 + lane] = (int)q[0]; out[320 + lane] = (int)q[1];
}
int main() {
    int* d; hipMalloc(&d, 384 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[6] = {"dpp s0 (old x0, src x1, ror8, banks 2,3)", "dpp s1 (old x1, src x0, ror8, banks 0,1)", "permlane32_swap [0]", "permlane32_swap [1]", "+ permlane16_swap [0]", "+ permlane16_swap [1]"};
    for (int t = 0; t < 6; ++t) {
        printf("%s\n", names[t]);
        for (int l = 0; l < 64; ++l) printf("%5d%s", h[t * 64 + l], (l & 15) == 15 ? "\n" : "");
    }
    return 0;
}
