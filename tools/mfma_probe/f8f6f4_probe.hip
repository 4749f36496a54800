// Which lane holds which k of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands, and what do zero scale operands mean?
// hipcc -O2 --offload-arch=gfx950 f8f6f4_probe.hip -o f8f6f4_probe && ./f8f6f4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const uint8_t* A, const uint8_t* B, float* C, int scale) {
    // hypothesis: lane l holds row (col) l & 31, k = 32 * (l >> 5) + j, j = 0..31 in byte order
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v8i a, b;
    for (int w = 0; w < 8; ++w) {
        uint32_t wa = 0, wb = 0;
        for (int j = 0; j < 4; ++j) {
            wa |= (uint32_t)A[r * 64 + 32 * h + 4 * w + j] << (8 * j);        // A [32][64] row-major
            wb |= (uint32_t)B[(32 * h + 4 * w + j) * 32 + r] << (8 * j);      // B [64][32] row-major
        }
        a[w] = (int)wa; b[w] = (int)wb;
    }
    f32x16 c = {0};
    if (scale == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    // C/D layout of the 32x32 shapes: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    for (int g = 0; g < 16; ++g) C[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];
}
static float e4m3(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}
int main() {
    const uint8_t vals[7] = {0x00, 0x30, 0x38, 0x40, 0xB0, 0xB8, 0xC0};     // 0, .5, 1, 2, -.5, -1, -2
    uint8_t hA[32 * 64], hB[64 * 32];
    srand(1);
    for (auto& x : hA) x = vals[rand() % 7];
    for (auto& x : hB) x = vals[rand() % 7];
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, 32 * 32 * 4);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    for (int scale = 0; scale < 2; ++scale) {
        probe<<<1, 64>>>(dA, dB, dC, scale);
        float hC[32 * 32];
        hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
        int bad = 0; double ratio = 0; int nr = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                float ref = 0;
                for (int k = 0; k < 64; ++k) ref += e4m3(hA[i * 64 + k]) * e4m3(hB[k * 32 + j]);
                if (hC[i * 32 + j] != ref) ++bad;
                if (ref != 0) { ratio += hC[i * 32 + j] / ref; ++nr; }
            }
        printf("scale operands %s: %d of 1024 elements differ from the reference, mean C/ref %.6g, C[0][0..3] = %g %g %g %g\n",
               scale ? "0x7F7F7F7F" : "0", bad, ratio / (nr ? nr : 1), hC[0], hC[1], hC[2], hC[3]);
    }
    return 0;
}
