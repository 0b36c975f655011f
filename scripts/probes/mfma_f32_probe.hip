// Accumulator layout of v_mfma_f32_16x16x4_f32 on gfx950: A[i][0] = i + 1, B[0][j] = 100 (j + 1) -> D[i][j] = 100 (i+1)(j+1);
// every lane decodes (i, j) of its four result registers.  Expected: register r of lane l is row 4 (l >> 4) + r, column l & 15.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* o) {
    const int l = threadIdx.x;
    const float a = (l >> 4) == 0 ? (float)((l & 15) + 1) : 0.f;
    const float b = (l >> 4) == 0 ? 100.f * ((l & 15) + 1) : 0.f;
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) o[l * 4 + r] = c[r];
}
int main() {
    float* d; float h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * (l >> 4) + r, j = l & 15;
            if (h[l * 4 + r] != 100.f * (i + 1) * (j + 1)) ++bad;
        }
    printf("mfma_f32_16x16x4 layout row = 4*(lane>>4)+reg, col = lane&15: %s (%d mismatches); lane 17 regs: %g %g %g %g\n",
           bad ? "WRONG" : "confirmed", bad, h[68], h[69], h[70], h[71]);
    return bad != 0;
}
