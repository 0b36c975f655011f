// Does v_mfma_f64_16x16x4 overlap with VALU work of the same wave / a partner wave on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: MFMA only, 1: f64 FMA only, 2: MFMA + 12 f64 FMA, 3: i32 VALU only, 4: MFMA + 12 i32 ops, 5: readlane+fma mix, 6: MFMA + that mix
__global__ void k(double* out, int iters) {
    d4 acc = {0, 0, 0, 0};
    double a = threadIdx.x * 0.001 + 1.0, b = 0.5;
    double f[12]; int q[12];
    for (int i = 0; i < 12; ++i) { f[i] = a + i; q[i] = threadIdx.x + i; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 6) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 12; ++i) f[i] = fma(f[i], 1.0000001, 0.5);
        }
        if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int i = 0; i < 12; ++i) q[i] = q[i] * 3 + 1;
        }
        if (MODE == 5 || MODE == 6) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                int lo = __builtin_amdgcn_readlane(__double2loint(f[i]), i + 1), hi = __builtin_amdgcn_readlane(__double2hiint(f[i]), i + 1);
                f[i + 6] = fma(f[i + 6], __hiloint2double(hi, lo), 0.25);
            }
        }
    }
    double s = acc[0] + acc[3];
    for (int i = 0; i < 12; ++i) s += f[i] + q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> float run(int threads) {
    double* out; hipMalloc(&out, 8 * threads * 256);
    const int iters = 20000;
    k<MODE><<<256, threads>>>(out, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<MODE><<<256, threads>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(out); return ms;
}
int main() {
    for (int threads = 256; threads <= 512; threads *= 2) {
        printf("%d waves/SIMD: mfma %.3f | f64fma x12 %.3f | both %.3f || i32 x12 %.3f | mfma+i32 %.3f || readlane-fma x6 %.3f | mfma+that %.3f  (ms)\n", threads / 256,
               run<0>(threads), run<1>(threads), run<2>(threads), run<3>(threads), run<4>(threads), run<5>(threads), run<6>(threads));
    }
    return 0;
}
