// Microbenchmark: cycles per v_mfma_f64_16x16x4_f64 on gfx950 (per wave, per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void rate(double* out, unsigned long long* cyc, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = threadIdx.x * 0.001 + 1.0, b = 0.5 - threadIdx.x * 0.002;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC> void run(int threads, int blocks, const char* tag) {
    double* out; unsigned long long* cyc, h; hipMalloc(&out, 8 * threads * blocks); hipMalloc(&cyc, 8);
    const int iters = 20000;
    rate<NACC><<<blocks, threads>>>(out, cyc, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); rate<NACC><<<blocks, threads>>>(out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * NACC;
    double tf = n * 2048.0 * (threads / 64) * blocks / (ms * 1e-3) / 1e12;
    printf("%-28s acc=%d: %.1f cycles/MFMA/wave, %.3f ms, %.1f TFLOP/s chip-wide\n", tag, NACC, h / n, ms, tf);
}
int main() {
    run<1>(256, 256, "1 wave/SIMD"); run<2>(256, 256, "1 wave/SIMD"); run<4>(256, 256, "1 wave/SIMD");
    run<4>(512, 256, "2 waves/SIMD"); run<3>(512, 256, "2 waves/SIMD"); run<4>(1024, 256, "4 waves/SIMD");
    return 0;
}
