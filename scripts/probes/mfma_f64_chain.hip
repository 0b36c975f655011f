// Microbenchmark: v_mfma_f64_16x16x4_f64 issue rate vs. the number of independent accumulator chains and waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_f64_chain.hip -o scripts/probes/mfma_f64_chain ; run on the GPU box.
// Result (MI355X): 64 cycles per MFMA, a single wave per SIMD already saturates the pipe, independent accumulator
// chains change nothing; with several waves per SIMD the oldest wave is served first.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

template <int CH>
__global__ void k(double* out, unsigned long long* cyc, int iters) {
    d4 acc[CH];
    const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int c = 0; c < CH; ++c) acc[c] = d4{0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = MFMA64(a, b, acc[c]);
    }
    d4 s = acc[0];
    for (int c = 1; c < CH; ++c) s += acc[c];
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int CH>
void run(int waves, double* out, unsigned long long* cyc) {
    const int iters = 1 << 14;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CH>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 16);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<CH>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    (void)hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double nm = (double)iters * CH;
    printf("chains %d, waves/WG %2d (%.1f per SIMD): %.1f ticks per MFMA per wave; %.1f ns per MFMA per wave, %.2f ns per MFMA per SIMD (tick rate %.2f GHz)\n", CH, waves, waves / 4.0,
           (double)c / nm, ms * 1e6 / nm, ms * 1e6 / (nm * (waves > 4 ? waves / 4.0 : 1.0)), c / (ms * 1e6));
}

int main() {
    double* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 8);
    for (int waves : {1, 4, 8, 16}) {
        run<1>(waves, out, cyc); run<2>(waves, out, cyc); run<4>(waves, out, cyc); run<8>(waves, out, cyc);
    }
    return 0;
}
