// Probe: v_mfma_f64_4x4x4_4b_f64 on gfx950 -- issue rate and operand/result lane maps.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
template <int NACC>
__global__ void rate(double* out, unsigned long long* cyc, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a = threadIdx.x * 0.001 + 1.0, b = 0.5 - threadIdx.x * 0.002;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i];
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
    double tf = n * 512.0 * (threads / 64) * blocks / (ms * 1e-3) / 1e12;
    printf("%-16s acc=%d: %.1f cycles/MFMA/wave, %.3f ms, %.1f TFLOP/s chip-wide\n", tag, NACC, h / n, ms, tf);
}
// layout: one-hot probing.  a one-hot on lane la, b one-hot on lane lb -> which lanes of D are nonzero
__global__ void probe(const double* a, const double* b, double* d) {
    d[threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], 0.0, 0, 0, 0);
}
int main() {
    run<1>(256, 256, "1 wave/SIMD"); run<2>(256, 256, "1 wave/SIMD"); run<4>(256, 256, "1 wave/SIMD"); run<8>(256, 256, "1 wave/SIMD");
    run<4>(512, 256, "2 waves/SIMD");
    double *da, *db, *dd; hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
    double ha[64], hb[64], hd[64];
    // for each (la, lb) pair record which D lanes light up; print a compact table
    printf("pairs (la,lb) -> D lanes (value 1):\n");
    for (int la = 0; la < 64; ++la) {
        for (int i = 0; i < 64; ++i) ha[i] = (i == la);
        int cnt = 0;
        printf("la=%2d:", la);
        for (int lb = 0; lb < 64; ++lb) {
            for (int i = 0; i < 64; ++i) hb[i] = (i == lb);
            hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
            probe<<<1, 64>>>(da, db, dd); hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
            for (int i = 0; i < 64; ++i) if (hd[i] != 0.0) { printf(" b%d->d%d", lb, i); ++cnt; }
        }
        printf("\n");
    }
    return 0;
}
