// Microbenchmark: v_mfma_f64_16x16x4_f64 fed from LDS (B operand) and from global memory (A operand), the inner loop of the
// dense-form product (fmpc_kernel_inv.hip).  Prints cycles per MFMA and wave for a few feeding patterns.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_lds_feed.hip -o scripts/probes/mfma_lds_feed && scripts/probes/mfma_lds_feed
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// MODE 0: A, B in registers.  1: B from LDS every k-step (64 consecutive doubles), A in registers.
// 2: B from LDS, A from global (one 512-byte load per k-step, 16 k-steps requested ahead).  3: like 2 plus a barrier per 16 k-steps.
template <int MODE, int NACC>
__global__ void feed(const double* __restrict__ A, double* out, unsigned long long* cyc, int chunks) {
    __shared__ double B[2][16 * 66];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 16 * 66; i += blockDim.x) (&B[0][0])[i] = 1.0 + 1e-3 * i;
    __syncthreads();
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    // MODE >= 4: A shared by many workgroups (L2-resident, like the image of J): 13 x 4 row tiles
    const double* ap = A + (size_t)((MODE >= 4 ? blockIdx.x % 13 : blockIdx.x) * 4 + wv) * chunks * 16 * 64 + lane;
    const double* wp = A + (size_t)(blockIdx.x % 200) * 4096 + threadIdx.x;      // MODE 5: a chunk of d staged per 16 k-steps
    double a0[16], a1[16];
    for (int u = 0; u < 16; ++u) { a0[u] = 1.0 + lane * 1e-3 + u; a1[u] = a0[u] * 0.5; }
    if (MODE >= 2) for (int u = 0; u < 16; ++u) a0[u] = ap[u * 64];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int ch = 0; ch < chunks; ch += 2) {
        if (MODE >= 2) for (int u = 0; u < 16; ++u) a1[u] = ap[((ch + 1) * 16 + u) * 64];
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                const double b = MODE >= 1 ? B[0][u * 66 + ((64 * j + lane) & 63)] : a1[u];
                acc[j] = MFMA(a0[u], b, acc[j]);
            }
        if (MODE >= 5) { double t[8]; for (int i = 0; i < 8; ++i) t[i] = wp[(ch * 8 + i) * 256 % 3000]; for (int i = 0; i < 8; ++i) B[1][(lane >> 2) * 66 + 16 * (lane & 3) + ((wv * 2 + i) & 15)] = t[i]; }
        if (MODE >= 3) __syncthreads();
        if (MODE >= 2) { const int c2 = ch + 2 < chunks ? ch + 2 : ch; for (int u = 0; u < 16; ++u) a0[u] = ap[(c2 * 16 + u) * 64]; }
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                const double b = MODE >= 1 ? B[1][u * 66 + ((64 * j + lane) & 63)] : a0[u] * 0.25;
                acc[j] = MFMA(a1[u], b, acc[j]);
            }
        if (MODE >= 3) __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE, int NACC> void run(int blocks, const char* tag) {
    const int chunks = 64, threads = 256;
    double *A, *out; unsigned long long *cyc, h;
    const size_t na = (size_t)blocks * 4 * chunks * 16 * 64;
    hipMalloc(&A, na * 8); hipMalloc(&out, 8 * threads * blocks); hipMalloc(&cyc, 8);
    hipMemset(A, 0, na * 8);
    feed<MODE, NACC><<<blocks, threads>>>(A, out, cyc, chunks); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); feed<MODE, NACC><<<blocks, threads>>>(A, out, cyc, chunks); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)chunks * 16 * NACC;
    printf("%-34s mode %d acc %d blocks %4d: %6.1f cycles/MFMA/wave, %.3f ms, %.1f TFLOP/s\n", tag, MODE, NACC, blocks, h / n, ms,
           n * 2048.0 * 4 * blocks / (ms * 1e-3) / 1e12);
    hipFree(A); hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 1>(256, "registers only"); run<0, 2>(256, "registers only");
    run<1, 1>(256, "B from LDS"); run<1, 2>(256, "B from LDS");
    run<2, 1>(256, "B LDS, A global"); run<2, 2>(256, "B LDS, A global");
    run<3, 1>(256, "B LDS, A global, barriers"); run<3, 2>(256, "B LDS, A global, barriers");
    run<3, 1>(768, "same, 3 workgroups per CU"); run<3, 2>(768, "same, 3 workgroups per CU");
    run<4, 1>(256, "A shared (L2), barriers"); run<4, 2>(256, "A shared (L2), barriers"); run<4, 1>(832, "A shared, 3.25 wg/CU"); run<4, 2>(832, "A shared, 3.25 wg/CU");
    run<5, 1>(832, "+ d chunk staged"); run<5, 2>(416, "+ d chunk staged");
    return 0;
}
