// Probe (round 4): how many vector-memory wave-instructions a CU retires per cycle -- the factor phase of fmpc_newton_wave
// issues ~130 of them per stage and wave, mostly 8-byte-per-lane accesses of 27-28 active lanes.  Per CU: 8 wavefronts (one
// 512-thread workgroup), each streaming through its own region (HBM / L2 resident), 8 independent requests in flight.
//   modes: load x2 64 lanes | load x2 28 lanes | load x4 64 lanes | store x2 64 | store x2 28 lanes | store x4 64 lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define IT 512
typedef double d2 __attribute__((ext_vector_type(2)));
template <int MODE, int SHARED>
__global__ void __launch_bounds__(512, 1) probe(double* buf, size_t per_wave, unsigned long long* cyc, double* sink) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // shared != 0: every CU streams the SAME 8 regions (L2-resident): the instruction rate, not the HBM rate
    double* base = buf + ((size_t)(SHARED ? 0 : blockIdx.x) * 8 + wv) * per_wave;
    double acc = 0.0;
    const bool act = (MODE == 1 || MODE == 4) ? lane < 28 : true;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < IT; it += 8) {
        if (MODE == 0 || MODE == 1) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = act ? base[(size_t)(it + q) * 64 + lane] : 0.0;      // 512 B (224 B) per instruction
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
        } else if (MODE == 2) {
            d2 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = ((const d2*)base)[(size_t)(it + q) * 64 + lane];     // 1 KB per instruction
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q].x + v[q].y;
        } else if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int q = 0; q < 8; ++q) if (act) base[(size_t)(it + q) * 64 + lane] = acc + q;
        } else if (MODE == 5) {
#pragma unroll
            for (int q = 0; q < 8; ++q) ((d2*)base)[(size_t)(it + q) * 64 + lane] = (d2){acc + q, acc - q};
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int MODE, int SHARED>
void run(const char* name, size_t per_wave, const char* where, int bytes_per_instr) {
    double *buf, *sink; unsigned long long* cyc;
    const int grid = 256;
    hipMalloc(&buf, (size_t)grid * 8 * per_wave * 8); hipMalloc(&sink, grid * 512 * 8); hipMalloc(&cyc, grid * 8 * 8);
    hipMemset(buf, 0, (size_t)grid * 8 * per_wave * 8);
    for (int rep = 0; rep < 2; ++rep) { probe<MODE, SHARED><<<grid, 512>>>(buf, per_wave, cyc, sink); hipDeviceSynchronize(); }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); probe<MODE, SHARED><<<grid, 512>>>(buf, per_wave, cyc, sink); hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(grid * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= h.size();
    printf("%-30s %-12s %7.1f cycles per instruction per wave = %5.1f per CU (8 waves); %6.2f TB/s chip-wide, kernel %.1f us\n", name, where, m / IT, m / IT / 8,
           (double)grid * 8 * IT * bytes_per_instr / (ms * 1e-3) / 1e12, ms * 1e3);
    hipFree(buf); hipFree(sink); hipFree(cyc);
}
int main() {
    // per_wave doubles: IT * 64 * (1 or 2)
    run<0, 0>("load  dwordx2, 64 lanes", IT * 64, "(64 MB)", 512);
    run<1, 0>("load  dwordx2, 28 lanes", IT * 64, "(64 MB)", 224);
    run<2, 0>("load  dwordx4, 64 lanes", IT * 128, "(128 MB)", 1024);
    run<3, 0>("store dwordx2, 64 lanes", IT * 64, "(64 MB)", 512);
    run<4, 0>("store dwordx2, 28 lanes", IT * 64, "(64 MB)", 224);
    run<5, 0>("store dwordx4, 64 lanes", IT * 128, "(128 MB)", 1024);
    run<0, 1>("load  dwordx2, 64 lanes", IT * 64, "(L2, 2 MB)", 512);
    run<1, 1>("load  dwordx2, 28 lanes", IT * 64, "(L2, 2 MB)", 224);
    run<2, 1>("load  dwordx4, 64 lanes", IT * 128, "(L2, 4 MB)", 1024);
    return 0;
}
