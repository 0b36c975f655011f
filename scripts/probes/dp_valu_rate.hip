// Probe (round 4): issue cost of fp64 vector instructions on one SIMD of gfx950 -- independent v_fma_f64 streams, a dependent
// chain, v_readlane + use, uniform-address ds_read2_b64 + use -- with one and with two wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o dp_valu_rate dp_valu_rate.hip && ./dp_valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define IT 2000
template <int MODE>
__global__ void __launch_bounds__(512, 1) probe(double* out, unsigned long long* cyc, double seed) {
    __shared__ double lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = seed * i;
    __syncthreads();
    double a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = seed * (q + 1) + threadIdx.x;
    const double b = seed * 1.0000001, c = seed * 0.5;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < IT; ++it) {
        if (MODE == 0) {            // 16 independent chains
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = fma(a[q], b, c);
        } else if (MODE == 1) {     // one dependent chain of 16
#pragma unroll
            for (int q = 0; q < 16; ++q) a[0] = fma(a[0], b, c);
        } else if (MODE == 2) {     // 4 chains
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q & 3] = fma(a[q & 3], b, c);
        } else if (MODE == 3) {     // readlane broadcast feeding an fma (the pivot chain's shape): 8 x (2 readlane + fma)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                int lo = __builtin_amdgcn_readlane(__double2loint(a[0]), q), hi = __builtin_amdgcn_readlane(__double2hiint(a[0]), q);
                a[0] = fma(a[0], __hiloint2double(hi, lo), c);
            }
        } else if (MODE == 4) {     // 8 uniform-address ds_read2_b64 feeding 16 independent fma
            const double* p = lds + (it & 7) * 16;
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = p[q];
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = fma(a[q], v[q], c);
        } else if (MODE == 5) {     // fp32 reference: 16 independent v_fma_f32
            float* f = (float*)a;
#pragma unroll
            for (int q = 0; q < 16; ++q) f[q] = fmaf(f[q], 1.0000001f, 0.5f);
        } else if (MODE == 6) {     // v_mul_f64
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = a[q] * b;
        } else if (MODE == 7) {     // 16 fma with an SGPR operand (broadcast value in scalar registers)
            double s = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(b)), __builtin_amdgcn_readfirstlane(__double2loint(b)));
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = fma(a[q], s, c);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += a[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE>
void run(const char* name, int per_it) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8 * 8);
    for (int threads : {256, 512}) {
        probe<MODE><<<256, threads>>>(out, cyc, 1e-9);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double m = 0; for (auto v : h) m += v; m /= h.size();
        printf("%-44s %d wave(s)/SIMD: %7.2f cycles per instruction per wave (%d per iteration)\n", name, threads / 256, m / IT / per_it, per_it);
    }
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("v_fma_f64, 16 independent", 16);
    run<2>("v_fma_f64, 4 chains", 16);
    run<1>("v_fma_f64, 1 dependent chain", 16);
    run<6>("v_mul_f64, 16 independent", 16);
    run<7>("v_fma_f64 with SGPR operand, 16 independent", 16);
    run<3>("2 v_readlane + dependent v_fma_f64 (per triple)", 8);
    run<4>("8 ds_read2_b64 uniform + 16 fma (per fma)", 16);
    run<5>("v_fma_f32, 16 independent", 16);
    return 0;
}
