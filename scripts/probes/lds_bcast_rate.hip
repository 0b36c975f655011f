// Probe (round 5): what a UNIFORM-address LDS read (the column broadcasts of the wave kernel's potrf loop) costs the CU's LDS pipe
// with 1 and with 8 wavefronts per CU: ds_read2_b64 (two doubles), ds_read_b128 (two doubles, aligned), ds_read_b64, and -- for
// comparison -- the same 16 doubles fetched by v_readlane pairs.
//   hipcc --offload-arch=gfx950 -O3 -o lds_bcast_rate lds_bcast_rate.hip && ./lds_bcast_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define IT 4000
typedef double d2 __attribute__((ext_vector_type(2)));
template <int MODE, int NFMA>
__global__ void __launch_bounds__(512, 1) probe(double* out, unsigned long long* cyc, double seed) {
    __shared__ __attribute__((aligned(16))) double lds[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = seed * (i + 1);
    __syncthreads();
    double a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = seed * (q + 1) + threadIdx.x;
    const double c = seed * 0.5;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) double*)lds;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < IT; ++it) {
        const unsigned p = base + (it & 7) * 128;
        double v[16];
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { d2 t; asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(t) : "v"(p), "n"(2 * q), "n"(2 * q + 1)); v[2 * q] = t.x; v[2 * q + 1] = t.y; }
        } else if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { d2 t; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(p), "n"(16 * q)); v[2 * q] = t.x; v[2 * q + 1] = t.y; }
        } else if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < 16; ++q) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[q]) : "v"(p), "n"(8 * q));
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                int lo = __builtin_amdgcn_readlane(__double2loint(a[q]), q), hi = __builtin_amdgcn_readlane(__double2hiint(a[q]), q);
                v[q] = __hiloint2double(hi, lo);
            }
        }
        if (MODE < 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                                   "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
#pragma unroll
        for (int q = 0; q < NFMA; ++q) a[q & 15] = fma(a[q & 15], v[q & 15], c);
        if (NFMA == 0) a[0] += v[it & 15];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += a[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE, int NFMA>
void run(const char* name) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8 * 8);
    for (int threads : {64, 256, 512}) {
        probe<MODE, NFMA><<<256, threads>>>(out, cyc, 1e-9);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double m = 0; for (auto v : h) m += v; m /= h.size();
        printf("%-52s %d wave(s)/CU: %8.1f cycles per iteration (16 doubles broadcast%s) per wave\n", name, threads / 64, m / IT, NFMA ? " + fma" : "");
    }
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 0>("8 ds_read2_b64 uniform");
    run<1, 0>("8 ds_read_b128 uniform (aligned)");
    run<2, 0>("16 ds_read_b64 uniform");
    run<3, 0>("32 v_readlane");
    run<0, 32>("8 ds_read2_b64 uniform + 32 v_fma_f64");
    run<1, 32>("8 ds_read_b128 uniform + 32 v_fma_f64");
    run<3, 32>("32 v_readlane + 32 v_fma_f64");
    return 0;
}
