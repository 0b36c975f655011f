// Microbenchmark: what a kernel costs beyond the lifetime of its wavefronts, as a function of the bytes it wrote.
// A streaming-write kernel records the first start and the last end of its wavefronts (constant 100 MHz clock); the launch
// is also timed with HIP events in a stream of back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/kernel_boundary.hip -o scripts/probes/kernel_boundary && scripts/probes/kernel_boundary
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void writer(double* out, size_t n_per_block, unsigned long long* tmin, unsigned long long* tmax) {
    const unsigned long long t0 = wall_clock64();
    double* p = out + (size_t)blockIdx.x * n_per_block;
    for (size_t i = threadIdx.x; i < n_per_block; i += blockDim.x) p[i] = (double)i;
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { atomicMin(tmin, t0); atomicMax(tmax, t1); }
}
int main() {
    unsigned long long *tmin, *tmax;
    hipMalloc(&tmin, 8); hipMalloc(&tmax, 8);
    double* buf; hipMalloc(&buf, (size_t)256 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double mbs[] = {0.05, 1, 4, 13.4, 40, 82, 200};
    for (double mb : mbs) {
        const int blocks = 1024, threads = 256;
        const size_t n_per_block = (size_t)(mb * 1e6 / 8 / blocks);
        for (int rep = 0; rep < 3; ++rep) writer<<<blocks, threads>>>(buf, n_per_block, tmin, tmax);
        hipDeviceSynchronize();
        const unsigned long long big = ~0ull, zero = 0;
        float ms_sum = 0; double life_sum = 0; const int R = 10;
        for (int rep = 0; rep < R; ++rep) {
            hipMemcpy(tmin, &big, 8, hipMemcpyHostToDevice); hipMemcpy(tmax, &zero, 8, hipMemcpyHostToDevice);
            hipEventRecord(e0); writer<<<blocks, threads>>>(buf, n_per_block, tmin, tmax); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms_sum += ms;
            unsigned long long a, b; hipMemcpy(&a, tmin, 8, hipMemcpyDeviceToHost); hipMemcpy(&b, tmax, 8, hipMemcpyDeviceToHost);
            life_sum += (double)(b - a) * 0.01;
        }
        printf("%7.2f MB written: events %7.2f us, first wavefront start .. last wavefront end %7.2f us, difference %6.2f us\n",
               mb, ms_sum / R * 1e3, life_sum / R, ms_sum / R * 1e3 - life_sum / R);
    }
    return 0;
}
