// Microbenchmark: how fast do 125 / 256 workgroups drain 656 KB of stores each, for three write patterns?
//   0: fully coalesced (each wave instruction writes 512 contiguous bytes)
//   1: panel kernel, transposed epilogue: per instruction 4 problems x 128 contiguous bytes (problems 41 KB apart)
//   2: panel kernel, D-layout epilogue: per instruction 16 problems x 32 contiguous bytes
//   3: as 1 with 16-byte stores: per instruction 4 problems x 256 contiguous bytes
#include <hip/hip_runtime.h>
#include <stdio.h>
#define NZ 5130
__global__ void __launch_bounds__(512) wr(double* out, int mode, int reps) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    double* base = out + (size_t)blockIdx.x * 16 * NZ;
    const double v = threadIdx.x;
    for (int rep = 0; rep < reps; ++rep) {
        if (mode == 0) {
            for (int i = threadIdx.x; i < 16 * NZ; i += 512) base[i] = v + rep;
        } else if (mode == 1) {
            for (int j = wv; j < 30; j += 8)
                for (int J = 0; J < 10; ++J)
                    for (int r = 0; r < 4; ++r) base[(size_t)(4 * r + g) * NZ + j * 171 + 16 * J + c] = v + rep;
        } else if (mode == 2) {
            for (int j = wv; j < 30; j += 8)
                for (int J = 0; J < 10; ++J)
                    for (int r = 0; r < 4; ++r) base[(size_t)c * NZ + j * 171 + 16 * J + 4 * r + g] = v + rep;
        } else {
            typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
            for (int j = wv; j < 30; j += 8)
                for (int J = 0; J < 5; ++J)
                    for (int r = 0; r < 4; ++r) *(d2*)(base + (size_t)(4 * r + g) * NZ + j * 171 + 32 * J + 2 * c) = (d2){v + rep, v};
        }
    }
}
int main() {
    double* out; hipMalloc(&out, (size_t)256 * 16 * NZ * 8 + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {125, 256})
        for (int mode = 0; mode < 4; ++mode) {
            wr<<<grid, 512>>>(out, mode, 1); hipDeviceSynchronize();
            float best = 1e9;
            for (int t = 0; t < 5; ++t) {
                hipEventRecord(e0); wr<<<grid, 512>>>(out, mode, 1); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double bytes = (double)grid * 16 * (mode == 0 ? NZ : 30 * 160) * 8;
            printf("grid %3d mode %d: %.1f us, %.2f TB/s\n", grid, mode, best * 1e3, bytes / (best * 1e-3) / 1e12);
        }
    return 0;
}
