// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for 8-byte-per-lane coalesced access
// (the pattern of the fastMPC kernels); the microarch guide calibrates only 16 B/lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void rd8(const double* __restrict__ in, double* out, size_t n) {
    double s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += in[i];
    if (s == 12345.678) out[0] = s;
}
__global__ void wr8(double* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (double)i;
}
__global__ void rd8_rows27(const double* __restrict__ in, double* out, size_t rows) {   // 27 lanes x 8 B = 216-B pieces, row stride 224 B
    double s = 0; const int lane = threadIdx.x & 63; const size_t w = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = w; r < rows; r += nw) if (lane < 27) s += in[r * 28 + lane];
    if (s == 12345.678) out[0] = s;
}
int main() {
    const size_t n = (size_t)1 << 27;  // 1 GiB of doubles
    double *a, *o; (void)hipMalloc(&a, n * 8); (void)hipMalloc(&o, 64); (void)hipMemset(a, 0, n * 8);
    rd8<<<4096, 256>>>(a, o, n); (void)hipDeviceSynchronize();
    wr8<<<4096, 256>>>(a, n); (void)hipDeviceSynchronize();
    rd8_rows27<<<4096, 256>>>(a, o, n / 28); (void)hipDeviceSynchronize();
    printf("bytes read by rd8 = %zu, written by wr8 = %zu, useful bytes rd8_rows27 = %zu (rows span %zu)\n", n * 8, n * 8, (n / 28) * 27 * 8, (n / 28) * 224);
    return 0;
}
