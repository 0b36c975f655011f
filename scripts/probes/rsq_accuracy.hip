// Accuracy of v_rsq_f64 / v_rcp_f64 and of their Newton refinements (MI355X): max relative error over 1e6 random arguments.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/rsq_accuracy.hip -o /tmp/rsq_accuracy && /tmp/rsq_accuracy
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const double* d, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = d[i];
    double y0 = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    double y1 = y0 * fma(-h * y0, y0, 1.5);
    double y2 = y1 * fma(-h * y1, y1, 1.5);
    double r0 = __builtin_amdgcn_rcp(x);
    double r1 = fma(r0, fma(-x, r0, 1.0), r0);
    double r2 = fma(r1, fma(-x, r1, 1.0), r1);
    out[6 * i + 0] = y0; out[6 * i + 1] = y1; out[6 * i + 2] = y2; out[6 * i + 3] = r0; out[6 * i + 4] = r1; out[6 * i + 5] = r2;
}
int main() {
    const int n = 1 << 20;
    double* h = (double*)malloc(n * sizeof(double)); double* ho = (double*)malloc(6 * n * sizeof(double));
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = exp(40.0 * ((double)rand() / RAND_MAX - 0.5));
    double *d, *o; hipMalloc(&d, n * sizeof(double)); hipMalloc(&o, 6 * n * sizeof(double));
    hipMemcpy(d, h, n * sizeof(double), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o, n);
    hipMemcpy(ho, o, 6 * n * sizeof(double), hipMemcpyDeviceToHost);
    double e[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const long double x = h[i], ys = 1.0L / sqrtl(x), rs = 1.0L / x;
        for (int q = 0; q < 3; ++q) { const double r = (double)fabsl((ho[6 * i + q] - ys) / ys); if (r > e[q]) e[q] = r; }
        for (int q = 3; q < 6; ++q) { const double r = (double)fabsl((ho[6 * i + q] - rs) / rs); if (r > e[q]) e[q] = r; }
    }
    printf("max relative error: v_rsq_f64 %.3e, +1 Newton %.3e, +2 Newton %.3e | v_rcp_f64 %.3e, +1 Newton %.3e, +2 Newton %.3e  (2^-52 = 2.2e-16)\n",
           e[0], e[1], e[2], e[3], e[4], e[5]);
    return 0;
}
