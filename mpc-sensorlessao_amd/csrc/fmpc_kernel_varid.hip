// VAR(2) identification on the device (first half; reference: README.md:108-130):
//     AA(i-2, :) = [ad_acc(i-1, :), ad_acc(i-2, :)],  BB(i-2, :) = ad_acc(i, :),   i = 3 .. num_train
//     PARA = (AA'*AA) \ AA'*BB ;   A1 = PARA(1:n, :)' ;  A2 = PARA(n+1:2n, :)'
// One workgroup per coefficient series (realisation).  The Gram matrices G = AA'AA (2n x 2n) and H = AA'BB (2n x n) are
// X'Z products with the time samples as the contraction index: fp64 MFMA tiles of 16 x 16 with both operands read
// straight from the series (a row of AA is two consecutive samples), then a Cholesky solve of the normal equations in
// LDS -- the reference's `\` on a symmetric positive definite matrix is a Cholesky solve as well.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/fastmpc.h"

typedef double vi_d4 __attribute__((ext_vector_type(4)));
#define VI_THREADS 256
#define VI_MAXP 64                         // 2n <= 64

// series: [batch][num_samples][n] (one n-vector per time step, contiguous); A1, A2: [batch] n x n column-major
__global__ void __launch_bounds__(VI_THREADS, 2)
fmpc_var_identify_kernel(int n, int num_train, int num_samples, int batch, const double* __restrict__ series,
                         double* __restrict__ A1, double* __restrict__ A2, int* __restrict__ status) {
    __shared__ double G[VI_MAXP][VI_MAXP + 1];      // Gram matrix, then its Cholesky factor (lower)
    __shared__ double H[VI_MAXP][33];               // right-hand sides AA'BB, then PARA
    __shared__ int sbad;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;
    const int p2 = 2 * n, PT = (p2 + 15) / 16, NT_ = (n + 15) / 16, rows = num_train - 2;
    for (int r = blockIdx.x; r < batch; r += gridDim.x) {
        const double* a = series + (size_t)r * num_samples * n;
        if (tid == 0) sbad = 0;
        // ---- G (upper-triangular tiles) and H tiles: out[p][q] = sum_k AA[k][p] * (AA | BB)[k][q]
        // AA[k][p] = a[(k + 1 - p / n) * n + p % n] for p < 2n (0 beyond), BB[k][q] = a[(k + 2) * n + q]
        const int ngt = PT * (PT + 1) / 2, items = ngt + PT * NT_;
        for (int item = wv; item < items; item += 4) {
            int I, J; bool isH = item >= ngt;
            if (!isH) { int t = item; I = 0; while (t >= PT - I) { t -= PT - I; ++I; } J = I + t; }
            else { const int t = item - ngt; I = t / NT_; J = t - I * NT_; }
            const int pa = 16 * I + c, pb = 16 * J + c;                       // this lane's column of the A / B operand
            const bool aok = pa < p2, bok = isH ? pb < n : pb < p2;
            // unconditional loads from clamped addresses, zeros by a factor (a conditional load is a branch)
            const int pac = aok ? pa : 0, pbc = bok ? pb : 0;
            const double* xa = a + (size_t)(1 - pac / n) * n + pac % n;
            const double* zb = isH ? a + (size_t)2 * n + pbc : a + (size_t)(1 - pbc / n) * n + pbc % n;
            const double fa = aok ? 1.0 : 0.0, fb = bok ? 1.0 : 0.0;
            vi_d4 acc = {0, 0, 0, 0};
            int k0 = 0;
            for (; k0 + 16 <= rows; k0 += 16) {
                double xv[4], zv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { xv[u] = xa[(size_t)(k0 + 4 * u + g) * n]; zv[u] = zb[(size_t)(k0 + 4 * u + g) * n]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[u] * fa, zv[u] * fb, acc, 0, 0, 0);
            }
            for (; k0 < rows; k0 += 4) {
                const int k = k0 + g;
                const int kc = k < rows ? k : rows - 1;
                const double fk = k < rows ? 1.0 : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[(size_t)kc * n] * fa * fk, zb[(size_t)kc * n] * fb, acc, 0, 0, 0);
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = 16 * I + g + 4 * rr, col = 16 * J + c;        // accumulator layout: row g + 4 rr, column c
                if (!isH) { if (row < VI_MAXP && col < VI_MAXP) { G[row][col] = acc[rr]; G[col][row] = acc[rr]; } }
                else if (row < VI_MAXP && col < 32) H[row][col] = acc[rr];
            }
        }
        __syncthreads();
        // ---- Cholesky of G (lower), right-looking, one barrier per column
        bool fail = false;
        for (int k = 0; k < p2; ++k) {
            const double piv = G[k][k];
            if (!(piv > 0.0) || isinf(piv)) { fail = true; break; }           // uniform
            const double ip = 1.0 / piv;
            const int rem = p2 - k - 1;
            for (int idx = tid; idx < rem * rem; idx += VI_THREADS) {
                const int rr = k + 1 + idx / rem, cc = k + 1 + idx % rem;
                if (cc <= rr) G[rr][cc] -= G[rr][k] * G[cc][k] * ip;
            }
            __syncthreads();
        }
        if (!fail) {
            for (int idx = tid; idx < p2 * p2; idx += VI_THREADS) {
                const int rr = idx / p2, cc = idx - rr * p2;
                if (cc < rr) G[rr][cc] /= sqrt(G[cc][cc]);
            }
            __syncthreads();
            if (tid < p2) G[tid][tid] = sqrt(G[tid][tid]);
            __syncthreads();
            // ---- PARA = G^-1 H: one thread per right-hand side column
            if (tid < n) {
                for (int rr = 0; rr < p2; ++rr) {
                    double v = H[rr][tid];
                    for (int j = 0; j < rr; ++j) v -= G[rr][j] * H[j][tid];
                    H[rr][tid] = v / G[rr][rr];
                }
                for (int rr = p2 - 1; rr >= 0; --rr) {
                    double v = H[rr][tid];
                    for (int j = rr + 1; j < p2; ++j) v -= G[j][rr] * H[j][tid];
                    H[rr][tid] = v / G[rr][rr];
                }
            }
            __syncthreads();
            // A1 = PARA(1:n, :)', A2 = PARA(n+1:2n, :)'   (column-major n x n: A[i + j n] = PARA[j][i])
            for (int idx = tid; idx < n * n; idx += VI_THREADS) {
                const int i = idx % n, j = idx / n;
                A1[(size_t)r * n * n + idx] = H[j][i];
                A2[(size_t)r * n * n + idx] = H[n + j][i];
            }
        }
        if (tid == 0 && status) status[r] = fail ? FMPC_E_NOT_PD_SCHUR : FMPC_OK;
        __syncthreads();
    }
}

hipError_t fmpc_launch_var_identify(int n, int num_train, int num_samples, int batch, const double* series, double* A1,
                                    double* A2, int* status, hipStream_t stream) {
    if (2 * n > VI_MAXP || n > 32) return hipErrorInvalidValue;
    int grid = batch < 1024 ? batch : 1024;
    hipLaunchKernelGGL(fmpc_var_identify_kernel, dim3(grid), dim3(VI_THREADS), 0, stream, n, num_train, num_samples, batch,
                       series, A1, A2, status);
    return hipGetLastError();
}
