// Dense input weight R, shared by the tiled kernel (scratch in LDS) and the generic kernel's workspace instance (scratch in HBM).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

// Dense R (fast_mpc_objective.m:51-54 takes any square R; inf_newton_KKT_H.m:13, inf_newton_solver.m:24): the u block of
// Phi at stage j, Rt_j = 2R + k diag(1/s+^2 + 1/s-^2), is a dense symmetric positive definite m x m matrix.  Per stage, the
// whole workgroup: Cholesky in LDS on the packed lower triangle (columns kept UNSCALED, L[i][k] = a[i][k] d[k] with
// d[k] = 1/sqrt(pivot): one barrier per column), then the two substitutions on the m x (n + 1) right-hand sides
// [B' | r_d[u_j]] in right-looking form (one barrier per row).  Out: zt[j][q][0..n] = [Rt_j^-1 B' | Rt_j^-1 r_d[u_j]] -- what
// the Schur complement (B Rt_j^-1 B'), the right-hand side and d_u need.  Returns 1 if a pivot is not positive.
// (A generality path: ~4 m barriers and m^3/3 + 2 m^2 (n + 1) flops on the vector units per stage.)
static __device__ __noinline__ int ft_dense_r(double* sL, const double* R2P, int MP, const double* Bt, const double* hess,
                                       const double* rdu, double* zt, int n, int m, int T) {
    const int tid = threadIdx.x, NT = blockDim.x;
    const int ZLD = n + 1;
    double* sX = sL + (size_t)m * (m + 1) / 2;
    double* sD = sX + (size_t)m * ZLD;
    const int ty = tid >> 4, tx = tid & 15, NY = NT >> 4;          // triangle work: rows by ty, columns by tx
    const int cy = tid >> 5, cx = tid & 31, NC = NT >> 5;          // right-hand sides: rows by cy, columns by cx (< ZLD)
    int bad = 0;
    for (int j = 0; j < T; ++j) {
        __syncthreads();
        for (int i = ty; i < m; i += NY)
            for (int k = tx; k <= i; k += 16)
                sL[(size_t)i * (i + 1) / 2 + k] = R2P[(size_t)i * MP + k] + (i == k ? hess[(size_t)j * m + i] : 0.0);
        for (int e = tid; e < m * ZLD; e += NT) {
            const int q = e / ZLD, c = e - q * ZLD;
            sX[e] = c < n ? Bt[(size_t)q * n + c] : rdu[(size_t)j * m + q];
        }
        // ---- factor: a[i][jj] -= a[i][k] a[jj][k] / a[k][k]  for k < jj <= i
        for (int k = 0; k < m; ++k) {
            __syncthreads();
            double dkk = sL[(size_t)k * (k + 1) / 2 + k];
            if (!(dkk > 0.0) || isinf(dkk)) { bad = 1; dkk = 1.0; }
            const double inv2 = 1.0 / dkk;
            if (tid == 0) sD[k] = 1.0 / sqrt(dkk);
            for (int i = k + 1 + ty; i < m; i += NY) {
                const double aik = sL[(size_t)i * (i + 1) / 2 + k] * inv2;
                for (int jj = k + 1 + tx; jj <= i; jj += 16)
                    sL[(size_t)i * (i + 1) / 2 + jj] -= aik * sL[(size_t)jj * (jj + 1) / 2 + k];
            }
        }
        // ---- forward: x[i] -= a[i][k] x[k] / a[k][k]  (i > k); then y[k] = x[k] d[k]
        for (int k = 0; k < m; ++k) {
            __syncthreads();
            for (int cq = cx; cq < ZLD; cq += 32) {                // (32 columns per pass: n + 1 may exceed 32 -- round 5 fix, n >= 32)
                const double d = sD[k];
                const double xk = sX[k * ZLD + cq] * d * d;
                for (int i = k + 1 + cy; i < m; i += NC) sX[i * ZLD + cq] -= sL[(size_t)i * (i + 1) / 2 + k] * xk;
            }
        }
        __syncthreads();
        for (int e = tid; e < m * ZLD; e += NT) sX[e] *= sD[e / ZLD];
        // ---- backward: z[k] = v[k] d[k];  v[i] -= L[k][i] z[k] = a[k][i] d[i] z[k]  (i < k)
        for (int k = m - 1; k >= 0; --k) {
            __syncthreads();
            for (int cq = cx; cq < ZLD; cq += 32) {
                const double zk = sX[k * ZLD + cq] * sD[k];
                for (int i = cy; i < k; i += NC) sX[i * ZLD + cq] -= sL[(size_t)k * (k + 1) / 2 + i] * sD[i] * zk;
            }
        }
        __syncthreads();
        for (int e = tid; e < m * ZLD; e += NT) zt[(size_t)j * m * ZLD + e] = sX[e] * sD[e / ZLD];
    }
    __syncthreads();
    return bad;
}
