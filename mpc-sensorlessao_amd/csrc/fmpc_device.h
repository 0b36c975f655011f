// Device-side view of the shared model and of the per-problem workspace.
// Internal to the library (the public boundary is include/fastmpc.h).
#pragma once
#include <stddef.h>
#include <hip/hip_runtime.h>

struct FmpcDevModel {
    int n, m, T, nb;          // nb = block rows of C / Y = T + (xf ? 1 : 0)
    int has_xf, var2;         // var2: A2 present (VAR(2)); 0 -> Y is block-tridiagonal
    const double* A1;         // row-major n x n           A1[r*n+c]
    const double* A2;
    const double* A1t;        // transposes, row-major     A1t[c*n+r] = A1[r][c]
    const double* A2t;
    const double* Bt;         // m x n                     Bt[c*n+r]  = B[r][c]
    const double* R2;         // 2*diag(R)  (m)            Phi u-block without the barrier term
    const double* Q2;         // 2*diag(Q)  (n)
    const double* Qf2;        // 2*diag(Qf) (n)
    int denseQ;               // Q or Qf not diagonal (generic kernel, workspace instance only): 2Q, 2Qf and their inverses, n x n row-major
    const double* Q2m; const double* Qf2m; const double* Xm; const double* Xfm;
    int denseR;               // R not diagonal (same instance): 2R, m x m row-major
    const double* R2m;
    const double* rl;         // linear cost r (m), q (n), qf (n)
    const double* ql;
    const double* qfl;
    const double* umin;
    const double* umax;
    const double* umid;       // cold start (fast_mpc_init.m:19-20)
    const double* xmid;
    const double* xf;
    const double* Yblk;       // unique iteration-invariant Y blocks, each n x n row-major
    const int* idxD;          // per block row: index into Yblk of the constant part of Y_ii
    const int* idx1;          //                of Y_{i,i+1} (-1: none)
    const int* idx2;          //                of Y_{i,i+2} (-1: none)
};

// Per-workgroup scratch in HBM (doubles).  The factor tiles are written during the forward
// sweep and streamed back once, in reverse, by the backward sweep.
struct FmpcWsLayout {
    size_t b, nu, hess, winv, rdu, rdx, rp, y, dnu, fac, tiles, drs, zt, total;
};

__host__ __device__ static inline FmpcWsLayout fmpc_ws_layout(int n, int m, int T, int nb, bool big = false, bool dense_r = false) {
    FmpcWsLayout L;
    size_t o = 0;
    const size_t nbn = (size_t)nb * n, Tm = (size_t)T * m, Tn = (size_t)T * n;
    L.b = o;    o += nbn;
    L.nu = o;   o += nbn;
    L.hess = o; o += Tm;
    L.winv = o; o += Tm;
    L.rdu = o;  o += Tm;
    L.rdx = o;  o += Tn;
    L.rp = o;   o += nbn;
    L.y = o;    o += nbn;
    L.dnu = o;  o += nbn;
    L.fac = o;  o += (size_t)nb * 3 * n * (n + 1);
    L.tiles = o; if (big) o += 6 * (size_t)n * (n + 1);          // the generic kernel's tiles when they do not fit the LDS (any n)
    // dense R in that instance: the scratch of ft_dense_r (packed triangle, right-hand sides, pivots) and [Rt_j^-1 B' | Rt_j^-1 r_d[u_j]] per stage
    L.drs = o; if (big && dense_r) o += (size_t)m * (m + 1) / 2 + (size_t)m * (n + 2);
    L.zt = o; if (big && dense_r) o += (size_t)T * m * (n + 1);
    L.total = (o + 15) & ~(size_t)15;
    return L;
}

// Extras of the one-wave-per-problem kernel (fmpc_kernel_wave.hip), device pointers.
struct FwModel {
    int mp;                 // m rounded up to a multiple of 4 (MFMA k-steps)
    const double* BtP;      // mp x 33, zero padded:  BtP[c*33 + r] = B[r][c]
    const double* img;      // per unique block: MFMA-layout images (see FwCfg in the kernel file)
    const int* iD;          // per block row: block id of the constant part of Y_ii
    const int* i1;          //                of Y_{i,i+1}   (zero block if none)
    const int* i2;          //                of Y_{i,i+2}   (zero block if none)
};
