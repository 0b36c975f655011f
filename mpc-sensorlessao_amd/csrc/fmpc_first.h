// Parameter block of the first-move kernel (fmpc_kernel_first.hip).  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include "fmpc_host.h"

struct FmParams {
    int n, m, T, nb, var2, has_xf, step_ld;
    const double* a_k; const double* x0_last; const double* u1; const double* u2; const double* nu0;   // per realisation (u1, u2, x0_last, nu0 nullable)
    double* x0; double* x0_pre; double* w; double* u0out;
    int* status; int* iters; double* step;
    int* need;                          // per realisation: 1 = not clear-cut, the exact path redoes it
    int* handed;                        // diagnostic counter of the exact path (fmpc_last_dispatch): zeroed here
    const double* bt;                   // m x n: bt[c*n + r] = B[r][c]
    const double* K0t; const double* u0c;               // [4n][m], [m]
    const double* E; const double* e;                   // [4n][4n], [4n]
    const double* Ep; const double* ep;
    const double* m12t;                 // [2n][T n]: columns of M1, then of M2 (closed-loop prediction matrices)
    const double* dx0T;                 // 2 Qf xbar + qf (n): x entries of r_d at the last stage without nu
    double e0, ep0, normE, norme, normEp, normep, rd2_0;
};

hipError_t fmpc_launch_first_move(const FmParams& P, int batch, hipStream_t stream);
