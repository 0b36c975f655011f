// Parameter block of the first-move kernel (fmpc_kernel_first.hip).  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include "fmpc_host.h"

struct FmParams {
    int n, m, T, nb, var2, has_xf, step_ld;
    int x0_given;                       // 1: a_k IS x0 and x0_last IS x0_pre (the loop with its estimator: x0 = ad_est, README.md:482-488); x0 / x0_pre are not written
    const double* a_k; const double* x0_last; const double* u1; const double* u2; const double* nu0;   // per realisation (u1, u2, x0_last, nu0 nullable)
    double* x0; double* x0_pre; double* w; double* u0out;
    int* status; int* iters; double* step;
    int* need;                          // per realisation: 1 = not clear-cut, the exact path redoes it
    int* handed;                        // diagnostic counter of the exact path (fmpc_last_dispatch): zeroed here
    const double* bt;                   // m x n: bt[c*n + r] = B[r][c]
    const double* K0t; const double* u0c;               // [4n][m], [m]
    const double* E; const double* e;                   // circulant half of E: [2n + 1][4n] (fmpc_host_build_first_move: Ec), [4n]
    const double* Ep; const double* ep;
    double* forms;                      // diagnostic (nullable): per realisation the bounds the decision used [e2, rp2, rho2]
    const double* m12t;                 // [2n][T n]: columns of M1, then of M2 (closed-loop prediction matrices)
    const double* dx0T;                 // 2 Qf xbar + qf (n): x entries of r_d at the last stage without nu
    double e0, ep0, normE, norme, normEp, normep, rd2_0;
};

// A stretch of the loop in ONE launch (fmpc_first_move_run): per realisation the steps start[p] .. steps-1, until one is not
// clear-cut (stop[p] = that step, steps when none was).
struct FmRun {
    int steps, batch, have_x0_last;
    const int* start; int* stop;
    const double* a;                    // [steps][batch][n]
    const double* nu0;                  // [steps][batch][nb n] or NULL
    double* U0;                         // [steps][batch][m]
    double* X0;                         // [steps][batch][n] or NULL
    const double* ub1; const double* ub2;   // first moves of the two steps before the stretch (nullable)
};

// The stopped realisations of a walk as one compact batch for the one-step call, and its results back (idx: realisation,
// stp: its step; compact arrays [cnt][.]).
struct FmCompact {
    int cnt, n, m, T, nb, batch, have_x0_last;
    const int* idx; const int* stp;
    const double* a; const double* nu0; double* U0; double* X0; const double* ub1; const double* ub2;    // the stretch (FmRun)
    double* x0; double* x0_pre; double* w; int* status; int* iters;                                     // per realisation [batch][.]
    double* ca; double* cnu; double* cu1; double* cu2; double* cu0; double* cx0; double* cx0p; double* cw; int* cst; int* cit;
};
size_t fmpc_compact_doubles(int n, int m, int T, int nb, int cap);
void fmpc_compact_carve(FmCompact& C, double* base, int cap);
hipError_t fmpc_launch_walk_gather(const FmCompact& C, hipStream_t stream);
hipError_t fmpc_launch_walk_scatter(const FmCompact& C, hipStream_t stream);

hipError_t fmpc_launch_first_move(const FmParams& P, int batch, hipStream_t stream);
hipError_t fmpc_launch_first_move_run(const FmParams& P, const FmRun& R, hipStream_t stream);
