// Parameter block of the many-realisation first-move kernel (fmpc_kernel_loopu0.hip).  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>

#define FL_KS 28                        // k-steps of 4: d = [x0 ; x0_pre ; B u1 ; B u2] (4 n = 108), column 108 = the constant 1, then zeros

struct FlParams {
    int n, m, T, nb, has_xf, var2, batch, step_ld;
    const double* x0; const double* x0_pre; const double* v;     // per realisation: n, n, 2 n (the loop-input kernel's outputs)
    const double* nu0;
    double* u0out; int* status; int* iters; double* step;
    double* x0w; double* x0pw;          // fused step (fmpc_loop_step27): x0, x0_pre are OUTPUTS of the launch (x0w may alias the input x0_last)
    int* need; int* handed;
    const double* imgU;                 // [m / 16][FL_KS][64]: operand images of [K0 | u0c | 0 0 0]
    const double* imgE; const double* imgEp;    // [7][FL_KS][64]: E, Ep (4 n x 4 n, zero padded to 112 x 112)
    const double* dx0T;
    double e0, ep0, normE, norme, normEp, normep, rd2_0;
};

hipError_t fmpc_launch_loop_u0(const FlParams& P, hipStream_t stream);

// The fused closed-loop step (fmpc_loop_step27): loop inputs + first moves + forms in one launch.  P.imgU / imgE / imgEp are
// then the images in the fused column order (blocks of 28: 27 entries + a pad, the constant in column 111).
struct FlStepIn {
    int rs, rows;                       // w: rs slices of `rows` <= 64 rows
    int x0_given;                       // 1: a IS x0, x0_last IS x0_pre (the loop with its estimator); x0 / x0_pre are not written
    const double* imgB;                 // [2][ceil(m / 4)][64]: operand images of B (n x m)
    const double* M1; const double* M2; // T n x n row-major
    const double* a; const double* x0_last; const double* u1; const double* u2;
    double* w;
};
hipError_t fmpc_launch_loop_step27(const FlParams& P, const FlStepIn& I, hipStream_t stream);
