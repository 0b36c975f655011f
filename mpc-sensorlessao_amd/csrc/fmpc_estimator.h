// Parameter block of the estimator kernels (fmpc_kernel_estimator.hip).  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>

#define FE_MAXDIV 3                     // phase diversities per measurement (the reference: zd_list = [-3 0 3])

struct FeParams {
    int len, d, ndiv, nx, batch;
    double scale;                       // dx^4 AU
    const double* scrn;                 // [batch][len x len column-major]
    const double* noise;                // [batch][ndiv d^2] or NULL
    const double* Dre; const double* Dim;   // [ndiv][len x len column-major]: pupil .* exp(1i zd_k W)
    const int* qrange;                  // per row block of 16: [first, last + 1) of the k-steps (4 columns) with a pixel inside the pupil
    const double* Fimg;                 // DFT factors of the window as operand images (fmpc_host_estimator_dft_images)
    const double* G; const double* bs;  // nx x p row-major, p
    double* part;                       // workspace [batch][ndiv][len / 16][2][32][32]
    double* shares;                     // [batch][ndiv][nshare][nx] shares of ad_est per diversity (and window quarter: few screens)
    size_t shares_cap, part_cap;        // doubles allocated behind `shares` / `part`
    int nshare;                         // set by the launcher: 1, or 4 for the few-screen finish
    double* ad_est; double* Yout;       // [batch][nx], [batch][p] or NULL
};

hipError_t fmpc_launch_estimator(const FeParams& P, hipStream_t stream);
