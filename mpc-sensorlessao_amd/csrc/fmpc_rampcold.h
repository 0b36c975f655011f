// Cold-start Newton step WITH the ramp-rate rows in its Woodbury form (fmpc_ramp_cold, fmpc_kernel_ramp.hip; constants built by
// fmpc_host_build_ramp_cold, fmpc_host.h).  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include "fmpc_device.h"

struct FrColdParams {
    FmpcDevModel M;
    const double* dumin; const double* dumax;
    // constants per (handle, k, ramp bounds): see FmpcRampColdOut
    const double *g0, *Gf, *phib_u, *phib_x, *gbar_u, *gbar_x, *hd, *erb, *cpb, *betab, *Yinv, *G, *Gt, *Xiu0t, *y0c;
    const double *imgBk, *imgBb;        // operand images of B' (m x n, k = state entry) and of B (n x m, k = actuator): fmpc_host_mfma_images      // Gt: G as the packed upper tile triangle of [G | 0] (FR_TIDX order)
    int batch;
    double kbar;
    const double *x0, *x0p, *w, *uprev, *nu0;
    double *zout, *nuout, *u0out;       // zout may be NULL when u0out is given (first moves only)
    int* status; int* iters; double* step; int step_ld;
    double* ws; size_t ws_stride;       // per workgroup: the 16 x 16 tiles of [M | rhs] and the inverse diagonal tiles
};
size_t fmpc_ramp_cold_lds_bytes(int n, int m, int T, int nb);
size_t fmpc_ramp_cold_ws_doubles(int m);
hipError_t fmpc_ramp_cold_prepare(size_t lds_bytes);
hipError_t fmpc_launch_ramp_cold(const FrColdParams& P, int grid, hipStream_t stream);
