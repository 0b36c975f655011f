// Parameter block of the affine cold-start kernel (fmpc_kernel_affine.hip).  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include "fmpc_host.h"

struct FaParams {
    int n, m, T, nb, has_xf, batch, rows, tiles, step_ld;
    int nu_rows, nu_tiles;              // nu+ as further tiles behind those of z (written when nuout != NULL)
    int ldz;                            // doubles between the z rows of consecutive problems (>= rows; launcher: rows unless set)
    int tiles_used, wgs_per_group;    // set by the launcher (flags: knock-out experiments, FMPC_AFFINE_FLAGS)
    const double* img;                  // [tiles][FA_KS][64]: A-operand images of [Kz | zc | 0]
    const double* imgE; const double* imgEp;               // [4][FA_KS][64]: the (x0, x0_pre) blocks of E, Ep (64 rows, zero padded)
    const double* elin; const double* eplin;               // 2 e, -2 ep (64 entries, zero padded)
    const double* dx0T;                 // 2 Qf xbar + qf (n)
    double e0, ep0, normE, norme, normEp, normep, rd2_0;
    const double* x0; const double* x0p; const double* nu0;
    double* zout; double* nuout; double* u0out; int* status; int* iters; double* step;
    int* need; int* handed;
    int* nflag;                         // += 1 per problem flagged in `need` (a running device counter, never reset: the exact-path launch behind
                                        // this kernel compares it with the count it has dealt with and leaves at once when they agree)
    double* dump;                       // 4096 doubles nobody reads: where lanes without a valid target store (no branch around a store)
};

hipError_t fmpc_launch_affine(FaParams P, int num_cu, hipStream_t stream);
// The same step as TWO chained products per stage (nu+_s = J_s d, u_s = Bw nu+_s; the x rows directly): 20 % fewer matrix
// instructions and one task per wavefront.  z_out required, nu_out not served (the caller takes fmpc_launch_affine then).
