// Shared between the panel kernel (fmpc_kernel_panel.hip) and the host code that builds its constants
// (fmpc_api.hip).  Internal to the library.
#pragma once
#include <stddef.h>
#include <hip/hip_runtime.h>

#define FP_N 27                         // states per stage (the AO configuration)
#define FP_NP 16                        // problems per panel = MFMA N dimension
#define FP_KS 7                         // k-steps of 4 covering 27 (28) entries
#define FP_IMG (2 * FP_KS * 64)         // doubles per A-operand image of a 27 x 27 matrix
// A-operand image of a matrix M (rows x 27): element [(I*7 + ks)*64 + l] = M[16 I + (l & 15)][4 ks + (l >> 4)],
// zero outside the matrix: one coalesced 512-byte load per MFMA operand.  The result register r of lane group
// g = l >> 4 then holds row 16 I + 4 r + g, which is the B-operand layout of the next product.
// The same image is a valid B operand of the TRANSPOSED product (panel registers as A operand, problems as
// rows): lane (g, c) then holds M'[4 ks + g][16 I + c].

// per-stage sweep images, [stage][6][FP_IMG]
#define FP_SIMG_LINV 0                  //  Linv_i
#define FP_SIMG_W1 1                    // -Linv_i U_{i-1,i}'
#define FP_SIMG_W2 2                    // -Linv_i U_{i-2,i}'
#define FP_SIMG_LINVT 3                 //  Linv_i'
#define FP_SIMG_V1 4                    // -Linv_i' U_{i,i+1}
#define FP_SIMG_V2 5                    // -Linv_i' U_{i,i+2}
// model images, [5][FP_IMG]
#define FP_AIMG_A1 0
#define FP_AIMG_A2 1
#define FP_AIMG_A1T 2
#define FP_AIMG_A2T 3
#define FP_AIMG_BBT 4

// row-indexed constants (leading dimension 32), offsets in doubles into FpParams::vec
struct FpVec { int ct, cp, xc, iq, dx0, bcu, total; };
__host__ __device__ static inline FpVec fp_vec_layout(int nb, int T) {
    FpVec v; int o = 0;
    v.ct = o; o += nb * 32;            // rhs_i = ct_i - b_i
    v.cp = o; o += nb * 32;            // r_p,i = cp_i - b_i
    v.xc = o; o += T * 32;             // xbar - (2Q_j)^-1 dx0_j
    v.iq = o; o += T * 32;             // (2Q_j)^-1
    v.dx0 = o; o += T * 32;            // 2Q_j xbar + q_j
    v.bcu = o; o += 32;                // B cu
    v.total = o;
    return v;
}

struct FpParams {
    int m, mp, T, nb, has_xf, var2;
    int batch, npanels, step_ld;
    const double* x0; const double* x0p; const double* w; const double* nu0;
    double* zout; double* nuout; int* status; int* iters; double* step;
    const double* simg;                 // sweep images (k-dependent), nb + 1 stage slots, the last all zero
    const double* btimg;                // B' images, [mp/16][7][64]
    const double* aimg;                 // model images
    const double* vec;                  // FpVec
    const double* ucon;                 // [cu | wc | hc | ubar], each mp
    double rd2_0;                       // ||r_d||^2 at nu = 0
    double sa_cu;                       // |cu|^2
    int* sel; int* sel_count;           // problems handed to the exact path
    int dbg;
    double* dump;                       // T (n+m) + nb n doubles: target of the lanes beyond the batch
};

size_t fmpc_panel_lds_bytes(int nb, int mp);
hipError_t fmpc_panel_prepare(size_t lds_bytes);
hipError_t fmpc_launch_panel(const FpParams& P, int grid, size_t lds_bytes, hipStream_t stream);
