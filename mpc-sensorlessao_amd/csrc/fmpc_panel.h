// Shared between the panel kernel (fmpc_kernel_panel.hip) and the host code that builds its constants
// (fmpc_api.hip).  Internal to the library.
#pragma once
#include <stddef.h>
#include <hip/hip_runtime.h>

#include "fmpc_panel_layout.h"

// One parameter block for both kernels of the panel path.
struct FpParams {
    int m, mp, T, nb, has_xf, var2;
    int batch, npanels, step_ld;
    const double* x0; const double* x0p; const double* w; const double* nu0;
    double* zout; int* status; int* iters; double* step;
    double* nuws;                       // nu+ in panel layout [panel][stage row][16]: written by the panel kernel, read by d_z
    double* nuout;                      // the caller's nu_out (may be NULL): written by d_z
    const double* simg;                 // Linv per stage (standard layout), then -Linv_0 A1, -Linv_0 A2, -Linv_1 A2
    const double* limg;                 // lane-major images: Linv' per stage, zero, edges
    const int* sched_f; const int* sched_b; int nsf, nsb;        // sweep schedules and their step counts
    const double* btimg;                // B' images, [mp/16][7][64]
    const double* aimg;                 // model images
    const double* vec;                  // FpVec
    const double* ucon;                 // [cu | wc | hc | ubar], each mp
    double rd2_0;                       // ||r_d||^2 at nu = 0
    double rp2c;                        // sum_{i>=2} |cp_i|^2: ||r_p||^2 of the stages >= 2 when w = NULL
    double* gate;                       // per problem: ||r_p||^2, lower bound of rho^2   (read by fmpc_newton_wave)
    double* epsp;                       // per (panel, stage, problem): partial ||e||^2   (read by fmpc_newton_wave)
    const double* dzimg;                // LDS image of the d_z kernel: [B' | A1' | A2' | c1 wc hc ubar | xc xc' iq iq' | c2 2R hp hm]
    int dzimg_len;
    double kbar;                        // barrier weight (the d_z variant that evaluates the next exit test needs it)
    double* rnp;                        // per (panel, stage, problem): partial ||r_d(z+, nu+)||^2, barrier terms re-evaluated
    int* handed;                        // number of problems the exact path had to solve (diagnostic), zeroed here
    double* dump;                       // T (n+m) + nb n doubles: target of the lanes beyond the batch
    const double* jimg; const double* nuc; int jks, jksp;        // dense form (fmpc_cold_inv): image of J (jks k-steps, rows padded with zeros to jksp), nu+ at d = 0
    const double* gw; int gwn;                                   // the data behind [x0 ; x0_pre ; 0 0] in d and its row length: w (T n), or
                                                                 // [B u1 ; B u2] (2 n) with the image J' = [J_x | -J_w M1 | -J_w M2] (fmpc_loop_step_device)
    const double* jst; const double* nucst;                      // per stage: image of the rows of J_x (2 row tiles x FP_XKS k-steps) and nuc, padded to 32 (fmpc_cold_dz<.., true>)
    int gate_only;                                               // fmpc_cold_inv_rg: only the gate tasks (the dual solve itself is fused into d_z)
    const double* eimg;                                          // image of E = [A1 A2 ; A2 0] (4 row tiles x FP_XKS k-steps): b_0, b_1
    double* u0out;                                               // fmpc_cold_dz<.., .., true>: first moves (m x batch); z is not written
};

// d[kk] of problem p, d = [x0 (27) ; x0_pre (27) ; 0 0 ; w (T n)]: ONE load from a selected address; a missing x0_pre / w
// reads x0 and is zeroed, kk beyond the end reads a finite value (J has a zero column there).  Address and zero flag are
// separate so that a caller can request all its values before it touches the first (a select right behind each load
// makes the compiler wait for every load in turn).
__device__ __forceinline__ const double* fi_addr(const double* x0, const double* x0p, const double* w, size_t p, int kk, int TN) {
    const double* a = x0 + p * FP_N + (kk < FP_N ? kk : FP_N - 1);
    if (x0p) { int kb = kk - FP_N; kb = kb < 0 ? 0 : (kb < FP_N ? kb : FP_N - 1); a = kk >= FP_N ? x0p + p * FP_N + kb : a; }
    if (w) { int wi = kk - 4 * FP_XKS; wi = wi < 0 ? 0 : (wi < TN ? wi : TN - 1); a = kk >= 4 * FP_XKS ? w + p * (size_t)TN + wi : a; }
    return a;
}
__device__ __forceinline__ bool fi_zero(bool has_x0p, bool has_w, int kk) {
    return (kk >= 2 * FP_N && kk < 4 * FP_XKS) || (!has_x0p && kk >= FP_N && kk < 2 * FP_N) || (!has_w && kk >= 4 * FP_XKS);
}

size_t fmpc_panel_lds_bytes(int nb, int mp);
size_t fmpc_panel_lds_used(int nb, int mp, int nsteps);
hipError_t fmpc_panel_prepare(size_t lds_bytes);
hipError_t fmpc_launch_panel(const FpParams& P, int grid, size_t lds_bytes, hipStream_t stream);
int fmpc_inv_variant(int npanels, int has_w, int jks);
hipError_t fmpc_launch_inv(const FpParams& P, hipStream_t stream);
size_t fmpc_dz_lds_bytes(int mp, int next);
hipError_t fmpc_dz_prepare(int mp);
hipError_t fmpc_launch_dz(const FpParams& P, int grid, int next, hipStream_t stream, int fused = 0, int u0only = 0);
