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

// FpParams::simg: Linv_i per stage (standard layout; S1 with w), then -Linv_0 A1, -Linv_0 A2, -Linv_1 A2.
// FpParams::limg: LANE-MAJOR images, FP_IMGL doubles each, element [(I*64 + l)*8 + ks]: the 7 (+1 pad) values of a
// lane are contiguous, 4 x 16-byte loads instead of 7 x 8-byte ones (the sweeps are bound by the ISSUE of their
// operand loads).  Image id i < nb: Linv_i'; id nb: all zero; then one image per edge of the two sweeps.
#define FP_IMGL (2 * 64 * 8)
// Dense form of the same dual solve (fmpc_kernel_inv.hip): nu+ = nuc + J d with d = [x0 ; x0_pre ; 0 0 ; w].
// FpParams::jimg: A-operand image of J by 16-row tile, element [(rt * jksp + ks) * 64 + l] = J[16 rt + (l & 15)][4 ks + (l >> 4)],
// jks = FP_XKS + ceil(T n / 4) k-steps, rows padded with zero k-steps to jksp = 16 (ceil(jks / 16) + 3): the k groups of a workgroup may run past the end; the first FP_XKS k-steps are the columns of [x0 ; x0_pre ; 0 0].
#define FP_XKS 14
// Sweep schedules (host: fmpc_upload_panel).  The block Cholesky factor of Y is computed on the host in an
// elimination order chosen for a short dependency chain; a sweep is then a list of EDGES  y_tgt += IMG y_src
// executed in steps with one workgroup barrier per step.  Per step and wave (8 waves; wave 2p + I does row
// block I of the p-th target of the step): one entry {target stage, source stage, image id} (id < 0: none).
#define FP_STEP_INTS (8 * 3)
#define FP_MAX_STEPS(nb) (2 * (nb) + 4)
// model images, [5][FP_IMG]
#define FP_AIMG_A1 0
#define FP_AIMG_A2 1
#define FP_AIMG_A1T 2
#define FP_AIMG_A2T 3
#define FP_AIMG_BBT 4

// row-indexed constants (leading dimension 32), offsets in doubles into FpParams::vec
struct FpVec { int ct, cp, xc, iq, dx0, bcu, rt, total; };
__host__ __device__ static inline FpVec fp_vec_layout(int nb, int T) {
    FpVec v; int o = 0;
    v.ct = o; o += nb * 32;            // rhs_i = ct_i - b_i
    v.cp = o; o += nb * 32;            // r_p,i = cp_i - b_i
    v.xc = o; o += T * 32;             // xbar - (2Q_j)^-1 dx0_j
    v.iq = o; o += T * 32;             // (2Q_j)^-1
    v.dx0 = o; o += T * 32;            // 2Q_j xbar + q_j
    v.bcu = o; o += 32;                // B cu
    v.rt = o; o += nb * 32;            // Linv_i ct_i
    v.total = o;
    return v;
}

// LDS map of the d_z kernel (doubles); the host packs FpParams::dzimg in exactly this order
struct FdLds { int BT, A1T, A2T, UC, XQ, total, UX, total_next; };
#define FD_WAVES 8                      // wavefronts (tasks) per d_z workgroup
#define FD_SCR (FP_N * FP_NP)           // per wave: nu+_j as [row][problem] (swizzled) for the transposed read-back
__host__ __device__ static inline FdLds fd_lds_layout(int mp) {
    FdLds L; int o = 0;
    L.BT = o;  o += (mp / 16) * FP_KS * 64;         // B' images
    L.A1T = o; o += FP_IMG;
    L.A2T = o; o += FP_IMG;
    L.UC = o;  o += 4 * mp;                         // [c1 | wc | hc | ubar], c1 = -wc cu
    L.XQ = o;  o += 4 * 32;                         // [xc | xc(last stage) | iq | iq(last stage)]
    L.total = o;                                    // what the plain kernel copies of FpParams::dzimg
    L.UX = o;  o += 4 * mp;                         // [c2 | 2R | hp | hm]: only the variant that also evaluates the NEXT
    L.total_next = o;                               //   exit test (Newton budgets > 1) copies and uses these
    return L;                                       // (behind it: FD_WAVES x FD_SCR doubles of per-wave scratch)
}

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
