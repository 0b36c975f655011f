// Layout constants of the panel path that host-only code shares with the kernels: no HIP types in here, so that the
// host builders (fmpc_host.cpp) also compile with a plain C++ compiler (the sanitizer build of tests/host_san).
#pragma once
#include <stddef.h>
#if defined(__HIPCC__)
#define FMPC_HD __host__ __device__
#else
#define FMPC_HD
#endif

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
FMPC_HD static inline FpVec fp_vec_layout(int nb, int T) {
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
FMPC_HD static inline FdLds fd_lds_layout(int mp) {
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

