// Host-only builders of the library: everything that is computed on the CPU once per handle or per (handle, k) and then
// uploaded -- the iteration-invariant Y blocks with their de-duplication, the panel path's twisted block factorisation in
// long double with its operator images and sweep schedules, and the least-recently-used cache of the one-shot entry.
// No HIP in here: fmpc_api.hip calls these and does the allocation / upload; tests/host_san builds the same file with
// g++ -fsanitize=address,undefined and checks the factorisation against a dense solve (sanitizers on the CPU build).
#pragma once
#include <stddef.h>
#include <string.h>
#include <vector>

#include "fmpc_panel_layout.h"

// out (row-major n x n) += sign * A X B'   with A, B, X row-major n x n (X dense)
void fmpc_host_add_AXBt(std::vector<double>& out, const std::vector<double>& A, const std::vector<double>& X,
                        const std::vector<double>& B, int n, double sign);

// Iteration-invariant blocks of Y = C Phi^-1 C' (SURVEY.md App. A.4), de-duplicated byte for byte:
//   Yd_i = X_{i+1} + [i>=1] A1 X_i A1' + [i>=2] A2 X_{i-1} A2' ,  Y1_i = -X_{i+1} A1' + [i>=1] A1 X_i A2' ,  Y2_i = -X_{i+1} A2'
//   (X_j = (2Q)^-1, X_T = (2Qf)^-1; terminal rows: Yd_T = Xf, Y1_{T-1} = Xf).  idx*[i] = block id or -1 (none).
void fmpc_host_y_blocks(int n, int T, bool var2, bool has_xf, const std::vector<double>& a1, const std::vector<double>& a2,
                        const std::vector<double>& X, const std::vector<double>& Xf,
                        std::vector<std::vector<double>>& blocks, std::vector<int>& idxD, std::vector<int>& idx1, std::vector<int>& idx2);

// Inputs of the panel-path builder: host copies of the model (row-major) and the layout of the constant pool.
struct FmpcPanelIn {
    int n, m, T, nb, mp, var_order;
    size_t pool_doubles, o_simg, o_limg, o_bt, o_aimg, o_vec, o_ucon, o_dz;
    int limg_cap;
    const double *umax, *umin, *umid, *xmid, *R2, *rl, *Q2, *Qf2, *ql, *qfl, *xf;
    const double *bt;                   // bt[c*n + r] = B[r][c]
    const double *a1, *a2;              // n x n row-major
    const double *blocks;               // unique Y blocks, n x n row-major each
    const int *idxD, *idx1, *idx2;
};
struct FmpcPanelOut {
    std::vector<double> pool;           // [simg | limg | btimg | aimg | vec | ucon | dump | dzimg] per the offsets above
    std::vector<int> sched;             // forward schedule, then backward (FP_MAX_STEPS(nb) x FP_STEP_INTS each)
    int nsf, nsb, nimg, valid;
    double rp2c, rd2_0;
};
// Layout of the constant pool of the panel path (offsets in doubles): fills the o_* / pool_doubles / limg_cap fields of L;
// *o_dump = offset of the dump area behind the uploaded constants, *total = doubles to allocate, *dz_len = length of the
// d_z kernel's LDS image.
void fmpc_host_panel_layout(int n, int m, int T, int nb, int mp, FmpcPanelIn& L, size_t* o_dump, size_t* total, int* dz_len);
// Returns 0; Out.valid = 0 when Y is not positive definite at the start point (the exact path reports that) or the
// edges do not fit the image capacity.
int fmpc_host_build_panel(const FmpcPanelIn& In, double k, FmpcPanelOut& Out);

// Key of the one-shot entry's model cache: every model argument byte for byte, NULL and present arguments distinct.
inline void fmpc_host_key_push(std::vector<double>& key, const double* p, size_t cnt) {
    key.push_back(p ? (double)cnt : -1.0);
    if (p) key.insert(key.end(), p, p + cnt);
}

// Least-recently-used cache of a few payloads (device handles in the library, anything in the tests).
template <class H>
struct FmpcLru {
    struct Entry { std::vector<double> key; H h; std::vector<double> ramp; unsigned long long stamp; };
    std::vector<Entry> items;
    unsigned long long clock = 0;
    size_t capacity;
    explicit FmpcLru(size_t cap) : capacity(cap) {}
    Entry* find(const std::vector<double>& key) {
        for (Entry& e : items)
            if (e.key.size() == key.size() && memcmp(e.key.data(), key.data(), key.size() * sizeof(double)) == 0) return &e;
        return nullptr;
    }
    template <class D>
    Entry* insert(std::vector<double>&& key, H h, D destroy) {       // evicts the least recently used entry when full
        if (items.size() >= capacity) {
            size_t old = 0;
            for (size_t i = 1; i < items.size(); ++i) if (items[i].stamp < items[old].stamp) old = i;
            destroy(items[old].h);
            items.erase(items.begin() + old);
        }
        items.push_back(Entry{std::move(key), h, {}, 0});
        return &items.back();
    }
    void touch(Entry* e) { e->stamp = ++clock; }
    template <class D>
    void clear(D destroy) { for (Entry& e : items) destroy(e.h); items.clear(); }
};

// ---- first-move form of the cold-start step (fmpc_kernel_first.hip; closed-loop steps of a few realisations, z_out = NULL)
// With the data d = [x0 ; x0_pre ; B u1 ; B u2] (4 n numbers; w = -M1 B u1 - M2 B u2, README.md:490-497) the new dual
// variable of the full step is nu+ = nuc + J d (fmpc_kernel_inv.hip).  What the caller applies is the first move only:
//     u0 = ubar + wc o (B' nu+_0 - cu) = u0c + K0 d ,            K0 = diag(wc) B' J_0              (m x 4n)
// and the step-length / exit decision (backtracking_inf_newton.m:2-11, inf_newton_solver.m:19-22) needs
//     ||e||^2   = sum_j || hc o wc o (B' nu+_j - cu) ||^2 = d'E d + 2 e'd + e0
//     ||r_p||^2 = sum_i || cp_i - b_i ||^2               = d'Ep d - 2 ep'd + ep0       (b affine in d: fast_mpc_eq_const.m:39-47)
// Matrices are stored [column][row] (consecutive threads = consecutive rows of a product).
#define FM_NC_MAX 108
struct FmpcFirstIn {
    int n, m, T, nb, var2, has_xf;
    const double *bt, *umax, *umin, *umid, *xmid, *R2, *rl, *a1, *a2, *m1, *m2, *xf;
    const double *J;                    // nb n x 4n row-major: columns [x0 | x0_pre | B u1 | B u2]
    const double *nuc;                  // nb n
    double k;
};
struct FmpcFirstOut {
    int nc;                             // 4 n
    std::vector<double> K0t, u0c, E, e, Ep, ep;     // E, Ep: full symmetric nc x nc (checks); the kernel reads the circulant halves:
    std::vector<double> Ec, Epc;                    // [nc/2 + 1][nc]: Ec[j][r] = w_j E[r][(r + j) mod nc]
    double e0, ep0, normE, norme, normEp, normep;
};
void fmpc_host_build_first_move(const FmpcFirstIn& In, FmpcFirstOut& Out);

// ---- affine form of the whole cold-start step without w (fmpc_kernel_affine.hip):  z+ = zc + Kz [x0 ; x0_pre]
// (nu+ = nuc + J d is affine in the data, and so is d_z = -Phi^-1 (r_d + C' nu+) from the fixed cold-start point).
#define FA_KS 14                        // k-steps of 4: 2 n = 54 data columns, column 54 = the constant zc (times 1), column 55 = 0
#define FA_KC (4 * FA_KS)
struct FmpcAffineIn {
    int n, m, T, nb, has_xf, ncJ;       // J: nb n rows x ncJ columns row-major, the first 2 n columns are used
    const double *bt, *umax, *umin, *umid, *xmid, *R2, *rl, *Q2, *Qf2, *ql, *qfl, *a1, *a2;
    const double *J, *nuc;
    double k;
};
struct FmpcAffineOut {
    int rows, tiles;                    // T (n + m); 16-row tiles
    int nu_rows, nu_tiles;              // nb n rows of nu+ = nuc + J d, as further tiles behind those of z
    std::vector<double> Kz;             // rows x FA_KC row-major (checks)
    std::vector<double> img;            // matrix-core operand images: [tile][k-step][lane = 16 (k mod 4) + (row mod 16)], z tiles then nu tiles
};
void fmpc_host_build_affine(const FmpcAffineIn& In, FmpcAffineOut& Out);
// operand images [tile][k-step][lane = 16 (k mod 4) + (row mod 16)] of a rows x cols row-major matrix, ks k-steps of 4 columns
// (rows padded to tiles of 16, columns to 4 ks, with zeros)
void fmpc_host_mfma_images(const double* M, int rows, int cols, int ks, std::vector<double>& img);
// ---- the first-move form as products over many realisations (fmpc_kernel_loopu0.hip): images over d of ks k-steps
//   imgU  [ceil(m / 16)][ks][64]   rows of [K0 | u0c]: column of the constant holds u0c
//   imgE, imgEp [ks / 4][ks][64]   E, Ep (symmetric, 4 n x 4 n) with only the BLOCK-UPPER triangle of 16 x 16 blocks kept
//                                  (blocks above the diagonal doubled, those below dropped) and the linear term (2 e / -2 ep) in
//                                  the column of the constant, whose own row is zero: d'E^ d = d'E d + 2 e'd for d[const] = 1
// Column order of d: fused = false: [x0; x0_pre; B u1; B u2] contiguous (4 n entries), the constant at 4 n;
//                    fused = true : four blocks of ks (27 entries + a zero pad each, ks = 28), the constant in the last pad slot
//                                   (column 4 ks - 1), and imgB [2][ceil(m / 4)][64] = operand images of B (n x m) from bt.
struct FmpcLoopImages { std::vector<double> imgU, imgE, imgEp, imgB; };
void fmpc_host_build_loop_images(const FmpcFirstOut& O, int n, int m, int ks, bool fused, const double* bt, FmpcLoopImages& out);
// operand images of a rows x FA_KC row-major matrix (rows padded to tiles of 16 with zeros)
void fmpc_host_mfma_a_images(const double* M, int rows, std::vector<double>& img);

// ---- cold-start step WITH the ramp-rate rows (VAR_1/fast_mpc_ineq_const.m:58-76; fmpc_ramp_cold in fmpc_kernel_ramp.hip)
// From the mid-box start u_j = ubar, x_j = xbar every ramp slack u_j - u_{j-1} (j >= 1) is zero -- constant -- and only the rows
// of stage 0, u_0 - u_prev, depend on the problem.  So Phi = Phibar + E diag(delta) E' with E the columns of u_0, Phibar constant,
// delta_c = k (1/sr+^2 + 1/sr-^2) of stage 0: the KKT matrix of the first Newton step is a constant matrix Kbar plus a
// rank-m diagonal term, and
//     [d_z ; nu+] = Kbar^-1 f - Kbar^-1 [E; 0] q ,   (diag(1/delta) + G) q = (Kbar^-1 f)_{u_0} ,   G = (Kbar^-1)_{u_0,u_0}
// (Woodbury), f = -[gbar + E rho ; r_p], rho_c = k (1/sr+ - 1/sr-) of stage 0.  Per problem: ONE m x m Cholesky factorisation
// and two passes through constant operators, where the general path factors the dense (T n)^2 Schur complement (SURVEY 8 a6').
// Everything below is constant per (handle, k, du bounds) and built here in long double.  Kbar^-1 is applied through
//     phi = Phibar^-1 f_z ,  nu = Ybar^-1 (C phi - f_nu) ,  z = phi - Phibar^-1 C' nu ,   Ybar = C Phibar^-1 C'
// with Phibar^-1 = per actuator the inverse Gf of its T x T tridiagonal, per state entry 1 / (2Q).
struct FmpcRampColdIn {
    int n, m, T, nb, var2, has_xf;
    const double *bt, *a1, *a2;         // bt[c*n + r] = B[r][c]; a1, a2 row-major n x n
    const double *umax, *umin, *umid, *xmid, *R2, *rl, *Q2, *Qf2, *ql, *qfl, *xf, *dumin, *dumax;
    double k;
};
struct FmpcRampColdOut {
    int valid;                          // 0: Phibar or Ybar not positive definite / not finite (the general path reports that)
    std::vector<double> g0;             // [T][m]      Gf_c[j][0]: column 0 of actuator c's inverse tridiagonal
    std::vector<double> Gf;             // [T*T][m]    Gf_c[i][j] at (i T + j) m + c
    std::vector<double> phib_u, phib_x; // [T][m], [T][n]   Phibar^-1 (-gbar)
    std::vector<double> gbar_u, gbar_x; // [T][m], [T][n]   gradient at the start point without the stage-0 ramp rows and without C' nu
    std::vector<double> hd;             // [T][m]      diagonal of k P'DP without the stage-0 ramp rows
    std::vector<double> erb;            // [m]         k (1/du_max^2 + 1/du_min^2): minus the off-diagonal of k P'DP between stages
    std::vector<double> cpb;            // [nb n]      r_p at the start point for b = 0 (terminal rows: xbar - xf)
    std::vector<double> betab;          // [nb n]      C phibar + cpb
    std::vector<double> Yinv;           // [nb n][ldy] symmetric, rows ldy = nb n rounded up to even apart (pad column zero)
    std::vector<double> G;              // [m][ldg] symmetric, ldg = m rounded up to even (pad column zero)
    std::vector<double> Xiu0t;          // [T n][ldg]  column-major transpose: Xiu0t[col ldg + r] = d (Kbar^-1 f)_{u_0, r} / d bhat_col
    std::vector<double> y0c;            // [m]         (Kbar^-1 (-[gbar ; cpb]))_{u_0}
};
void fmpc_host_build_ramp_cold(const FmpcRampColdIn& In, FmpcRampColdOut& Out);

// ---- estimator (README.md:456-480): ad_est = lsqminnorm(A_s'*A_s, A_s'*(Y_M - b_s)) = G (Y_M - b_s), G = pinv(A_s' A_s) A_s'
// A_s: p x nx COLUMN-major (MATLAB).  G: nx x p row-major.  Eigenvalues of A_s'A_s below nx * eps * max are treated as zero
// (minimum-norm solution, as lsqminnorm).  Returns the numerical rank.
int fmpc_host_estimator_gain(const double* A_s, int p, int nx, std::vector<double>& G);
// DFT factors of the d-point window starting at frequency index first (0-based, of the fftshift-ed len-point transform):
// F[y][j] = exp(-2 pi i (first + j - len/2) (y - len/2) / len), as matrix-core operand images [len/4][2 tiles][re, im][64 lanes]
// (lane = 16 (y mod 4) + (j mod 16); columns j >= d are zero).
void fmpc_host_estimator_dft_images(int len, int d, int first, std::vector<double>& img);
