// Generic fastMPC Newton kernel for gfx950: any (n <= 64, m, T) whose tiles fit the LDS -- and, as the instance BIG, ANY (n, m, T)
// with diagonal weights (the reference checks shapes only, fast_mpc_objective.m:17-47): the same code with B' read from the model
// and the six n x (n + 1) tiles in the workgroup's slot of the HBM workspace (L2-resident at the sizes in question) instead of LDS.
// BIG is the size fallback behind the tiled kernel (n <= 79): correctness at any size, not speed.
//
// One 256-thread workgroup owns one problem at a time and runs the whole
// `inf_newton_solver` loop for it (reference: Fast_MPC/VAR_2/inf_newton_solver.m:10-41):
//   P1  residuals r_d, r_p with the barrier terms        inf_newton_solver.m:11-17,
//                                                         inf_newton_KKT_H.m:3-13
//   P2  rhs = r_p - C Phi^-1 r_d                          inf_newton_solver.m:28-29
//   P3  block-penta-diagonal Cholesky of Y = C Phi^-1 C' fused with the forward sweep
//                                                         inf_newton_solver.m:27,30-31
//   P4  backward sweep -> d_nu                            inf_newton_solver.m:32
//   P5  d_z = Phi^-1(-r_d - C' d_nu), line search, update inf_newton_solver.m:34-38,
//                                                         backtracking_inf_newton.m:2-11
// Phi is block diagonal (diagonal for diagonal Q, R) and is never formed; Y is handled as
// n x n blocks; the factor tiles are streamed to an HBM workspace in the forward sweep and
// read back once in the backward sweep.  Matrix conventions: SURVEY.md App. A.4, in the
// transposed ("U") form  U_{i,i+1} = L_ii^-1 (Y_{i,i+1} - U_{i-1,i}' U_{i-1,i+1}),
// U_{i,i+2} = L_ii^-1 Y_{i,i+2},  S_ii = Y_ii - U_{i-1,i}' U_{i-1,i} - U_{i-2,i}' U_{i-2,i}.
// Every LDS tile is n x (n+1): column n carries the right-hand side / y_i so that the forward
// substitution rides along with the block operations.
#include <hip/hip_runtime.h>
#include <math.h>
#include "fmpc_device.h"
#include "fmpc_dense_r.h"
#include "../../include/fastmpc.h"

#define FMPC_THREADS 256
#define FMPC_MAX_HALVINGS 64

__device__ __forceinline__ double fmpc_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Sum over the workgroup, result to every thread; fixed order -> bitwise reproducible.
__device__ __forceinline__ double fmpc_block_sum(double v, double* red) {
    v = fmpc_wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < (FMPC_THREADS >> 6); ++i) s += red[i];
    return s;
}

template <bool BIG>
__global__ void __launch_bounds__(FMPC_THREADS)
fmpc_newton_generic(FmpcDevModel M, int batch,
                    const double* __restrict__ x0, const double* __restrict__ x0p,
                    const double* __restrict__ w, const double* zinit,
                    const double* __restrict__ nu0, int max_iter, double kbar,
                    double* zout, double* __restrict__ nuout,
                    int* __restrict__ status, int* __restrict__ iters,
                    double* __restrict__ step, int step_ld,
                    double* __restrict__ ws, size_t ws_stride) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = M.n, m = M.m, T = M.T, nb = M.nb;
    const int s = n + m, Nz = T * s, nbn = nb * n, ldt = n + 1, tsz = n * ldt;
    const int tid = threadIdx.x;
    const bool var2 = M.var2 != 0;

    const FmpcWsLayout L = fmpc_ws_layout(n, m, T, nb, BIG, BIG && M.denseR != 0);
    const bool DR = BIG && M.denseR != 0;       // dense R: Rt_j^-1 applied through zt (ft_dense_r), not through winv
    const int ZLD = n + 1;
    double* wsp = ws + (size_t)blockIdx.x * ws_stride;
    // ---- LDS carve (BIG: B' stays in the model, the tiles live in the workspace slot)
    const double* sBt = BIG ? M.Bt : lds;                                 // m*n
    double* tile0 = BIG ? wsp + L.tiles : lds + (size_t)m * n;            // 6 tiles n*ldt
    double* sw = BIG ? lds : tile0 + 6 * tsz;                             // m     winv of the current stage
    double* sv1 = sw + m;                 // n     d_nu_{i+1}
    double* sv2 = sv1 + n;                // n     d_nu_{i+2}
    double* srs = sv2 + n;                // n     1/sqrt(pivot)
    double* red = srs + n;                // 8
    int* sflag = (int*)(red + 8);         // 2

    if (!BIG)
        for (int i = tid; i < m * n; i += FMPC_THREADS) lds[i] = M.Bt[i];

    double* b = wsp + L.b;
    double* nu = wsp + L.nu;
    double* hess = wsp + L.hess;
    double* winv = wsp + L.winv;
    double* rdu = wsp + L.rdu;
    double* rdx = wsp + L.rdx;
    double* rp = wsp + L.rp;
    double* y = wsp + L.y;
    double* dnu = wsp + L.dnu;
    double* fac = wsp + L.fac;
    double* zt = wsp + L.zt;

    for (int p = blockIdx.x; p < batch; p += gridDim.x) {
        double* zp = zout + (size_t)p * Nz;
        const double* x0v = x0 + (size_t)p * n;
        const double* x0pv = x0p ? x0p + (size_t)p * n : nullptr;
        __syncthreads();
        // ================= P0: start point, nu, b  (fast_mpc_init.m:12-27,
        //                  fast_mpc_eq_const.m:39,44,47,68)
        for (int idx = tid; idx < Nz; idx += FMPC_THREADS) {
            const int e = idx % s;
            zp[idx] = zinit ? zinit[(size_t)p * Nz + idx] : (e < m ? M.umid[e] : M.xmid[e - m]);
        }
        for (int idx = tid; idx < nbn; idx += FMPC_THREADS) {
            nu[idx] = nu0 ? nu0[(size_t)p * nbn + idx] : 0.0;
            const int i = idx / n, r = idx - i * n;
            double v = (i < T && w) ? w[(size_t)p * T * n + idx] : 0.0;
            if (i == 0) {
                for (int c = 0; c < n; ++c) v += M.A1t[c * n + r] * x0v[c];
                if (var2 && x0pv)
                    for (int c = 0; c < n; ++c) v += M.A2t[c * n + r] * x0pv[c];
            } else if (i == 1 && i < T && var2) {
                for (int c = 0; c < n; ++c) v += M.A2t[c * n + r] * x0v[c];
            }
            if (i == T) v = M.xf[r];
            b[idx] = v;
        }
        if (step)
            for (int idx = tid; idx < step_ld; idx += FMPC_THREADS)
                step[(size_t)p * step_ld + idx] = -1.0;
        __syncthreads();

        int st = FMPC_OK, nsteps = 0;
        for (int it = 0; it < max_iter; ++it) {
            // ================= P1: residuals
            double acc_d = 0.0, acc_p = 0.0;
            int bad = 0;
            for (int idx = tid; idx < T * m; idx += FMPC_THREADS) {
                const int j = idx / m, c = idx - j * m;
                const double u = zp[j * s + c];
                const double dp = 1.0 / (M.umax[c] - u), dm = 1.0 / (u - M.umin[c]);
                const double hs = kbar * (dp * dp + dm * dm);
                const double rt = (DR ? 0.0 : M.R2[c]) + hs;
                if (DR ? (!(hs >= 0.0) || isinf(hs)) : (!(rt > 0.0) || isinf(rt))) bad = 1;
                double dot = 0.0;
                const double* bt = sBt + c * n;
                const double* nj = nu + j * n;
                for (int r = 0; r < n; ++r) dot += bt[r] * nj[r];
                double ru2;                                             // (2R u_j)_c
                if (DR) {
                    const double* rr = M.R2m + (size_t)c * m;
                    const double* uj = zp + j * s;
                    ru2 = 0.0;
                    for (int q = 0; q < m; ++q) ru2 += rr[q] * uj[q];
                } else {
                    ru2 = M.R2[c] * u;
                }
                const double rd = ru2 + M.rl[c] + kbar * (dp - dm) - dot;
                hess[idx] = hs;
                winv[idx] = DR ? 0.0 : 1.0 / rt;
                rdu[idx] = rd;
                acc_d += rd * rd;
            }
            for (int idx = tid; idx < T * n; idx += FMPC_THREADS) {
                const int jj = idx / n, r = idx - jj * n, j = jj + 1;   // x_j, j = 1..T
                const double x = zp[jj * s + m + r];
                double v;
                if (BIG && M.denseQ) {                                  // dense Q, Qf (fast_mpc_objective.m:52-55): (2Q x_j)_r
                    const double* qr = (j == T ? M.Qf2m : M.Q2m) + (size_t)r * n;
                    const double* xj = zp + jj * s + m;
                    double t = 0.0;
                    for (int c = 0; c < n; ++c) t += qr[c] * xj[c];
                    v = t + (j == T ? M.qfl[r] : M.ql[r]) + nu[jj * n + r];
                } else {
                    v = (j == T ? M.Qf2[r] * x + M.qfl[r] : M.Q2[r] * x + M.ql[r]) + nu[jj * n + r];
                }
                if (j < T) {
                    const double* nj = nu + j * n;
                    for (int c = 0; c < n; ++c) v -= M.A1[c * n + r] * nj[c];
                }
                if (var2 && j + 1 < T) {
                    const double* nj = nu + (j + 1) * n;
                    for (int c = 0; c < n; ++c) v -= M.A2[c * n + r] * nj[c];
                }
                if (j == T && M.has_xf) v += nu[T * n + r];
                rdx[idx] = v;
                if (!(BIG && M.denseQ)) dnu[idx] = v / (j == T ? M.Qf2[r] : M.Q2[r]);   // Phi^-1 r_d on x_j (temp)
                acc_d += v * v;
            }
            if (BIG && M.denseQ) {
                __syncthreads();
                for (int idx = tid; idx < T * n; idx += FMPC_THREADS) {
                    const int jj = idx / n, r = idx - jj * n;
                    const double* xr = (jj + 1 == T ? M.Xfm : M.Xm) + (size_t)r * n;
                    const double* vj = rdx + jj * n;
                    double t = 0.0;
                    for (int c = 0; c < n; ++c) t += xr[c] * vj[c];
                    dnu[idx] = t;
                }
            }
            for (int idx = tid; idx < nbn; idx += FMPC_THREADS) {
                const int i = idx / n, r = idx - i * n;
                double v;
                if (i < T) {
                    v = zp[i * s + m + r] - b[idx];
                    const double* ui = zp + i * s;
                    for (int c = 0; c < m; ++c) v -= sBt[c * n + r] * ui[c];
                    if (i >= 1) {
                        const double* xi = zp + (i - 1) * s + m;
                        for (int c = 0; c < n; ++c) v -= M.A1t[c * n + r] * xi[c];
                    }
                    if (var2 && i >= 2) {
                        const double* xi = zp + (i - 2) * s + m;
                        for (int c = 0; c < n; ++c) v -= M.A2t[c * n + r] * xi[c];
                    }
                } else {
                    v = zp[(T - 1) * s + m + r] - b[idx];
                }
                rp[idx] = v;
                acc_p += v * v;
            }
            const double rp2 = fmpc_block_sum(acc_p, red);
            const double rho2 = fmpc_block_sum(acc_d, red) + rp2;
            const double badsum = fmpc_block_sum((double)bad, red);
            // early exit, tested before the step (inf_newton_solver.m:19-22)
            if (sqrt(rho2) <= 1e-6 && sqrt(rp2) <= 1e-8) break;
            if (badsum > 0.0) { st = FMPC_E_NOT_PD_PHI; break; }
            if (DR) {
                // dense R (fast_mpc_objective.m:51-54): factor Rt_j = 2R + k diag(..) of every stage, zt[j] = [Rt_j^-1 B' | Rt_j^-1 r_d[u_j]]
                __syncthreads();
                if (ft_dense_r(wsp + L.drs, M.R2m, m, M.Bt, hess, rdu, zt, n, m, T)) { st = FMPC_E_NOT_PD_PHI; break; }
            }

            // ================= P2: rhs_i = r_p,i - (C Phi^-1 r_d)_i   (into y)
            for (int idx = tid; idx < nbn; idx += FMPC_THREADS) {
                const int i = idx / n, r = idx - i * n;
                double cv;
                if (i < T) {
                    const double* phx = dnu;                    // written in P1
                    cv = phx[i * n + r];
                    const double* ru = rdu + i * m;
                    const double* wi = winv + i * m;
                    if (DR) { const double* zj = zt + (size_t)i * m * ZLD + n; for (int c = 0; c < m; ++c) cv -= sBt[c * n + r] * zj[(size_t)c * ZLD]; }
                    else for (int c = 0; c < m; ++c) cv -= sBt[c * n + r] * (ru[c] * wi[c]);
                    if (i >= 1) {
                        const double* px = phx + (i - 1) * n;
                        for (int c = 0; c < n; ++c) cv -= M.A1t[c * n + r] * px[c];
                    }
                    if (var2 && i >= 2) {
                        const double* px = phx + (i - 2) * n;
                        for (int c = 0; c < n; ++c) cv -= M.A2t[c * n + r] * px[c];
                    }
                } else {
                    cv = dnu[(T - 1) * n + r];
                }
                y[idx] = rp[idx] - cv;
            }
            __syncthreads();

            // ================= P3: factor + forward sweep
            double* tS = tile0;
            double* tM1 = tile0 + tsz;
            double* tM2 = tile0 + 2 * tsz;
            double* tUa = tile0 + 3 * tsz;   // U_{i-1,i}   (col n: y_{i-1})
            double* tUb = tile0 + 4 * tsz;   // U_{i-1,i+1}
            double* tUc = tile0 + 5 * tsz;   // U_{i-2,i}   (col n: y_{i-2})
            bool vA = false, vB = false, vC = false;
            bool fail = false;
            for (int i = 0; i < nb; ++i) {
                const bool hasB = i < T;
                const bool has1 = M.idx1[i] >= 0;     // a block row i+1 exists
                const bool has2 = M.idx2[i] >= 0;
                if (hasB)
                    for (int c = tid; c < m; c += FMPC_THREADS) sw[c] = winv[i * m + c];
                __syncthreads();
                // ---- S (lower triangle) and its rhs column
                {
                    const double* Yd = M.Yblk + (size_t)M.idxD[i] * n * n;
                    const int ntri = n * (n + 1) / 2;
                    for (int idx = tid; idx < ntri + n; idx += FMPC_THREADS) {
                        if (idx < ntri) {
                            int a = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
                            while (a * (a + 1) / 2 > idx) --a;
                            while ((a + 1) * (a + 2) / 2 <= idx) ++a;
                            const int bb = idx - a * (a + 1) / 2;
                            double acc = Yd[a * n + bb];
                            if (hasB) {
                                double t = 0.0;
                                if (DR) { const double* zi_ = zt + (size_t)i * m * ZLD + bb; for (int c = 0; c < m; ++c) t += sBt[c * n + a] * zi_[(size_t)c * ZLD]; }
                                else for (int c = 0; c < m; ++c) t += sBt[c * n + a] * sw[c] * sBt[c * n + bb];
                                acc += t;
                            }
                            if (vA) {
                                double t = 0.0;
                                for (int k = 0; k < n; ++k) t += tUa[k * ldt + a] * tUa[k * ldt + bb];
                                acc -= t;
                            }
                            if (vC) {
                                double t = 0.0;
                                for (int k = 0; k < n; ++k) t += tUc[k * ldt + a] * tUc[k * ldt + bb];
                                acc -= t;
                            }
                            tS[a * ldt + bb] = acc;
                        } else {
                            const int a = idx - ntri;
                            double acc = y[i * n + a];
                            if (vA)
                                for (int k = 0; k < n; ++k) acc -= tUa[k * ldt + a] * tUa[k * ldt + n];
                            if (vC)
                                for (int k = 0; k < n; ++k) acc -= tUc[k * ldt + a] * tUc[k * ldt + n];
                            tS[a * ldt + n] = acc;
                        }
                    }
                }
                // ---- M1 = Y_{i,i+1} - Ua' Ub ; M2 = Y_{i,i+2}
                if (has1) {
                    const double* Y1 = M.Yblk + (size_t)M.idx1[i] * n * n;
                    const bool upd = vA && vB;
                    for (int idx = tid; idx < n * n; idx += FMPC_THREADS) {
                        const int a = idx / n, bb = idx - a * n;
                        double acc = Y1[idx];
                        if (upd) {
                            double t = 0.0;
                            for (int k = 0; k < n; ++k) t += tUa[k * ldt + a] * tUb[k * ldt + bb];
                            acc -= t;
                        }
                        tM1[a * ldt + bb] = acc;
                    }
                }
                if (has2) {
                    const double* Y2 = M.Yblk + (size_t)M.idx2[i] * n * n;
                    for (int idx = tid; idx < n * n; idx += FMPC_THREADS) {
                        const int a = idx / n, bb = idx - a * n;
                        tM2[a * ldt + bb] = Y2[idx];
                    }
                }
                if (tid == 0) sflag[0] = 0;
                __syncthreads();
                // ---- potrf(S), right-looking, one barrier per column; scaling deferred
                for (int k = 0; k < n; ++k) {
                    const double piv = tS[k * ldt + k];
                    if (!(piv > 0.0) || isinf(piv)) { fail = true; break; }   // uniform
                    const double ip = 1.0 / piv;
                    if (tid == 0) srs[k] = 1.0 / sqrt(piv);
                    const int rem = n - k - 1;
                    for (int idx = tid; idx < rem * rem; idx += FMPC_THREADS) {
                        const int r = k + 1 + idx / rem, c = k + 1 + idx % rem;
                        if (c <= r) tS[r * ldt + c] -= tS[r * ldt + k] * tS[c * ldt + k] * ip;
                    }
                    __syncthreads();
                }
                if (fail) break;
                for (int idx = tid; idx < n * n; idx += FMPC_THREADS) {
                    const int r = idx / n, c = idx - r * n;
                    if (c <= r) tS[r * ldt + c] *= srs[c];     // L[r][c] = S[r][c]/sqrt(p_c)
                }
                __syncthreads();
                // ---- [U1 | U2 | y_i] = L^-1 [M1 | M2 | s] : one thread per column
                {
                    const int ncol = 2 * n + 1;
                    for (int cc = tid; cc < ncol; cc += FMPC_THREADS) {
                        double* X;
                        int col;
                        if (cc < n) { if (!has1) continue; X = tM1; col = cc; }
                        else if (cc < 2 * n) { if (!has2) continue; X = tM2; col = cc - n; }
                        else { X = tS; col = n; }
                        for (int r = 0; r < n; ++r) {
                            double v = X[r * ldt + col];
                            for (int j = 0; j < r; ++j) v -= tS[r * ldt + j] * X[j * ldt + col];
                            X[r * ldt + col] = v / tS[r * ldt + r];
                        }
                    }
                }
                __syncthreads();
                // ---- y_i into the rhs columns of U1/U2 and to HBM; factor tiles to HBM
                for (int r = tid; r < n; r += FMPC_THREADS) {
                    const double yi = tS[r * ldt + n];
                    y[i * n + r] = yi;
                    tM1[r * ldt + n] = yi;
                    tM2[r * ldt + n] = yi;
                }
                {
                    double* f = fac + (size_t)i * 3 * tsz;
                    for (int idx = tid; idx < tsz; idx += FMPC_THREADS) {
                        f[idx] = tS[idx];
                        if (has1) f[tsz + idx] = tM1[idx];
                        if (has2) f[2 * tsz + idx] = tM2[idx];
                    }
                }
                // ---- rotate: Ua <- U1, Ub <- U2, Uc <- old Ub
                double* oUa = tUa; double* oUc = tUc;
                tUc = tUb; vC = vB;
                tUa = tM1; vA = has1;
                tUb = tM2; vB = has2;
                tM1 = oUa; tM2 = oUc;
                __syncthreads();
            }
            if (fail) { st = FMPC_E_NOT_PD_SCHUR; break; }

            // ================= P4: backward sweep, d_nu_i = L^-T (y_i - U1 d_nu_{i+1} - U2 d_nu_{i+2})
            {
                double* tL = tile0;
                double* tU1 = tile0 + tsz;
                double* tU2 = tile0 + 2 * tsz;
                double* tv = tile0 + 3 * tsz;     // n: rhs of the triangular solve
                for (int i = nb - 1; i >= 0; --i) {
                    const bool has1 = M.idx1[i] >= 0, has2 = M.idx2[i] >= 0;
                    const double* f = fac + (size_t)i * 3 * tsz;
                    for (int idx = tid; idx < tsz; idx += FMPC_THREADS) {
                        tL[idx] = f[idx];
                        if (has1) tU1[idx] = f[tsz + idx];
                        if (has2) tU2[idx] = f[2 * tsz + idx];
                    }
                    __syncthreads();
                    for (int r = tid; r < n; r += FMPC_THREADS) {
                        double v = tL[r * ldt + n];                 // y_i
                        if (has1)
                            for (int c = 0; c < n; ++c) v -= tU1[r * ldt + c] * sv1[c];
                        if (has2)
                            for (int c = 0; c < n; ++c) v -= tU2[r * ldt + c] * sv2[c];
                        tv[r] = v;
                    }
                    __syncthreads();
                    if (BIG) {
                        // any n: column-oriented substitution by the whole workgroup, one barrier per row
                        for (int r = n - 1; r >= 0; --r) {
                            const double xr = tv[r] / tL[r * ldt + r];
                            for (int c = tid; c < r; c += FMPC_THREADS) tv[c] -= tL[r * ldt + c] * xr;
                            if (tid == 0) dnu[i * n + r] = xr;
                            __syncthreads();
                        }
                        for (int r = tid; r < n; r += FMPC_THREADS) { sv2[r] = sv1[r]; sv1[r] = dnu[i * n + r]; }
                    } else if (tid < 64) {                           // wave 0: lane j <-> entry j
                        double v = (tid < n) ? tv[tid] : 0.0;
                        for (int r = n - 1; r >= 0; --r) {
                            const double xr = __shfl(v, r, 64) / tL[r * ldt + r];
                            if (tid < r) v -= tL[r * ldt + tid] * xr;
                            else if (tid == r) v = xr;
                        }
                        if (tid < n) {
                            dnu[i * n + tid] = v;
                            sv2[tid] = sv1[tid];
                            sv1[tid] = v;
                        }
                    }
                    __syncthreads();
                }
            }

            // ================= P5: d_z, line-search scalars, update
            double be = 0.0, e2 = 0.0;
            for (int idx = tid; idx < T * m; idx += FMPC_THREADS) {
                const int j = idx / m, c = idx - j * m;
                double dot = 0.0;
                const double* bt = sBt + c * n;
                const double* dj = dnu + j * n;
                for (int r = 0; r < n; ++r) dot += bt[r] * dj[r];
                const double rd = rdu[idx];
                double du;
                if (DR) {                                               // d_u = Rt^-1 (B' d_nu - r_d[u]) = Z d_nu - Rt^-1 r_d[u]
                    const double* zc = zt + ((size_t)j * m + c) * ZLD;
                    du = -zc[n];
                    for (int r = 0; r < n; ++r) du += zc[r] * dj[r];
                } else {
                    du = (dot - rd) * winv[idx];
                }
                const double e = hess[idx] * du;        // k P'DP dz
                be += rd * e;
                e2 += e * e;
                rdu[idx] = du;                          // reuse as d_u
            }
            for (int idx = tid; idx < T * n; idx += FMPC_THREADS) {
                const int jj = idx / n, r = idx - jj * n, j = jj + 1;
                double v = -rdx[idx] - dnu[jj * n + r];
                if (j < T) {
                    const double* dj = dnu + j * n;
                    for (int c = 0; c < n; ++c) v += M.A1[c * n + r] * dj[c];
                }
                if (var2 && j + 1 < T) {
                    const double* dj = dnu + (j + 1) * n;
                    for (int c = 0; c < n; ++c) v += M.A2[c * n + r] * dj[c];
                }
                if (j == T && M.has_xf) v -= dnu[T * n + r];
                if (BIG && M.denseQ) rp[idx] = v;               // (r_p is dead since P2: takes the right-hand side of d_x)
                else rdx[idx] = v / (j == T ? M.Qf2[r] : M.Q2[r]);   // reuse as d_x
            }
            if (BIG && M.denseQ) {
                __syncthreads();
                for (int idx = tid; idx < T * n; idx += FMPC_THREADS) {
                    const int jj = idx / n, r = idx - jj * n;
                    const double* xr = (jj + 1 == T ? M.Xfm : M.Xm) + (size_t)r * n;
                    const double* vj = rp + jj * n;
                    double t = 0.0;
                    for (int c = 0; c < n; ++c) t += xr[c] * vj[c];
                    rdx[idx] = t;
                }
            }
            const double beta_e = fmpc_block_sum(be, red);
            const double eps2 = fmpc_block_sum(e2, red);
            // closed form of backtracking_inf_newton.m:2-11 with the frozen barrier gradient:
            // ||r(t)||^2 - ((1-al t) rho)^2 = t * gq(t)
            double t = 1.0;
            {
                const double al = 1e-4;
                int halv = 0;
                while (true) {
                    const double gq = (t - 2.0 + 2.0 * al - al * al * t) * rho2
                                      - 2.0 * (1.0 - t) * beta_e + t * eps2;
                    if (gq <= 0.0) break;
                    t *= 0.5;
                    if (++halv >= FMPC_MAX_HALVINGS) { t = 0.0; st = FMPC_W_LINESEARCH; break; }
                }
            }
            for (int idx = tid; idx < Nz; idx += FMPC_THREADS) {
                const int j = idx / s, e = idx - j * s;
                zp[idx] += t * (e < m ? rdu[j * m + e] : rdx[j * n + e - m]);
            }
            for (int idx = tid; idx < nbn; idx += FMPC_THREADS) nu[idx] += t * dnu[idx];
            if (step && tid == 0 && it < step_ld) step[(size_t)p * step_ld + it] = t;
            ++nsteps;
            __syncthreads();
        }
        if (nuout)
            for (int idx = tid; idx < nbn; idx += FMPC_THREADS) nuout[(size_t)p * nbn + idx] = nu[idx];
        if (tid == 0) {
            if (status) status[p] = st;
            if (iters) iters[p] = nsteps;
        }
    }
}

// Caller-side unpack (README.md:558-570, :589): z -> U, X, u0.
extern "C" __global__ void __launch_bounds__(256)
fmpc_unpack_kernel(int n, int m, int T, int batch, const double* __restrict__ z,
                   double* __restrict__ U, double* __restrict__ X, double* __restrict__ u0) {
    const int s = n + m;
    const size_t Nz = (size_t)T * s, total = Nz * batch;
    if (!U && !X) {                                   // only the first move: read m entries per problem, not all of z
        if (!u0) return;
        for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < (size_t)batch * m;
             g += (size_t)gridDim.x * blockDim.x) {
            const size_t p = g / m;
            u0[g] = z[p * Nz + (g - p * m)];
        }
        return;
    }
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (size_t)gridDim.x * blockDim.x) {
        const size_t p = g / Nz;
        const int idx = (int)(g - p * Nz);
        const int j = idx / s, e = idx - j * s;
        const double v = z[g];
        if (e < m) {
            if (U) U[p * (size_t)T * m + (size_t)j * m + e] = v;
            if (u0 && j == 0) u0[p * m + e] = v;
        } else if (X) {
            X[p * (size_t)T * n + (size_t)j * n + (e - m)] = v;
        }
    }
}

// Closed-loop inputs (fmpc_loop_inputs_device in include/fastmpc.h; README.md:482-497):
//   x0 = a + B u1 ,  x0_pre = x0_last ,  w = -M1 (B u1) - M2 (B u2)      M1, M2: (T n) x n row-major
// A workgroup takes a tile of LI_PT problems and one of LI_RS row slices of w: B u1, B u2 of its problems go to LDS, then
// every thread keeps one row of M1 and of M2 in registers and applies it to all problems of the tile (a row of M is
// read once per LI_PT problems, not once per problem: the first version, one workgroup per problem, spent 61 us per
// step at 512 problems re-reading the 350 KB of M1, M2).
#define LI_PT 16
#define LI_RS 8
#define LI_NMAX 64
#ifdef FW_TIMING
__device__ unsigned long long li_timing[8];
extern "C" int fmpc_debug_loop_inputs_timing(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(li_timing), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#define LI_TICK(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) li_timing[k] = (unsigned long long)wall_clock64(); } while (0)
#else
#define LI_TICK(k)
#endif
// NC: n at compile time (27: the AO configuration), or 0 = any n <= LI_NMAX (arrays sized LI_NMAX, predicated loops)
template <int NC>
__global__ void __launch_bounds__(256)
fmpc_loop_inputs_kernel(int n_rt, int m, int T, int batch, const double* __restrict__ Bt, const double* __restrict__ M1,
                        const double* __restrict__ M2, const double* __restrict__ a, const double* x0_last,
                        const double* __restrict__ u1, const double* __restrict__ u2,
                        double* x0, double* __restrict__ x0_pre, double* __restrict__ w) {
    extern __shared__ double sh[];      // bu1[LI_PT][n], bu2[LI_PT][n], this row slice of M1 and M2, B' (m x n), u1 and u2 of the tile
    const int n = NC ? NC : n_rt;
    constexpr int NQ = NC ? NC : LI_NMAX;
    double* bu1 = sh; double* bu2 = sh + LI_PT * n;
    const int rows_ = (T * n + LI_RS - 1) / LI_RS;
    double* sBt = sh + 2 * LI_PT * n + 2 * (size_t)rows_ * (n + 1);
    double* su = sBt + (size_t)m * n;
    const int p0 = blockIdx.x * LI_PT, np = batch - p0 < LI_PT ? batch - p0 : LI_PT;
    const int tid = threadIdx.x;
    // everything the products B u need goes to LDS first with independent, coalesced loads (a dependent chain of global
    // loads per output cost 40 us here)
    LI_TICK(0);
    // (copy loops in batches of 8 loads before the 8 LDS stores: left to itself the compiler waits for every load)
    for (int base = 0; base < m * n; base += 8 * 256) {
        double t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int idx = base + k * 256 + tid; t[k] = Bt[idx < m * n ? idx : 0]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int idx = base + k * 256 + tid; if (idx < m * n) sBt[idx] = t[k]; }
    }
    for (int which = 0; which < 2; ++which) {
        const double* u = which ? u2 : u1;
        const int len = np * m;                          // the tile's rows of u are contiguous
        for (int base = 0; base < LI_PT * m; base += 8 * 256) {
            double t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int idx = base + k * 256 + tid; t[k] = (u && idx < len) ? u[(size_t)p0 * m + idx] : 0.0; }
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int idx = base + k * 256 + tid; if (idx < LI_PT * m) su[which * LI_PT * m + idx] = t[k]; }
        }
    }
    __syncthreads();
    LI_TICK(1);
    // B u1, B u2: 2 x LI_PT x n outputs, each a sum over the m actuators.  A thread works on its (up to 4) outputs
    // INTERLEAVED -- four independent accumulation chains instead of one dependent chain after the other
    {
        const int nout = 2 * LI_PT * n;
        const double* bp[4]; const double* upp[4]; double acc4[4]; bool act[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + k * 256;
            const int ic = idx < nout ? idx : 0;
            const int which = ic / (LI_PT * n), rem = ic - which * LI_PT * n, pp = rem / n, r = rem - pp * n;
            act[k] = idx < nout && pp < np;
            bp[k] = sBt + r; upp[k] = su + (size_t)(which * LI_PT + pp) * m; acc4[k] = 0.0;
        }
        if (act[0] || act[1] || act[2] || act[3]) {
#pragma unroll 4
            for (int c = 0; c < m; ++c) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc4[k] += bp[k][(size_t)c * n] * upp[k][c];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int idx = tid + k * 256; if (idx < nout) sh[idx] = act[k] ? acc4[k] : 0.0; }
    }
    __syncthreads();
    LI_TICK(2);
    if (blockIdx.y == 0)
        for (int idx = tid; idx < np * n; idx += blockDim.x) {
            const int pp = idx / n, r = idx - pp * n;
            const size_t g = (size_t)(p0 + pp) * n + r;
            const double xl = x0_last ? x0_last[g] : 0.0;        // (x0 may alias x0_last: read before the write below)
            x0_pre[g] = xl;
            x0[g] = a[g] + bu1[pp * n + r];
        }
    // thread = (row lane er = tid & 15, problem pp = tid >> 4): B u1, B u2 of its problem in registers; rows of M1, M2 staged
    // through LDS 16 at a time (coalesced reads), every staged row used by the 16 problems of the tile
    const int Tn = T * n;
    const int rows = (Tn + LI_RS - 1) / LI_RS, e0 = blockIdx.y * rows, e1 = e0 + rows < Tn ? e0 + rows : Tn;
    const int er = tid & 15, pp = tid >> 4;
    double b1[NQ], b2[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        b1[q] = q < n ? bu1[pp * n + q] : 0.0;
        b2[q] = q < n ? bu2[pp * n + q] : 0.0;
    }
    // this workgroup's rows of M1 and M2 into LDS in one go (one memory latency), leading dimension n + 1
    double* t1 = sh + 2 * LI_PT * n;
    double* t2 = t1 + (size_t)rows * (n + 1);
    {
        const int len = (e1 - e0) * n;                   // the slice's rows are contiguous in M1, M2
        for (int base = 0; base < rows * n; base += 4 * 256) {
            double ta[4], tb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = base + k * 256 + tid;
                const bool ok = idx < len;
                ta[k] = ok ? M1[(size_t)e0 * n + idx] : 0.0;
                tb[k] = ok ? M2[(size_t)e0 * n + idx] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = base + k * 256 + tid;
                if (idx < rows * n) { const int rr = idx / n, q = idx - rr * n; t1[rr * (n + 1) + q] = ta[k]; t2[rr * (n + 1) + q] = tb[k]; }
            }
        }
    }
    __syncthreads();
    LI_TICK(3);
    for (int eb = 0; e0 + eb < e1; eb += 16) {
        const int rr = eb + er < rows ? eb + er : 0;
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (q < n) acc -= t1[rr * (n + 1) + q] * b1[q] + t2[rr * (n + 1) + q] * b2[q];
        if (e0 + eb + er < e1 && pp < np) w[(size_t)(p0 + pp) * Tn + e0 + eb + er] = acc;
    }
    LI_TICK(4);
}

// The same for n = 27 on the matrix cores (v_mfma_f64_16x16x4_f64), 16 problems = the N dimension:
//   B u1, B u2 :  rows q of B (2 row tiles) x actuators (k) x problems: result tile register r of lane (lk, li) = (B u)[16 I + 4 r + lk]
//                 of problem li -- which is directly the B operand (k-step 4 I + r) of
//   w          :  rows e of M1 | M2 (A operand straight from L2) x q (7 k-steps each) x problems.
// Every wave computes B u itself (144 MFMAs, all four SIMDs in parallel) and then its own 16-row tiles of the slice.
typedef double li_d4 __attribute__((ext_vector_type(4)));
#define LI_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
// (the second launch bound keeps the register budget at 256: with 512 the compiler puts the MFMA results into AGPRs and
//  copies all accumulators AGPR <-> VGPR around every k-step)
__global__ void __launch_bounds__(256, 2)
fmpc_loop_inputs_mfma27(int m, int T, int batch, int rows, const double* __restrict__ Bt, const double* __restrict__ M1,
                        const double* __restrict__ M2, const double* __restrict__ a, const double* x0_last,
                        const double* __restrict__ u1, const double* __restrict__ u2,
                        double* x0, double* __restrict__ x0_pre, double* __restrict__ w, double* __restrict__ lv) {
    constexpr int n = 27;
    extern __shared__ double sh[];                      // B' (m x n), u1 and u2 of the tile (LI_PT x (m + 1) each)
    double* sBt = sh; double* su = sh + (size_t)m * n;
    const int ldu = m + 1;                              // padded rows of u: the 16 problems of a k-step read different banks
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int p0 = blockIdx.x * LI_PT, np = batch - p0 < LI_PT ? batch - p0 : LI_PT;
    const int Tn = T * n, e0 = blockIdx.y * rows, e1 = e0 + rows < Tn ? e0 + rows : Tn;
    LI_TICK(0);
    // ---- this wave's first row tile of M1, M2 as A operands: requested now, used last
    double am1[7], am2[7];
    {
        const int e = e0 + 16 * wv + li;
        const bool eok = e < e1;
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            const int q = 4 * ks + lk;
            const bool ok = eok && q < n;
            const size_t off = (size_t)(eok ? e : e0) * n + (q < n ? q : 0);
            const double t1 = M1[off], t2 = M2[off];
            am1[ks] = ok ? t1 : 0.0; am2[ks] = ok ? t2 : 0.0;
        }
    }
    // B' and the tile's u1, u2 into LDS: every load of the workgroup is requested before the first LDS write (one round trip
    // instead of one per 2048 entries): 16 + 2 x 9 loads per thread cover m n <= 4096 and 16 m <= 2304 (m = 144); larger
    // sizes take further rounds.
    for (int base = 0; base < m * n; base += 16 * 256) {
        double t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int idx = base + k * 256 + tid; t[k] = Bt[idx < m * n ? idx : 0]; }
        double tu[2][9];
        const int len = np * m;
        if (base == 0) {
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const double* u = which ? u2 : u1;
#pragma unroll
                for (int k = 0; k < 9; ++k) { const int idx = k * 256 + tid; tu[which][k] = (u && idx < len) ? u[(size_t)p0 * m + idx] : 0.0; }
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int idx = base + k * 256 + tid; if (idx < m * n) sBt[idx] = t[k]; }
        if (base == 0) {
#pragma unroll
            for (int which = 0; which < 2; ++which)
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int idx = k * 256 + tid;
                    if (idx < LI_PT * m) { const int pp = idx / m; su[(which * LI_PT + pp) * ldu + idx - pp * m] = tu[which][k]; }
                }
        }
    }
    for (int which = 0; which < 2; ++which) {                  // (m > 144: the rest of u)
        const double* u = which ? u2 : u1;
        const int len = np * m;
        for (int base = 9 * 256; base < LI_PT * m; base += 8 * 256) {
            double t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int idx = base + k * 256 + tid; t[k] = (u && idx < len) ? u[(size_t)p0 * m + idx] : 0.0; }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = base + k * 256 + tid;
                if (idx < LI_PT * m) { const int pp = idx / m; su[(which * LI_PT + pp) * ldu + idx - pp * m] = t[k]; }
            }
        }
    }
    __syncthreads();
    LI_TICK(1);
    // ---- B u1, B u2
    li_d4 bu[2][2];
#pragma unroll
    for (int wh = 0; wh < 2; ++wh)
#pragma unroll
        for (int I = 0; I < 2; ++I) bu[wh][I] = li_d4{0, 0, 0, 0};
    {
        const int q0 = li, q1 = 16 + li < n ? 16 + li : n - 1;
        const bool q1ok = 16 + li < n;
        const double* ua = su + (size_t)li * ldu;
        const double* ub = su + (size_t)(LI_PT + li) * ldu;
        // the four wavefronts split the actuators (k-steps of 4) and add their partial tiles in LDS in a fixed order: a
        // quarter of the dependent products each instead of the same 4 x m / 4 on every wavefront
        const int ksteps = (m + 3) / 4, per = (ksteps + 3) / 4;
        const int k0 = wv * per, k1 = k0 + per < ksteps ? k0 + per : ksteps;
        for (int ks = k0; ks < k1; ++ks) {
            const int c = 4 * ks + lk;
            const bool cok = c < m;
            const int cc = cok ? c : m - 1;
            const double r0 = sBt[(size_t)cc * n + q0], r1 = sBt[(size_t)cc * n + q1];   // unconditional loads, then selects:
            const double b0 = cok ? r0 : 0.0;                                            // (a conditional load becomes a branch
            const double b1 = (cok && q1ok) ? r1 : 0.0;                                  //  and the accumulators bounce AGPR <-> VGPR)
            const double v1 = ua[cc], v2 = ub[cc];
            bu[0][0] = LI_MFMA(b0, v1, bu[0][0]); bu[0][1] = LI_MFMA(b1, v1, bu[0][1]);
            bu[1][0] = LI_MFMA(b0, v2, bu[1][0]); bu[1][1] = LI_MFMA(b1, v2, bu[1][1]);
        }
    }
    __syncthreads();                                    // (u in LDS is dead: its space takes the partial tiles, [wave][tile][reg][lane])
#pragma unroll
    for (int wh = 0; wh < 2; ++wh)
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) su[((wv * 4 + wh * 2 + I) * 4 + r) * 64 + lane] = bu[wh][I][r];
    __syncthreads();
#pragma unroll
    for (int wh = 0; wh < 2; ++wh)
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double sacc = 0.0;
#pragma unroll
                for (int v = 0; v < 4; ++v) sacc += su[((v * 4 + wh * 2 + I) * 4 + r) * 64 + lane];
                bu[wh][I][r] = sacc;
            }
    LI_TICK(2);
    // ---- x0 = a + B u1 ,  x0_pre = x0_last    (one wave of the first row slice)
    if (blockIdx.y == 0 && wv == 0 && li < np) {
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * I + 4 * r + lk;
                if (q < n) {
                    const size_t g = (size_t)(p0 + li) * n + q;
                    const double xl = x0_last ? x0_last[g] : 0.0;    // (x0 may alias x0_last: read before the write below)
                    x0_pre[g] = xl;
                    x0[g] = a[g] + bu[0][I][r];
                    // [B u1 ; B u2]: the 2 n numbers w depends on (fmpc_loop_step_device: the dense form of the dual solve
                    // takes them instead of the T n entries of w)
                    if (lv) { lv[(size_t)(p0 + li) * 2 * n + q] = bu[0][I][r]; lv[(size_t)(p0 + li) * 2 * n + n + q] = bu[1][I][r]; }
                }
            }
    }
    LI_TICK(3);
    // ---- w = -M1 (B u1) - M2 (B u2): this wave's row tiles of the slice
    for (int t = wv; 16 * t < e1 - e0; t += 4) {
        if (t != wv) {                                   // (slices longer than 64 rows: further tiles, loaded here)
            const int e = e0 + 16 * t + li;
            const bool eok = e < e1;
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {
                const int q = 4 * ks + lk;
                const bool ok = eok && q < n;
                const size_t off = (size_t)(eok ? e : e0) * n + (q < n ? q : 0);
                const double t1 = M1[off], t2 = M2[off];
                am1[ks] = ok ? t1 : 0.0; am2[ks] = ok ? t2 : 0.0;
            }
        }
        li_d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            acc = LI_MFMA(am1[ks], bu[0][ks >> 2][ks & 3], acc);
            acc = LI_MFMA(am2[ks], bu[1][ks >> 2][ks & 3], acc);
        }
        if (li < np) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = e0 + 16 * t + 4 * r + lk;
                if (e < e1) w[(size_t)(p0 + li) * Tn + e] = -acc[r];
            }
        }
    }
    LI_TICK(4);
}

// Any size (n > LI_NMAX, or B' and a row slice beyond the LDS): one workgroup per problem, B u1 and B u2 in LDS (2 n doubles), a thread
// per row of w.  The size fallback of the two kernels above -- same outputs, no tiling.
__global__ void __launch_bounds__(256)
fmpc_loop_inputs_any(int n, int m, int T, const double* __restrict__ Bt, const double* __restrict__ M1, const double* __restrict__ M2,
                     const double* __restrict__ a, const double* x0_last, const double* __restrict__ u1, const double* __restrict__ u2,
                     double* x0, double* __restrict__ x0_pre, double* __restrict__ w, double* __restrict__ lv) {
    extern __shared__ double sh[];
    double* bu1 = sh; double* bu2 = sh + n;
    const int p = blockIdx.x, tid = threadIdx.x;
    const double* up1 = u1 ? u1 + (size_t)p * m : nullptr;
    const double* up2 = u2 ? u2 + (size_t)p * m : nullptr;
    for (int r = tid; r < n; r += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int c = 0; c < m; ++c) {
            const double b = Bt[(size_t)c * n + r];
            if (up1) s1 += b * up1[c];
            if (up2) s2 += b * up2[c];
        }
        bu1[r] = s1; bu2[r] = s2;
    }
    __syncthreads();
    for (int r = tid; r < n; r += 256) {
        const size_t g = (size_t)p * n + r;
        const double xl = x0_last ? x0_last[g] : 0.0;            // (x0 may alias x0_last: read before the write below)
        x0_pre[g] = xl;
        x0[g] = a[g] + bu1[r];
        if (lv) { lv[(size_t)p * 2 * n + r] = bu1[r]; lv[(size_t)p * 2 * n + n + r] = bu2[r]; }
    }
    const int Tn = T * n;
    for (int e = tid; e < Tn; e += 256) {
        const double* r1 = M1 + (size_t)e * n; const double* r2 = M2 + (size_t)e * n;
        double acc = 0.0;
        for (int q = 0; q < n; ++q) acc -= r1[q] * bu1[q] + r2[q] * bu2[q];
        w[(size_t)p * Tn + e] = acc;
    }
}

hipError_t fmpc_launch_loop_inputs(int n, int m, int T, int batch, const double* Bt, const double* M1, const double* M2,
                                   const double* a, const double* x0_last, const double* u1, const double* u2,
                                   double* x0, double* x0_pre, double* w, hipStream_t stream, double* lv) {
    const size_t lds_plain = (2 * LI_PT * n + 2 * (size_t)((T * n + LI_RS - 1) / LI_RS) * (n + 1) + (size_t)m * n + 2 * LI_PT * m) * sizeof(double);
    size_t lds27 = 0;
    if (n == 27) { size_t su_d = 2 * (size_t)LI_PT * (m + 1); if (su_d < 4096) su_d = 4096; lds27 = ((size_t)m * n + su_d) * sizeof(double); }
    if (n > LI_NMAX || (n != 27 && lds_plain > 160 * 1024) || (n == 27 && lds27 > 160 * 1024)) {
        if (2 * (size_t)n * sizeof(double) > 64 * 1024) return hipErrorInvalidValue;
        hipLaunchKernelGGL(fmpc_loop_inputs_any, dim3(batch), dim3(256), 2 * (size_t)n * sizeof(double), stream,
                           n, m, T, Bt, M1, M2, a, x0_last, u1, u2, x0, x0_pre, w, lv);
        return hipGetLastError();
    }
    if (n == 27) {
        const int Tn = T * n, rs = (Tn + 63) / 64, rows = (Tn + rs - 1) / rs;      // <= 64 rows = 4 tiles per workgroup
        size_t su_d = 2 * (size_t)LI_PT * (m + 1);                 // u1, u2 of the tile; afterwards the partial tiles of B u (4 x 4 x 256)
        if (su_d < 4096) su_d = 4096;
        const size_t lds = ((size_t)m * n + su_d) * sizeof(double);
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        hipError_t ea = hipFuncSetAttribute((const void*)fmpc_loop_inputs_mfma27, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return ea;
        hipLaunchKernelGGL(fmpc_loop_inputs_mfma27, dim3((batch + LI_PT - 1) / LI_PT, rs), dim3(256), lds, stream,
                           m, T, batch, rows, Bt, M1, M2, a, x0_last, u1, u2, x0, x0_pre, w, lv);
        return hipGetLastError();
    }
    const auto kern = fmpc_loop_inputs_kernel<0>;
    const size_t lds = (2 * LI_PT * n + 2 * (size_t)((T * n + LI_RS - 1) / LI_RS) * (n + 1) + (size_t)m * n + 2 * LI_PT * m) * sizeof(double);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL(kern, dim3((batch + LI_PT - 1) / LI_PT, LI_RS), dim3(256), lds, stream,
                       n, m, T, batch, Bt, M1, M2, a, x0_last, u1, u2, x0, x0_pre, w);
    return hipGetLastError();
}

size_t fmpc_generic_lds_bytes(int n, int m) {
    const size_t d = (size_t)m * n + 6 * (size_t)n * (n + 1) + m + 3 * (size_t)n + 8 + 2;
    return d * sizeof(double);
}
size_t fmpc_generic_big_lds_bytes(int n, int m) { return ((size_t)m + 3 * (size_t)n + 8 + 2) * sizeof(double); }

hipError_t fmpc_launch_generic(const FmpcDevModel& M, int batch, int grid, const double* x0,
                               const double* x0p, const double* w, const double* zinit,
                               const double* nu0, int max_iter, double kbar, double* zout,
                               double* nuout, int* status, int* iters, double* step, int step_ld,
                               double* ws, size_t ws_stride, hipStream_t stream, int big) {
    if (big) {
        hipLaunchKernelGGL(fmpc_newton_generic<true>, dim3(grid), dim3(FMPC_THREADS), fmpc_generic_big_lds_bytes(M.n, M.m), stream, M, batch,
                           x0, x0p, w, zinit, nu0, max_iter, kbar, zout, nuout, status, iters, step, step_ld, ws, ws_stride);
        return hipGetLastError();
    }
    const size_t lds = fmpc_generic_lds_bytes(M.n, M.m);
    hipLaunchKernelGGL(fmpc_newton_generic<false>, dim3(grid), dim3(FMPC_THREADS), lds, stream, M, batch,
                       x0, x0p, w, zinit, nu0, max_iter, kbar, zout, nuout, status, iters, step,
                       step_ld, ws, ws_stride);
    return hipGetLastError();
}

hipError_t fmpc_generic_prepare(size_t lds_bytes, int big) {
    if (big) return hipFuncSetAttribute((const void*)fmpc_newton_generic<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    return hipFuncSetAttribute((const void*)fmpc_newton_generic<false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

hipError_t fmpc_launch_unpack(int n, int m, int T, int batch, const double* z, double* U,
                              double* X, double* u0, hipStream_t stream) {
    const size_t total = (size_t)T * (n + m) * batch;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(fmpc_unpack_kernel, dim3(grid), dim3(256), 0, stream, n, m, T, batch, z, U, X, u0);
    return hipGetLastError();
}
