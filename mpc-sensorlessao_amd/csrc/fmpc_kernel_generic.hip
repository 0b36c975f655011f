// Generic fastMPC Newton kernel for gfx950: any (n <= 64, m, T) whose tiles fit the LDS.
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

extern "C" __global__ void __launch_bounds__(FMPC_THREADS)
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

    // ---- LDS carve
    double* sBt = lds;                    // m*n
    double* tile0 = sBt + (size_t)m * n;  // 6 tiles n*ldt
    double* sw = tile0 + 6 * tsz;         // m     winv of the current stage
    double* sv1 = sw + m;                 // n     d_nu_{i+1}
    double* sv2 = sv1 + n;                // n     d_nu_{i+2}
    double* srs = sv2 + n;                // n     1/sqrt(pivot)
    double* red = srs + n;                // 8
    int* sflag = (int*)(red + 8);         // 2

    for (int i = tid; i < m * n; i += FMPC_THREADS) sBt[i] = M.Bt[i];

    const FmpcWsLayout L = fmpc_ws_layout(n, m, T, nb);
    double* wsp = ws + (size_t)blockIdx.x * ws_stride;
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
                const double rt = M.R2[c] + hs;
                if (!(rt > 0.0) || isinf(rt)) bad = 1;
                double dot = 0.0;
                const double* bt = sBt + c * n;
                const double* nj = nu + j * n;
                for (int r = 0; r < n; ++r) dot += bt[r] * nj[r];
                const double rd = M.R2[c] * u + M.rl[c] + kbar * (dp - dm) - dot;
                hess[idx] = hs;
                winv[idx] = 1.0 / rt;
                rdu[idx] = rd;
                acc_d += rd * rd;
            }
            for (int idx = tid; idx < T * n; idx += FMPC_THREADS) {
                const int jj = idx / n, r = idx - jj * n, j = jj + 1;   // x_j, j = 1..T
                const double x = zp[jj * s + m + r];
                double v = (j == T ? M.Qf2[r] * x + M.qfl[r] : M.Q2[r] * x + M.ql[r]) + nu[jj * n + r];
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
                dnu[idx] = v / (j == T ? M.Qf2[r] : M.Q2[r]);   // Phi^-1 r_d on x_j (temp)
                acc_d += v * v;
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

            // ================= P2: rhs_i = r_p,i - (C Phi^-1 r_d)_i   (into y)
            for (int idx = tid; idx < nbn; idx += FMPC_THREADS) {
                const int i = idx / n, r = idx - i * n;
                double cv;
                if (i < T) {
                    const double* phx = dnu;                    // written in P1
                    cv = phx[i * n + r];
                    const double* ru = rdu + i * m;
                    const double* wi = winv + i * m;
                    for (int c = 0; c < m; ++c) cv -= sBt[c * n + r] * (ru[c] * wi[c]);
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
                                for (int c = 0; c < m; ++c) t += sBt[c * n + a] * sw[c] * sBt[c * n + bb];
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
                    if (tid < 64) {                                  // wave 0: lane j <-> entry j
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
                const double du = (dot - rd) * winv[idx];
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
                rdx[idx] = v / (j == T ? M.Qf2[r] : M.Q2[r]);   // reuse as d_x
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

// Closed-loop inputs (fmpc_loop_inputs_device in include/fastmpc.h; README.md:482-497): one workgroup per problem.
//   x0 = a + B u1 ,  x0_pre = x0_last ,  w = -M1 (B u1) - M2 (B u2)      M1, M2: (T n) x n row-major
extern "C" __global__ void __launch_bounds__(256)
fmpc_loop_inputs_kernel(int n, int m, int T, const double* __restrict__ Bt, const double* __restrict__ M1,
                        const double* __restrict__ M2, const double* __restrict__ a, const double* x0_last,
                        const double* __restrict__ u1, const double* __restrict__ u2,
                        double* x0, double* __restrict__ x0_pre, double* __restrict__ w) {
    extern __shared__ double sh[];                      // bu1[n], bu2[n]
    double* bu1 = sh; double* bu2 = sh + n;
    const size_t p = blockIdx.x;
    const int tid = threadIdx.x;
    for (int r = tid; r < 2 * n; r += blockDim.x) {
        const int rr = r < n ? r : r - n;
        const double* u = r < n ? u1 : u2;
        double acc = 0.0;
        if (u) for (int c = 0; c < m; ++c) acc += Bt[(size_t)c * n + rr] * u[p * m + c];
        sh[r] = acc;
    }
    __syncthreads();
    for (int r = tid; r < n; r += blockDim.x) {
        const double xl = x0_last ? x0_last[p * n + r] : 0.0;
        x0_pre[p * n + r] = xl;
        x0[p * n + r] = a[p * n + r] + bu1[r];
    }
    for (int e = tid; e < T * n; e += blockDim.x) {
        double acc = 0.0;
        const double* r1 = M1 + (size_t)e * n; const double* r2 = M2 + (size_t)e * n;
        for (int q = 0; q < n; ++q) acc -= r1[q] * bu1[q] + r2[q] * bu2[q];
        w[p * (size_t)T * n + e] = acc;
    }
}

hipError_t fmpc_launch_loop_inputs(int n, int m, int T, int batch, const double* Bt, const double* M1, const double* M2,
                                   const double* a, const double* x0_last, const double* u1, const double* u2,
                                   double* x0, double* x0_pre, double* w, hipStream_t stream) {
    hipLaunchKernelGGL(fmpc_loop_inputs_kernel, dim3(batch), dim3(256), 2 * n * sizeof(double), stream,
                       n, m, T, Bt, M1, M2, a, x0_last, u1, u2, x0, x0_pre, w);
    return hipGetLastError();
}

size_t fmpc_generic_lds_bytes(int n, int m) {
    const size_t d = (size_t)m * n + 6 * (size_t)n * (n + 1) + m + 3 * (size_t)n + 8 + 2;
    return d * sizeof(double);
}

hipError_t fmpc_launch_generic(const FmpcDevModel& M, int batch, int grid, const double* x0,
                               const double* x0p, const double* w, const double* zinit,
                               const double* nu0, int max_iter, double kbar, double* zout,
                               double* nuout, int* status, int* iters, double* step, int step_ld,
                               double* ws, size_t ws_stride, hipStream_t stream) {
    const size_t lds = fmpc_generic_lds_bytes(M.n, M.m);
    hipLaunchKernelGGL(fmpc_newton_generic, dim3(grid), dim3(FMPC_THREADS), lds, stream, M, batch,
                       x0, x0p, w, zinit, nu0, max_iter, kbar, zout, nuout, status, iters, step,
                       step_ld, ws, ws_stride);
    return hipGetLastError();
}

hipError_t fmpc_generic_prepare(size_t lds_bytes) {
    return hipFuncSetAttribute((const void*)fmpc_newton_generic,
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
