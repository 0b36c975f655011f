/*
 * TEST / BASELINE INFRASTRUCTURE -- structured CPU solver for the fastMPC hot path.  PARITY UNPINNED (see
 * oracle/dense_ref.py: the reference is MATLAB only and holds no vectors).
 *
 * The same Newton iteration as oracle/banded_ref.py (itself pinned to the op-for-op dense restatement of the
 * reference in tests/test_oracle_banded.py), in plain C with OpenMP over the batch: the fair ALGORITHMIC CPU baseline
 * SURVEY.md §8(d) asks for ("B-banded": the GPU algorithm on the host cores), next to the dense restatement that
 * stands in for MATLAB.  Diagonal Q, R, Qf (the device's scope).  Only tests/ and bench.py's cpu_baseline leg load
 * this library; the product package never does.
 *
 * Reference lines restated (under /root/reference/Fast_MPC/VAR_2):
 *   residuals r_d, r_p, early exit     inf_newton_solver.m:11-22
 *   Phi = 2H + k P'DP                  inf_newton_KKT_H.m:3-13
 *   Y = C Phi^-1 C', chol, dnu, dz     inf_newton_solver.m:24-35   (block-penta-diagonal form: SURVEY App. A.4)
 *   line search                        backtracking_inf_newton.m:2-11 (closed form: SURVEY App. A.5)
 *   problem data                       fast_mpc_eq_const.m:38-71, fast_mpc_objective.m:50-65,
 *                                      fast_mpc_ineq_const.m:46-56, fast_mpc_init.m:19-25
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAX_HALVINGS 64
#define ST_OK 0
#define ST_W_LINESEARCH 1
#define ST_E_NOT_PD_PHI (-4)
#define ST_E_NOT_PD_SCHUR (-5)

typedef struct {
    int n, m, T, nb, var2, has_xf;
    const double *A1, *A2, *B;            /* row-major n x n, n x n, n x m */
    const double *Q2, *R2, *Qf2;          /* 2 diag(Q), 2 diag(R), 2 diag(Qf) */
    const double *q, *r, *qf, *umin, *umax, *xmin, *xmax, *xf;
    double *Yd, *Y1, *Y2;                 /* per block row: constant parts of Y_ii, Y_{i,i+1}, Y_{i,i+2} (n x n each) */
    char *has1, *has2;
} model_t;

static double Xj(const model_t* M, int j, int a) { return 1.0 / (j == M->T ? M->Qf2[a] : M->Q2[a]); }

/* out (n x n) += sign * A diag(x) B'    */
static void add_AxBt(double* out, const double* A, const model_t* M, int j, const double* Bm, int n, double sign) {
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            double t = 0.0;
            for (int c = 0; c < n; ++c) t += A[a * n + c] * Xj(M, j, c) * Bm[b * n + c];
            out[a * n + b] += sign * t;
        }
}

static void const_blocks(model_t* M) {
    const int n = M->n, T = M->T, nb = M->nb, nn = n * n;
    memset(M->Yd, 0, sizeof(double) * nb * nn); memset(M->Y1, 0, sizeof(double) * nb * nn); memset(M->Y2, 0, sizeof(double) * nb * nn);
    memset(M->has1, 0, nb); memset(M->has2, 0, nb);
    for (int i = 0; i < T; ++i) {
        double* d = M->Yd + (size_t)i * nn;
        for (int a = 0; a < n; ++a) d[a * n + a] = Xj(M, i + 1, a);
        if (i >= 1) add_AxBt(d, M->A1, M, i, M->A1, n, 1.0);
        if (i >= 2 && M->var2) add_AxBt(d, M->A2, M, i - 1, M->A2, n, 1.0);
        if (i + 1 < T) {
            double* o = M->Y1 + (size_t)i * nn;
            for (int a = 0; a < n; ++a)
                for (int b = 0; b < n; ++b) o[a * n + b] = -Xj(M, i + 1, a) * M->A1[b * n + a];
            if (i >= 1 && M->var2) add_AxBt(o, M->A1, M, i, M->A2, n, 1.0);
            M->has1[i] = 1;
        }
        if (i + 2 < T && M->var2) {
            double* o = M->Y2 + (size_t)i * nn;
            for (int a = 0; a < n; ++a)
                for (int b = 0; b < n; ++b) o[a * n + b] = -Xj(M, i + 1, a) * M->A2[b * n + a];
            M->has2[i] = 1;
        }
    }
    if (M->has_xf) {
        double* d = M->Yd + (size_t)T * nn;
        for (int a = 0; a < n; ++a) d[a * n + a] = Xj(M, T, a);
        memcpy(M->Y1 + (size_t)(T - 1) * nn, d, sizeof(double) * nn);
        M->has1[T - 1] = 1;
    }
}

/* lower Cholesky in place (row-major, lower triangle used); returns 0 when not positive definite */
static int chol_lower(double* S, int n) {
    for (int c = 0; c < n; ++c) {
        double d = S[c * n + c];
        for (int k = 0; k < c; ++k) d -= S[c * n + k] * S[c * n + k];
        if (!(d > 0.0) || isinf(d)) return 0;
        const double l = sqrt(d);
        S[c * n + c] = l;
        for (int r = c + 1; r < n; ++r) {
            double t = S[r * n + c];
            for (int k = 0; k < c; ++k) t -= S[r * n + k] * S[c * n + k];
            S[r * n + c] = t / l;
        }
    }
    return 1;
}

/* X (n x n, row-major) <- X L^-T, i.e. every ROW x of X solves L x' = x (forward substitution along the row) */
static void rows_times_LinvT(double* X, const double* L, int n) {
    for (int r = 0; r < n; ++r) {
        double* x = X + (size_t)r * n;
        for (int c = 0; c < n; ++c) {
            double t = x[c];
            for (int k = 0; k < c; ++k) t -= L[c * n + k] * x[k];
            x[c] = t / L[c * n + c];
        }
    }
}

typedef struct {
    double *b, *U, *Xs, *NU, *hess, *bar, *rdu, *rdx, *rp, *winv, *phx, *rhs, *Ld, *L1, *L2, *y, *dnu, *du, *dx, *BW;
} work_t;

static size_t work_doubles(int n, int m, int T, int nb) {
    const size_t nbn = (size_t)nb * n, Tm = (size_t)T * m, Tn = (size_t)T * n, nn = (size_t)n * n;
    return nbn * 6 + Tm * 7 + Tn * 5 + 3 * nb * nn + (size_t)n * m + 64;
}

static void work_carve(work_t* W, double* p, int n, int m, int T, int nb) {
    const size_t nbn = (size_t)nb * n, Tm = (size_t)T * m, Tn = (size_t)T * n, nn = (size_t)n * n;
    W->b = p; p += nbn; W->NU = p; p += nbn; W->rp = p; p += nbn; W->rhs = p; p += nbn; W->y = p; p += nbn; W->dnu = p; p += nbn;
    W->U = p; p += Tm; W->hess = p; p += Tm; W->bar = p; p += Tm; W->rdu = p; p += Tm; W->winv = p; p += Tm; W->du = p; p += Tm;
    p += Tm;
    W->Xs = p; p += Tn; W->rdx = p; p += Tn; W->phx = p; p += Tn; W->dx = p; p += Tn; p += Tn;
    W->Ld = p; p += nb * nn; W->L1 = p; p += nb * nn; W->L2 = p; p += nb * nn;
    W->BW = p;
}

/* one inf_newton_solver call; returns the status, *iters_out = Newton steps taken */
static int solve_one(const model_t* M, work_t* W, const double* x0, const double* x0p, const double* w,
                     const double* z_init, const double* nu0, int n_newton, double k, double* z, double* nu_out,
                     int* iters_out, double* step, int step_ld) {
    const int n = M->n, m = M->m, T = M->T, nb = M->nb, s = n + m, nn = n * n;
    const double *A1 = M->A1, *A2 = M->A2, *B = M->B;
    /* b  (fast_mpc_eq_const.m:39,44,47,68) */
    memset(W->b, 0, sizeof(double) * nb * n);
    if (w) memcpy(W->b, w, sizeof(double) * T * n);
    for (int r = 0; r < n; ++r) {
        double t = 0.0, t2 = 0.0;
        for (int c = 0; c < n; ++c) { t += A1[r * n + c] * x0[c]; if (M->var2) { if (x0p) t += A2[r * n + c] * x0p[c]; t2 += A2[r * n + c] * x0[c]; } }
        W->b[r] += t;
        if (T > 1) W->b[n + r] += t2;
    }
    if (M->has_xf) memcpy(W->b + (size_t)T * n, M->xf, sizeof(double) * n);
    /* start point (fast_mpc_init.m:12-27) */
    for (int j = 0; j < T; ++j) {
        for (int c = 0; c < m; ++c) W->U[j * m + c] = z_init ? z_init[j * s + c] : 0.5 * (M->umin[c] + M->umax[c]);
        for (int r = 0; r < n; ++r) W->Xs[j * n + r] = z_init ? z_init[j * s + m + r] : 0.5 * (M->xmin[r] + M->xmax[r]);
    }
    for (int i = 0; i < nb * n; ++i) W->NU[i] = nu0 ? nu0[i] : 0.0;
    if (step) for (int i = 0; i < step_ld; ++i) step[i] = -1.0;
    const int max_iter = n_newton > 0 ? n_newton : 1000;
    int status = ST_OK, steps = 0;
    for (int it = 0; it < max_iter; ++it) {
        double acc_d = 0.0, acc_p = 0.0;
        int bad = 0;
        for (int j = 0; j < T; ++j)
            for (int c = 0; c < m; ++c) {
                const double u = W->U[j * m + c];
                const double dp = 1.0 / (M->umax[c] - u), dm = 1.0 / (u - M->umin[c]);
                const double hs = k * (dp * dp + dm * dm);
                double dot = 0.0;
                for (int r = 0; r < n; ++r) dot += B[r * m + c] * W->NU[j * n + r];
                const double rd = M->R2[c] * u + (M->r ? M->r[c] : 0.0) + k * (dp - dm) - dot;
                const double rt = M->R2[c] + hs;
                if (!(rt > 0.0) || isinf(rt)) bad = 1;
                W->hess[j * m + c] = hs; W->winv[j * m + c] = 1.0 / rt; W->rdu[j * m + c] = rd;
                acc_d += rd * rd;
            }
        for (int jj = 0; jj < T; ++jj) {
            const int j = jj + 1;
            for (int r = 0; r < n; ++r) {
                const double q2 = j == T ? M->Qf2[r] : M->Q2[r];
                const double ql = j == T ? (M->qf ? M->qf[r] : 0.0) : (M->q ? M->q[r] : 0.0);
                double v = q2 * W->Xs[jj * n + r] + ql + W->NU[jj * n + r];
                if (j < T) for (int c = 0; c < n; ++c) v -= A1[c * n + r] * W->NU[j * n + c];
                if (M->var2 && j + 1 < T) for (int c = 0; c < n; ++c) v -= A2[c * n + r] * W->NU[(j + 1) * n + c];
                if (j == T && M->has_xf) v += W->NU[T * n + r];
                W->rdx[jj * n + r] = v; W->phx[jj * n + r] = v / q2;
                acc_d += v * v;
            }
        }
        for (int i = 0; i < nb; ++i)
            for (int r = 0; r < n; ++r) {
                double v;
                if (i < T) {
                    v = W->Xs[i * n + r] - W->b[i * n + r];
                    for (int c = 0; c < m; ++c) v -= B[r * m + c] * W->U[i * m + c];
                    if (i >= 1) for (int c = 0; c < n; ++c) v -= A1[r * n + c] * W->Xs[(i - 1) * n + c];
                    if (M->var2 && i >= 2) for (int c = 0; c < n; ++c) v -= A2[r * n + c] * W->Xs[(i - 2) * n + c];
                } else {
                    v = W->Xs[(T - 1) * n + r] - W->b[i * n + r];
                }
                W->rp[i * n + r] = v;
                acc_p += v * v;
            }
        const double rho2 = acc_d + acc_p;
        if (sqrt(rho2) <= 1e-6 && sqrt(acc_p) <= 1e-8) break;          /* inf_newton_solver.m:19-22 */
        if (bad) { status = ST_E_NOT_PD_PHI; break; }
        /* rhs = r_p - C Phi^-1 r_d */
        for (int i = 0; i < nb; ++i)
            for (int r = 0; r < n; ++r) {
                double cv;
                if (i < T) {
                    cv = W->phx[i * n + r];
                    for (int c = 0; c < m; ++c) cv -= B[r * m + c] * (W->rdu[i * m + c] * W->winv[i * m + c]);
                    if (i >= 1) for (int c = 0; c < n; ++c) cv -= A1[r * n + c] * W->phx[(i - 1) * n + c];
                    if (M->var2 && i >= 2) for (int c = 0; c < n; ++c) cv -= A2[r * n + c] * W->phx[(i - 2) * n + c];
                } else {
                    cv = W->phx[(T - 1) * n + r];
                }
                W->rhs[i * n + r] = W->rp[i * n + r] - cv;
            }
        /* block-penta-diagonal Cholesky fused with the forward sweep */
        int ok = 1;
        for (int i = 0; i < nb && ok; ++i) {
            double* S = W->Ld + (size_t)i * nn;
            memcpy(S, M->Yd + (size_t)i * nn, sizeof(double) * nn);
            if (i < T) {
                for (int r = 0; r < n; ++r)
                    for (int c = 0; c < m; ++c) W->BW[r * m + c] = B[r * m + c] * W->winv[i * m + c];
                for (int a = 0; a < n; ++a)
                    for (int bb = 0; bb <= a; ++bb) {
                        double t = 0.0;
                        for (int c = 0; c < m; ++c) t += W->BW[a * m + c] * B[bb * m + c];
                        S[a * n + bb] += t;
                    }
            }
            double* yi = W->y + (size_t)i * n;
            memcpy(yi, W->rhs + (size_t)i * n, sizeof(double) * n);
            if (i >= 1 && M->has1[i - 1]) {
                const double* L1p = W->L1 + (size_t)(i - 1) * nn;          /* L_{i,i-1} */
                for (int a = 0; a < n; ++a) {
                    for (int bb = 0; bb <= a; ++bb) {
                        double t = 0.0;
                        for (int c = 0; c < n; ++c) t += L1p[a * n + c] * L1p[bb * n + c];
                        S[a * n + bb] -= t;
                    }
                    double t = 0.0;
                    for (int c = 0; c < n; ++c) t += L1p[a * n + c] * W->y[(i - 1) * n + c];
                    yi[a] -= t;
                }
            }
            if (i >= 2 && M->has2[i - 2]) {
                const double* L2p = W->L2 + (size_t)(i - 2) * nn;          /* L_{i,i-2} */
                for (int a = 0; a < n; ++a) {
                    for (int bb = 0; bb <= a; ++bb) {
                        double t = 0.0;
                        for (int c = 0; c < n; ++c) t += L2p[a * n + c] * L2p[bb * n + c];
                        S[a * n + bb] -= t;
                    }
                    double t = 0.0;
                    for (int c = 0; c < n; ++c) t += L2p[a * n + c] * W->y[(i - 2) * n + c];
                    yi[a] -= t;
                }
            }
            if (!chol_lower(S, n)) { ok = 0; break; }
            for (int c = 0; c < n; ++c) {                                  /* y_i = L^-1 s */
                double t = yi[c];
                for (int kk = 0; kk < c; ++kk) t -= S[c * n + kk] * yi[kk];
                yi[c] = t / S[c * n + c];
            }
            if (i + 1 < nb && M->has1[i]) {                                /* L_{i+1,i} = (Y_{i+1,i} - L_{i+1,i-1} L_{i,i-1}') L^-T */
                double* X = W->L1 + (size_t)i * nn;
                const double* Y1 = M->Y1 + (size_t)i * nn;
                for (int a = 0; a < n; ++a)
                    for (int bb = 0; bb < n; ++bb) X[a * n + bb] = Y1[bb * n + a];
                if (i >= 1 && M->has2[i - 1] && M->has1[i - 1]) {
                    const double* L2p = W->L2 + (size_t)(i - 1) * nn;      /* L_{i+1,i-1} */
                    const double* L1p = W->L1 + (size_t)(i - 1) * nn;      /* L_{i,i-1} */
                    for (int a = 0; a < n; ++a)
                        for (int bb = 0; bb < n; ++bb) {
                            double t = 0.0;
                            for (int c = 0; c < n; ++c) t += L2p[a * n + c] * L1p[bb * n + c];
                            X[a * n + bb] -= t;
                        }
                }
                rows_times_LinvT(X, S, n);
            }
            if (i + 2 < nb && M->has2[i]) {                                /* L_{i+2,i} = Y_{i+2,i} L^-T */
                double* X = W->L2 + (size_t)i * nn;
                const double* Y2 = M->Y2 + (size_t)i * nn;
                for (int a = 0; a < n; ++a)
                    for (int bb = 0; bb < n; ++bb) X[a * n + bb] = Y2[bb * n + a];
                rows_times_LinvT(X, S, n);
            }
        }
        if (!ok) { status = ST_E_NOT_PD_SCHUR; break; }
        /* backward sweep */
        for (int i = nb - 1; i >= 0; --i) {
            double* v = W->dnu + (size_t)i * n;
            const double* L = W->Ld + (size_t)i * nn;
            memcpy(v, W->y + (size_t)i * n, sizeof(double) * n);
            if (i + 1 < nb && M->has1[i]) {
                const double* X = W->L1 + (size_t)i * nn;
                for (int a = 0; a < n; ++a) { const double d = W->dnu[(i + 1) * n + a]; for (int c = 0; c < n; ++c) v[c] -= X[a * n + c] * d; }
            }
            if (i + 2 < nb && M->has2[i]) {
                const double* X = W->L2 + (size_t)i * nn;
                for (int a = 0; a < n; ++a) { const double d = W->dnu[(i + 2) * n + a]; for (int c = 0; c < n; ++c) v[c] -= X[a * n + c] * d; }
            }
            for (int c = n - 1; c >= 0; --c) {                             /* L' x = v */
                double t = v[c];
                for (int kk = c + 1; kk < n; ++kk) t -= L[kk * n + c] * v[kk];
                v[c] = t / L[c * n + c];
            }
        }
        /* d_z and the line-search scalars */
        double beta_e = 0.0, eps2 = 0.0;
        for (int j = 0; j < T; ++j)
            for (int c = 0; c < m; ++c) {
                double dot = 0.0;
                for (int r = 0; r < n; ++r) dot += B[r * m + c] * W->dnu[j * n + r];
                const double du = (dot - W->rdu[j * m + c]) * W->winv[j * m + c];
                const double e = W->hess[j * m + c] * du;
                beta_e += W->rdu[j * m + c] * e; eps2 += e * e;
                W->du[j * m + c] = du;
            }
        for (int jj = 0; jj < T; ++jj) {
            const int j = jj + 1;
            for (int r = 0; r < n; ++r) {
                double v = -W->rdx[jj * n + r] - W->dnu[jj * n + r];
                if (j < T) for (int c = 0; c < n; ++c) v += A1[c * n + r] * W->dnu[j * n + c];
                if (M->var2 && j + 1 < T) for (int c = 0; c < n; ++c) v += A2[c * n + r] * W->dnu[(j + 1) * n + c];
                if (j == T && M->has_xf) v -= W->dnu[T * n + r];
                W->dx[jj * n + r] = v / (j == T ? M->Qf2[r] : M->Q2[r]);
            }
        }
        double t = 1.0;
        {
            const double al = 1e-4;
            int halv = 0;
            for (;;) {
                const double gq = (t - 2.0 + 2.0 * al - al * al * t) * rho2 - 2.0 * (1.0 - t) * beta_e + t * eps2;
                if (gq <= 0.0) break;
                t *= 0.5;
                if (++halv >= MAX_HALVINGS) { t = 0.0; status = ST_W_LINESEARCH; break; }
            }
        }
        for (int i = 0; i < T * m; ++i) W->U[i] += t * W->du[i];
        for (int i = 0; i < T * n; ++i) W->Xs[i] += t * W->dx[i];
        for (int i = 0; i < nb * n; ++i) W->NU[i] += t * W->dnu[i];
        if (step && it < step_ld) step[it] = t;
        ++steps;
    }
    for (int j = 0; j < T; ++j) {
        memcpy(z + (size_t)j * s, W->U + (size_t)j * m, sizeof(double) * m);
        memcpy(z + (size_t)j * s + m, W->Xs + (size_t)j * n, sizeof(double) * n);
    }
    if (nu_out) memcpy(nu_out, W->NU, sizeof(double) * nb * n);
    *iters_out = steps;
    return status;
}

/*
 * Batch entry (ctypes).  Matrices ROW-major: A1, A2 n x n (A2 NULL: VAR(1)), B n x m; Q2, R2, Qf2 the DOUBLED
 * diagonals; q, r, qf, xf, x0_pre, w, z_init, nu0 nullable; per-problem arrays problem-major.  nthreads <= 0: all.
 * Returns 0, or -1 on allocation failure.
 */
int banded_cpu_solve_batch(int n, int m, int T, const double* A1, const double* A2, const double* B, const double* Q2,
                           const double* R2, const double* Qf2, const double* q, const double* r, const double* qf,
                           const double* umin, const double* umax, const double* xmin, const double* xmax,
                           const double* xf, int batch, const double* x0, const double* x0_pre, const double* w,
                           const double* z_init, const double* nu0, int n_newton, double k, double* z, double* nu,
                           int* iters, int* status, double* step, int step_ld, int nthreads) {
    model_t M;
    M.n = n; M.m = m; M.T = T; M.has_xf = xf != NULL; M.nb = T + M.has_xf; M.var2 = A2 != NULL;
    M.A1 = A1; M.A2 = A2; M.B = B; M.Q2 = Q2; M.R2 = R2; M.Qf2 = Qf2; M.q = q; M.r = r; M.qf = qf;
    M.umin = umin; M.umax = umax; M.xmin = xmin; M.xmax = xmax; M.xf = xf;
    const size_t nn = (size_t)n * n;
    M.Yd = (double*)malloc(sizeof(double) * 3 * M.nb * nn);
    M.has1 = (char*)malloc(2 * (size_t)M.nb);
    if (!M.Yd || !M.has1) { free(M.Yd); free(M.has1); return -1; }
    M.Y1 = M.Yd + (size_t)M.nb * nn; M.Y2 = M.Y1 + (size_t)M.nb * nn; M.has2 = M.has1 + M.nb;
    const_blocks(&M);
    const size_t Nz = (size_t)T * (n + m), nbn = (size_t)M.nb * n, Tn = (size_t)T * n;
    int fail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        double* buf = (double*)malloc(sizeof(double) * work_doubles(n, m, T, M.nb));
        work_t W;
        if (!buf) {
#pragma omp atomic write
            fail = 1;
        } else {
            work_carve(&W, buf, n, m, T, M.nb);
#pragma omp for schedule(dynamic, 1)
            for (int p = 0; p < batch; ++p) {
                int it = 0;
                const int st = solve_one(&M, &W, x0 + (size_t)p * n, x0_pre ? x0_pre + (size_t)p * n : NULL,
                                         w ? w + (size_t)p * Tn : NULL, z_init ? z_init + (size_t)p * Nz : NULL,
                                         nu0 ? nu0 + (size_t)p * nbn : NULL, n_newton, k, z + (size_t)p * Nz,
                                         nu ? nu + (size_t)p * nbn : NULL, &it, step ? step + (size_t)p * step_ld : NULL, step_ld);
                if (iters) iters[p] = it;
                if (status) status[p] = st;
            }
            free(buf);
        }
    }
    free(M.Yd); free(M.has1);
    return fail ? -1 : 0;
}

int banded_cpu_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
