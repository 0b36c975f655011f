// CPU test of fmpc_host_build_ramp_cold (mpc-sensorlessao_amd/csrc/fmpc_host.cpp), built by tests/test_host_sanitizers.py with
// g++ -fsanitize=address,undefined:   ramp_cold_test n m T has_xf var_order seed
// For one synthetic model with ramp-rate bounds it builds the constants of the cold-start step's Woodbury form, then does in
// plain double exactly what the device kernel (fmpc_ramp_cold, fmpc_kernel_ramp.hip) does with them for a few problems
// (u_prev, x0, x0_pre, w at random): delta, rho, y_u0, (diag(1/delta) + G) q = y_u0, the pass through the constant operators --
// and compares [d_z ; nu+] with an independent dense solve of the FULL KKT system [[Phi, C'], [C, 0]] of that problem
// (Gaussian elimination with partial pivoting in long double; Phi assembled from the rows of VAR_1/fast_mpc_ineq_const.m:58-76).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

#include "../../mpc-sensorlessao_amd/csrc/fmpc_host.h"

typedef long double ld;
static int fail(const char* what) { fprintf(stderr, "FAIL: %s\n", what); return 1; }

static bool solve_dense(std::vector<ld>& A, std::vector<ld>& b, int N) {       // in place, partial pivoting
    for (int k = 0; k < N; ++k) {
        int piv = k;
        for (int i = k + 1; i < N; ++i) if (fabsl(A[(size_t)i * N + k]) > fabsl(A[(size_t)piv * N + k])) piv = i;
        if (A[(size_t)piv * N + k] == 0.0L) return false;
        if (piv != k) { for (int c = 0; c < N; ++c) std::swap(A[(size_t)k * N + c], A[(size_t)piv * N + c]); std::swap(b[k], b[piv]); }
        for (int i = k + 1; i < N; ++i) {
            const ld f = A[(size_t)i * N + k] / A[(size_t)k * N + k];
            if (f == 0.0L) continue;
            for (int c = k; c < N; ++c) A[(size_t)i * N + c] -= f * A[(size_t)k * N + c];
            b[i] -= f * b[k];
        }
    }
    for (int k = N - 1; k >= 0; --k) {
        ld v = b[k];
        for (int c = k + 1; c < N; ++c) v -= A[(size_t)k * N + c] * b[c];
        b[k] = v / A[(size_t)k * N + k];
    }
    return true;
}

int main(int argc, char** argv) {
    if (argc < 7) return fail("usage: n m T has_xf var_order seed");
    const int n = atoi(argv[1]), m = atoi(argv[2]), T = atoi(argv[3]), has_xf = atoi(argv[4]), var_order = atoi(argv[5]);
    const unsigned seed = (unsigned)atoi(argv[6]);
    const int nb = T + has_xf, nn = n * n, s = n + m, Nz = T * s, nbn = nb * n;
    const bool var2 = var_order == 2;
    const double k = 0.01;
    std::mt19937_64 rng(seed);
    std::normal_distribution<double> N01(0.0, 1.0);
    std::uniform_real_distribution<double> U01(0.0, 1.0);
    std::vector<double> a1(nn), a2(nn, 0.0), bt((size_t)m * n), R2(m), rl(m), Q2(n), Qf2(n), ql(n), qfl(n), umin(m), umax(m), umid(m), xmid(n), xf(n, 0.0), dumin(m), dumax(m);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            a1[r * n + c] = (r == c ? 0.9 + 0.09 * U01(rng) : 0.0) + 0.02 * N01(rng);
            if (var2) a2[r * n + c] = (r == c ? -0.3 : 0.0) + 0.02 * N01(rng);
        }
    for (size_t i = 0; i < bt.size(); ++i) bt[i] = 0.3 * N01(rng);
    for (int j = 0; j < m; ++j) {
        R2[j] = 2.0 * (1.0 + 0.1 * U01(rng)); rl[j] = 0.01 * N01(rng); umin[j] = -2.0 - U01(rng); umax[j] = 2.0 + U01(rng); umid[j] = 0.5 * (umin[j] + umax[j]);
        dumin[j] = -0.4 * (0.5 + U01(rng)); dumax[j] = 0.4 * (0.5 + U01(rng));
    }
    for (int r = 0; r < n; ++r) { Q2[r] = 2.0 * (1.0 + 0.5 * U01(rng)); Qf2[r] = 50.0 * Q2[r]; ql[r] = 0.1 * N01(rng); qfl[r] = 0.1 * N01(rng); xmid[r] = 0.01 * N01(rng); xf[r] = 0.1 * N01(rng); }

    FmpcRampColdIn In;
    In.n = n; In.m = m; In.T = T; In.nb = nb; In.var2 = var2 ? 1 : 0; In.has_xf = has_xf;
    In.bt = bt.data(); In.a1 = a1.data(); In.a2 = a2.data(); In.umax = umax.data(); In.umin = umin.data(); In.umid = umid.data(); In.xmid = xmid.data();
    In.R2 = R2.data(); In.rl = rl.data(); In.Q2 = Q2.data(); In.Qf2 = Qf2.data(); In.ql = ql.data(); In.qfl = qfl.data(); In.xf = xf.data();
    In.dumin = dumin.data(); In.dumax = dumax.data(); In.k = k;
    FmpcRampColdOut O;
    fmpc_host_build_ramp_cold(In, O);
    if (!O.valid) return fail("builder reports invalid");
    if ((int)O.G.size() != m * ((m + 1) & ~1) || (int)O.Yinv.size() != nbn * ((nbn + 1) & ~1) || (int)O.Xiu0t.size() != T * n * ((m + 1) & ~1)) return fail("sizes");

    double worst = 0.0;
    for (int prob = 0; prob < 3; ++prob) {
        // ---- data of one problem
        std::vector<double> uprev(m), x0(n), x0p(n), w((size_t)T * n);
        for (int c = 0; c < m; ++c) uprev[c] = umid[c] + 0.5 * (dumin[c] + (dumax[c] - dumin[c]) * U01(rng));
        for (int r = 0; r < n; ++r) { x0[r] = N01(rng); x0p[r] = N01(rng); }
        for (size_t i = 0; i < w.size(); ++i) w[i] = prob == 0 ? 0.0 : 0.1 * N01(rng);
        // ---- what the device does (plain double)
        std::vector<double> delta(m), rho(m), bh(nbn, 0.0);
        for (int c = 0; c < m; ++c) {
            const double dl = umid[c] - uprev[c], a = 1.0 / (dumax[c] - dl), b = 1.0 / (dl - dumin[c]);
            delta[c] = k * (a * a + b * b); rho[c] = k * (a - b);
        }
        for (int i = 0; i < T; ++i)
            for (int r = 0; r < n; ++r) {
                double v = w[(size_t)i * n + r];
                if (i == 0) for (int c = 0; c < n; ++c) v += a1[r * n + c] * x0[c] + (var2 ? a2[r * n + c] * x0p[c] : 0.0);
                if (i == 1 && var2) for (int c = 0; c < n; ++c) v += a2[r * n + c] * x0[c];
                bh[(size_t)i * n + r] = v;
            }
        std::vector<double> yu0(m);
        for (int r = 0; r < m; ++r) {
            double v = O.y0c[r];
            for (int c = 0; c < m; ++c) v -= O.G[(size_t)c * ((m + 1) & ~1) + r] * rho[c];
            for (int col = 0; col < T * n; ++col) v += O.Xiu0t[(size_t)col * ((m + 1) & ~1) + r] * bh[col];
            yu0[r] = v;
        }
        std::vector<ld> Mq((size_t)m * m), q(m);
        for (int r = 0; r < m; ++r) { for (int c = 0; c < m; ++c) Mq[(size_t)r * m + c] = O.G[(size_t)r * ((m + 1) & ~1) + c]; Mq[(size_t)r * m + r] += 1.0L / delta[r]; q[r] = yu0[r]; }
        if (!solve_dense(Mq, q, m)) return fail("M singular");
        std::vector<double> sv(m), phiu((size_t)T * m), beta(nbn), nup(nbn), kap((size_t)T * m), dz(Nz);
        for (int c = 0; c < m; ++c) sv[c] = rho[c] + (double)q[c];
        for (int j = 0; j < T; ++j) for (int c = 0; c < m; ++c) phiu[(size_t)j * m + c] = O.phib_u[(size_t)j * m + c] - O.g0[(size_t)j * m + c] * sv[c];
        for (int a = 0; a < nbn; ++a) {
            double v = O.betab[a] - bh[a];
            const int i = a / n, r = a % n;
            if (i < T) for (int c = 0; c < m; ++c) v += bt[(size_t)c * n + r] * O.g0[(size_t)i * m + c] * sv[c];
            beta[a] = v;
        }
        for (int a = 0; a < nbn; ++a) { double v = 0.0; for (int b = 0; b < nbn; ++b) v += O.Yinv[(size_t)b * ((nbn + 1) & ~1) + a] * beta[b]; nup[a] = v; }
        for (int j = 0; j < T; ++j) for (int c = 0; c < m; ++c) { double v = 0.0; for (int r = 0; r < n; ++r) v += bt[(size_t)c * n + r] * nup[(size_t)j * n + r]; kap[(size_t)j * m + c] = v; }
        for (int j = 0; j < T; ++j) {
            for (int c = 0; c < m; ++c) {
                double v = phiu[(size_t)j * m + c];
                for (int i = 0; i < T; ++i) v += O.Gf[((size_t)j * T + i) * m + c] * kap[(size_t)i * m + c];
                dz[(size_t)j * s + c] = v;
            }
            const int jx = j + 1;
            for (int r = 0; r < n; ++r) {
                double v = nup[(size_t)(jx - 1) * n + r];
                if (jx < T) for (int c = 0; c < n; ++c) v -= a1[c * n + r] * nup[(size_t)jx * n + c];
                if (var2 && jx + 1 < T) for (int c = 0; c < n; ++c) v -= a2[c * n + r] * nup[(size_t)(jx + 1) * n + c];
                if (jx == T && has_xf) v += nup[(size_t)T * n + r];
                dz[(size_t)j * s + m + r] = O.phib_x[(size_t)j * n + r] - v / (jx == T ? Qf2[r] : Q2[r]);
            }
        }
        // ---- the full KKT system of this problem, assembled from scratch
        const int N = Nz + nbn;
        std::vector<ld> K((size_t)N * N, 0.0L), f(N, 0.0L);
        std::vector<ld> z0(Nz);
        for (int j = 0; j < T; ++j) { for (int c = 0; c < m; ++c) z0[(size_t)j * s + c] = umid[c]; for (int r = 0; r < n; ++r) z0[(size_t)j * s + m + r] = xmid[r]; }
        for (int j = 0; j < T; ++j) {
            for (int c = 0; c < m; ++c) {
                const int idx = j * s + c;
                const ld u = umid[c], dp = 1.0L / ((ld)umax[c] - u), dm = 1.0L / (u - (ld)umin[c]);
                const ld dl = j == 0 ? u - (ld)uprev[c] : 0.0L, rp = 1.0L / ((ld)dumax[c] - dl), rm = 1.0L / (dl - (ld)dumin[c]);
                const ld er = (ld)k * (rp * rp + rm * rm), gr = (ld)k * (rp - rm);
                K[(size_t)idx * N + idx] += (ld)R2[c] + (ld)k * (dp * dp + dm * dm) + er;
                f[idx] -= (ld)R2[c] * u + (ld)rl[c] + (ld)k * (dp - dm) + gr;
                if (j >= 1) {                                     // the row u_j - u_{j-1} also acts on u_{j-1}
                    const int pi = (j - 1) * s + c;
                    K[(size_t)pi * N + pi] += er; K[(size_t)pi * N + idx] -= er; K[(size_t)idx * N + pi] -= er;
                    f[pi] += gr;
                }
            }
            for (int r = 0; r < n; ++r) {
                const int idx = j * s + m + r;
                const bool last = j + 1 == T;
                K[(size_t)idx * N + idx] = last ? Qf2[r] : Q2[r];
                f[idx] -= last ? (ld)Qf2[r] * (ld)xmid[r] + (ld)qfl[r] : (ld)Q2[r] * (ld)xmid[r] + (ld)ql[r];
            }
        }
        auto setC = [&](int row, int col, ld v) { K[(size_t)(Nz + row) * N + col] = v; K[(size_t)col * N + Nz + row] = v; };
        for (int i = 0; i < T; ++i)
            for (int r = 0; r < n; ++r) {
                const int row = i * n + r;
                setC(row, i * s + m + r, 1.0L);
                for (int c = 0; c < m; ++c) setC(row, i * s + c, -(ld)bt[(size_t)c * n + r]);
                if (i >= 1) for (int c = 0; c < n; ++c) setC(row, (i - 1) * s + m + c, -(ld)a1[r * n + c]);
                if (var2 && i >= 2) for (int c = 0; c < n; ++c) setC(row, (i - 2) * s + m + c, -(ld)a2[r * n + c]);
            }
        if (has_xf) for (int r = 0; r < n; ++r) setC(T * n + r, (T - 1) * s + m + r, 1.0L);
        for (int a = 0; a < nbn; ++a) {                           // f_nu = -r_p = -(C z0 - b)
            ld cz = 0.0L;
            for (int c = 0; c < Nz; ++c) cz += K[(size_t)(Nz + a) * N + c] * z0[c];
            const ld b = a < T * n ? (ld)bh[a] : (ld)xf[a - T * n];
            f[Nz + a] = -(cz - b);
        }
        if (!solve_dense(K, f, N)) return fail("KKT singular");
        ld num = 0.0L, den = 0.0L, numn = 0.0L, denn = 0.0L;
        for (int i = 0; i < Nz; ++i) { num += ((ld)dz[i] - f[i]) * ((ld)dz[i] - f[i]); den += f[i] * f[i]; }
        for (int a = 0; a < nbn; ++a) { numn += ((ld)nup[a] - f[Nz + a]) * ((ld)nup[a] - f[Nz + a]); denn += f[Nz + a] * f[Nz + a]; }
        const double ez = (double)sqrtl(num / den), en = (double)sqrtl(numn / denn);
        if (ez > worst) worst = ez;
        if (en > worst) worst = en;
        if (!(ez <= 1e-11) || !(en <= 1e-11)) { fprintf(stderr, "problem %d: rel err d_z %.3e nu+ %.3e\n", prob, ez, en); return fail("Woodbury form differs from the dense KKT solve"); }
    }
    printf("ramp cold form ok: n=%d m=%d T=%d xf=%d var=%d worst rel err %.2e\n", n, m, T, has_xf, var_order, worst);
    return 0;
}
