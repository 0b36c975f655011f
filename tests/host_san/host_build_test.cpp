// CPU test of the library's host-only builders (mpc-sensorlessao_amd/csrc/fmpc_host.cpp), built by
// tests/test_host_sanitizers.py with  g++ -fsanitize=address,undefined  (sanitizers on the CPU build).
//   host_build_test n m T has_xf var_order seed
// For one synthetic model it runs the Y-block builder with de-duplication, the panel-path builder (twisted block
// factorisation in long double, operator images, sweep schedules) and checks NUMERICALLY what the device would do with the
// result: it decodes the images, executes the two sweeps step by step exactly as the schedules prescribe (reading only
// sources that are final) and compares nu = Y^-1 rhs with an independent dense Cholesky solve of the assembled Y.
// Also exercises the least-recently-used cache of the one-shot entry.  Exit code 0 = all checks passed.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

#include "../../mpc-sensorlessao_amd/csrc/fmpc_host.h"

typedef long double ld;

static int fail(const char* what) { fprintf(stderr, "FAIL: %s\n", what); return 1; }

int main(int argc, char** argv) {
    if (argc < 7) return fail("usage: n m T has_xf var_order seed");
    const int n = atoi(argv[1]), m = atoi(argv[2]), T = atoi(argv[3]), has_xf = atoi(argv[4]), var_order = atoi(argv[5]);
    const unsigned seed = (unsigned)atoi(argv[6]);
    if (n != FP_N) return fail("the panel path is built for n = 27");
    const int nb = T + has_xf, nn = n * n, mp = (m + 15) & ~15;
    const bool var2 = var_order == 2;
    std::mt19937_64 rng(seed);
    std::normal_distribution<double> N01(0.0, 1.0);
    std::uniform_real_distribution<double> U01(0.0, 1.0);
    // model: per-mode AR(2)-like dynamics + weak coupling, B, diagonal weights, box
    std::vector<double> a1(nn), a2(nn, 0.0), bt((size_t)m * n), R2(m), rl(m), Q2(n), Qf2(n), ql(n), qfl(n), umin(m), umax(m), umid(m), xmid(n), xf(n, 0.0);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            a1[r * n + c] = (r == c ? 1.8 * (0.9 + 0.09 * U01(rng)) : 0.0) + 0.002 * N01(rng);
            if (var2) a2[r * n + c] = (r == c ? -0.85 : 0.0) + 0.002 * N01(rng);
        }
    for (size_t i = 0; i < bt.size(); ++i) bt[i] = 0.05 * N01(rng);
    for (int j = 0; j < m; ++j) { R2[j] = 2.0 * (1.0 + 0.1 * U01(rng)); rl[j] = 0.01 * N01(rng); umin[j] = -28.0 - U01(rng); umax[j] = 28.0 + U01(rng); umid[j] = 0.5 * (umin[j] + umax[j]); }
    for (int r = 0; r < n; ++r) { Q2[r] = 3.0e4 * (1.0 + 0.1 * U01(rng)); Qf2[r] = 2.0 * Q2[r]; ql[r] = 0.1 * N01(rng); qfl[r] = 0.1 * N01(rng); xmid[r] = 0.01 * N01(rng); xf[r] = 0.1 * N01(rng); }
    std::vector<double> X(nn, 0.0), Xf(nn, 0.0);
    for (int r = 0; r < n; ++r) { X[r * n + r] = 1.0 / Q2[r]; Xf[r * n + r] = 1.0 / Qf2[r]; }

    // ---- Y blocks + de-duplication
    std::vector<std::vector<double>> blocks; std::vector<int> idxD, idx1, idx2;
    fmpc_host_y_blocks(n, T, var2, has_xf != 0, a1, a2, X, Xf, blocks, idxD, idx1, idx2);
    if ((int)idxD.size() != nb) return fail("idxD size");
    // interior stages share their blocks: at most 3 distinct diagonal, 3 off-diagonal-1 and 1 off-diagonal-2 blocks, plus the terminal ones
    if (blocks.size() > 12) return fail("de-duplication: too many unique blocks");
    for (int i = 0; i < nb; ++i) {
        if (idxD[i] < 0 || idxD[i] >= (int)blocks.size()) return fail("idxD range");
        const bool e1 = (i + 1 < T) || (has_xf && i == T - 1), e2 = var2 && i + 2 < T;
        if ((idx1[i] >= 0) != e1) return fail("idx1 presence");
        if ((idx2[i] >= 0) != e2) return fail("idx2 presence");
    }
    if (T >= 6 && (idxD[3] != idxD[4] || idx1[2] != idx1[3])) return fail("interior stages must share their blocks");
    std::vector<double> yall;
    for (auto& b : blocks) yall.insert(yall.end(), b.begin(), b.end());

    // ---- panel-path builder
    FmpcPanelIn In;
    In.n = n; In.m = m; In.T = T; In.nb = nb; In.mp = mp; In.var_order = var_order;
    size_t o_dump = 0, total = 0; int dz_len = 0;
    fmpc_host_panel_layout(n, m, T, nb, mp, In, &o_dump, &total, &dz_len);
    if (!(In.pool_doubles == o_dump && total > o_dump && dz_len > 0)) return fail("pool layout");
    In.umax = umax.data(); In.umin = umin.data(); In.umid = umid.data(); In.xmid = xmid.data(); In.R2 = R2.data(); In.rl = rl.data();
    In.Q2 = Q2.data(); In.Qf2 = Qf2.data(); In.ql = ql.data(); In.qfl = qfl.data(); In.xf = xf.data();
    In.bt = bt.data(); In.a1 = a1.data(); In.a2 = a2.data(); In.blocks = yall.data();
    In.idxD = idxD.data(); In.idx1 = idx1.data(); In.idx2 = idx2.data();
    const double k = 1e-2;
    FmpcPanelOut Out;
    if (fmpc_host_build_panel(In, k, Out) != 0 || !Out.valid) return fail("panel builder did not produce a valid result");
    if (Out.pool.size() != In.pool_doubles) return fail("pool size");
    for (double v : Out.pool) if (!(v == v) || fabs(v) > 1e300) return fail("non-finite value in the pool");
    if (Out.nsf < 0 || Out.nsb < 0 || Out.nsf > FP_MAX_STEPS(nb) || Out.nsb > FP_MAX_STEPS(nb)) return fail("step counts");
    if ((nb > 1) != (Out.nsf > 0) || (nb > 1) != (Out.nsb > 0)) return fail("a single block row has no edges; more have");
    if (Out.nimg > In.limg_cap) return fail("image capacity");

    // ---- assemble Y (dense, long double) independently and solve Y nu = rhs by dense Cholesky
    const int N = nb * n;
    std::vector<ld> Y((size_t)N * N, 0.0L), G(nn, 0.0L);
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            ld t = 0.0L;
            for (int j = 0; j < m; ++j) {
                const double sp = umax[j] - umid[j], sm = umid[j] - umin[j];
                const double hc = k * (1.0 / (sp * sp) + 1.0 / (sm * sm));
                t += (ld)bt[(size_t)j * n + a] * (ld)(1.0 / (R2[j] + hc)) * (ld)bt[(size_t)j * n + b];
            }
            G[a * n + b] = t;
        }
    auto put = [&](int i, int j, const double* blk) {       // Y_ij = blk, Y_ji = blk'
        for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) { Y[(size_t)(i * n + r) * N + j * n + c] = blk[r * n + c]; Y[(size_t)(j * n + c) * N + i * n + r] = blk[r * n + c]; }
    };
    for (int i = 0; i < nb; ++i) {
        put(i, i, yall.data() + (size_t)idxD[i] * nn);
        if (i < T) for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) Y[(size_t)(i * n + r) * N + i * n + c] += G[r * n + c];
        if (idx1[i] >= 0) put(i, i + 1, yall.data() + (size_t)idx1[i] * nn);
        if (idx2[i] >= 0) put(i, i + 2, yall.data() + (size_t)idx2[i] * nn);
    }
    std::vector<ld> rhs(N), ref(N);
    for (int i = 0; i < N; ++i) rhs[i] = N01(rng);
    {
        std::vector<ld> L((size_t)N * N, 0.0L);
        for (int c = 0; c < N; ++c) {
            ld d = Y[(size_t)c * N + c];
            for (int q = 0; q < c; ++q) d -= L[(size_t)c * N + q] * L[(size_t)c * N + q];
            if (!(d > 0.0L)) return fail("reference Cholesky: Y not positive definite");
            L[(size_t)c * N + c] = sqrtl(d);
            for (int r = c + 1; r < N; ++r) {
                ld t = Y[(size_t)r * N + c];
                for (int q = 0; q < c; ++q) t -= L[(size_t)r * N + q] * L[(size_t)c * N + q];
                L[(size_t)r * N + c] = t / L[(size_t)c * N + c];
            }
        }
        std::vector<ld> y(N);
        for (int r = 0; r < N; ++r) { ld t = rhs[r]; for (int q = 0; q < r; ++q) t -= L[(size_t)r * N + q] * y[q]; y[r] = t / L[(size_t)r * N + r]; }
        for (int r = N - 1; r >= 0; --r) { ld t = y[r]; for (int q = r + 1; q < N; ++q) t -= L[(size_t)q * N + r] * ref[q]; ref[r] = t / L[(size_t)r * N + r]; }
    }

    // ---- what the device does with the builder's output
    const double* simg = Out.pool.data() + In.o_simg;            // Linv_i, standard image: [(I*7 + ks)*64 + l] = M[16 I + (l & 15)][4 ks + (l >> 4)]
    const double* limg = Out.pool.data() + In.o_limg;            // lane-major: [(I*64 + l)*8 + ks]
    auto apply_std = [&](const double* img, const std::vector<ld>& v, std::vector<ld>& out) {
        for (int I = 0; I < 2; ++I) for (int ks = 0; ks < FP_KS; ++ks) for (int l = 0; l < 64; ++l) {
            const int r = 16 * I + (l & 15), c = 4 * ks + (l >> 4);
            if (r < n && c < n) out[r] += (ld)img[(I * FP_KS + ks) * 64 + l] * v[c];
        }
    };
    auto apply_lane = [&](const double* img, const std::vector<ld>& v, std::vector<ld>& out) {
        for (int I = 0; I < 2; ++I) for (int l = 0; l < 64; ++l) for (int ks = 0; ks < FP_KS; ++ks) {
            const int r = 16 * I + (l & 15), c = 4 * ks + (l >> 4);
            if (r < n && c < n) out[r] += (ld)img[(I * 64 + l) * 8 + ks] * v[c];
        }
    };
    std::vector<std::vector<ld>> Yv(nb, std::vector<ld>(n, 0.0L));
    for (int i = 0; i < nb; ++i) {                               // S1: y_i = Linv_i rhs_i
        std::vector<ld> v(rhs.begin() + i * n, rhs.begin() + (i + 1) * n);
        apply_std(simg + (size_t)i * FP_IMG, v, Yv[i]);
    }
    auto sweep = [&](const int* sched, int nsteps, const char* name) -> int {
        std::vector<int> written(nb, 0);                         // step in which a stage was last a target
        for (int st = 0; st < nsteps; ++st) {
            const int* row = sched + (size_t)st * FP_STEP_INTS;
            std::vector<std::vector<ld>> add(4, std::vector<ld>(n, 0.0L));
            int tg[4] = {-1, -1, -1, -1};
            for (int gi = 0; gi < 4; ++gi) {
                const int* e0 = row + (2 * gi) * 3; const int* e1 = row + (2 * gi + 1) * 3;
                if (e0[2] < 0) continue;
                if (e0[0] != e1[0] || e0[1] != e1[1] || e0[2] != e1[2]) { fprintf(stderr, "%s: wave pair mismatch\n", name); return 1; }
                const int tgt = e0[0], src = e0[1], img = e0[2];
                if (tgt < 0 || tgt >= nb || src < 0 || src >= nb || img <= nb || img >= Out.nimg) { fprintf(stderr, "%s: entry out of range\n", name); return 1; }
                for (int q = 0; q < gi; ++q) if (tg[q] == tgt) { fprintf(stderr, "%s: two edges into one target in a step\n", name); return 1; }
                if (written[src] > st) { fprintf(stderr, "%s: source not final\n", name); return 1; }
                tg[gi] = tgt;
                apply_lane(limg + (size_t)img * FP_IMGL, Yv[src], add[gi]);
            }
            for (int gi = 0; gi < 4; ++gi) if (tg[gi] >= 0) { for (int r = 0; r < n; ++r) Yv[tg[gi]][r] += add[gi][r]; written[tg[gi]] = st + 1; }
            // a source used in this step must not be a target of this or a later step: checked through `written` on later use
            for (int gi = 0; gi < 4; ++gi) if (tg[gi] >= 0) for (int q = 0; q < 4; ++q) {
                const int* e = row + (2 * q) * 3;
                if (e[2] >= 0 && e[1] == tg[gi]) { fprintf(stderr, "%s: a stage is source and target in the same step\n", name); return 1; }
            }
        }
        return 0;
    };
    if (sweep(Out.sched.data(), Out.nsf, "forward sweep")) return 1;
    for (int i = 0; i < nb; ++i) {                               // S3: y_i <- Linv_i' y_i  (lane-major image id = stage)
        std::vector<ld> v = Yv[i], o(n, 0.0L);
        apply_lane(limg + (size_t)i * FP_IMGL, v, o);
        Yv[i] = o;
    }
    if (sweep(Out.sched.data() + (size_t)FP_MAX_STEPS(nb) * FP_STEP_INTS, Out.nsb, "backward sweep")) return 1;
    ld err = 0.0L, nrm = 0.0L;
    for (int i = 0; i < nb; ++i) for (int r = 0; r < n; ++r) { const ld d = Yv[i][r] - ref[i * n + r]; err += d * d; nrm += ref[i * n + r] * ref[i * n + r]; }
    const double rel = (double)sqrtl(err / nrm);
    printf("n %d m %d T %d xf %d var %d: %zu unique blocks, %d images, %d / %d sweep steps, nu vs dense solve: rel. error %.2e\n",
           n, m, T, has_xf, var_order, blocks.size(), Out.nimg, Out.nsf, Out.nsb, rel);
    if (!(rel <= 1e-10)) return fail("sweeps through the built factor do not reproduce the dense solve");
    // a zero image (id nb) must be all zero
    for (int q = 0; q < FP_IMGL; ++q) if (limg[(size_t)nb * FP_IMGL + q] != 0.0) return fail("zero image");

    // ---- first-move form (fmpc_host_build_first_move): sizes, symmetry of E / Ep, and the circulant halves the kernel reads:
    //      sum_r d_r sum_j Ec[j][r] d[(r + j) mod nc] == d'E d  (a random J: the identity does not depend on what J means)
    {
        const int nc = 4 * n, H = nc / 2 + 1;
        std::vector<double> J((size_t)nb * n * nc), nuc((size_t)nb * n), m1((size_t)T * nn), m2((size_t)T * nn);
        for (auto& v : J) v = 0.1 * N01(rng);
        for (auto& v : nuc) v = 0.1 * N01(rng);
        for (auto& v : m1) v = 0.3 * N01(rng);
        for (auto& v : m2) v = 0.3 * N01(rng);
        FmpcFirstIn Fi;
        Fi.n = n; Fi.m = m; Fi.T = T; Fi.nb = nb; Fi.var2 = var2 ? 1 : 0; Fi.has_xf = has_xf;
        Fi.bt = bt.data(); Fi.umax = umax.data(); Fi.umin = umin.data(); Fi.umid = umid.data(); Fi.xmid = xmid.data(); Fi.R2 = R2.data(); Fi.rl = rl.data();
        Fi.a1 = a1.data(); Fi.a2 = a2.data(); Fi.m1 = m1.data(); Fi.m2 = m2.data(); Fi.xf = xf.data(); Fi.J = J.data(); Fi.nuc = nuc.data(); Fi.k = k;
        FmpcFirstOut Fo;
        fmpc_host_build_first_move(Fi, Fo);
        if (Fo.nc != nc || Fo.K0t.size() != (size_t)nc * m || Fo.u0c.size() != (size_t)m || Fo.E.size() != (size_t)nc * nc || Fo.Ep.size() != (size_t)nc * nc ||
            Fo.e.size() != (size_t)nc || Fo.ep.size() != (size_t)nc || Fo.Ec.size() != (size_t)H * nc || Fo.Epc.size() != (size_t)H * nc) return fail("first-move form: sizes");
        for (int pass = 0; pass < 2; ++pass) {
            const std::vector<double>& F = pass ? Fo.Ep : Fo.E; const std::vector<double>& Fc = pass ? Fo.Epc : Fo.Ec;
            ld asym = 0.0L, nrmF = 0.0L;
            for (int r = 0; r < nc; ++r) for (int c = 0; c < nc; ++c) { const ld d = (ld)F[(size_t)r * nc + c] - (ld)F[(size_t)c * nc + r]; asym += d * d; nrmF += (ld)F[(size_t)r * nc + c] * F[(size_t)r * nc + c]; }
            if (!(nrmF > 0.0L) || sqrtl(asym / nrmF) > 1e-12L) return fail("first-move form: E / Ep not symmetric");
            std::vector<ld> d(nc);
            for (auto& v : d) v = N01(rng);
            ld full = 0.0L, circ = 0.0L, scale = 0.0L;
            for (int r = 0; r < nc; ++r) for (int c = 0; c < nc; ++c) { const ld t = d[r] * (ld)F[(size_t)r * nc + c] * d[c]; full += t; scale += fabsl(t); }
            for (int r = 0; r < nc; ++r) { ld srow = 0.0L; for (int j = 0; j < H; ++j) srow += (ld)Fc[(size_t)j * nc + r] * d[(r + j) % nc]; circ += d[r] * srow; }
            if (fabsl(full - circ) > 1e-12L * scale) { fprintf(stderr, "d'Ed %.17Lg vs circulant %.17Lg\n", full, circ); return fail("first-move form: circulant half does not reproduce d'E d"); }
        }
        // ---- the same form as matrix-core images over many realisations (fmpc_host_build_loop_images; fmpc_kernel_loopu0.hip), in
        // both column orders: decoded exactly as the kernels use them -- k-steps 4 t .. ks - 1 of row tile t for the forms (the
        // block-upper triangle), all k-steps for the first moves -- they must reproduce u0c + K0 d, d'E d + 2 e'd, d'Ep d - 2 ep'd
        for (int fused = 0; fused < 2; ++fused) {
            const int ks = 28, kc = 4 * ks, cst = fused ? kc - 1 : nc;
            FmpcLoopImages LI;
            fmpc_host_build_loop_images(Fo, n, m, ks, fused != 0, bt.data(), LI);
            const int mt = (m + 15) / 16;
            if (LI.imgU.size() != (size_t)mt * ks * 64 || LI.imgE.size() != (size_t)(kc / 16) * ks * 64 || LI.imgEp.size() != LI.imgE.size()) return fail("loop images: sizes");
            if (fused ? LI.imgB.size() != (size_t)2 * ((m + 3) / 4) * 64 : !LI.imgB.empty()) return fail("loop images: B images");
            std::vector<ld> d(nc), dd(kc, 0.0L);
            for (auto& v : d) v = N01(rng);
            for (int c = 0; c < nc; ++c) dd[fused ? ks * (c / n) + c % n : c] = d[c];
            dd[cst] = 1.0L;
            auto at = [&](const std::vector<double>& img, int row, int col) { return (ld)img[((size_t)(row / 16) * ks + col / 4) * 64 + (col % 4) * 16 + row % 16]; };
            for (int j = 0; j < m; ++j) {
                ld want = Fo.u0c[j], got = 0.0L, sc = fabsl(want);
                for (int c = 0; c < nc; ++c) { want += (ld)Fo.K0t[(size_t)c * m + j] * d[c]; sc += fabsl((ld)Fo.K0t[(size_t)c * m + j] * d[c]); }
                for (int c = 0; c < kc; ++c) got += at(LI.imgU, j, c) * dd[c];
                if (fabsl(want - got) > 1e-13L * (sc + 1e-300L)) return fail("loop images: first moves");
            }
            for (int pass = 0; pass < 2; ++pass) {
                const std::vector<double>& F = pass ? Fo.Ep : Fo.E; const std::vector<double>& lin = pass ? Fo.ep : Fo.e;
                const std::vector<double>& img = pass ? LI.imgEp : LI.imgE;
                ld want = 0.0L, got = 0.0L, sc = 0.0L;
                for (int r = 0; r < nc; ++r) {
                    for (int c = 0; c < nc; ++c) { const ld t = d[r] * (ld)F[(size_t)r * nc + c] * d[c]; want += t; sc += fabsl(t); }
                    const ld t = (pass ? -2.0L : 2.0L) * (ld)lin[r] * d[r]; want += t; sc += fabsl(t);
                }
                for (int row = 0; row < kc; ++row)
                    for (int c = 16 * (row / 16); c < kc; ++c) got += dd[row] * at(img, row, c) * dd[c];      // k-steps 4 t .. ks - 1 only
                if (fabsl(want - got) > 1e-12L * sc) { fprintf(stderr, "form %d fused %d: %.17Lg vs %.17Lg\n", pass, fused, want, got); return fail("loop images: triangular form"); }
                for (int c = 0; c < kc; ++c) if (at(img, cst, c) != 0.0L) return fail("loop images: the row of the constant must be zero");
            }
            if (fused)
                for (int q = 0; q < n; ++q)
                    for (int c = 0; c < m; ++c)
                        if (LI.imgB[((size_t)(q / 16) * ((m + 3) / 4) + c / 4) * 64 + (c % 4) * 16 + q % 16] != bt[(size_t)c * n + q]) return fail("loop images: B");
        }
    }

    // ---- estimator builders: G = pinv(A'A) A' (full rank: G A = I; a repeated column: the minimum-norm solution treats both alike),
    //      DFT factor images (unit modulus inside the window, zero padding, the centre sample is 1)
    {
        const int pe = 60, nxe = 7;
        std::vector<double> A((size_t)pe * nxe), G;
        for (auto& v : A) v = N01(rng);
        if (fmpc_host_estimator_gain(A.data(), pe, nxe, G) != nxe || G.size() != (size_t)nxe * pe) return fail("estimator gain: rank / size");
        for (int a = 0; a < nxe; ++a)
            for (int b = 0; b < nxe; ++b) {
                ld s = 0.0L;
                for (int i = 0; i < pe; ++i) s += (ld)G[(size_t)a * pe + i] * (ld)A[(size_t)b * pe + i];
                if (fabsl(s - (a == b ? 1.0L : 0.0L)) > 1e-12L) return fail("estimator gain: G A != I");
            }
        for (int i = 0; i < pe; ++i) A[(size_t)6 * pe + i] = A[(size_t)5 * pe + i];           // columns 5 and 6 equal: rank 6
        if (fmpc_host_estimator_gain(A.data(), pe, nxe, G) != nxe - 1) return fail("estimator gain: rank of a deficient model");
        for (int i = 0; i < pe; ++i) if (fabs(G[(size_t)5 * pe + i] - G[(size_t)6 * pe + i]) > 1e-12) return fail("estimator gain: minimum norm");
        std::vector<double> img;
        const int L = 128, dw = 31, first = 49;
        fmpc_host_estimator_dft_images(L, dw, first, img);
        if (img.size() != (size_t)(L / 4) * 256) return fail("dft images: size");
        for (int y = 0; y < L; ++y)
            for (int j = 0; j < 32; ++j) {
                const size_t base = (((size_t)(y / 4) * 2 + (j / 16)) * 2) * 64 + (size_t)(y % 4) * 16 + (j % 16);
                const double re = img[base], im = img[base + 64];
                if (j >= dw) { if (re != 0.0 || im != 0.0) return fail("dft images: padding"); continue; }
                if (fabs(re * re + im * im - 1.0) > 1e-14) return fail("dft images: modulus");
                const double ang = -2.0 * M_PI * (double)(first + j - L / 2) * (double)(y - L / 2) / L;
                if (fabs(re - cos(ang)) > 1e-12 || fabs(im - sin(ang)) > 1e-12) return fail("dft images: value");
            }
    }

    // ---- the model cache of the one-shot entry: capacity 4, least recently used goes first
    {
        int destroyed = 0, last = -1;
        FmpcLru<int> C(4);
        auto kill = [&](int h) { ++destroyed; last = h; };
        for (int q = 0; q < 4; ++q) {
            std::vector<double> key; double v = q; fmpc_host_key_push(key, &v, 1); fmpc_host_key_push(key, nullptr, 3);
            C.touch(C.insert(std::move(key), q, kill));
        }
        std::vector<double> k1; { double v = 1.0; fmpc_host_key_push(k1, &v, 1); fmpc_host_key_push(k1, nullptr, 3); }
        std::vector<double> k0; { double v = 0.0; fmpc_host_key_push(k0, &v, 1); fmpc_host_key_push(k0, nullptr, 3); }
        auto* e = C.find(k0);
        if (!e || e->h != 0) return fail("cache lookup");
        C.touch(e);                                               // 0 is now the most recent; 1 the oldest
        std::vector<double> k9; { double v = 9.0; fmpc_host_key_push(k9, &v, 1); fmpc_host_key_push(k9, nullptr, 3); }
        C.touch(C.insert(std::move(k9), 9, kill));
        if (destroyed != 1 || last != 1 || C.find(k1) != nullptr || C.find(k0) == nullptr || C.items.size() != 4) return fail("cache eviction order");
        std::vector<double> kn; { double v = 0.0; fmpc_host_key_push(kn, &v, 1); double w3[3] = {0, 0, 0}; fmpc_host_key_push(kn, w3, 3); }
        if (C.find(kn) != nullptr) return fail("NULL and present arguments must give different keys");
        C.clear(kill);
        if (destroyed != 5 || !C.items.empty()) return fail("cache clear");
    }
    return 0;
}
