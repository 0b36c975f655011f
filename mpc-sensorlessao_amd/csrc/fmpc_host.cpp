// Host-only builders (see fmpc_host.h).  Plain C++: also compiled by g++ with the address and undefined-behaviour
// sanitizers (tests/host_san).
#include "fmpc_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

void fmpc_host_add_AXBt(std::vector<double>& out, const std::vector<double>& A, const std::vector<double>& X,
                        const std::vector<double>& B, int n, double sign) {
    std::vector<double> AX((size_t)n * n, 0.0);
    for (int a = 0; a < n; ++a)
        for (int c = 0; c < n; ++c) {
            double t = 0.0;
            for (int k = 0; k < n; ++k) t += A[a * n + k] * X[k * n + c];
            AX[a * n + c] = t;
        }
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            double t = 0.0;
            for (int c = 0; c < n; ++c) t += AX[a * n + c] * B[b * n + c];
            out[a * n + b] += sign * t;
        }
}

void fmpc_host_y_blocks(int n, int T, bool var2, bool has_xf, const std::vector<double>& a1, const std::vector<double>& a2,
                        const std::vector<double>& X, const std::vector<double>& Xf,
                        std::vector<std::vector<double>>& blocks, std::vector<int>& idxD, std::vector<int>& idx1, std::vector<int>& idx2) {
    const int nn = n * n, nb = T + (has_xf ? 1 : 0);
    blocks.clear();
    idxD.assign(nb, -1); idx1.assign(nb, -1); idx2.assign(nb, -1);
    auto intern = [&](const std::vector<double>& blk) {
        for (size_t k = 0; k < blocks.size(); ++k)
            if (memcmp(blocks[k].data(), blk.data(), nn * sizeof(double)) == 0) return (int)k;
        blocks.push_back(blk);
        return (int)blocks.size() - 1;
    };
    auto Xj = [&](int j) -> const std::vector<double>& { return j == T ? Xf : X; };
    std::vector<double> eye(nn, 0.0);
    for (int a = 0; a < n; ++a) eye[a * n + a] = 1.0;
    for (int i = 0; i < T; ++i) {
        std::vector<double> d = Xj(i + 1);
        if (i >= 1) fmpc_host_add_AXBt(d, a1, Xj(i), a1, n, 1.0);
        if (i >= 2 && var2) fmpc_host_add_AXBt(d, a2, Xj(i - 1), a2, n, 1.0);
        idxD[i] = intern(d);
        if (i + 1 < T) {
            std::vector<double> o(nn, 0.0);
            fmpc_host_add_AXBt(o, eye, Xj(i + 1), a1, n, -1.0);              // -X_{i+1} A1'
            if (i >= 1 && var2) fmpc_host_add_AXBt(o, a1, Xj(i), a2, n, 1.0);
            idx1[i] = intern(o);
        }
        if (i + 2 < T && var2) {
            std::vector<double> o(nn, 0.0);
            fmpc_host_add_AXBt(o, eye, Xj(i + 1), a2, n, -1.0);              // -X_{i+1} A2'
            idx2[i] = intern(o);
        }
    }
    if (has_xf) {
        idxD[T] = intern(Xf);
        idx1[T - 1] = idxD[T];
    }
}

void fmpc_host_panel_layout(int n, int m, int T, int nb, int mp, FmpcPanelIn& L, size_t* o_dump, size_t* total, int* dz_len) {
    const FpVec V = fp_vec_layout(nb, T);
    size_t o = 0;
    L.o_simg = o; o += (size_t)nb * FP_IMG + 3 * FP_IMG;          // Linv per stage, then the prediction images
    L.limg_cap = 9 * nb + 2;                                        // Linv' per stage, a zero image, the edges
    L.o_limg = o; o += (size_t)L.limg_cap * FP_IMGL;
    L.o_bt = o;   o += (size_t)(mp / 16) * FP_KS * 64;
    L.o_aimg = o; o += 5 * FP_IMG;
    L.o_vec = o;  o += V.total;
    L.o_ucon = o; o += 4 * (size_t)mp;
    *dz_len = fd_lds_layout(mp).total_next;
    L.o_dz = o;   o += (size_t)*dz_len;
    L.pool_doubles = o;
    *o_dump = o;  o += (size_t)T * (n + m) + (size_t)nb * n;
    *total = o;
}

// Constants of the panel kernel for barrier weight k (fmpc_panel_layout.h): Y at the cold start is assembled from the
// de-duplicated blocks and G = B diag(wc) B', factored in long double by a block-sparse Cholesky in a twisted elimination
// order, and every operator the device needs becomes an MFMA A-operand image; a list scheduler turns the edges of the
// two sweeps into steps.  30 stages x 27^3: microseconds.
int fmpc_host_build_panel(const FmpcPanelIn& In, double k, FmpcPanelOut& Out) {
    typedef long double ld;
    const int n = In.n, m = In.m, T = In.T, nb = In.nb, mp = In.mp;
    const int nn = n * n;
    Out.valid = 0; Out.nsf = 0; Out.nsb = 0;
    std::vector<double>& pool = Out.pool;
    pool.assign(In.pool_doubles, 0.0);
    auto image = [&](const std::vector<ld>& M, double sign, double* out) {      // standard A-operand image of M (n x n row-major)
        for (int I = 0; I < 2; ++I)
            for (int ks = 0; ks < FP_KS; ++ks)
                for (int l = 0; l < 64; ++l) {
                    const int r = 16 * I + (l & 15), c = 4 * ks + (l >> 4);
                    out[(I * FP_KS + ks) * 64 + l] = (r < n && c < n) ? (double)(sign * M[r * n + c]) : 0.0;
                }
    };
    auto limage = [&](const std::vector<ld>& M, double sign, double* out) {     // the same, lane-major (FP_IMGL doubles)
        for (int I = 0; I < 2; ++I)
            for (int l = 0; l < 64; ++l)
                for (int ks = 0; ks < 8; ++ks) {
                    const int r = 16 * I + (l & 15), c = 4 * ks + (l >> 4);
                    out[(I * 64 + l) * 8 + ks] = (ks < FP_KS && r < n && c < n) ? (double)(sign * M[r * n + c]) : 0.0;
                }
    };
    // ---- u constants (needed for G = B diag(wc) B')
    double* uc = pool.data() + In.o_ucon;
    double sa_cu = 0.0;
    for (int j = 0; j < m; ++j) {
        const double sp = In.umax[j] - In.umid[j], sm = In.umid[j] - In.umin[j];
        const double dp = 1.0 / sp, dm = 1.0 / sm;
        const double hc = k * (dp * dp + dm * dm);
        const double cu = In.R2[j] * In.umid[j] + In.rl[j] + k * (dp - dm);
        uc[j] = cu; uc[mp + j] = 1.0 / (In.R2[j] + hc); uc[2 * mp + j] = hc; uc[3 * mp + j] = In.umid[j];
        sa_cu += cu * cu;
    }
    const double* bt = In.bt;                           // bt[c*n + r] = B[r][c]
    // ---- Y = C Phi^-1 C' at the cold start, block by block: Y_ii = const + G (i < T), Y_{i,i+1}, Y_{i,i+2}
    std::vector<ld> G(nn, 0.0L);
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            ld t = 0.0L;
            for (int j = 0; j < m; ++j) t += (ld)bt[(size_t)j * n + a] * (ld)uc[mp + j] * (ld)bt[(size_t)j * n + b];
            G[a * n + b] = t;
        }
    // Elimination order ("twisted" factorisation): the stages 0 .. hs-1 top-down, then nb-1 .. hs+2 bottom-up, then
    // hs, hs+1.  The two chains are independent until the middle, which halves the serial length of both sweeps.
    const int hs = nb >= 8 ? nb / 2 - 1 : nb;
    std::vector<int> perm, rank(nb);
    for (int i = 0; i < (hs < nb ? hs : nb); ++i) perm.push_back(i);
    if (hs < nb) { for (int i = nb - 1; i >= hs + 2; --i) perm.push_back(i); perm.push_back(hs); perm.push_back(hs + 1); }
    for (int a = 0; a < nb; ++a) rank[perm[a]] = a;
    // P[a][b] (a <= b in elimination rank): the block Y[perm[a]][perm[b]]
    std::vector<std::vector<std::vector<ld>>> Pm(nb, std::vector<std::vector<ld>>(nb));
    auto yblock = [&](int i, int j, std::vector<ld>& out) -> bool {     // Y_ij (|i-j| <= 2), false if structurally zero
        out.assign(nn, 0.0L);
        const int lo = i < j ? i : j, d = i < j ? j - i : i - j;
        int id = -1;
        if (d == 0) id = In.idxD[lo]; else if (d == 1) id = In.idx1[lo]; else if (d == 2) id = In.idx2[lo];
        if (d > 2 || id < 0) return false;
        const double* src = In.blocks + (size_t)id * nn;
        for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) out[r * n + c] = i <= j ? (ld)src[r * n + c] : (ld)src[c * n + r];
        if (d == 0 && i < T) for (int q = 0; q < nn; ++q) out[q] += G[q];
        return true;
    };
    for (int a = 0; a < nb; ++a)
        for (int b = a; b < nb; ++b) {
            std::vector<ld> blk;
            if (yblock(perm[a], perm[b], blk)) Pm[a][b] = blk;
        }
    std::vector<std::vector<ld>> Linv(nb, std::vector<ld>(nn, 0.0L));        // by rank
    std::vector<std::vector<std::vector<ld>>> Um(nb, std::vector<std::vector<ld>>(nb));   // U[a][b] = L_a^-1 P[a][b], b > a
    for (int a = 0; a < nb; ++a) {
        // L = chol(P[a][a]) (lower), X = L^-1
        std::vector<ld> Lm(nn, 0.0L);
        const std::vector<ld>& S = Pm[a][a];
        for (int c = 0; c < n; ++c) {
            ld d = S[c * n + c];
            for (int q = 0; q < c; ++q) d -= Lm[c * n + q] * Lm[c * n + q];
            if (!(d > 0.0L)) return 0;                          // not PD at the start point: the exact path reports it
            const ld ldiag = sqrtl(d);
            Lm[c * n + c] = ldiag;
            for (int r = c + 1; r < n; ++r) {
                ld t = S[r * n + c];
                for (int q = 0; q < c; ++q) t -= Lm[r * n + q] * Lm[c * n + q];
                Lm[r * n + c] = t / ldiag;
            }
        }
        std::vector<ld>& X = Linv[a];
        for (int c = 0; c < n; ++c) {
            X[c * n + c] = 1.0L / Lm[c * n + c];
            for (int r = c + 1; r < n; ++r) {
                ld sacc = 0.0L;
                for (int q = c; q < r; ++q) sacc += Lm[r * n + q] * X[q * n + c];
                X[r * n + c] = -sacc / Lm[r * n + r];
            }
        }
        for (int b = a + 1; b < nb; ++b) {
            if (Pm[a][b].empty()) continue;
            std::vector<ld> u(nn);
            for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) {
                ld t = 0.0L;
                for (int q = 0; q <= r; ++q) t += X[r * n + q] * Pm[a][b][q * n + c];
                u[r * n + c] = t;
            }
            Um[a][b] = u;
        }
        for (int b1 = a + 1; b1 < nb; ++b1) {
            if (Um[a][b1].empty()) continue;
            for (int b2 = b1; b2 < nb; ++b2) {
                if (Um[a][b2].empty()) continue;
                if (Pm[b1][b2].empty()) Pm[b1][b2].assign(nn, 0.0L);
                std::vector<ld>& dst = Pm[b1][b2];
                const std::vector<ld>& u1 = Um[a][b1]; const std::vector<ld>& u2 = Um[a][b2];
                for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) {
                    ld t = 0.0L;
                    for (int q = 0; q < n; ++q) t += u1[q * n + r] * u2[q * n + c];
                    dst[r * n + c] -= t;
                }
            }
        }
    }
    // ---- images: Linv (standard, by stage), Linv' (lane-major id = stage), a zero image (id nb), one per edge
    std::vector<ld> M(nn);
    double* limg = pool.data() + In.o_limg;
    for (int i = 0; i < nb; ++i) {
        const std::vector<ld>& X = Linv[rank[i]];
        image(X, 1.0, pool.data() + In.o_simg + (size_t)i * FP_IMG);
        for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) M[r * n + c] = X[c * n + r];
        limage(M, 1.0, limg + (size_t)i * FP_IMGL);
    }
    int nimg = nb + 1;                                            // id nb stays all zero
    struct Edge { int tgt, src, img; };
    std::vector<Edge> ef, eb;                                     // forward: y_tgt += img y_src ; backward: nu_tgt += img nu_src
    for (int a = 0; a < nb; ++a)
        for (int b = a + 1; b < nb; ++b) {
            if (Um[a][b].empty()) continue;
            if (nimg + 2 > In.limg_cap) return 0;        // cannot happen for the penta-diagonal structure
            const std::vector<ld>& u = Um[a][b];
            // forward: -Linv_b U_ab'          (target rank b, source rank a)
            const std::vector<ld>& Xb = Linv[b];
            for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) {
                ld t = 0.0L;
                for (int q = 0; q <= r; ++q) t += Xb[r * n + q] * u[c * n + q];
                M[r * n + c] = t;
            }
            limage(M, -1.0, limg + (size_t)nimg * FP_IMGL);
            ef.push_back({perm[b], perm[a], nimg++});
            // backward: -Linv_a' U_ab         (target rank a, source rank b)
            const std::vector<ld>& Xa = Linv[a];
            for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) {
                ld t = 0.0L;
                for (int q = r; q < n; ++q) t += Xa[q * n + r] * u[q * n + c];
                M[r * n + c] = t;
            }
            limage(M, -1.0, limg + (size_t)nimg * FP_IMGL);
            eb.push_back({perm[a], perm[b], nimg++});
        }
    // ---- schedules: list scheduling of the edges, <= 4 targets per step (one wave pair each), one edge per target and step
    std::vector<int>& sched = Out.sched;
    sched.assign((size_t)2 * FP_MAX_STEPS(nb) * FP_STEP_INTS, -1);
    auto schedule = [&](const std::vector<Edge>& E, int* out) -> int {
        const int ne = (int)E.size();
        std::vector<int> done_step(nb, -1), pending(nb, 0), estep(ne, -1);
        for (const Edge& e : E) pending[e.tgt]++;
        for (int i = 0; i < nb; ++i) if (pending[i] == 0) done_step[i] = 0;       // final before the sweep starts
        int left = ne, step = 0;
        while (left > 0) {
            ++step;
            if (step > FP_MAX_STEPS(nb)) return -1;
            int groups = 0; int gt[4];
            int* row = out + (size_t)(step - 1) * FP_STEP_INTS;
            std::vector<int> newly;
            for (int e = 0; e < ne && true; ++e) {
                if (estep[e] >= 0) continue;
                const int sd = done_step[E[e].src];
                if (sd < 0 || sd >= step) continue;                   // source not final before this step
                bool taken = false;                                   // one edge per target and step
                for (int q = 0; q < groups; ++q) if (gt[q] == E[e].tgt) taken = true;
                if (taken || groups == 4) continue;
                const int gi = groups; gt[groups++] = E[e].tgt;
                for (int I = 0; I < 2; ++I) {
                    int* ent = row + (2 * gi + I) * 3;
                    ent[0] = E[e].tgt; ent[1] = E[e].src; ent[2] = E[e].img;
                }
                estep[e] = step; --left;
                if (--pending[E[e].tgt] == 0) newly.push_back(E[e].tgt);
            }
            for (int t : newly) done_step[t] = step;
        }
        return step;
    };
    Out.nsf = schedule(ef, sched.data());
    Out.nsb = schedule(eb, sched.data() + (size_t)FP_MAX_STEPS(nb) * FP_STEP_INTS);
    if (Out.nsf < 0 || Out.nsb < 0) return 0;
    if (getenv("FMPC_PANEL_VERBOSE")) fprintf(stderr, "fastmpc panel: nb %d, split %d, %d images, %d forward / %d backward edges, %d / %d steps\n", nb, hs, nimg, (int)ef.size(), (int)eb.size(), Out.nsf, Out.nsb);
    // ---- the prediction terms of stages 0 and 1 folded into product images: -Linv_0 A1, -Linv_0 A2, -Linv_1 A2
    {
        double* dst = pool.data() + In.o_simg + (size_t)nb * FP_IMG;
        auto prod = [&](const std::vector<ld>& X, const double* A) {
            for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) {
                ld a = 0.0L;
                for (int q = 0; q <= r; ++q) a += X[r * n + q] * (ld)A[q * n + c];
                M[r * n + c] = a;
            }
        };
        prod(Linv[rank[0]], In.a1); image(M, -1.0, dst);
        prod(Linv[rank[0]], In.a2); image(M, -1.0, dst + FP_IMG);
        if (nb > 1) { prod(Linv[rank[1]], In.a2); image(M, -1.0, dst + 2 * FP_IMG); }
    }
    // ---- model images
    {
        double* dst = pool.data() + In.o_bt;
        for (int J = 0; J < mp / 16; ++J)
            for (int ks = 0; ks < FP_KS; ++ks)
                for (int l = 0; l < 64; ++l) {
                    const int c = 16 * J + (l & 15), q = 4 * ks + (l >> 4);
                    dst[(J * FP_KS + ks) * 64 + l] = (c < m && q < n) ? bt[(size_t)c * n + q] : 0.0;
                }
        double* a = pool.data() + In.o_aimg;
        std::vector<ld> A(nn);
        for (int i = 0; i < nn; ++i) A[i] = In.a1[i];
        image(A, 1.0, a + FP_AIMG_A1 * FP_IMG);
        for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) M[r * n + c] = A[c * n + r];
        image(M, 1.0, a + FP_AIMG_A1T * FP_IMG);
        for (int i = 0; i < nn; ++i) A[i] = In.a2[i];
        image(A, 1.0, a + FP_AIMG_A2 * FP_IMG);
        for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) M[r * n + c] = A[c * n + r];
        image(M, 1.0, a + FP_AIMG_A2T * FP_IMG);
        for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) {
            ld t = 0.0L;
            for (int j = 0; j < m; ++j) t += (ld)bt[(size_t)j * n + r] * (ld)bt[(size_t)j * n + c];
            M[r * n + c] = t;
        }
        image(M, 1.0, a + FP_AIMG_BBT * FP_IMG);
    }
    // ---- vectors
    const FpVec V = fp_vec_layout(nb, T);
    double* vec = pool.data() + In.o_vec;
    std::vector<double> cbu(n), bu(n), a1x(n), a2x(n);
    for (int a = 0; a < n; ++a) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0;
        for (int j = 0; j < m; ++j) {
            t0 += bt[(size_t)j * n + a] * uc[mp + j] * uc[j];
            t1 += bt[(size_t)j * n + a] * In.umid[j];
            t2 += bt[(size_t)j * n + a] * uc[j];
        }
        cbu[a] = t0; bu[a] = t1; vec[V.bcu + a] = t2;
        double s1 = 0.0, s2 = 0.0;
        for (int q = 0; q < n; ++q) { s1 += In.a1[a * n + q] * In.xmid[q]; s2 += In.a2[a * n + q] * In.xmid[q]; }
        a1x[a] = s1; a2x[a] = s2;
    }
    std::vector<double> phx((size_t)T * n);
    double rd2_0 = T * sa_cu;
    for (int j = 0; j < T; ++j)
        for (int r = 0; r < n; ++r) {
            const bool last = j + 1 == T;
            const double q2 = last ? In.Qf2[r] : In.Q2[r];
            const double d0 = q2 * In.xmid[r] + (last ? In.qfl[r] : In.ql[r]);
            vec[V.dx0 + j * 32 + r] = d0;
            vec[V.iq + j * 32 + r] = 1.0 / q2;
            phx[(size_t)j * n + r] = d0 / q2;
            vec[V.xc + j * 32 + r] = In.xmid[r] - d0 / q2;
            rd2_0 += d0 * d0;
        }
    const bool var2 = In.var_order == 2;
    for (int i = 0; i < nb; ++i)
        for (int r = 0; r < n; ++r) {
            double cp, c0;
            if (i < T) {
                cp = In.xmid[r] - bu[r] - (i >= 1 ? a1x[r] : 0.0) - (i >= 2 ? a2x[r] : 0.0);
                c0 = phx[(size_t)i * n + r] - cbu[r];
                if (i >= 1) for (int q = 0; q < n; ++q) c0 -= In.a1[r * n + q] * phx[(size_t)(i - 1) * n + q];
                if (i >= 2 && var2) for (int q = 0; q < n; ++q) c0 -= In.a2[r * n + q] * phx[(size_t)(i - 2) * n + q];
            } else {
                cp = In.xmid[r] - In.xf[r];
                c0 = phx[(size_t)(T - 1) * n + r];
            }
            vec[V.cp + i * 32 + r] = cp;
            vec[V.ct + i * 32 + r] = cp - c0;
        }
    // rt_i = Linv_i ct_i and the w-free part of ||r_p||^2
    double rp2c = 0.0;
    for (int i = 0; i < nb; ++i)
        for (int r = 0; r < n; ++r) {
            ld a = 0.0L;
            for (int q = 0; q <= r; ++q) a += Linv[rank[i]][r * n + q] * (ld)vec[V.ct + i * 32 + q];
            vec[V.rt + i * 32 + r] = (double)a;
            if (i >= 2) rp2c += vec[V.cp + i * 32 + r] * vec[V.cp + i * 32 + r];
        }
    Out.rp2c = rp2c;
    Out.rd2_0 = rd2_0;
    {   // LDS image of the d_z kernel, in LDS order
        const FdLds D = fd_lds_layout(mp);
        double* dz = pool.data() + In.o_dz;
        memcpy(dz + D.BT, pool.data() + In.o_bt, (size_t)(mp / 16) * FP_KS * 64 * sizeof(double));
        memcpy(dz + D.A1T, pool.data() + In.o_aimg + FP_AIMG_A1T * FP_IMG, FP_IMG * sizeof(double));
        memcpy(dz + D.A2T, pool.data() + In.o_aimg + FP_AIMG_A2T * FP_IMG, FP_IMG * sizeof(double));
        for (int j = 0; j < mp; ++j) {
            dz[D.UC + j] = -uc[mp + j] * uc[j];
            dz[D.UC + mp + j] = uc[mp + j]; dz[D.UC + 2 * mp + j] = uc[2 * mp + j]; dz[D.UC + 3 * mp + j] = uc[3 * mp + j];
        }
        for (int j = 0; j < mp; ++j) {                               // [c2 | 2R | hp | hm] for the next-exit-test variant
            const bool in = j < m;
            dz[D.UX + j] = in ? In.R2[j] * In.umid[j] + In.rl[j] : 0.0;
            dz[D.UX + mp + j] = in ? In.R2[j] : 0.0;
            dz[D.UX + 2 * mp + j] = in ? In.umax[j] - In.umid[j] : 1.0;
            dz[D.UX + 3 * mp + j] = in ? In.umid[j] - In.umin[j] : 1.0;
        }
        for (int r = 0; r < 32; ++r) {
            dz[D.XQ + r] = vec[V.xc + r]; dz[D.XQ + 32 + r] = vec[V.xc + (T - 1) * 32 + r];
            dz[D.XQ + 64 + r] = vec[V.iq + r]; dz[D.XQ + 96 + r] = vec[V.iq + (T - 1) * 32 + r];
        }
    }
    Out.valid = 1;
    Out.nimg = nimg;
    return 0;
}


void fmpc_host_build_first_move(const FmpcFirstIn& In, FmpcFirstOut& Out) {
    typedef long double ld;
    const int n = In.n, m = In.m, T = In.T, nb = In.nb, nc = 4 * n;
    const bool var2 = In.var2 != 0;
    Out.nc = nc;
    std::vector<double> cu(m), wc(m), av(m);
    for (int j = 0; j < m; ++j) {
        const double sp = In.umax[j] - In.umid[j], sm = In.umid[j] - In.umin[j];
        const double dp = 1.0 / sp, dm = 1.0 / sm;
        const double hc = In.k * (dp * dp + dm * dm);
        cu[j] = In.R2[j] * In.umid[j] + In.rl[j] + In.k * (dp - dm);
        wc[j] = 1.0 / (In.R2[j] + hc);
        av[j] = hc * wc[j];
    }
    const double* bt = In.bt;                                     // bt[c*n + r] = B[r][c]
    // ---- u0 = u0c + K0 d
    Out.K0t.assign((size_t)nc * m, 0.0); Out.u0c.assign(m, 0.0);
    for (int j = 0; j < m; ++j) {
        ld t = 0.0L;
        for (int r = 0; r < n; ++r) t += (ld)bt[(size_t)j * n + r] * (ld)In.nuc[r];
        Out.u0c[j] = (double)((ld)In.umid[j] + (ld)wc[j] * (t - (ld)cu[j]));
        for (int c = 0; c < nc; ++c) {
            ld s = 0.0L;
            for (int r = 0; r < n; ++r) s += (ld)bt[(size_t)j * n + r] * (ld)In.J[(size_t)r * nc + c];
            Out.K0t[(size_t)c * m + j] = (double)((ld)wc[j] * s);
        }
    }
    // ---- ||e||^2 = d'E d + 2 e'd + e0 :  per stage s, g_s = a o (B' nuc_s - cu), G_s = diag(a) B' J_s
    //      E = sum J_s' Ma2 J_s, e = sum J_s' B (a o g_s), e0 = sum |g_s|^2,  Ma2 = B diag(a^2) B'
    std::vector<ld> Ma2((size_t)n * n, 0.0L);
    for (int r = 0; r < n; ++r)
        for (int q = 0; q < n; ++q) {
            ld t = 0.0L;
            for (int j = 0; j < m; ++j) t += (ld)bt[(size_t)j * n + r] * (ld)av[j] * (ld)av[j] * (ld)bt[(size_t)j * n + q];
            Ma2[(size_t)r * n + q] = t;
        }
    std::vector<ld> E((size_t)nc * nc, 0.0L), e(nc, 0.0L), tmp((size_t)n * nc), bg(n), g(m);
    ld e0 = 0.0L;
    for (int s = 0; s < T; ++s) {
        const double* Js = In.J + (size_t)s * n * nc;
        const double* ncs = In.nuc + (size_t)s * n;
        for (int j = 0; j < m; ++j) {
            ld t = 0.0L;
            for (int r = 0; r < n; ++r) t += (ld)bt[(size_t)j * n + r] * (ld)ncs[r];
            g[j] = (ld)av[j] * (t - (ld)cu[j]);
            e0 += g[j] * g[j];
        }
        for (int r = 0; r < n; ++r) {
            ld t = 0.0L;
            for (int j = 0; j < m; ++j) t += (ld)bt[(size_t)j * n + r] * (ld)av[j] * g[j];
            bg[r] = t;
        }
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < nc; ++c) {
                ld t = 0.0L;
                for (int q = 0; q < n; ++q) t += Ma2[(size_t)r * n + q] * (ld)Js[(size_t)q * nc + c];
                tmp[(size_t)r * nc + c] = t;
            }
        for (int c1 = 0; c1 < nc; ++c1) {
            ld t = 0.0L;
            for (int r = 0; r < n; ++r) t += (ld)Js[(size_t)r * nc + c1] * bg[r];
            e[c1] += t;
            for (int c2 = 0; c2 < nc; ++c2) {
                ld u = 0.0L;
                for (int r = 0; r < n; ++r) u += (ld)Js[(size_t)r * nc + c1] * tmp[(size_t)r * nc + c2];
                E[(size_t)c1 * nc + c2] += u;
            }
        }
    }
    // ---- ||r_p||^2 = |cp - Pb d|^2:  rows of stage 0 [A1 | A2 | -M1_0 | -M2_0], stage 1 [A2 | 0 | -M1_1 | -M2_1], i >= 2 [0 | 0 | -M1_i | -M2_i]
    std::vector<ld> Ep((size_t)nc * nc, 0.0L), ep(nc, 0.0L), row(nc);
    ld ep0 = 0.0L;
    std::vector<ld> bu(n, 0.0L), a1x(n, 0.0L), a2x(n, 0.0L);
    for (int r = 0; r < n; ++r) {
        for (int j = 0; j < m; ++j) bu[r] += (ld)bt[(size_t)j * n + r] * (ld)In.umid[j];
        for (int q = 0; q < n; ++q) { a1x[r] += (ld)In.a1[r * n + q] * (ld)In.xmid[q]; a2x[r] += (ld)In.a2[r * n + q] * (ld)In.xmid[q]; }
    }
    for (int i = 0; i < nb; ++i)
        for (int r = 0; r < n; ++r) {
            ld cp;
            for (int c = 0; c < nc; ++c) row[c] = 0.0L;
            if (i < T) {
                cp = (ld)In.xmid[r] - bu[r] - (i >= 1 ? a1x[r] : 0.0L) - (i >= 2 ? a2x[r] : 0.0L);
                if (i == 0) for (int q = 0; q < n; ++q) { row[q] = In.a1[r * n + q]; row[n + q] = var2 ? (ld)In.a2[r * n + q] : 0.0L; }
                if (i == 1 && var2) for (int q = 0; q < n; ++q) row[q] = In.a2[r * n + q];
                for (int q = 0; q < n; ++q) {
                    row[2 * n + q] = -(ld)In.m1[((size_t)i * n + r) * n + q];
                    row[3 * n + q] = -(ld)In.m2[((size_t)i * n + r) * n + q];
                }
            } else {
                cp = (ld)In.xmid[r] - (ld)In.xf[r];
            }
            ep0 += cp * cp;
            for (int c1 = 0; c1 < nc; ++c1) {
                if (row[c1] == 0.0L) continue;
                ep[c1] += row[c1] * cp;
                for (int c2 = 0; c2 < nc; ++c2) Ep[(size_t)c1 * nc + c2] += row[c1] * row[c2];
            }
        }
    auto fro = [](const std::vector<ld>& v) { ld s = 0.0L; for (ld x : v) s += x * x; return (double)sqrtl(s); };
    Out.E.resize(E.size()); Out.Ep.resize(Ep.size()); Out.e.resize(nc); Out.ep.resize(nc);
    for (size_t i = 0; i < E.size(); ++i) { Out.E[i] = (double)E[i]; Out.Ep[i] = (double)Ep[i]; }   // symmetric: [c][r] == [r][c]
    for (int c = 0; c < nc; ++c) { Out.e[c] = (double)e[c]; Out.ep[c] = (double)ep[c]; }
    Out.e0 = (double)e0; Out.ep0 = (double)ep0;
    // circulant halves for the kernel: d'E d = sum_r d_r sum_{j=0}^{nc/2} Ec[j][r] d[(r + j) mod nc]  (every unordered pair once;
    // the pairs at distance nc/2 would come twice: kept for r < nc/2)
    const int H = nc / 2 + 1;
    Out.Ec.assign((size_t)H * nc, 0.0); Out.Epc.assign((size_t)H * nc, 0.0);
    for (int j = 0; j < H; ++j)
        for (int r = 0; r < nc; ++r) {
            const int c = (r + j) % nc;
            const ld wgt = j == 0 ? 1.0L : ((j == nc / 2 && r >= nc / 2) ? 0.0L : 2.0L);
            const ld se = (E[(size_t)r * nc + c] + E[(size_t)c * nc + r]) * 0.5L, sp = (Ep[(size_t)r * nc + c] + Ep[(size_t)c * nc + r]) * 0.5L;
            Out.Ec[(size_t)j * nc + r] = (double)(wgt * se);
            Out.Epc[(size_t)j * nc + r] = (double)(wgt * sp);
        }
    Out.normE = fro(E); Out.norme = fro(e); Out.normEp = fro(Ep); Out.normep = fro(ep);
}

void fmpc_host_mfma_images(const double* M, int rows, int cols, int ks, std::vector<double>& img) {
    const int tiles = (rows + 15) / 16;
    img.assign((size_t)tiles * ks * 64, 0.0);
    for (int t = 0; t < tiles; ++t)
        for (int q = 0; q < ks; ++q)
            for (int g = 0; g < 4; ++g)
                for (int r = 0; r < 16; ++r) {
                    const int row = 16 * t + r, col = 4 * q + g;
                    if (row < rows && col < cols) img[((size_t)t * ks + q) * 64 + g * 16 + r] = M[(size_t)row * cols + col];
                }
}

void fmpc_host_build_loop_images(const FmpcFirstOut& O, int n, int m, int ks, bool fused, const double* bt, FmpcLoopImages& out) {
    const int nc = 4 * n, kc = 4 * ks, cst = fused ? kc - 1 : nc;
    auto col = [&](int c) { return fused ? ks * (c / n) + c % n : c; };          // column of d -> column of the images
    std::vector<double> U((size_t)m * kc, 0.0), E2((size_t)kc * kc, 0.0), Ep2((size_t)kc * kc, 0.0);
    for (int j = 0; j < m; ++j) {
        for (int c = 0; c < nc; ++c) U[(size_t)j * kc + col(c)] = O.K0t[(size_t)c * m + j];
        U[(size_t)j * kc + cst] = O.u0c[j];
    }
    for (int r = 0; r < nc; ++r) {
        const int rr = col(r);
        for (int c = 0; c < nc; ++c) {
            const int cc = col(c);
            const double wgt = cc / 16 > rr / 16 ? 2.0 : (cc / 16 == rr / 16 ? 1.0 : 0.0);
            E2[(size_t)rr * kc + cc] = wgt * O.E[(size_t)r * nc + c];
            Ep2[(size_t)rr * kc + cc] = wgt * O.Ep[(size_t)r * nc + c];
        }
        E2[(size_t)rr * kc + cst] = 2.0 * O.e[r];
        Ep2[(size_t)rr * kc + cst] = -2.0 * O.ep[r];
    }
    fmpc_host_mfma_images(U.data(), m, kc, ks, out.imgU);
    fmpc_host_mfma_images(E2.data(), kc, kc, ks, out.imgE);
    fmpc_host_mfma_images(Ep2.data(), kc, kc, ks, out.imgEp);
    out.imgB.clear();
    if (fused && bt) {
        std::vector<double> Brm((size_t)n * m);
        for (int q = 0; q < n; ++q)
            for (int c = 0; c < m; ++c) Brm[(size_t)q * m + c] = bt[(size_t)c * n + q];
        fmpc_host_mfma_images(Brm.data(), n, m, (m + 3) / 4, out.imgB);
    }
}

void fmpc_host_mfma_a_images(const double* M, int rows, std::vector<double>& img) {
    const int tiles = (rows + 15) / 16;
    img.assign((size_t)tiles * FA_KS * 64, 0.0);
    for (int t = 0; t < tiles; ++t)
        for (int q = 0; q < FA_KS; ++q)
            for (int g = 0; g < 4; ++g)
                for (int r = 0; r < 16; ++r) {
                    const int row = 16 * t + r;
                    if (row < rows) img[((size_t)t * FA_KS + q) * 64 + g * 16 + r] = M[(size_t)row * FA_KC + 4 * q + g];
                }
}

// z+ = zc + Kz d, d = [x0 ; x0_pre], from nu+ = nuc + J d (rows of stage s: u_s, then x_{s+1}):
//   u_s+   = umid + wc o (B' nu+_s - cu)                                            (inf_newton_solver.m:34-35 on the u entries)
//   x_j+   = xmid + X_j (-(2 Q_j xmid + q_j) - nu+_{j-1} + A1' nu+_j + A2' nu+_{j+1} [- nu+_T])   j = 1..T, X_j = (2 Q_j)^-1
void fmpc_host_build_affine(const FmpcAffineIn& In, FmpcAffineOut& Out) {
    typedef long double ld;
    const int n = In.n, m = In.m, T = In.T, s = n + m, nd = 2 * n, ncJ = In.ncJ;
    Out.rows = T * s; Out.tiles = (Out.rows + 15) / 16;
    Out.Kz.assign((size_t)Out.rows * FA_KC, 0.0);
    std::vector<ld> cu(m), wc(m);
    for (int j = 0; j < m; ++j) {
        const ld sp = (ld)In.umax[j] - (ld)In.umid[j], sm = (ld)In.umid[j] - (ld)In.umin[j];
        const ld dp = 1.0L / sp, dm = 1.0L / sm;
        cu[j] = (ld)In.R2[j] * (ld)In.umid[j] + (ld)In.rl[j] + (ld)In.k * (dp - dm);
        wc[j] = 1.0L / ((ld)In.R2[j] + (ld)In.k * (dp * dp + dm * dm));
    }
    // nu+ as an affine function of d: column c < nd of J, column nd = nuc
    auto nuv = [&](int blk, int r, int c) -> ld {
        return c < nd ? (ld)In.J[((size_t)blk * n + r) * ncJ + c] : (ld)In.nuc[(size_t)blk * n + r];
    };
    for (int st = 0; st < T; ++st) {
        for (int j = 0; j < m; ++j) {
            double* row = &Out.Kz[((size_t)st * s + j) * FA_KC];
            for (int c = 0; c <= nd; ++c) {
                ld t = 0.0L;
                for (int r = 0; r < n; ++r) t += (ld)In.bt[(size_t)j * n + r] * nuv(st, r, c);
                if (c == nd) t = (ld)In.umid[j] + wc[j] * (t - cu[j]); else t *= wc[j];
                row[c] = (double)t;
            }
        }
        const int jx = st + 1;                                     // x_jx lives in stage st
        const bool last = jx == T;
        for (int r = 0; r < n; ++r) {
            double* row = &Out.Kz[((size_t)st * s + m + r) * FA_KC];
            const ld q2 = last ? (ld)In.Qf2[r] : (ld)In.Q2[r], ql = last ? (ld)In.qfl[r] : (ld)In.ql[r];
            for (int c = 0; c <= nd; ++c) {
                ld v = -nuv(jx - 1, r, c);
                if (jx < T) for (int q = 0; q < n; ++q) v += (ld)In.a1[q * n + r] * nuv(jx, q, c);
                if (jx + 1 < T) for (int q = 0; q < n; ++q) v += (ld)In.a2[q * n + r] * nuv(jx + 1, q, c);
                if (last && In.has_xf) v -= nuv(T, r, c);
                if (c == nd) v = (ld)In.xmid[r] + (v - (q2 * (ld)In.xmid[r] + ql)) / q2; else v /= q2;
                row[c] = (double)v;
            }
        }
    }
    fmpc_host_mfma_a_images(Out.Kz.data(), Out.rows, Out.img);
    // nu+ = nuc + J d as further rows (requested by callers that want the multipliers)
    Out.nu_rows = In.nb * n; Out.nu_tiles = (Out.nu_rows + 15) / 16;
    std::vector<double> Jn((size_t)Out.nu_rows * FA_KC, 0.0), imgn;
    for (int r = 0; r < Out.nu_rows; ++r) {
        for (int c = 0; c < nd; ++c) Jn[(size_t)r * FA_KC + c] = In.J[(size_t)r * ncJ + c];
        Jn[(size_t)r * FA_KC + nd] = In.nuc[r];
    }
    fmpc_host_mfma_a_images(Jn.data(), Out.nu_rows, imgn);
    Out.img.insert(Out.img.end(), imgn.begin(), imgn.end());
}

// ---- cold-start step with the ramp-rate rows: constants of the Woodbury form (fmpc_host.h)
void fmpc_host_build_ramp_cold(const FmpcRampColdIn& In, FmpcRampColdOut& Out) {
    typedef long double ld;
    const int n = In.n, m = In.m, T = In.T, nb = In.nb, s = n + m, Nz = T * s, nbn = nb * n;
    const bool var2 = In.var2 != 0, has_xf = In.has_xf != 0;
    Out.valid = 0;
    // ---- per actuator: box and ramp terms at the start point, the tridiagonal and its inverse
    std::vector<ld> hb(m), gb(m), erb(m), grb(m);
    for (int c = 0; c < m; ++c) {
        const ld dp = 1.0L / ((ld)In.umax[c] - (ld)In.umid[c]), dm = 1.0L / ((ld)In.umid[c] - (ld)In.umin[c]);
        hb[c] = (ld)In.k * (dp * dp + dm * dm); gb[c] = (ld)In.k * (dp - dm);
        const ld rp = 1.0L / (ld)In.dumax[c], rm = 1.0L / (-(ld)In.dumin[c]);       // slacks of u_j - u_{j-1} = 0
        erb[c] = (ld)In.k * (rp * rp + rm * rm); grb[c] = (ld)In.k * (rp - rm);
    }
    Out.g0.assign((size_t)T * m, 0.0); Out.Gf.assign((size_t)T * T * m, 0.0); Out.hd.assign((size_t)T * m, 0.0);
    Out.erb.resize(m);
    std::vector<ld> Gl((size_t)T * T * m);                        // Gf in long double, same layout
    std::vector<ld> dg(T), lo(T);
    for (int c = 0; c < m; ++c) {
        Out.erb[c] = (double)erb[c];
        // diag_j = 2R + hb + [j >= 1] erb + [j + 1 < T] erb ; off_{j,j+1} = -erb    (stage 0's own ramp term is the per-problem delta)
        for (int j = 0; j < T; ++j) {
            const ld hdj = hb[c] + (j >= 1 ? erb[c] : 0.0L) + (j + 1 < T ? erb[c] : 0.0L);
            Out.hd[(size_t)j * m + c] = (double)hdj;
            ld d = (ld)In.R2[c] + hdj;
            if (j > 0) d -= lo[j - 1] * lo[j - 1] * dg[j - 1];
            if (!(d > 0.0L) || !std::isfinite((double)d)) return;
            dg[j] = d;
            lo[j] = j + 1 < T ? -erb[c] / d : 0.0L;
        }
        // inverse of L D L' (L unit lower bidiagonal with sub-diagonal lo): column by column
        for (int col = 0; col < T; ++col) {
            std::vector<ld> x(T, 0.0L);
            x[col] = 1.0L;
            for (int j = 1; j < T; ++j) x[j] -= lo[j - 1] * x[j - 1];
            for (int j = 0; j < T; ++j) x[j] /= dg[j];
            for (int j = T - 2; j >= 0; --j) x[j] -= lo[j] * x[j + 1];
            for (int j = 0; j < T; ++j) Gl[((size_t)j * T + col) * m + c] = x[j];
        }
        for (int i = 0; i < T; ++i)
            for (int j = 0; j < T; ++j) {                          // (symmetrised: the two triangles agree to rounding)
                const ld v = 0.5L * (Gl[((size_t)i * T + j) * m + c] + Gl[((size_t)j * T + i) * m + c]);
                Out.Gf[((size_t)i * T + j) * m + c] = (double)v;
            }
        for (int j = 0; j < T; ++j) Out.g0[(size_t)j * m + c] = Out.Gf[((size_t)j * T + 0) * m + c];
    }
    for (size_t i = 0; i < Gl.size(); ++i) Gl[i] = (ld)Out.Gf[i];   // (the device applies the rounded values: keep the algebra consistent)
    auto qinv = [&](int jx, int r) -> ld { return 1.0L / (jx == T ? (ld)In.Qf2[r] : (ld)In.Q2[r]); };   // x_jx, jx = 1..T
    // ---- gbar, phibar = Phibar^-1 (-gbar)
    Out.gbar_u.assign((size_t)T * m, 0.0); Out.gbar_x.assign((size_t)T * n, 0.0);
    Out.phib_u.assign((size_t)T * m, 0.0); Out.phib_x.assign((size_t)T * n, 0.0);
    std::vector<ld> gu((size_t)T * m), gx((size_t)T * n), pu((size_t)T * m), px((size_t)T * n);
    for (int j = 0; j < T; ++j) {
        for (int c = 0; c < m; ++c)
            gu[(size_t)j * m + c] = (ld)In.R2[c] * (ld)In.umid[c] + (ld)In.rl[c] + gb[c] + (j >= 1 ? grb[c] : 0.0L) - (j + 1 < T ? grb[c] : 0.0L);
        for (int r = 0; r < n; ++r) {
            const bool last = j + 1 == T;
            gx[(size_t)j * n + r] = last ? (ld)In.Qf2[r] * (ld)In.xmid[r] + (ld)In.qfl[r] : (ld)In.Q2[r] * (ld)In.xmid[r] + (ld)In.ql[r];
        }
    }
    for (int j = 0; j < T; ++j) {
        for (int c = 0; c < m; ++c) {
            ld t = 0.0L;
            for (int i = 0; i < T; ++i) t -= Gl[((size_t)j * T + i) * m + c] * gu[(size_t)i * m + c];
            pu[(size_t)j * m + c] = t;
        }
        for (int r = 0; r < n; ++r) px[(size_t)j * n + r] = -gx[(size_t)j * n + r] * qinv(j + 1, r);
    }
    for (size_t i = 0; i < gu.size(); ++i) { Out.gbar_u[i] = (double)gu[i]; Out.phib_u[i] = (double)pu[i]; }
    for (size_t i = 0; i < gx.size(); ++i) { Out.gbar_x[i] = (double)gx[i]; Out.phib_x[i] = (double)px[i]; }
    // ---- C as sparse rows (column, value); columns: u_j[c] = j s + c, x_{j+1}[r] = j s + m + r
    struct Ent { int col; ld v; };
    std::vector<std::vector<Ent>> Cr(nbn);
    for (int i = 0; i < T; ++i)
        for (int r = 0; r < n; ++r) {
            std::vector<Ent>& row = Cr[(size_t)i * n + r];
            row.push_back({i * s + m + r, 1.0L});
            for (int c = 0; c < m; ++c) row.push_back({i * s + c, -(ld)In.bt[(size_t)c * n + r]});
            if (i >= 1) for (int q = 0; q < n; ++q) row.push_back({(i - 1) * s + m + q, -(ld)In.a1[(size_t)r * n + q]});
            if (var2 && i >= 2) for (int q = 0; q < n; ++q) row.push_back({(i - 2) * s + m + q, -(ld)In.a2[(size_t)r * n + q]});
        }
    if (has_xf) for (int r = 0; r < n; ++r) Cr[(size_t)T * n + r].push_back({(T - 1) * s + m + r, 1.0L});
    auto zstart = [&](int col) -> ld { const int e = col % s; return e < m ? (ld)In.umid[e] : (ld)In.xmid[e - m]; };
    // cpb = C z0 (- xf on the terminal rows), betab = C phibar + cpb
    Out.cpb.assign(nbn, 0.0); Out.betab.assign(nbn, 0.0);
    auto phib_at = [&](int col) -> ld { const int j = col / s, e = col % s; return e < m ? pu[(size_t)j * m + e] : px[(size_t)j * n + (e - m)]; };
    std::vector<ld> cpl(nbn), betal(nbn);
    for (int a = 0; a < nbn; ++a) {
        ld cz = 0.0L, cp = 0.0L;
        for (const Ent& e : Cr[a]) { cz += e.v * zstart(e.col); cp += e.v * phib_at(e.col); }
        if (a >= T * n) cz -= (ld)In.xf[a - T * n];
        cpl[a] = cz; betal[a] = cp + cz;
        Out.cpb[a] = (double)cz; Out.betab[a] = (double)(cp + cz);
    }
    // ---- P = Phibar^-1 C' (Nz x nbn), Ybar = C P
    std::vector<ld> P((size_t)Nz * nbn, 0.0L);
    for (int b = 0; b < nbn; ++b)
        for (const Ent& e : Cr[b]) {
            const int j = e.col / s, el = e.col % s;
            if (el < m) { for (int jj = 0; jj < T; ++jj) P[((size_t)jj * s + el) * nbn + b] += Gl[((size_t)jj * T + j) * m + el] * e.v; }
            else P[(size_t)e.col * nbn + b] += e.v * qinv(j + 1, el - m);
        }
    std::vector<ld> Y((size_t)nbn * nbn, 0.0L);
    for (int a = 0; a < nbn; ++a)
        for (const Ent& e : Cr[a]) {
            const ld* prow = &P[(size_t)e.col * nbn];
            ld* yrow = &Y[(size_t)a * nbn];
            for (int b = 0; b < nbn; ++b) yrow[b] += e.v * prow[b];
        }
    // Cholesky Y = L L' (lower, in place), then Yinv = L^-T L^-1
    for (int j = 0; j < nbn; ++j) {
        ld d = Y[(size_t)j * nbn + j];
        for (int q = 0; q < j; ++q) d -= Y[(size_t)j * nbn + q] * Y[(size_t)j * nbn + q];
        if (!(d > 0.0L) || !std::isfinite((double)d)) return;
        const ld l = sqrtl(d);
        Y[(size_t)j * nbn + j] = l;
        for (int i = j + 1; i < nbn; ++i) {
            ld v = Y[(size_t)i * nbn + j];
            for (int q = 0; q < j; ++q) v -= Y[(size_t)i * nbn + q] * Y[(size_t)j * nbn + q];
            Y[(size_t)i * nbn + j] = v / l;
        }
    }
    std::vector<ld> Yi((size_t)nbn * nbn, 0.0L), x(nbn);
    for (int col = 0; col < nbn; ++col) {
        for (int i = 0; i < nbn; ++i) {
            ld v = i == col ? 1.0L : 0.0L;
            if (i < col) { x[i] = 0.0L; continue; }
            for (int q = col; q < i; ++q) v -= Y[(size_t)i * nbn + q] * x[q];
            x[i] = v / Y[(size_t)i * nbn + i];
        }
        for (int i = nbn - 1; i >= 0; --i) {
            ld v = x[i];
            for (int q = i + 1; q < nbn; ++q) v -= Y[(size_t)q * nbn + i] * x[q];
            x[i] = v / Y[(size_t)i * nbn + i];
        }
        for (int i = 0; i < nbn; ++i) Yi[(size_t)i * nbn + col] = x[i];
    }
    const int ldy = (nbn + 1) & ~1;                               // rows an even number of doubles apart (16-byte loads), pad column zero
    Out.Yinv.assign((size_t)nbn * ldy, 0.0);
    for (int a = 0; a < nbn; ++a)
        for (int b = 0; b < nbn; ++b) Out.Yinv[(size_t)a * ldy + b] = (double)(0.5L * (Yi[(size_t)a * nbn + b] + Yi[(size_t)b * nbn + a]));
    for (int a = 0; a < nbn; ++a)
        for (int b = 0; b < nbn; ++b) Yi[(size_t)a * nbn + b] = (ld)Out.Yinv[(size_t)a * ldy + b];
    // ---- Xi_u0 = P_{u0,:} Yinv (m x nbn);  G = Gf[0][0] - P_{u0,:} Yinv P_{u0,:}';  y0c = (Kbar^-1 (-[gbar; cpb]))_{u0}
    std::vector<ld> Xi((size_t)m * nbn, 0.0L);
    for (int r = 0; r < m; ++r)
        for (int a = 0; a < nbn; ++a) {
            const ld pv = P[(size_t)r * nbn + a];                 // (rows 0 .. m-1 of P are the u_0 rows)
            if (pv == 0.0L) continue;
            for (int b = 0; b < nbn; ++b) Xi[(size_t)r * nbn + b] += pv * Yi[(size_t)a * nbn + b];
        }
    const int ldg = (m + 1) & ~1;                                 // (rows of G and of Xiu0t an even number of doubles apart: 16-byte loads)
    Out.Xiu0t.assign((size_t)T * n * ldg, 0.0);
    for (int col = 0; col < T * n; ++col)
        for (int r = 0; r < m; ++r) Out.Xiu0t[(size_t)col * ldg + r] = (double)Xi[(size_t)r * nbn + col];
    Out.G.assign((size_t)m * ldg, 0.0);
    for (int r = 0; r < m; ++r)
        for (int c = r; c < m; ++c) {
            ld v = r == c ? Gl[((size_t)0 * T + 0) * m + r] : 0.0L;
            for (int b = 0; b < nbn; ++b) v -= Xi[(size_t)r * nbn + b] * P[(size_t)c * nbn + b];
            Out.G[(size_t)r * ldg + c] = (double)v; Out.G[(size_t)c * ldg + r] = (double)v;
        }
    // y0c: f_z = -gbar, f_nu = -cpb:  phi = phibar, nu = Yinv (C phibar + cpb) = Yinv betab, z_u0 = phibar_u0 - P_{u0,:} nu
    Out.y0c.assign(m, 0.0);
    {
        std::vector<ld> nu(nbn, 0.0L);
        for (int a = 0; a < nbn; ++a) { ld v = 0.0L; for (int b = 0; b < nbn; ++b) v += Yi[(size_t)a * nbn + b] * betal[b]; nu[a] = v; }
        for (int r = 0; r < m; ++r) {
            ld v = pu[r];
            for (int b = 0; b < nbn; ++b) v -= P[(size_t)r * nbn + b] * nu[b];
            Out.y0c[r] = (double)v;
        }
    }
    for (double v : Out.Yinv) if (!std::isfinite(v)) return;
    for (double v : Out.G) if (!std::isfinite(v)) return;
    Out.valid = 1;
}

int fmpc_host_estimator_gain(const double* A_s, int p, int nx, std::vector<double>& G) {
    typedef long double ld;
    std::vector<ld> M((size_t)nx * nx, 0.0L), V((size_t)nx * nx, 0.0L);
    for (int a = 0; a < nx; ++a)
        for (int b = a; b < nx; ++b) {
            ld s = 0.0L;
            for (int i = 0; i < p; ++i) s += (ld)A_s[(size_t)a * p + i] * (ld)A_s[(size_t)b * p + i];
            M[(size_t)a * nx + b] = s; M[(size_t)b * nx + a] = s;
        }
    for (int a = 0; a < nx; ++a) V[(size_t)a * nx + a] = 1.0L;
    // cyclic Jacobi on the symmetric M: M <- J' M J, V <- V J
    for (int sweep = 0; sweep < 60; ++sweep) {
        ld off = 0.0L, dia = 0.0L;
        for (int a = 0; a < nx; ++a) for (int b = 0; b < nx; ++b) { if (a != b) off += M[(size_t)a * nx + b] * M[(size_t)a * nx + b]; else dia += M[(size_t)a * nx + a] * M[(size_t)a * nx + a]; }
        if (off <= 1e-60L * dia || off == 0.0L) break;
        for (int a = 0; a < nx - 1; ++a)
            for (int b = a + 1; b < nx; ++b) {
                const ld apq = M[(size_t)a * nx + b];
                if (apq == 0.0L) continue;
                const ld theta = (M[(size_t)b * nx + b] - M[(size_t)a * nx + a]) / (2.0L * apq);
                const ld t = (theta >= 0.0L ? 1.0L : -1.0L) / (fabsl(theta) + sqrtl(theta * theta + 1.0L));
                const ld c = 1.0L / sqrtl(t * t + 1.0L), s = t * c;
                for (int k = 0; k < nx; ++k) {                       // columns a, b
                    const ld mka = M[(size_t)k * nx + a], mkb = M[(size_t)k * nx + b];
                    M[(size_t)k * nx + a] = c * mka - s * mkb; M[(size_t)k * nx + b] = s * mka + c * mkb;
                }
                for (int k = 0; k < nx; ++k) {                       // rows a, b
                    const ld mak = M[(size_t)a * nx + k], mbk = M[(size_t)b * nx + k];
                    M[(size_t)a * nx + k] = c * mak - s * mbk; M[(size_t)b * nx + k] = s * mak + c * mbk;
                }
                for (int k = 0; k < nx; ++k) {
                    const ld vka = V[(size_t)k * nx + a], vkb = V[(size_t)k * nx + b];
                    V[(size_t)k * nx + a] = c * vka - s * vkb; V[(size_t)k * nx + b] = s * vka + c * vkb;
                }
            }
    }
    ld lmax = 0.0L;
    for (int a = 0; a < nx; ++a) lmax = fmaxl(lmax, fabsl(M[(size_t)a * nx + a]));
    const ld tol = (ld)nx * 2.220446049250313e-16L * lmax;
    int rank = 0;
    std::vector<ld> inv(nx, 0.0L);
    for (int a = 0; a < nx; ++a) if (M[(size_t)a * nx + a] > tol) { inv[a] = 1.0L / M[(size_t)a * nx + a]; ++rank; }
    // pinv = V diag(inv) V' ;  G = pinv A_s'
    std::vector<ld> Pm((size_t)nx * nx, 0.0L);
    for (int a = 0; a < nx; ++a)
        for (int b = 0; b < nx; ++b) {
            ld s = 0.0L;
            for (int k = 0; k < nx; ++k) s += V[(size_t)a * nx + k] * inv[k] * V[(size_t)b * nx + k];
            Pm[(size_t)a * nx + b] = s;
        }
    G.assign((size_t)nx * p, 0.0);
    for (int a = 0; a < nx; ++a)
        for (int i = 0; i < p; ++i) {
            ld s = 0.0L;
            for (int b = 0; b < nx; ++b) s += Pm[(size_t)a * nx + b] * (ld)A_s[(size_t)b * p + i];
            G[(size_t)a * p + i] = (double)s;
        }
    return rank;
}

void fmpc_host_estimator_dft_images(int len, int d, int first, std::vector<double>& img) {
    typedef long double ld;
    const ld two_pi = 6.283185307179586476925286766559L;
    img.assign((size_t)(len / 4) * 2 * 2 * 64, 0.0);
    for (int y = 0; y < len; ++y)
        for (int j = 0; j < d && j < 32; ++j) {
            // the exponent reduced mod len in integers: the argument of cos / sin stays in [0, 2 pi)
            const long long e = ((long long)(first + j - len / 2) * (long long)(y - len / 2)) % len;
            const ld ang = -two_pi * (ld)((e + len) % len) / (ld)len;
            const size_t base = (((size_t)(y / 4) * 2 + (j / 16)) * 2) * 64 + (size_t)(y % 4) * 16 + (j % 16);
            img[base] = (double)cosl(ang);
            img[base + 64] = (double)sinl(ang);
        }
}
