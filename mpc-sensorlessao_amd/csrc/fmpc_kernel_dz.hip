// CDNA4 fastMPC, cold-start Newton step on panels of 16 problems (n = 27): the primal step from nu+.
//
// Second kernel of the panel path (the first, fmpc_kernel_panel.hip, produced nu+ = nu + d_nu of the full
// step; same reference correspondence: inf_newton_solver.m:33-36, backtracking_inf_newton.m:2-11):
//     d_u_j  = wc o (B' nu+_j - cu) ,  d_x_j = (2Q_j)^-1 (-dx0_j - nu+_j + A1' nu+_{j+1} + A2' nu+_{j+2} [- nu+_T])
//     z = zbar + d_z ,  ||e||^2 = sum (k P'DP d_u)^2
// One independent TASK per (panel of 16 problems, stage j), one wavefront per task, no barrier: the whole
// chip works on it whatever the batch size.  The products are only consumed element-wise, so they are computed
// TRANSPOSED: the nu+ panel in B-operand layout (lane (g, c) holds nu+[4 ks + g] of problem c) is also a valid
// A operand with the problems as rows, and the images of B', A1', A2' (LDS) serve as B operands with the
// entries as columns.  Result register r of lane (g, c) is then (problem 4 r + g, entry 16 J + c): each store
// instruction writes, for 4 problems, 16 consecutive entries (128 contiguous bytes), and the per-entry
// constants are one LDS read per lane.  A wave does ONE task: all its loads precede all its stores (a load behind a
// store waits for the store: vmcnt is in order), and the hardware dispatcher balances the workgroups.
//
// Per task it also leaves the partial ||e||^2 of its 16 problems; the exact-path launch that follows
// (fmpc_newton_wave in panel mode) sums them and decides the step length of every problem.
#include <hip/hip_runtime.h>
#include <math.h>
#include "fmpc_device.h"
#include "fmpc_panel.h"
#include "../../include/fastmpc.h"

#define FD_THREADS (FD_WAVES * 64)

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
typedef __attribute__((address_space(3))) double* fd_lds_t;
typedef const __attribute__((address_space(3))) double* fd_clds_t;
typedef const FpParams* FdKP;          // points at a LOCAL copy of the kernel argument (fields loaded once at kernel entry)

#ifdef FW_TIMING
#define FD_TICK(k) do { _tr[k + 1] = (unsigned long long)wall_clock64(); } while (0)
// per-wave trace (constant 100 MHz clock): entry, LDS image ready, loads issued, done; of the LAST launch
__device__ unsigned long long fd_trace[4 * 8192];
extern "C" int fmpc_debug_dz_trace(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fd_trace), sizeof(unsigned long long) * 4 * (n < 8192 ? n : 8192)) == hipSuccess ? 0 : -1;
}
#else
#define FD_TICK(k)
#endif

// NEXT: also leave, per task, the partial ||r_d||^2 of the NEW point (z+, nu+) with the barrier terms re-evaluated there,
// i.e. what the exit test of the following Newton iteration (inf_newton_solver.m:19-22) needs; its x entries and r_p
// vanish for a full step (the x part of Phi is constant), so only the u entries are summed.  Used for budgets > 1.
// FUSED: nu+_j, nu+_{j+1}, nu+_{j+2} are not read from the panel workspace but computed here, from the dense form of the dual
// solve (fmpc_kernel_inv.hip): nu+_s = nuc_s + J_s [x0 ; x0_pre] with the 27 x 56 rows of J of stage s as two 16-row A
// tiles -- the result registers of the two products are exactly the B-operand layout the rest of the task works with.
// Taken without w, without the terminal row and for a Newton budget of 1 (nobody else reads nu+ then): the 13 MB round trip
// of nu+ through HBM and one launch disappear.
// U0: the caller asked for the first moves only (z_out == NULL, README.md:589 uses nothing but U(1:nu)): nothing of z is
// written -- the task of stage 0 writes u_0 of its 16 problems to u0out, every task still leaves its partial ||e||^2 (the
// step-length decision needs all stages), and the x entries (a third of the products, two thirds of the loads) are skipped.
template <bool NEXT, bool FUSED = false, bool U0 = false>
__global__ void __launch_bounds__(FD_THREADS, FUSED ? 2 : 4) fmpc_cold_dz(FpParams Pv) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const FpParams Q = Pv;
    const FdKP P = &Q;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, c16 = lane & 15;
    const int nb = P->nb, T = P->T, m = P->m, mp = P->mp, batch = P->batch, s = FP_N + m;
    const FdLds L = fd_lds_layout(mp);
#ifdef FW_TIMING
    unsigned long long _tr[4] = {(unsigned long long)wall_clock64(), 0, 0, 0};
#endif
    {   // the LDS image is packed by the host in LDS order: all loads first, then the stores
        const double* src = P->dzimg;
        const int len = NEXT ? L.total_next : L.total;
        constexpr int NB = 7168 / FD_THREADS;           // NB x FD_THREADS doubles = 56 KB >= the image for m <= 160
        double t[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) { const int i = k * FD_THREADS + tid; t[k] = src[i < len ? i : 0]; }
#pragma unroll
        for (int k = 0; k < NB; ++k) { const int i = k * FD_THREADS + tid; if (i < len) lds[i] = t[k]; }
        for (int i = NB * FD_THREADS + tid; i < len; i += FD_THREADS) lds[i] = src[i];
    }
    __syncthreads();                                 // the only barrier: the waves are independent below
    FD_TICK(0);
    const fd_clds_t BT = (fd_clds_t)lds + L.BT + lane;
    const fd_clds_t A1T = (fd_clds_t)lds + L.A1T + lane;
    const fd_clds_t A2T = (fd_clds_t)lds + L.A2T + lane;
    const fd_clds_t UC = (fd_clds_t)lds + L.UC + c16;
    const fd_clds_t XQ = (fd_clds_t)lds + L.XQ;
    const bool has_xf = P->has_xf != 0, var2 = P->var2 != 0;
    const int NJ = mp / 16, NJF = m / 16;            // column blocks, full column blocks
    // Task order: workgroups b and b + 8 share an XCD (round-robin dispatch; for speed only), so the tasks of a
    // panel all go to the workgroups of ONE XCD: the cache lines shared by consecutive stages of a problem are
    // then merged in that XCD's L2 instead of being written back half-filled from two.
    const int xcd = blockIdx.x & 7;
    const int npx = (P->npanels - xcd + 7) >> 3;       // panels xcd, xcd + 8, ...
    const int ntasks = npx * T;
    const int t = (blockIdx.x >> 3) * FD_WAVES + wv;   // this wave's task among those of its XCD
    if (t >= ntasks) return;
    const int pl = t / T, j = t - pl * T;
    const int panel = xcd + 8 * pl;
    const double* nuws = P->nuws;
    const size_t nus = (size_t)nb * FP_N;

    // ---- everything this task reads from HBM / L2.  The instruction arbiter serves the oldest wave of a SIMD first:
    // without help the stores of the older waves (already computing) starve the loads of the younger ones (measured:
    // the 4th wave of a SIMD had its operands 8 us after the 1st).  Loading waves therefore run at raised priority.
    __builtin_amdgcn_s_setprio(3);
    double v0[FP_KS], v1[FP_KS], v2[FP_KS];        // nu+_j, nu+_{j+1}, nu+_{j+2} in B-operand layout (problem = lane % 16)
    double nx[2][4];                                // nu+_T at (problem 4 r + g, row 16 I + lane % 16): last stage with xf only
    if (FUSED) {
        const bool h1 = j + 1 < T, h2 = j + 2 < T && var2;
        // d = [x0 ; x0_pre ; 0 0] of the panel's problems in B-operand layout: all requests first, the zeros afterwards
        double dv[FP_XKS];
        const long pp = (long)panel * FP_NP + c16;
        const size_t pd = (size_t)(pp < batch ? pp : batch - 1);
#pragma unroll
        for (int ks = 0; ks < FP_XKS; ++ks) dv[ks] = *fi_addr(P->x0, P->x0p, nullptr, pd, 4 * ks + g, 0);
        // the rows of J and nuc of the three stages: two register sets, the next stage requested before the products of the current
        struct JS { double a[2][FP_XKS]; double nc[2][4]; };
        auto jload = [&](int st, JS& S) {
            const double* im = P->jst + (size_t)st * 2 * FP_XKS * 64 + lane;
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int ks = 0; ks < FP_XKS; ++ks) S.a[I][ks] = im[(I * FP_XKS + ks) * 64];
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int r = 0; r < 4; ++r) S.nc[I][r] = P->nucst[st * 32 + 16 * I + 4 * r + g];
        };
        auto jmul = [&](const JS& S, double v[FP_KS], bool on) {
            d4 a0 = {S.nc[0][0], S.nc[0][1], S.nc[0][2], S.nc[0][3]}, a1 = {S.nc[1][0], S.nc[1][1], S.nc[1][2], S.nc[1][3]};
#pragma unroll
            for (int ks = 0; ks < FP_XKS; ++ks) { a0 = MFMA64(S.a[0][ks], dv[ks], a0); a1 = MFMA64(S.a[1][ks], dv[ks], a1); }
            // result register r of tile I = row 16 I + 4 r + g = k-step 4 I + r of the B-operand layout (row 27 is a zero row of the image)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = on ? a0[r] : 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r) v[4 + r] = on ? a1[r] : 0.0;
        };
        JS s0, s1;
        jload(j, s0);
        if (!U0) jload(h1 ? j + 1 : j, s1);
#pragma unroll
        for (int ks = 0; ks < FP_XKS; ++ks) dv[ks] = fi_zero(P->x0p != nullptr, false, 4 * ks + g) ? 0.0 : dv[ks];
        jmul(s0, v0, true);
        if (!U0) {
            jload(h2 ? j + 2 : j, s0);
            jmul(s1, v1, h1);
            jmul(s0, v2, h2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int I = 0; I < 2; ++I) nx[I][r] = 0.0;
    } else {
        // nu+ arrives in panel layout, [stage row][16 problems]: the B-operand loads are 512 contiguous bytes
        const double* pnl = nuws + (size_t)panel * nus * FP_NP;
        const bool h1 = j + 1 < T, h2 = j + 2 < T && var2, xfl = j + 1 == T && has_xf;
#pragma unroll
        for (int ks = 0; ks < FP_KS; ++ks) {
            const bool kok = 4 * ks + g < FP_N;
            const int ko = (kok ? 4 * ks + g : 0) * FP_NP + c16;
            const double t0 = pnl[j * FP_N * FP_NP + ko];
            v0[ks] = kok ? t0 : 0.0;
            if (!U0) {
                const double t1 = pnl[(h1 ? j + 1 : j) * FP_N * FP_NP + ko];
                const double t2 = pnl[(h2 ? j + 2 : j) * FP_N * FP_NP + ko];
                v1[ks] = (kok && h1) ? t1 : 0.0; v2[ks] = (kok && h2) ? t2 : 0.0;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int I = 0; I < 2; ++I) nx[I][r] = 0.0;
        if (xfl && !U0) {                           // (wave-uniform; a strided load: 16 cache lines per instruction)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int I = 0; I < 2; ++I) {
                    const int row = 16 * I + c16;
                    nx[I][r] = pnl[(T * FP_N + (row < FP_N ? row : 0)) * FP_NP + 4 * r + g];
                }
        }
    }
    // nu+_j is also needed TRANSPOSED, at (problem 4 r + g, row 16 I + lane % 16).  Loading it that way costs 16 cache
    // lines per instruction; the values are already here in B-operand layout, so they go through this wave's LDS
    // scratch instead: element (row, problem) at row * 16 + (problem ^ (row & 15)) -- conflict-free both ways.
    double nj[2][4];
    if (!U0) {
    const fd_lds_t scr = (fd_lds_t)lds + (NEXT ? L.total_next : L.total) + wv * FD_SCR;
#pragma unroll
    for (int ks = 0; ks < FP_KS; ++ks) {
        const int row = 4 * ks + g;
        if (row < FP_N) scr[row * FP_NP + (c16 ^ (row & 15))] = v0[ks];
    }
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + c16 < FP_N ? 16 * I + c16 : 0;
            nj[I][r] = scr[row * FP_NP + ((4 * r + g) ^ (row & 15))];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);                     // operands are here (the LDS round trip above needed v0): compute and store
    __builtin_amdgcn_sched_barrier(0);
    FD_TICK(1);
    double* zq[4]; double* nq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int p = panel * FP_NP + 4 * r + g;         // problem 4 r + g of the panel
        if (U0) zq[r] = (p < batch && j == 0 ? P->u0out + (size_t)p * m : P->dump) + c16;      // only stage 0 has something to write
        else zq[r] = (p < batch ? P->zout + (size_t)p * T * s : P->dump) + (size_t)j * s + c16;
        nq[r] = (P->nuout && !U0) ? (p < batch ? P->nuout + (size_t)p * nus : P->dump + (size_t)T * s) + c16 : nullptr;
    }
    double eps2[4] = {0.0, 0.0, 0.0, 0.0}, rn2[4] = {0.0, 0.0, 0.0, 0.0};
    const fd_clds_t UX = (fd_clds_t)lds + L.UX + c16;
    const double kbar = P->kbar;
    // ---- x entries first (they free 30 of the 37 loaded values): d_x = (2Q)^-1 (-dx0 - nu+_j + A1' nu+_{j+1} + A2' nu+_{j+2} [- nu+_T])
    const bool last = j + 1 == T;
    const fd_clds_t xcv = XQ + (last ? 32 : 0), iqv = XQ + 64 + (last ? 32 : 0);
#pragma unroll
    for (int I = 0; I < (U0 ? 0 : 2); ++I) {
        d4 hh = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < FP_KS; ++ks) hh = MFMA64(v1[ks], A1T[(I * FP_KS + ks) * 64], hh);
#pragma unroll
        for (int ks = 0; ks < FP_KS; ++ks) hh = MFMA64(v2[ks], A2T[(I * FP_KS + ks) * 64], hh);
        const int row = 16 * I + c16;
        const bool rok = row < FP_N;
        const int rc = rok ? row : 0;
        const double xc = xcv[rc], iq = iqv[rc];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double zx = xc + iq * (hh[r] - nj[I][r] - nx[I][r]);
            if (rok) {                                  // rows 27..31 of the second row block do not exist
                zq[r][m + 16 * I] = zx;
                if (nq[r]) {                            // nu_out: one contiguous vector per problem
                    nq[r][j * FP_N + 16 * I] = nj[I][r];
                    if (last && has_xf) nq[r][T * FP_N + 16 * I] = nx[I][r];
                }
            }
        }
    }
    // ---- u entries: d_u = wc o (B' nu+_j - cu); the MFMAs of the next column block are issued before the
    // element-wise work of the current one
    auto mm = [&](int J) {
        d4 acc = {0, 0, 0, 0};
        const fd_clds_t im = BT + J * FP_KS * 64;
#pragma unroll
        for (int ks = 0; ks < FP_KS; ++ks) acc = MFMA64(v0[ks], im[ks * 64], acc);
        return acc;
    };
    auto epi = [&](int J, d4 acc) {
        const fd_clds_t uc = UC + 16 * J;
        const double c1 = uc[0], wc = uc[mp], hc = uc[2 * mp], ub = uc[3 * mp];
        const bool cok = J < NJF || 16 * J + c16 < m;    // partial last column block when m % 16 != 0
        double c2 = 0.0, r2 = 0.0, hp = 1.0, hm = 1.0;
        if (NEXT) { const fd_clds_t ux = UX + 16 * J; c2 = ux[0]; r2 = ux[mp]; hp = ux[2 * mp]; hm = ux[3 * mp]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double du = fma(wc, acc[r], c1);
            const double e = hc * du;
            if (cok) {
                eps2[r] = fma(e, e, eps2[r]);
                if (!U0 || j == 0) zq[r][16 * J] = ub + du;
            }
            if (NEXT) {
                // r_d[u] at the new point: 2R u+ + r + k (1/(umax - u+) - 1/(u+ - umin)) - B' nu+ ,  u+ = ubar + du
                const double rn = fma(r2, du, c2) + kbar * (1.0 / (hp - du) - 1.0 / (hm + du)) - acc[r];
                if (cok) rn2[r] = fma(rn, rn, rn2[r]);
            }
        }
    };
    {
        d4 a0 = mm(0);
        int J = 0;
        for (; J + 2 < NJ; J += 2) {
            const d4 a1 = mm(J + 1);
            epi(J, a0);
            a0 = mm(J + 2);
            epi(J + 1, a1);
        }
        if (J + 1 < NJ) {
            const d4 a1 = mm(J + 1);
            epi(J, a0);
            epi(J + 1, a1);
        } else {
            epi(J, a0);
        }
    }
    // ---- ||e||^2 of this stage per problem
    double* ep = P->epsp + ((size_t)panel * T + j) * FP_NP;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double v = eps2[r];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        if (c16 == 0) ep[4 * r + g] = v;
    }
    if (NEXT) {
        double* rq = P->rnp + ((size_t)panel * T + j) * FP_NP;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v = rn2[r];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (c16 == 0) rq[4 * r + g] = v;
        }
    }
    FD_TICK(2);
#ifdef FW_TIMING
    if (lane == 0) {       // (no atomics here: thousands of waves adding to one address take longer than the kernel)
        const int wid = blockIdx.x * FD_WAVES + wv;
        if (wid < 8192) for (int q = 0; q < 4; ++q) fd_trace[4 * wid + q] = _tr[q];
    }
#endif
}

// ---------------------------------------------------------------- host side
size_t fmpc_dz_lds_bytes(int mp, int next) {
    const FdLds L = fd_lds_layout(mp);
    return ((size_t)(next ? L.total_next : L.total) + FD_WAVES * FD_SCR) * sizeof(double);
}

hipError_t fmpc_dz_prepare(int mp) {
    hipError_t e = hipFuncSetAttribute((const void*)fmpc_cold_dz<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)fmpc_dz_lds_bytes(mp, 0));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)fmpc_cold_dz<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)fmpc_dz_lds_bytes(mp, 0));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)fmpc_cold_dz<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)fmpc_dz_lds_bytes(mp, 0));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)fmpc_cold_dz<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)fmpc_dz_lds_bytes(mp, 0));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)fmpc_cold_dz<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)fmpc_dz_lds_bytes(mp, 1));
}

hipError_t fmpc_launch_dz(const FpParams& P, int grid, int next, hipStream_t stream, int fused, int u0only) {
    if (u0only && !next) {
        if (fused) hipLaunchKernelGGL((fmpc_cold_dz<false, true, true>), dim3(grid), dim3(FD_THREADS), fmpc_dz_lds_bytes(P.mp, 0), stream, P);
        else hipLaunchKernelGGL((fmpc_cold_dz<false, false, true>), dim3(grid), dim3(FD_THREADS), fmpc_dz_lds_bytes(P.mp, 0), stream, P);
    } else if (fused && !next) hipLaunchKernelGGL((fmpc_cold_dz<false, true>), dim3(grid), dim3(FD_THREADS), fmpc_dz_lds_bytes(P.mp, 0), stream, P);
    else if (next) hipLaunchKernelGGL(fmpc_cold_dz<true>, dim3(grid), dim3(FD_THREADS), fmpc_dz_lds_bytes(P.mp, 1), stream, P);
    else hipLaunchKernelGGL(fmpc_cold_dz<false>, dim3(grid), dim3(FD_THREADS), fmpc_dz_lds_bytes(P.mp, 0), stream, P);
    return hipGetLastError();
}
