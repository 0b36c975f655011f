// CDNA4 fastMPC, closed-loop step of a FEW realisations in one launch: the first-move form of the cold-start step (n = 27).
//
// The reference's loop (README.md:444-626) runs, per timestep and realisation, the steps either side of the solver
//     x0 = a[k] + B u[k-1] ,  x0_pre = previous x0 ,  w = b_ref = -M1 B u[k-1] - M2 B u[k-2]          README.md:482-497
// then Fast_MPC2(...).mpc_fixed_log_newton(1, k) (README.md:548-556) and applies u[k] = U(1:nu) only (README.md:589).
// For one realisation that is a chain of four dependent launches today (loop inputs, dense dual solve, d_z, decision:
// 34 us per step); nothing in it is large, it is all launch latency.  From the cold start the step is an AFFINE map of the
// data d = [x0 ; x0_pre ; B u1 ; B u2] (fmpc_kernel_inv.hip: nu+ = nuc + J d), so the first move is
//     u0 = u0c + K0 d                                             K0 = diag(wc) B' J_0 ,  144 x 108
// and the two sums the step-length / exit decision needs (backtracking_inf_newton.m:2-11, inf_newton_solver.m:19-22;
// SURVEY App. A.5) are quadratic forms of d built once per (handle, k) on the host (fmpc_host_build_first_move):
//     ||e||^2 = d'E d + 2 e'd + e0 ,      ||r_p||^2 = d'Ep d - 2 ep'd + ep0 .
// One 512-thread workgroup per realisation does all of that; four more per realisation write w (the API's output, and what
// the exact path needs should it have to redo the problem).  The decision is the panel path's (fw_panel_decide): t = 1 is
// accepted only with a wide margin, ||e||^2 <= rho_lb^2 / 2 -- here additionally widened by a bound on the rounding error
// of the quadratic forms, so a form that cancels badly can only hand a problem over, never accept one wrongly.  A problem
// that is not clear-cut is flagged in `need` and redone exactly by the launch that follows (fmpc_newton_wave, flag mode),
// which returns at once when no flag is set.  Same algebra as the four-launch path, different rounding: first moves agree
// to ~1e-13 (tests/test_gpu_closed_loop.py), both match the oracle to 1e-9.
#include <hip/hip_runtime.h>
#include <math.h>
#include "fmpc_device.h"
#include "fmpc_first.h"
#include "../../include/fastmpc.h"

#ifdef FW_TIMING
// diagnostic build: time stamps (constant 100 MHz clock) of wavefront 0 of the role-0 workgroup of realisation 0, last launch
__device__ unsigned long long fm_trace[8];
extern "C" int fmpc_debug_first_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fm_trace), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#define FM_TICK(k) do { if (p == 0 && role == 0 && tid == 0) fm_trace[k] = (unsigned long long)wall_clock64(); } while (0)
#else
#define FM_TICK(k)
#endif
#define FM_THREADS 1024                 // role 0: 576 + 216 + 216 partial rows; every partial row = ONE batch of <= 28 loads (the kernel is a chain
                                        // of memory round trips: 12.9 us with 9-load chunks on 512 threads)
#define FM_WROWS (FM_THREADS / 2)       // rows of w per w-workgroup (two threads per row)

__device__ __forceinline__ double fm_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One partial row of  M d  with M stored [column][row] (ld = rows): FM_CH values per thread, requested up front.
//   K0 (144 x 108): a row in four quarters of 27 columns.
//   E, Ep (108 x 108, SYMMETRIC): d'E d = sum_r d_r sum_j Ec[j][r] d[(r + j) mod nc], j = 0..nc/2, with the circulant half
//   Ec[j][r] = w_j E[r][(r + j) mod nc] (w = 1 on the diagonal, 2 off it; the pairs at distance nc/2 are kept for r < nc/2 only):
//   55 terms per row, two threads of 28 and 27 (fmpc_host_build_first_move builds Ec).  A full row would be 108 terms.
#define FM_CH 28
// (entries beyond c1 are ZERO: the sums below then need no predicate, and fma(0, x, acc) == acc exactly)
__device__ __forceinline__ void fm_row_load(double (&v)[FM_CH], const double* Mt, int ld, int r, int c0, int c1) {
#pragma unroll
    for (int q = 0; q < FM_CH; ++q) { const double t = Mt[(size_t)(c0 + q < c1 ? c0 + q : c0) * ld + r]; v[q] = c0 + q < c1 ? t : 0.0; }
}
// sum_q v[q] dl[base + q]: dl holds d TWICE in a row (dl[i + nc] == dl[i]), so that the circulant index needs no wrap -- 28 LDS
// reads at immediate offsets from one address (with a wrap per entry the compiler kept 2 x 28 precomputed addresses and
// predicates alive across the walk's loop and spilled: every reload a memory round trip)
__device__ __forceinline__ double fm_row_fma(const double (&v)[FM_CH], const double* dl, int base) {
    const double* d0 = dl + base;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < FM_CH; ++q) acc = fma(v[q], d0[q], acc);
    return acc;
}

// What a thread of the role-0 workgroup computes: kind 0 = K0 quarter row, 1 = E half row, 2 = Ep half row (-1: nothing).
struct FmRole { int kind, rr, c0, c1, base; };
__device__ __forceinline__ FmRole fm_role(int tid, int m, int nc) {
    FmRole r; r.kind = -1; r.rr = 0; r.c0 = 0; r.c1 = 0; r.base = 0;
    const int qc = nc / 4, H = nc / 2 + 1, hsplit = (H + 1) / 2;          // 55 circulant rows: 28 + 27
    if (tid < 4 * m) { r.kind = 0; r.rr = tid % m; const int qu = tid / m; r.c0 = qu * qc; r.c1 = qu == 3 ? nc : (qu + 1) * qc; r.base = r.c0; }
    else if (tid < 4 * m + 4 * nc) {
        const int t2 = tid - 4 * m;
        r.kind = t2 < 2 * nc ? 1 : 2;
        const int t3 = t2 < 2 * nc ? t2 : t2 - 2 * nc;
        r.rr = t3 % nc; r.c0 = (t3 / nc) * hsplit; r.c1 = r.c0 ? H : hsplit; r.base = r.rr + r.c0;
    }
    return r;
}
// partial value for spart: K0: the quarter sum; E / Ep: d_r ((Ec d)_r + lin_r), lin = +-2 e_r in the first half, 0 in the second
__device__ __forceinline__ double fm_partial(const FmRole& ro, const double (&mv)[FM_CH], const double* sd, double lin, int nc) {
    const double t = fm_row_fma(mv, sd, ro.base);
    const double q = (t + lin) * sd[ro.rr < nc ? ro.rr : 0];
    return ro.kind == 0 ? t : (ro.kind > 0 ? q : 0.0);
}
// v = B u1, B u2: output (which, r) split over 16 threads (entries j0 .. j0 + per of the row), fixed-order shuffle sum;
// bv = 0 where a thread has no entry (no branch)
__device__ __forceinline__ void fm_bv_load(double (&bv)[10], const double* bt, int n, int m, bool von, int vr, int j0, int per) {
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        const int j = j0 + q;
        const bool on = von && j < m && q < per;
        const double v = bt[(size_t)(on ? j : 0) * n + (von ? vr : 0)];
        bv[q] = on ? v : 0.0;
    }
}
__device__ __forceinline__ double fm_bv_dot(const double (&bv)[10], const double* u, int m, int j0) {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 10; ++q) { const int j = j0 + q; acc = fma(bv[q], u[j < m ? j : 0], acc); }
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64); acc += __shfl_xor(acc, 8, 64);
    return acc;
}
// the decision (lane 0 of wave 0): fw_panel_decide's, on bounds of the two forms.  forms (nullable): [e2 upper, rp2 lower, rho2 lower]
__device__ __forceinline__ bool fm_decide(const FmParams& P, double qe, double qp, double rdl, double dn2, double* forms) {
    const double dn = sqrt(dn2);
    double e2 = qe + P.e0, rp2 = qp + P.ep0;
    // rounding of the forms: |error| <= c eps (|d|^2 |M|_F + 2 |v| |d| + |const|), c generous
    const double ce = 4096.0 * 2.220446049250313e-16;
    const double de = ce * (dn2 * P.normE + 2.0 * P.norme * dn + fabs(P.e0));
    const double dp = ce * (dn2 * P.normEp + 2.0 * P.normep * dn + fabs(P.ep0));
    e2 += de;                                                     // upper bound of ||e||^2
    rp2 = rp2 - dp > 0.0 ? rp2 - dp : 0.0;                        // lower bound of ||r_p||^2
    const double rho2 = rp2 + rdl;                                // lower bound of rho^2
    if (forms) { forms[0] = e2; forms[1] = rp2; forms[2] = rho2; }
    const bool fin = rp2 < 1e300 && rho2 < 1e300 && e2 < 1e300 && e2 >= 0.0;
    return fin && (rp2 > 4e-16 || rho2 > 4e-12) && e2 <= 0.5 * rho2;
}

__global__ void __launch_bounds__(FM_THREADS) fmpc_first_move(FmParams P) {
    __shared__ double sd[2 * FM_NC_MAX + 4];        // d = [x0 ; x0_pre ; B u1 ; B u2], twice in a row (fm_row_fma)
    __shared__ double su[2][160];                   // u1, u2
    __shared__ double sx[2][32];                    // a_k, x0_last
    __shared__ double sred[16];
    __shared__ double spart[FM_THREADS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = P.n, m = P.m, nc = 4 * n, T = P.T, TN = T * n;
    const int p = blockIdx.x;                        // realisation
    const int role = blockIdx.y;                     // 0: first move + decision + x0, x0_pre ; >= 1: rows of w
    // ---- EVERY global load of this thread is requested here, before the first barrier: the kernel is a chain of
    // dependent steps through LDS, and each memory round trip in that chain would cost more than all its arithmetic
    // (12.9 us with the loads where they are used, measured).  Inputs; the B entries of the v = B u products; this
    // thread's partial row of K0 / Ec / Epc (role 0) or of [M1 M2] (role >= 1).
    FM_TICK(0);
    double in_u1 = 0.0, in_u2 = 0.0, in_a = 0.0, in_xl = 0.0;
    if (tid < m) { in_u1 = P.u1 ? P.u1[(size_t)p * m + tid] : 0.0; in_u2 = P.u2 ? P.u2[(size_t)p * m + tid] : 0.0; }
    if (tid < n) { in_a = P.a_k[(size_t)p * n + tid]; in_xl = P.x0_last ? P.x0_last[(size_t)p * n + tid] : 0.0; }
    const int o = tid >> 4, part = tid & 15;         // v stage: output o in 0..63 (which = o / 32, r = o % 32), 16 threads each
    const int which = (o >> 5) & 1, vr = o & 31;
    const int per = (m + 15) / 16, j0 = part * per; // m <= 160: at most 10 entries per part
    const bool von = o < 64 && vr < n;
    double bv[10];
    fm_bv_load(bv, P.bt, n, m, von, vr, j0, per);
    double mv[FM_CH];
    FmRole ro; ro.kind = -1; ro.rr = 0; ro.c0 = 0; ro.c1 = 0; ro.base = 0;
    if (role == 0) {
        ro = fm_role(tid, m, nc);
        if (ro.kind == 0) fm_row_load(mv, P.K0t, m, ro.rr, ro.c0, ro.c1);
        else if (ro.kind == 1) fm_row_load(mv, P.E, nc, ro.rr, ro.c0, ro.c1);
        else if (ro.kind == 2) fm_row_load(mv, P.Ep, nc, ro.rr, ro.c0, ro.c1);
        else {
#pragma unroll
            for (int q = 0; q < FM_CH; ++q) mv[q] = 0.0;
        }
    }
    // role >= 1: row e of w = -[M1 M2] [B u1 ; B u2], its 2 n columns split over two neighbouring threads
    const int wt = (role - 1) * FM_THREADS + tid, we = wt >> 1, wh = wt & 1;
    const bool won = role >= 1 && we < TN;
    if (role >= 1) fm_row_load(mv, P.m12t, TN, won ? we : 0, wh * n, (wh + 1) * n);
    double lin = 0.0, nuT = 0.0, nuX = 0.0, dxT = 0.0;
    if (role == 0) {
        if (ro.kind == 1 && ro.c0 == 0) lin = 2.0 * P.e[ro.rr];
        if (ro.kind == 2 && ro.c0 == 0) lin = -2.0 * P.ep[ro.rr];
        if (wv == 15 && lane < n) {
            dxT = P.dx0T[lane];
            if (P.nu0) { const double* nu = P.nu0 + (size_t)p * P.nb * n; nuT = nu[(T - 1) * n + lane]; nuX = P.has_xf ? nu[T * n + lane] : 0.0; }
        }
    }
    double u0c = 0.0;
    if (role == 0 && tid < m) u0c = P.u0c[tid];
    // ---- inputs into LDS
    if (tid < m) { su[0][tid] = in_u1; su[1][tid] = in_u2; }
    if (tid < n) { sx[0][tid] = in_a; sx[1][tid] = in_xl; }
    __syncthreads();
    FM_TICK(1);
    {
        const double acc = fm_bv_dot(bv, su[which], m, j0);
        if (part == 0 && von) { sd[2 * n + which * n + vr] = acc; sd[nc + 2 * n + which * n + vr] = acc; }
    }
    __syncthreads();
    if (tid < n) {
        const double xv = P.x0_given ? sx[0][tid] : sx[0][tid] + sd[2 * n + tid], xp = P.var2 ? sx[1][tid] : 0.0;
        sd[tid] = xv; sd[n + tid] = xp; sd[nc + tid] = xv; sd[nc + n + tid] = xp;
    }
    __syncthreads();
    FM_TICK(2);
    if (role >= 1) {
        // ---- w = -M1 (B u1) - M2 (B u2), one row per thread pair (README.md:490-497)
        double acc = fm_row_fma(mv, sd + 2 * n, wh * n);
        acc += __shfl_xor(acc, 1, 64);
        if (won && wh == 0) P.w[(size_t)p * TN + we] = -acc;
        return;
    }
    // ---- role 0: partial rows
    spart[tid] = fm_partial(ro, mv, sd, lin, nc);
    FM_TICK(3);
    // lower bound of ||r_d(nu0)||^2: its x entries of the last stage (no product needed), as the gate of the panel path
    if (wv == 15) {
        double rdl = 0.0, d2 = 0.0;
        if (P.nu0) {
            if (lane < n) { const double x = dxT + nuT + nuX; rdl = x * x; }
            rdl = fm_wave_sum(rdl);
        } else {
            rdl = P.rd2_0;
        }
        for (int c = lane; c < nc; c += 64) d2 = fma(sd[c], sd[c], d2);
        d2 = fm_wave_sum(d2);
        if (lane == 0) { sred[0] = rdl; sred[1] = d2; }
    }
    __syncthreads();
    // first moves (every realisation: a problem that is handed over gets its u0 overwritten by the exact path)
    if (tid < m) P.u0out[(size_t)p * m + tid] = u0c + ((spart[tid] + spart[m + tid]) + (spart[2 * m + tid] + spart[3 * m + tid]));
    if (tid < n && !P.x0_given) { P.x0[(size_t)p * n + tid] = sd[tid]; P.x0_pre[(size_t)p * n + tid] = sx[1][tid]; }
    FM_TICK(4);
    if (p == 0 && tid == 0 && P.handed) *P.handed = 0;
    if (wv == 0) {
        // fixed-order sums of the two quadratic forms
        double qe = 0.0, qp = 0.0;
        for (int r = lane; r < 2 * nc; r += 64) { qe += spart[4 * m + r]; qp += spart[4 * m + 2 * nc + r]; }
        qe = fm_wave_sum(qe); qp = fm_wave_sum(qp);
        if (lane == 0) {
            const bool clear = fm_decide(P, qe, qp, sred[0], sred[1], P.forms ? P.forms + 3 * (size_t)p : nullptr);
            P.need[p] = clear ? 0 : 1;
            if (clear) {
                if (P.status) P.status[p] = FMPC_OK;
                if (P.iters) P.iters[p] = 1;
                if (P.step) for (int q = 0; q < P.step_ld; ++q) P.step[(size_t)p * P.step_ld + q] = q == 0 ? 1.0 : -1.0;
            }
        }
    }
    FM_TICK(5);
}

// ------------------------------------------------------------------------------------------------------------------------
// A recorded stretch of the loop in ONE launch (fmpc_loop_run_device): the workgroup of a realisation keeps its rows of K0, Ec,
// Epc and of B in REGISTERS and walks through the steps; u[k-1], u[k-2] and the previous residual stay in LDS, a[k+1] is
// requested a step ahead, so that a step is five barriers and no memory round trip (the one-step kernel above is 12-15 us of
// dependent loads at idle clocks, plus a launch).  Same arithmetic in the same order as the one-step kernel (shared device
// functions): bitwise the same first moves.  A step that is not clear-cut ends the walk of that realisation: stop[p] = the
// step, x0 = the residual of the step before it; the host has the exact path redo that step and starts the walk again
// behind it.  w is not written (the last step of a stretch is the one-step call's).
// Barrier of the walk: orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding GLOBAL accesses (its
// fence covers global memory): the request for a[k+1] and the store of u[k] would each put a memory round trip into every step.
__device__ __forceinline__ void fm_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 512 threads, each doing the work of the one-step kernel's threads tid and tid + 512: two waves per SIMD, 256 registers per
// thread -- with 1024 threads (128 registers) the two resident rows spilled, and every reload is a memory round trip per step.
#define FMR_THREADS 512
__global__ void __launch_bounds__(FMR_THREADS) fmpc_first_move_run(FmParams P, FmRun R) {
    __shared__ double sd[2 * FM_NC_MAX + 4];
    __shared__ double su[2][160];
    __shared__ double sx[2][32];
    __shared__ double sred[16];
    __shared__ double spart[FM_THREADS];
    __shared__ int sclear;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = P.n, m = P.m, nc = 4 * n, T = P.T;
    const int p = blockIdx.x;
    const int s0 = R.start[p];
    if (s0 >= R.steps) { if (tid == 0) R.stop[p] = R.steps; return; }
    const size_t sn = (size_t)R.batch * n, sm = (size_t)R.batch * m, snu = (size_t)R.batch * P.nb * n;
    // ---- constants of this thread: those of the one-step kernel's threads tid (A) and tid + 512 (B).  The v products of A and
    // B are the same row of B against u1 and u2: one set of B entries.
    const int o = tid >> 4, part = tid & 15;         // o in 0..31: outputs (which = 0, r = o) and (which = 1, r = o)
    const int vr = o & 31;
    const int per = (m + 15) / 16, j0 = part * per;
    const bool von = vr < n;
    double bv[10];
    fm_bv_load(bv, P.bt, n, m, von, vr, j0, per);
    double mvA[FM_CH], mvB[FM_CH];
    const FmRole roA = fm_role(tid, m, nc), roB = fm_role(tid + FMR_THREADS, m, nc);
    if (roA.kind == 0) fm_row_load(mvA, P.K0t, m, roA.rr, roA.c0, roA.c1);
    else if (roA.kind == 1) fm_row_load(mvA, P.E, nc, roA.rr, roA.c0, roA.c1);
    else fm_row_load(mvA, P.Ep, nc, roA.rr, roA.c0, roA.c1);                       // (tid < 512: a kind is always assigned)
    if (roB.kind == 0) fm_row_load(mvB, P.K0t, m, roB.rr, roB.c0, roB.c1);
    else if (roB.kind == 1) fm_row_load(mvB, P.E, nc, roB.rr, roB.c0, roB.c1);
    else if (roB.kind == 2) fm_row_load(mvB, P.Ep, nc, roB.rr, roB.c0, roB.c1);
    else {
#pragma unroll
        for (int q = 0; q < FM_CH; ++q) mvB[q] = 0.0;
    }
    double linA = 0.0, linB = 0.0, dxT = 0.0;
    if (roA.kind == 1 && roA.c0 == 0) linA = 2.0 * P.e[roA.rr];
    if (roA.kind == 2 && roA.c0 == 0) linA = -2.0 * P.ep[roA.rr];
    if (roB.kind == 1 && roB.c0 == 0) linB = 2.0 * P.e[roB.rr];
    if (roB.kind == 2 && roB.c0 == 0) linB = -2.0 * P.ep[roB.rr];
    if (wv == 7 && lane < n) dxT = P.dx0T[lane];
    double u0c = 0.0;
    if (tid < m) u0c = P.u0c[tid];
    // ---- state at the first step of this walk
    {
        const double* u1 = s0 >= 1 ? R.U0 + (size_t)(s0 - 1) * sm : R.ub1;
        const double* u2 = s0 >= 2 ? R.U0 + (size_t)(s0 - 2) * sm : (s0 == 1 ? R.ub1 : R.ub2);
        if (tid < m) { su[0][tid] = u1 ? u1[(size_t)p * m + tid] : 0.0; su[1][tid] = u2 ? u2[(size_t)p * m + tid] : 0.0; }
        if (tid < n) {
            sx[0][tid] = R.a[(size_t)s0 * sn + (size_t)p * n + tid];
            sx[1][tid] = (s0 >= 1 || R.have_x0_last) ? P.x0[(size_t)p * n + tid] : 0.0;
        }
    }
    double x0new = 0.0, x0pre = 0.0;
    int s = s0;
    bool stopped = false;
    for (; s < R.steps; ++s) {
        double a_next = 0.0, nuT = 0.0, nuX = 0.0;
        if (tid < n && s + 1 < R.steps) a_next = R.a[(size_t)(s + 1) * sn + (size_t)p * n + tid];
        if (R.nu0 && wv == 7 && lane < n) {
            const double* nu = R.nu0 + (size_t)s * snu + (size_t)p * P.nb * n;
            nuT = nu[(T - 1) * n + lane]; nuX = P.has_xf ? nu[T * n + lane] : 0.0;
        }
        fm_lds_barrier();
        {
            const double acc1 = fm_bv_dot(bv, su[0], m, j0), acc2 = fm_bv_dot(bv, su[1], m, j0);
            if (part == 0 && von) { sd[2 * n + vr] = acc1; sd[3 * n + vr] = acc2; sd[nc + 2 * n + vr] = acc1; sd[nc + 3 * n + vr] = acc2; }
        }
        fm_lds_barrier();
        if (tid < n) {
            const double xv = sx[0][tid] + sd[2 * n + tid], xp = P.var2 ? sx[1][tid] : 0.0;
            sd[tid] = xv; sd[n + tid] = xp; sd[nc + tid] = xv; sd[nc + n + tid] = xp;
        }
        fm_lds_barrier();
        spart[tid] = fm_partial(roA, mvA, sd, linA, nc);
        spart[tid + FMR_THREADS] = fm_partial(roB, mvB, sd, linB, nc);
        if (wv == 7) {
            double rdl = 0.0, d2 = 0.0;
            if (R.nu0) {
                if (lane < n) { const double x = dxT + nuT + nuX; rdl = x * x; }
                rdl = fm_wave_sum(rdl);
            } else {
                rdl = P.rd2_0;
            }
            for (int c = lane; c < nc; c += 64) d2 = fma(sd[c], sd[c], d2);
            d2 = fm_wave_sum(d2);
            if (lane == 0) { sred[0] = rdl; sred[1] = d2; }
        }
        fm_lds_barrier();
        double unew = 0.0;
        if (tid < m) {
            unew = u0c + ((spart[tid] + spart[m + tid]) + (spart[2 * m + tid] + spart[3 * m + tid]));
            R.U0[(size_t)s * sm + (size_t)p * m + tid] = unew;
        }
        if (tid < n) {
            x0new = sd[tid]; x0pre = sx[1][tid];
            if (R.X0) R.X0[(size_t)s * sn + (size_t)p * n + tid] = x0new;
        }
        if (wv == 0) {
            double qe = 0.0, qp = 0.0;
            for (int r = lane; r < 2 * nc; r += 64) { qe += spart[4 * m + r]; qp += spart[4 * m + 2 * nc + r]; }
            qe = fm_wave_sum(qe); qp = fm_wave_sum(qp);
            if (lane == 0) sclear = fm_decide(P, qe, qp, sred[0], sred[1], nullptr) ? 1 : 0;
        }
        fm_lds_barrier();
        if (!sclear) { stopped = true; break; }
        if (tid < m) { su[1][tid] = su[0][tid]; su[0][tid] = unew; }
        if (tid < n) { sx[1][tid] = x0new; sx[0][tid] = a_next; }
    }
    if (stopped) {
        // the exact path redoes step s from the loop inputs: x0 must hold the residual of step s - 1
        if (tid < n && (s >= 1 || R.have_x0_last)) P.x0[(size_t)p * n + tid] = x0pre;
        if (tid == 0) R.stop[p] = s;
    } else {
        if (tid < n) { P.x0[(size_t)p * n + tid] = x0new; P.x0_pre[(size_t)p * n + tid] = x0pre; }
        if (tid == 0) {
            R.stop[p] = R.steps;
            if (P.status) P.status[p] = FMPC_OK;
            if (P.iters) P.iters[p] = 1;
        }
    }
    if (p == 0 && tid == 0 && P.handed) *P.handed = 0;
}

// ---- stopped realisations -> compact batch -> back (one workgroup per stopped realisation)
__global__ void __launch_bounds__(256) fmpc_walk_gather(FmCompact C) {
    const int c = blockIdx.x, tid = threadIdx.x;
    const int p = C.idx[c], s = C.stp[c];
    const int n = C.n, m = C.m;
    const size_t sn = (size_t)C.batch * n, sm = (size_t)C.batch * m, snu = (size_t)C.batch * C.nb * n;
    const double* u1 = s >= 1 ? C.U0 + (size_t)(s - 1) * sm : C.ub1;
    const double* u2 = s >= 2 ? C.U0 + (size_t)(s - 2) * sm : (s == 1 ? C.ub1 : C.ub2);
    for (int i = tid; i < m; i += 256) {
        C.cu1[(size_t)c * m + i] = u1 ? u1[(size_t)p * m + i] : 0.0;          // a NULL u1 / u2 / x0_last of the one-step call means zeros
        C.cu2[(size_t)c * m + i] = u2 ? u2[(size_t)p * m + i] : 0.0;
    }
    for (int i = tid; i < n; i += 256) {
        C.ca[(size_t)c * n + i] = C.a[(size_t)s * sn + (size_t)p * n + i];
        C.cx0[(size_t)c * n + i] = (s >= 1 || C.have_x0_last) ? C.x0[(size_t)p * n + i] : 0.0;
    }
    if (C.nu0) for (int i = tid; i < C.nb * n; i += 256) C.cnu[(size_t)c * C.nb * n + i] = C.nu0[(size_t)s * snu + (size_t)p * C.nb * n + i];
}
__global__ void __launch_bounds__(256) fmpc_walk_scatter(FmCompact C) {
    const int c = blockIdx.x, tid = threadIdx.x;
    const int p = C.idx[c], s = C.stp[c];
    const int n = C.n, m = C.m, TN = C.T * n;
    const size_t sn = (size_t)C.batch * n, sm = (size_t)C.batch * m;
    for (int i = tid; i < m; i += 256) C.U0[(size_t)s * sm + (size_t)p * m + i] = C.cu0[(size_t)c * m + i];
    for (int i = tid; i < n; i += 256) {
        const double v = C.cx0[(size_t)c * n + i];
        C.x0[(size_t)p * n + i] = v;
        C.x0_pre[(size_t)p * n + i] = C.cx0p[(size_t)c * n + i];
        if (C.X0) C.X0[(size_t)s * sn + (size_t)p * n + i] = v;
    }
    for (int i = tid; i < TN; i += 256) C.w[(size_t)p * TN + i] = C.cw[(size_t)c * TN + i];
    if (tid == 0) { if (C.status) C.status[p] = C.cst[c]; if (C.iters) C.iters[p] = C.cit[c]; }
}
size_t fmpc_compact_doubles(int n, int m, int T, int nb, int cap) {
    return (size_t)cap * (3 * n + 3 * m + (size_t)T * n + (size_t)nb * n + 2);        // + status, iters (ints in the last 2 doubles' space)
}
void fmpc_compact_carve(FmCompact& C, double* base, int cap) {
    double* q = base;
    C.ca = q; q += (size_t)cap * C.n; C.cx0 = q; q += (size_t)cap * C.n; C.cx0p = q; q += (size_t)cap * C.n;
    C.cu1 = q; q += (size_t)cap * C.m; C.cu2 = q; q += (size_t)cap * C.m; C.cu0 = q; q += (size_t)cap * C.m;
    C.cw = q; q += (size_t)cap * C.T * C.n; C.cnu = q; q += (size_t)cap * C.nb * C.n;
    C.cst = (int*)q; C.cit = C.cst + cap;
}
hipError_t fmpc_launch_walk_gather(const FmCompact& C, hipStream_t stream) {
    hipLaunchKernelGGL(fmpc_walk_gather, dim3(C.cnt), dim3(256), 0, stream, C);
    return hipGetLastError();
}
hipError_t fmpc_launch_walk_scatter(const FmCompact& C, hipStream_t stream) {
    hipLaunchKernelGGL(fmpc_walk_scatter, dim3(C.cnt), dim3(256), 0, stream, C);
    return hipGetLastError();
}

hipError_t fmpc_launch_first_move_run(const FmParams& P, const FmRun& R, hipStream_t stream) {
    if (P.n != 27 || 4 * P.n > FM_NC_MAX || P.m > 160 || 4 * P.m + 16 * P.n > FM_THREADS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fmpc_first_move_run, dim3(R.batch), dim3(FMR_THREADS), 0, stream, P, R);
    return hipGetLastError();
}

hipError_t fmpc_launch_first_move(const FmParams& P, int batch, hipStream_t stream) {
    if (P.n != 27 || 4 * P.n > FM_NC_MAX || P.m > 160 || 4 * P.m + 16 * P.n > FM_THREADS) return hipErrorInvalidValue;
    const int wg_w = (P.T * P.n + FM_WROWS - 1) / FM_WROWS;
    hipLaunchKernelGGL(fmpc_first_move, dim3(batch, 1 + wg_w), dim3(FM_THREADS), 0, stream, P);
    return hipGetLastError();
}
