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
#define FM_THREADS 1024                 // role 0: 576 + 216 + 216 partial rows; every partial row = ONE batch of <= 27 loads (the kernel is a chain
                                        // of memory round trips: 12.9 us with 9-load chunks on 512 threads)
#define FM_WROWS (FM_THREADS / 2)       // rows of w per w-workgroup (two threads per row)

__device__ __forceinline__ double fm_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One partial row of  M d  with M stored [column][row] (ld = rows): FM_CH values per thread, requested up front.
#define FM_CH 27
__device__ __forceinline__ void fm_row_load(double (&v)[FM_CH], const double* Mt, int ld, int r, int c0, int c1) {
#pragma unroll
    for (int q = 0; q < FM_CH; ++q) v[q] = Mt[(size_t)(c0 + q < c1 ? c0 + q : c0) * ld + r];
}
__device__ __forceinline__ double fm_row_fma(const double (&v)[FM_CH], const double* dl, int c0, int c1) {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < FM_CH; ++q) acc = c0 + q < c1 ? fma(v[q], dl[c0 + q], acc) : acc;
    return acc;
}

__global__ void __launch_bounds__(FM_THREADS) fmpc_first_move(FmParams P) {
    __shared__ double sd[FM_NC_MAX + 4];            // d = [x0 ; x0_pre ; B u1 ; B u2]
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
    // thread's partial row of K0 / E / Ep (role 0) or of [M1 M2] (role >= 1).
    FM_TICK(0);
    double in_u1 = 0.0, in_u2 = 0.0, in_a = 0.0, in_xl = 0.0;
    if (tid < m) { in_u1 = P.u1 ? P.u1[(size_t)p * m + tid] : 0.0; in_u2 = P.u2 ? P.u2[(size_t)p * m + tid] : 0.0; }
    if (tid < n) { in_a = P.a_k[(size_t)p * n + tid]; in_xl = P.x0_last ? P.x0_last[(size_t)p * n + tid] : 0.0; }
    const int o = tid >> 4, part = tid & 15;         // v stage: output o in 0..63 (which = o / 32, r = o % 32), 16 threads each
    const int which = (o >> 5) & 1, vr = o & 31;
    const int per = (m + 15) / 16, j0 = part * per; // m <= 160: at most 10 entries per part
    const bool von = o < 64 && vr < n;
    double bv[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) { const int j = j0 + q; bv[q] = P.bt[(size_t)((von && j < m && q < per) ? j : 0) * n + (von ? vr : 0)]; }
    double mv[FM_CH];
    const int qc = nc / 4, hcol = nc / 2;
    int kind = -1, rr = 0, c0 = 0, c1 = 0;           // role 0: 0 = K0 quarter row, 1 = E half row, 2 = Ep half row; role >= 1: 3 = w row
    if (role == 0) {
        if (tid < 4 * m) { kind = 0; rr = tid % m; const int qu = tid / m; c0 = qu * qc; c1 = qu == 3 ? nc : (qu + 1) * qc; fm_row_load(mv, P.K0t, m, rr, c0, c1); }
        else if (tid < 4 * m + 2 * nc) { kind = 1; const int t2 = tid - 4 * m; rr = t2 % nc; c0 = (t2 / nc) * hcol; c1 = c0 ? nc : hcol; fm_row_load(mv, P.E, nc, rr, c0, c1); }
        else if (tid < 4 * m + 4 * nc) { kind = 2; const int t2 = tid - 4 * m - 2 * nc; rr = t2 % nc; c0 = (t2 / nc) * hcol; c1 = c0 ? nc : hcol; fm_row_load(mv, P.Ep, nc, rr, c0, c1); }
    }
    // role >= 1: row e of w = -[M1 M2] [B u1 ; B u2], its 2 n columns split over two neighbouring threads
    const int wt = (role - 1) * FM_THREADS + tid, we = wt >> 1, wh = wt & 1;
    const bool won = role >= 1 && we < TN;
    if (role >= 1) fm_row_load(mv, P.m12t, TN, won ? we : 0, wh * n, (wh + 1) * n);
    double ev = 0.0, nuT = 0.0, nuX = 0.0, dxT = 0.0;
    if (role == 0) {
        if (kind == 1 && c0 == 0) ev = P.e[rr];
        if (kind == 2 && c0 == 0) ev = P.ep[rr];
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
    // ---- v1 = B u1, v2 = B u2: output (which, r) split over 16 threads, fixed-order shuffle sum
    {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 10; ++q) { const int j = j0 + q; if (von && j < m && q < per) acc = fma(bv[q], su[which][j], acc); }
        acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64); acc += __shfl_xor(acc, 8, 64);
        if (part == 0 && von) sd[2 * n + which * n + vr] = acc;
    }
    __syncthreads();
    if (tid < n) { sd[tid] = sx[0][tid] + sd[2 * n + tid]; sd[n + tid] = P.var2 ? sx[1][tid] : 0.0; }
    __syncthreads();
    FM_TICK(2);
    if (role >= 1) {
        // ---- w = -M1 (B u1) - M2 (B u2), one row per thread (README.md:490-497)
        double acc = fm_row_fma(mv, sd + 2 * n, wh * n, (wh + 1) * n);
        acc += __shfl_xor(acc, 1, 64);
        if (won && wh == 0) P.w[(size_t)p * TN + we] = -acc;
        return;
    }
    // ---- role 0.  Partial rows: K0 d in four column quarters per row (4 m threads), E d and Ep d in two halves per row
    double partv = 0.0;
    if (kind == 0) partv = fm_row_fma(mv, sd, c0, c1);
    else if (kind == 1) partv = (fm_row_fma(mv, sd, c0, c1) + 2.0 * ev) * sd[rr];        // d_r ((E d)_r + 2 e_r), in two halves
    else if (kind == 2) partv = (fm_row_fma(mv, sd, c0, c1) - 2.0 * ev) * sd[rr];        // d_r ((Ep d)_r - 2 ep_r)
    spart[tid] = partv;
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
    if (tid < n) { P.x0[(size_t)p * n + tid] = sd[tid]; P.x0_pre[(size_t)p * n + tid] = sx[1][tid]; }
    FM_TICK(4);
    if (p == 0 && tid == 0 && P.handed) *P.handed = 0;
    if (wv == 0) {
        // fixed-order sums of the two quadratic forms
        double qe = 0.0, qp = 0.0;
        for (int r = lane; r < 2 * nc; r += 64) { qe += spart[4 * m + r]; qp += spart[4 * m + 2 * nc + r]; }
        qe = fm_wave_sum(qe); qp = fm_wave_sum(qp);
        if (lane == 0) {
            const double dn2 = sred[1], dn = sqrt(dn2);
            double e2 = qe + P.e0, rp2 = qp + P.ep0;
            // rounding of the forms: |error| <= c eps (|d|^2 |M|_F + 2 |v| |d| + |const|), c generous
            const double ce = 4096.0 * 2.220446049250313e-16;
            const double de = ce * (dn2 * P.normE + 2.0 * P.norme * dn + fabs(P.e0));
            const double dp = ce * (dn2 * P.normEp + 2.0 * P.normep * dn + fabs(P.ep0));
            e2 += de;                                                     // upper bound of ||e||^2
            rp2 = rp2 - dp > 0.0 ? rp2 - dp : 0.0;                        // lower bound of ||r_p||^2
            const double rho2 = rp2 + sred[0];                           // lower bound of rho^2
            const bool fin = rp2 < 1e300 && rho2 < 1e300 && e2 < 1e300 && e2 >= 0.0;
            const bool clear = fin && (rp2 > 4e-16 || rho2 > 4e-12) && e2 <= 0.5 * rho2;
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

hipError_t fmpc_launch_first_move(const FmParams& P, int batch, hipStream_t stream) {
    if (P.n != 27 || 4 * P.n > FM_NC_MAX || P.m > 160 || 4 * P.m + 16 * P.n > FM_THREADS) return hipErrorInvalidValue;
    const int wg_w = (P.T * P.n + FM_WROWS - 1) / FM_WROWS;
    hipLaunchKernelGGL(fmpc_first_move, dim3(batch, 1 + wg_w), dim3(FM_THREADS), 0, stream, P);
    return hipGetLastError();
}
