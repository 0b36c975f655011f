// CDNA4 fastMPC, cold-start Newton step: the dual solve as ONE dense product (n = 27).
//
// Regime: the reference's own call, Fast_MPC2(..., x_init = []).mpc_fixed_log_newton(1, k) (README.md:548-556;
// inf_newton_solver.m:10-41), like fmpc_cold_panel.  From the mid-box start Y = C Phi^-1 C' is the same for every
// problem and the new dual variable is an AFFINE function of the per-problem data (fmpc_kernel_panel.hip, header):
//     nu+ = Y^-1 (ct - b) ,   b_0 = A1 x0 + A2 x0_pre + w_0 ,  b_1 = A2 x0 + w_1 ,  b_i = w_i   (fast_mpc_eq_const.m:39-68)
//         = nuc + J d ,       d = [x0 ; x0_pre ; 0 0 ; w]          J = d nu+ / d d   (nb n  x  56 + T n)
// fmpc_cold_panel evaluates this map through the block factor of Y: two sweeps over the horizon, a chain of ~31 dependent
// steps on one CU per 16 problems, 41 us however few problems there are.  This kernel evaluates it as the product J d on
// the fp64 matrix cores: about three times the flops and no dependency at all -- a cold-start solve of up to 64 problems takes
// 25-33 us instead of 56-75 us, 512 problems fill the chip -- and without w (replay batches: w = NULL) only the 56 columns of
// [x0 ; x0_pre] are left (measured: DESIGN.md §7).
// J and nuc come from the panel kernel itself (fmpc_build_inverse in fmpc_api.hip runs it once per (handle, k) on unit
// vectors), so both paths share one factorisation.
//
// Work item = (16-row tile of nu+, PB panels of 16 problems): KSPLIT wavefronts of a workgroup split the k range and add
// their partial tiles in LDS in a fixed order (results do not depend on the launch shape).  A operand: one coalesced
// 512-byte load per k-step from the image of J; B operand: d straight from the caller's arrays (lane = (k, problem)).
// Items are dealt so that the workgroups of one XCD share a few row tiles of J (its L2 holds them) and sweep the panels.
// The first workgroups of the grid compute what the step-length decision needs per problem: ||r_p||^2 and a lower bound of
// rho^2 (as S1 of the panel kernel does).  Output: nu+ in PANEL layout and `gate`, exactly what fmpc_cold_dz and the
// decision pass read.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include "fmpc_device.h"
#include "fmpc_panel.h"
#include "../../include/fastmpc.h"

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

#ifdef FW_TIMING
// per-wavefront trace of the row-group kernel (constant 100 MHz clock): entry, operands in LDS, products done, stores issued
__device__ unsigned long long fi_trace[4 * 8192];
extern "C" int fmpc_debug_inv_trace(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fi_trace), sizeof(unsigned long long) * 4 * (n < 8192 ? n : 8192)) == hipSuccess ? 0 : -1;
}
#define FI_TICK(k) do { _tr[k] = (unsigned long long)wall_clock64(); } while (0)
#else
#define FI_TICK(k)
#endif
#define FI_CH 7                          // k-steps per register set (two sets in flight)

__device__ __forceinline__ double fi_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ||r_p||^2 and the lower bound of rho^2 of the 16 problems of panel pn, for the step-length decision (what S1 of the panel
// kernel produces).  First 256 threads of the workgroup = 4 wavefronts; sh: 56 x 17 + 6 x 16 doubles.
//   r_p,i = cp_i - b_i ,  b_0 = A1 x0 + A2 x0_pre + w_0 ,  b_1 = A2 x0 + w_1 ,  b_i = w_i  (i < T) ,  0 on the xf row
// The prediction terms are one product E [x0 ; x0_pre] on the matrix cores (E = [A1 A2 ; A2 0], image built by the host),
// every row of x0, x0_pre, w, nu0 is read along its problem (coalesced), and all reads are issued before the first use.
__device__ __forceinline__ void fi_gate(const FpParams* P, int pn, double* sh) {
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, c16 = lane & 15;
    const int T = P->T, nb = P->nb, batch = P->batch, TN = T * FP_N;
    const bool has_w = P->w != nullptr;
    const FpVec V = fp_vec_layout(nb, T);
    const double* cp = P->vec + V.cp;
    double* xs = sh;                        // [56 columns of d][17]
    double* part = sh + 56 * 17;            // [4 wavefronts][16]: stages 0, 1
    double* wsum = part + 64;               // [16]: the other block rows
    double* rds = wsum + 16;                // [16]: lower bound of ||r_d||^2
    const int n01 = (nb < 2 ? nb : 2) * FP_N;                       // rows of the stages 0 and 1
    double ea[FP_XKS], w01[4], rsum[4] = {0.0, 0.0, 0.0, 0.0}, rdv[4] = {0.0, 0.0, 0.0, 0.0};
    if (wv < 4) {
        // ---- requests: E image (this wavefront's row tile), the x panel, w / nu0 rows of this wavefront's 4 problems
#pragma unroll
        for (int ks = 0; ks < FP_XKS; ++ks) ea[ks] = P->eimg[(wv * FP_XKS + ks) * 64 + lane];
        size_t pr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const long p = (long)pn * FP_NP + 4 * wv + i; pr[i] = (size_t)(p < batch ? p : batch - 1); }
        double xv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[i] = *fi_addr(P->x0, P->x0p, nullptr, pr[i], lane, TN);      // (zeroed where it is stored)
        const size_t pcol = (size_t)((long)pn * FP_NP + c16 < batch ? (long)pn * FP_NP + c16 : batch - 1);
        // w at the rows of this lane's result registers (stages 0, 1); without w: a valid address, zeroed
        const double* wq = has_w ? P->w + pcol * (size_t)TN : P->x0 + pcol * FP_N;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * wv + 4 * r + g;
            const double t = wq[has_w ? (row < TN ? row : TN - 1) : 0];
            w01[r] = has_w ? t : 0.0;
        }
        if (has_w) {
            for (int t0 = 0; t0 < nb * FP_N; t0 += 64 * 4) {
                double wl[4][4]; double cl[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = t0 + 64 * u + lane;
                    const int ic = idx < nb * FP_N ? idx : nb * FP_N - 1;
                    const int i = ic / FP_N;
                    cl[u] = cp[i * 32 + ic - i * FP_N];
#pragma unroll
                    for (int q = 0; q < 4; ++q) wl[q][u] = P->w[pr[q] * (size_t)TN + (ic < TN ? ic : TN - 1)];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = t0 + 64 * u + lane;
                    const bool on = idx >= n01 && idx < nb * FP_N;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double v = cl[u] - (idx < TN ? wl[q][u] : 0.0);
                        rsum[q] = on ? fma(v, v, rsum[q]) : rsum[q];
                    }
                }
            }
        }
        {
            // ||r_d(nu0)||^2 >= its x entries of the last stage, dx0_{T-1} + nu0_{T-1} [+ nu0_T]   (without nu0: a host constant)
            const int rl = lane < FP_N ? lane : 0;
            const double dx = P->vec[V.dx0 + (T - 1) * 32 + rl];
            const bool hn = P->nu0 != nullptr, hx = hn && P->has_xf;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double* nu = hn ? P->nu0 + pr[q] * (size_t)nb * FP_N : P->x0 + pr[q] * FP_N;     // (a valid address either way)
                const double t0 = nu[hn ? (T - 1) * FP_N + rl : rl];
                const double t1 = nu[hx ? T * FP_N + rl : rl];
                const double x = dx + t0 + (hx ? t1 : 0.0);
                rdv[q] = lane < FP_N ? x * x : 0.0;
            }
        }
        // ---- the x panel in B-operand layout
#pragma unroll
        for (int i = 0; i < 4; ++i) if (lane < 56) xs[lane * 17 + 4 * wv + i] = fi_zero(P->x0p != nullptr, false, lane) ? 0.0 : xv[i];
    }
    __syncthreads();
    if (wv < 4) {
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < FP_XKS; ++ks) acc = MFMA64(ea[ks], xs[(4 * ks + g) * 17 + c16], acc);
        double sq = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * wv + 4 * r + g;
            const int rc = row < n01 ? row : 0;
            const int i = rc >= FP_N ? 1 : 0;
            const double v = cp[i * 32 + rc - i * FP_N] - (i < T ? acc[r] + w01[r] : 0.0);
            sq = row < n01 ? fma(v, v, sq) : sq;
        }
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        if (g == 0) part[wv * 16 + c16] = sq;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double a = fi_wave_sum(rsum[q]), b = fi_wave_sum(rdv[q]);
            if (lane == 0) { wsum[4 * wv + q] = a; rds[4 * wv + q] = b; }
        }
    }
    __syncthreads();
    if (tid < 16) {
        const int p = pn * FP_NP + tid;
        double ssum = (has_w || nb <= 2) ? 0.0 : P->rp2c;           // without w the stages >= 2 are a host constant
        for (int q = 0; q < 4; ++q) ssum += part[q * 16 + tid];
        ssum += wsum[tid];
        const double rd = P->nu0 ? rds[tid] : P->rd2_0;
        if (p < batch) { P->gate[2 * p] = ssum; P->gate[2 * p + 1] = ssum + rd; }
    }
}

template <int KSPLIT, int PB>
__global__ void __launch_bounds__((KSPLIT == 16 ? 16 : 4) * 64) fmpc_cold_inv(FpParams Pv) {
    constexpr int NWV = KSPLIT == 16 ? 16 : 4, IPW = NWV / KSPLIT;
    constexpr int RED = KSPLIT > 1 ? NWV * PB * 256 : 1;
    __shared__ double red[RED > 1048 ? RED : 1048];
    const FpParams Q = Pv;
    const FpParams* P = &Q;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, c16 = lane & 15;
    const int nb = P->nb, batch = P->batch, npanels = P->npanels;
    const int nrow = nb * FP_N;
    const int nrt = (nrow + 15) / 16, npg = (npanels + PB - 1) / PB, rt8 = (nrt + 7) / 8;
    const int qcount = rt8 * npg;
    const int gemm_blocks = 8 * ((qcount + IPW - 1) / IPW);
    if (blockIdx.x == 0 && tid == 0 && P->handed) { P->handed[0] = 0; P->handed[1] = 0; }   // counters of the exact-path launches
    // the FIRST workgroups of the grid (a multiple of 8: the XCD of the others is unchanged) are the gate tasks: behind the
    // products they would wait for a free CU and then run alone
    const int ngate = (npanels + 7) & ~7;
    if ((int)blockIdx.x < ngate) {
        if ((int)blockIdx.x < npanels) fi_gate(P, (int)blockIdx.x, red);
        return;
    }
    const int bid = (int)blockIdx.x - ngate;
    if (bid >= gemm_blocks || P->gate_only) return;
    // XCD x = block & 7 works on the row tiles x, x + 8, ...: one row tile at a time, all panel groups
    const int x = bid & 7;
    const int q = (bid >> 3) * IPW + (KSPLIT == 1 ? wv : 0);
    const int rtq = x + 8 * (q / npg), pg = q % npg;
    const bool live = q < qcount && rtq < nrt;                     // uniform per wave (per workgroup when the k range is split)
    const int rt = live ? rtq : 0;
    const bool has_w = P->gw != nullptr;
    const int kend = has_w ? P->jks : FP_XKS;
    const int per = (kend + KSPLIT - 1) / KSPLIT;
    const int part = KSPLIT == 1 ? 0 : wv;
    const int k0 = part * per;
    const int k1 = k0 + per < kend ? k0 + per : kend;
    size_t pc[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const long p = (long)(pg * PB + j) * FP_NP + c16;
        pc[j] = (size_t)(p < batch ? p : batch - 1);
    }
    const double* jrow = P->jimg + (size_t)rt * P->jksp * 64 + lane;
    const double* x0 = P->x0;
    const double* x0p = P->x0p;
    const double* w = P->gw;                       // the columns of d behind [x0 ; x0_pre ; 0 0]: w, or [B u1 ; B u2]
    const int WN = P->gwn;
    struct Set { double a[FI_CH]; double b[PB][FI_CH]; };
    auto load = [&](Set& S, int kb) {                               // requests only; zeros are applied in mma()
#pragma unroll
        for (int u = 0; u < FI_CH; ++u) {
            int ks = kb + u; ks = ks < k1 ? ks : k1 - 1;            // past the end: a valid k-step, loaded and dropped
            ks = ks < 0 ? 0 : ks;
            S.a[u] = jrow[(size_t)ks * 64];
#pragma unroll
            for (int j = 0; j < PB; ++j) S.b[j][u] = *fi_addr(x0, x0p, w, pc[j], 4 * ks + g, WN);
        }
    };
    d4 acc[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) acc[j] = (d4){0, 0, 0, 0};
    auto mma = [&](const Set& S, int kb) {
#pragma unroll
        for (int u = 0; u < FI_CH; ++u)
            if (kb + u < k1) {
#pragma unroll
                for (int j = 0; j < PB; ++j)
                    acc[j] = MFMA64(S.a[u], fi_zero(x0p != nullptr, w != nullptr, 4 * (kb + u) + g) ? 0.0 : S.b[j][u], acc[j]);
            }
    };
    if (live && k0 < k1) {
        Set s0, s1;
        load(s0, k0);
        for (int kb = k0; kb < k1; kb += 2 * FI_CH) {
            load(s1, kb + FI_CH);
            mma(s0, kb);
            load(s0, kb + 2 * FI_CH);
            mma(s1, kb + FI_CH);
        }
    }
    const size_t pstride = (size_t)nrow * FP_NP;
    if (KSPLIT == 1) {
        if (live) {
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int pn = pg * PB + j;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * rt + 4 * r + g;
                    if (pn < npanels && row < nrow) P->nuws[(size_t)pn * pstride + (size_t)row * FP_NP + c16] = P->nuc[row] + acc[j][r];
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < PB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wv * PB + j) * 256 + r * 64 + lane] = acc[j][r];
        __syncthreads();
        if (live) {
            for (int e = tid; e < PB * 256; e += NWV * 64) {
                const int j = e >> 8, r = (e >> 6) & 3, l = e & 63;
                const int row = 16 * rt + 4 * r + (l >> 4), pn = pg * PB + j;
                double sacc = 0.0;
#pragma unroll
                for (int v = 0; v < NWV; ++v) sacc += red[(v * PB + j) * 256 + (e & 255)];
                if (pn < npanels && row < nrow) P->nuws[(size_t)pn * pstride + (size_t)row * FP_NP + (l & 15)] = P->nuc[row] + sacc;
            }
        }
    }
}

// Row-group form: a workgroup = 4 wavefronts = 4 row tiles of nu+ for the same PB panels, all of k.  The panel's data d
// is read from memory ONCE per workgroup, coalesced along each problem's row (64 consecutive columns per wavefront
// load), transposed through LDS into the B-operand layout [column][problem] and shared by the four wavefronts; chunks of
// 16 k-steps, double buffered, the B operand of k-step u + 1 read from LDS before the products of k-step u are issued.
// (The k-split form above reads d in 32-byte pieces straight into registers: good for a handful of problems,
// L1/L2-bound beyond.)  HAS_W = false: the one chunk of [x0 ; x0_pre].
// A workgroup = RT row tiles x KS groups that split the chunks of k (partial tiles added in LDS in a fixed order), PB
// panels.  The matrix pipes bound this kernel -- (row tiles x panels x k-steps) products of 64 cycles over the SIMDs that
// have a wavefront -- so the shape is chosen per batch to put work on every SIMD: few panels -> one row tile per workgroup
// and the k groups on its four SIMDs; many panels -> more row tiles per workgroup share the staged panel.
template <int PB, int RT, int KS, bool HAS_W>
__global__ void __launch_bounds__(64 * RT * KS) fmpc_cold_inv_rg(FpParams Pv) {
    constexpr int NP = PB * FP_NP, PPW = NP / RT;                 // problems per workgroup, problems a wavefront loads
    // LDS image of a chunk of d: per k-step a row of PB x 64 doubles in B-OPERAND ORDER (element (column 4 u + g, problem 16 j + c)
    // at u RS + 64 j + 16 g + c): a wavefront's operand read is 64 consecutive doubles, free of bank conflicts (the matrix
    // pipes wait for nothing but these reads); the row stride RS = 64 PB + 2 keeps the transposing writes at two lanes per bank
    constexpr int RS = 64 * PB + 2;
    constexpr int KC = 16;                                         // k-steps per chunk = 64 columns of d, one per lane
    constexpr int NBUF = HAS_W ? 2 : 1;
    constexpr int BUFD = 16 * RS;
    constexpr int LDSD = KS * NBUF * BUFD > 1048 ? KS * NBUF * BUFD : 1048;
    static_assert(KS == 1 || (KS - 1) * RT * PB * 256 <= KS * NBUF * BUFD, "partial tiles reuse the staging buffers");
    __shared__ double lds[LDSD];
    const FpParams Q = Pv;
    const FpParams* P = &Q;
    const int tid = threadIdx.x, lane = tid & 63, wvr = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, c16 = lane & 15;
    const int kq = wvr / RT, wv = wvr % RT;                         // k group, row tile within the workgroup
#ifdef FW_TIMING
    unsigned long long _tr[4] = {(unsigned long long)wall_clock64(), 0, 0, 0};
#endif
    const int nb = P->nb, batch = P->batch, npanels = P->npanels;
    const int nrow = nb * FP_N;
    const int nrt = (nrow + 15) / 16, nrg = (nrt + RT - 1) / RT, npg = (npanels + PB - 1) / PB, rg8 = (nrg + 7) / 8;
    const int gemm_blocks = 8 * rg8 * npg;
    if (blockIdx.x == 0 && tid == 0 && P->handed) { P->handed[0] = 0; P->handed[1] = 0; }
    const int ngate = (npanels + 7) & ~7;                           // gate tasks first (see fmpc_cold_inv)
    if ((int)blockIdx.x < ngate) {
        if ((int)blockIdx.x < npanels) fi_gate(P, (int)blockIdx.x, lds);
#ifdef FW_TIMING
        if (tid == 0 && blockIdx.x < 1000) { fi_trace[4 * (7000 + blockIdx.x)] = _tr[0]; fi_trace[4 * (7000 + blockIdx.x) + 3] = (unsigned long long)wall_clock64(); }
#endif
        return;
    }
    const int bid = (int)blockIdx.x - ngate;
    if (bid >= gemm_blocks || P->gate_only) return;
    // With w the workgroups of an XCD stay on a few row groups (their rows of J, 111 KB per tile, stay in that XCD's L2) and
    // sweep the panels; without w a row tile needs 7 KB of J and the only concern is an even spread: 13 row groups over 8
    // XCDs the first way left three XCDs half empty and the others with a second round of workgroups (14.7 -> 9 us).
    const int x = bid & 7, q = bid >> 3;
    const int rg = HAS_W ? x + 8 * (q / npg) : bid % nrg, pg = HAS_W ? q % npg : bid / nrg;
    if (rg >= nrg || pg >= npg) return;
    const int rtq = RT * rg + wv;
    const bool live = rtq < nrt;
    const int rt = live ? rtq : nrt - 1;
    const int jks = P->jks;
    const int nch = (jks + KC - 1) / KC, cpg = (nch + KS - 1) / KS; // chunks of this k group: kq cpg .. kq cpg + cpg - 1 (the image is padded)
    const int cbase = kq * cpg;
    const double* jrow = P->jimg + (size_t)rt * P->jksp * 64 + lane;
    double* buf0 = lds + (size_t)kq * NBUF * BUFD;
    double* buf1 = buf0 + (NBUF - 1) * BUFD;
    // this wavefront loads the problems wv * PPW .. + PPW - 1 of the workgroup, lane = column within the chunk
    size_t pr[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const long p = (long)pg * NP + wv * PPW + i;
        pr[i] = (size_t)(p < batch ? p : batch - 1);
    }
    const double* x0 = P->x0; const double* x0p = P->x0p; const double* w = P->gw;
    const int WN = P->gwn;                                          // row length of the data behind [x0 ; x0_pre ; 0 0]: T n, or 2 n
    auto gload_w = [&](int ch, double v[PPW]) {                     // chunk ch >= 1: columns of w only
        int wi = 64 * ch + lane - 4 * FP_XKS;                       // beyond T n: a finite value times a zero column of J
        wi = wi < WN ? wi : WN - 1;
#pragma unroll
        for (int i = 0; i < PPW; ++i) v[i] = w[pr[i] * (size_t)WN + wi];
    };
    auto aload = [&](int ch, double a[KC]) {
#pragma unroll
        for (int u = 0; u < KC; ++u) a[u] = jrow[(size_t)(KC * ch + u) * 64];
    };
    d4 acc[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) acc[j] = (d4){0, 0, 0, 0};
    auto products = [&](const double* B, const double a[KC], int nks) {
        double bc[PB], bn[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) bc[j] = B[64 * j + lane];
#pragma unroll
        for (int u = 0; u < KC; ++u) {
            if (u < nks) {                                           // (compile-time after unrolling)
                if (u + 1 < nks) {
#pragma unroll
                    for (int j = 0; j < PB; ++j) bn[j] = B[(u + 1) * RS + 64 * j + lane];
                }
#pragma unroll
                for (int j = 0; j < PB; ++j) acc[j] = MFMA64(a[u], bc[j], acc[j]);
#pragma unroll
                for (int j = 0; j < PB; ++j) bc[j] = bn[j];
            }
        }
    };
    auto stage = [&](double* B, const double v[PPW]) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pl = wv * PPW + i;                                // problem within the workgroup; lane = column of the chunk
            B[(lane >> 2) * RS + 64 * (pl >> 4) + 16 * (lane & 3) + (pl & 15)] = v[i];
        }
    };
    double bv[PPW], a0[KC];
#pragma unroll
    for (int i = 0; i < PPW; ++i) bv[i] = *fi_addr(x0, x0p, HAS_W ? w : nullptr, pr[i], 64 * cbase + lane, WN);
    aload(cbase, a0);
#pragma unroll
    for (int i = 0; i < PPW; ++i) bv[i] = fi_zero(x0p != nullptr, HAS_W, 64 * cbase + lane) ? 0.0 : bv[i];
    stage(buf0, bv);
    __syncthreads();
    FI_TICK(1);
    if constexpr (!HAS_W) {
        products(buf0, a0, FP_XKS);
    } else {
        double a1[KC];
        const int cend = cbase + cpg;
        for (int ch = cbase; ch + 1 < cend; ch += 2) {
            gload_w(ch + 1, bv);
            aload(ch + 1, a1);
            products(buf0, a0, KC);
            stage(buf1, bv);
            __syncthreads();
            const int c2 = ch + 2 < cend ? ch + 2 : cend - 1;       // (past the last chunk: loaded again and dropped)
            gload_w(c2, bv);
            aload(c2, a0);
            products(buf1, a1, KC);
            stage(buf0, bv);
            __syncthreads();
        }
        if (cpg & 1) products(buf0, a0, KC);
    }
    if constexpr (KS > 1) {
        __syncthreads();                                            // the staging buffers become the partial tiles of the groups >= 1
        if (kq > 0) {
#pragma unroll
            for (int j = 0; j < PB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[(((kq - 1) * RT + wv) * PB + j) * 256 + r * 64 + lane] = acc[j][r];
        }
        __syncthreads();
        if (kq == 0) {
#pragma unroll
            for (int v = 1; v < KS; ++v)
#pragma unroll
                for (int j = 0; j < PB; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[j][r] += lds[(((v - 1) * RT + wv) * PB + j) * 256 + r * 64 + lane];
        }
    }
    FI_TICK(2);
    if (live && kq == 0) {
        const size_t pstride = (size_t)nrow * FP_NP;
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int pn = pg * PB + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rt + 4 * r + g;
                if (pn < npanels && row < nrow) P->nuws[(size_t)pn * pstride + (size_t)row * FP_NP + c16] = P->nuc[row] + acc[j][r];
            }
        }
    }
#ifdef FW_TIMING
    FI_TICK(3);
    if (lane == 0) {
        const int wid = blockIdx.x * RT * KS + wvr;
        if (wid < 8192) for (int q = 0; q < 4; ++q) fi_trace[4 * wid + q] = _tr[q];
    }
#endif
}

// ---------------------------------------------------------------- host side
// Launch shapes.  0: without w (one chunk): 4 row tiles x 2 panels.  With w --  1: <= 4 panels: 16 wavefronts split k, d read
// straight into registers;  2: <= 16 panels: one row tile, k over the 4 SIMDs;  3: <= 64 panels: 2 row tiles x 2 k groups;
// 4: beyond (only when the bound is raised): 4 row tiles x 2 k groups x 2 panels.
int fmpc_inv_variant(int npanels, int has_w, int jks) {
    if (!has_w) return 0;
    if (jks <= 64 && npanels > 4) return 6;                         // [B u1 ; B u2] instead of w: two chunks
    static int forced = -2;
    if (forced == -2) { const char* e = getenv("FMPC_INV_VARIANT"); forced = e && e[0] ? atoi(e) : -1; }   // experiments
    if (forced > 0) return forced;
    if (npanels <= 4) return 1;
    if (npanels <= 16) return 2;
    return npanels <= 64 ? 3 : 4;
}

hipError_t fmpc_launch_inv(const FpParams& P, hipStream_t stream) {
    const int nrow = P.nb * FP_N, nrt = (nrow + 15) / 16;
    const int variant = fmpc_inv_variant(P.npanels, P.gw != nullptr, P.jks);
    auto grid_for = [&](int RT, int PB) {
        return (P.gate_only ? 0 : 8 * (((nrt + RT - 1) / RT + 7) / 8) * ((P.npanels + PB - 1) / PB)) + ((P.npanels + 7) & ~7);
    };
    switch (variant) {
    case 0: {
        static int shape = -2;
        if (shape == -2) { const char* e = getenv("FMPC_INV_SHAPE0"); shape = e && e[0] ? atoi(e) : 0; }     // experiments
        if (shape == 1) hipLaunchKernelGGL((fmpc_cold_inv_rg<4, 4, 1, false>), dim3(grid_for(4, 4)), dim3(256), 0, stream, P);
        else if (shape == 2) hipLaunchKernelGGL((fmpc_cold_inv_rg<2, 2, 1, false>), dim3(grid_for(2, 2)), dim3(128), 0, stream, P);
        else if (shape == 3) hipLaunchKernelGGL((fmpc_cold_inv_rg<1, 4, 1, false>), dim3(grid_for(4, 1)), dim3(256), 0, stream, P);
        else if (shape == 4) hipLaunchKernelGGL((fmpc_cold_inv_rg<4, 2, 1, false>), dim3(grid_for(2, 4)), dim3(128), 0, stream, P);
        else hipLaunchKernelGGL((fmpc_cold_inv_rg<2, 4, 1, false>), dim3(grid_for(4, 2)), dim3(256), 0, stream, P);
        break;
    }
    case 1: hipLaunchKernelGGL((fmpc_cold_inv<16, 1>), dim3(8 * ((nrt + 7) / 8) * P.npanels + ((P.npanels + 7) & ~7)), dim3(1024), 0, stream, P); break;
    case 2: hipLaunchKernelGGL((fmpc_cold_inv_rg<1, 1, 4, true>), dim3(grid_for(1, 1)), dim3(256), 0, stream, P); break;
    case 3: hipLaunchKernelGGL((fmpc_cold_inv_rg<1, 2, 2, true>), dim3(grid_for(2, 1)), dim3(256), 0, stream, P); break;
    case 5: hipLaunchKernelGGL((fmpc_cold_inv_rg<2, 2, 2, true>), dim3(grid_for(2, 2)), dim3(256), 0, stream, P); break;
    case 6: hipLaunchKernelGGL((fmpc_cold_inv_rg<2, 4, 1, true>), dim3(grid_for(4, 2)), dim3(256), 0, stream, P); break;
    default: hipLaunchKernelGGL((fmpc_cold_inv_rg<2, 4, 2, true>), dim3(grid_for(4, 2)), dim3(512), 0, stream, P); break;
    }
    return hipGetLastError();
}
