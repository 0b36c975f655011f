// CDNA4 fastMPC, cold-start Newton step on PANELS of 16 problems (n = 27).
//
// Regime: the reference's own call, Fast_MPC2(..., x_init = []).mpc_fixed_log_newton(1, k)
// (README.md:548-556; inf_newton_solver.m:10-41 with one iteration).  From the mid-box start
// (fast_mpc_init.m:19-20) Phi, Y = C Phi^-1 C' and its block Cholesky factor are the SAME for every
// problem (SURVEY §7.2a regime (ii)), so the batch is a multi-right-hand-side solve: 16 problems are the N
// dimension of v_mfma_f64_16x16x4_f64 and every operation of the step is a 27 x 27 (or 144 x 27) matrix
// applied to a 27 x 16 panel.  An MFMA result tile (row = 4*reg + lane/16, col = lane%16) is directly the
// B operand of the next product (k-step 4*I + reg), so panels move between products in registers.
//
// With a constant primal start the Newton step collapses (r_d is affine in nu with matrix C', so
// Y (nu + d_nu) = r_p - C Phi^-1 r_d(nu = 0)):
//     b_i    = w_i + [i=0](A1 x0 + A2 x0_pre) + [i=1] A2 x0                  (fast_mpc_eq_const.m:39-68)
//     rhs_i  = ct_i - b_i ,  r_p,i = cp_i - b_i                                (ct, cp: host constants)
//     nu+    = Y^-1 rhs          block-penta-diagonal factor of the handle, in the product form
//                                y_i  = Linv_i rhs_i - W1_i y_{i-1} - W2_i y_{i-2}      (W = Linv U')
//                                nu+_i = Linv_i' y_i - V1_i nu+_{i+1} - V2_i nu+_{i+2}  (V = Linv' U)
//     d_u_j  = wc o (B' nu+_j - cu) ,  d_x_j = (2Q_j)^-1 (-dx0_j - nu+_j + A1' nu+_{j+1} + A2' nu+_{j+2} [- nu+_T])
//     z = zbar + d_z ,  nu = nu+          (full step t = 1)
// The line search (backtracking_inf_newton.m:2-11) accepts t = 1 iff ||e||^2 <= (1-alpha)^2 rho^2 with
// e = k P'DP d_z and the exit test (inf_newton_solver.m:19-22) needs rho, ||r_p||: the
// panel path decides both only with a wide margin (||e||^2 <= rho^2 / 2; ||r_p|| or rho a factor 2 above the
// exit thresholds) and hands every other problem, untouched, to the exact one-wave-per-problem path
// (fmpc_kernel_wave.hip) through a selection list.  No result of this file depends on the margin.
//
// One 512-thread workgroup per panel, one workgroup per CU.  LDS: the rhs -> y -> nu+ panel
// ((27 nb + 1) x 16 doubles, 107 KB), the MFMA images of B', A1', A2' (46 KB) and the u constants.
// Stage-parallel phases (S1 rhs, S3 Linv' y, S5 d_z) deal stages round-robin to the 8 waves; the two
// serial sweeps (S2, S4) run on 4 waves = (row block) x (lag 1 | lag 2 term), one barrier per stage.
#include <hip/hip_runtime.h>
#include <math.h>
#include "fmpc_device.h"
#include "fmpc_panel.h"
#include "../../include/fastmpc.h"

#define FP_WAVES 8
#define FP_THREADS (FP_WAVES * 64)
#define FP_FN __device__ __forceinline__

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
typedef __attribute__((address_space(3))) double* fp_lds_t;
typedef const __attribute__((address_space(3))) double* fp_clds_t;
typedef const FpParams __attribute__((address_space(4))) * FpKP;

#ifdef FW_TIMING
__device__ unsigned long long fp_timing[16];
extern "C" int fmpc_debug_panel_timing(unsigned long long* out) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fp_timing), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(fp_timing), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

__device__ __forceinline__ FpKP fp_uniform(FpKP P) {          // see fw_uniform (fmpc_kernel_wave.hip)
    const unsigned long long a = (unsigned long long)P;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (FpKP)(((unsigned long long)hi << 32) | lo);
}

// LDS map (doubles)
struct FpLds {
    int Y, BT, A1T, A2T, UC, XQ, RED, FLAG, total;
};
__host__ __device__ static inline FpLds fp_lds_layout(int nb, int mp) {
    FpLds L; int o = 0;
    L.Y = o;   o += (nb * FP_N + 1) * FP_NP;
    L.BT = o;  o += (mp / 16) * FP_KS * 64;
    L.A1T = o; o += FP_IMG;
    L.A2T = o; o += FP_IMG;
    o = (o + 1) & ~1;
    L.UC = o;  o += 4 * mp;
    L.XQ = o;  o += 4 * 32;                         // [xc | xc(last stage) | iq | iq(last stage)]
    L.RED = o; o += 3 * FP_WAVES * FP_NP;          // per wave and problem: ||r_p||^2, ||e||^2, ||r_d||^2 (+ guard)
    o += FP_WAVES * FP_NP;
    L.FLAG = o; o += 2;
    L.total = o;
    return L;
}

// ---- panel <-> LDS helpers.  Stage i of the panel occupies rows i*27 .. i*27+26, 16 problems per row.
// B-operand layout of a stage vector: k-step ks, lane (g = lane/16, c = lane%16) holds row 4 ks + g.
// Row 27 (ks = 6, g = 3) is the first row of the next stage: finite, and multiplied by a zero image column.
__device__ __forceinline__ void fp_load_b(fp_clds_t Y, int i, int g, int c16, double v[FP_KS]) {
    const fp_clds_t s = Y + (i * FP_N + g) * FP_NP + c16;
#pragma unroll
    for (int ks = 0; ks < FP_KS; ++ks) v[ks] = s[4 * ks * FP_NP];
}
// D layout of row block I: reg r holds row 16 I + 4 r + g
__device__ __forceinline__ d4 fp_load_d(fp_clds_t Y, int i, int I, int g, int c16) {
    d4 a;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + 4 * r + g;
        a[r] = Y[(i * FP_N + (row < FP_N ? row : 0)) * FP_NP + c16];
    }
    return a;
}
__device__ __forceinline__ void fp_store_d(fp_lds_t Y, int i, int I, int g, int c16, d4 a) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + 4 * r + g;
        if (row < FP_N) Y[(i * FP_N + row) * FP_NP + c16] = a[r];
    }
}
// acc += IMG[I] * v   (IMG: A-operand image of a 27 x 27 matrix in global memory)
__device__ __forceinline__ d4 fp_mm_g(const double* img, int I, int lane, const double v[FP_KS], d4 acc) {
    double a[FP_KS];
#pragma unroll
    for (int ks = 0; ks < FP_KS; ++ks) a[ks] = img[(I * FP_KS + ks) * 64 + lane];
#pragma unroll
    for (int ks = 0; ks < FP_KS; ++ks) acc = MFMA64(a[ks], v[ks], acc);
    return acc;
}
__device__ __forceinline__ d4 fp_mm_l(fp_clds_t img, int I, int lane, const double v[FP_KS], d4 acc) {
#pragma unroll
    for (int ks = 0; ks < FP_KS; ++ks) acc = MFMA64(img[(I * FP_KS + ks) * 64 + lane], v[ks], acc);
    return acc;
}
__device__ __forceinline__ double fp_sum_g(double v) {         // sum over the 4 lane groups (same problem)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
// S1: rhs_i = ct_i - b_i, r~_i = Linv_i rhs_i -> panel;  ||r_p||^2 per problem.
// The global operands of the next stage are loaded (unconditionally, index clamped) before the current
// one is processed: the waits in front of the MFMAs then leave those loads in flight.
struct FpS1 { double w[8], ct[8], cp[8], img[2][FP_KS]; };
FP_FN void fp_s1(FpKP Pin, double* lds_g, int panel) {
    const FpKP P = fp_uniform(Pin);
    panel = __builtin_amdgcn_readfirstlane(panel);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c16 = lane & 15;
    const int T = P->T, nb = P->nb, batch = P->batch;
    const FpLds L = fp_lds_layout(nb, P->mp);
    const fp_lds_t Y = (fp_lds_t)lds_g + L.Y;
    const fp_lds_t red = (fp_lds_t)lds_g + L.RED;
    const int p = panel * FP_NP + c16;
    const size_t pc = p < batch ? p : batch - 1;
    const double* w = P->w ? P->w + pc * (size_t)T * FP_N : nullptr;
    const double* x0 = P->x0 + pc * FP_N;
    const double* x0p = P->x0p ? P->x0p + pc * FP_N : nullptr;
    const FpVec V = fp_vec_layout(nb, T);
    const double* ct = P->vec + V.ct;
    const double* cp = P->vec + V.cp;
    const double* simg = P->simg + FP_SIMG_LINV * FP_IMG + lane;
    const bool var2 = P->var2 != 0;
    int rowc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int row = 16 * (e >> 2) + 4 * (e & 3) + g; rowc[e] = row < FP_N ? row : 0; }
    auto ld = [&](int i, FpS1& d) {
        const int ic = i < nb ? i : nb - 1;
        const int iw = ic < T ? ic : T - 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            d.w[e] = w ? w[iw * FP_N + rowc[e]] : 0.0;
            d.ct[e] = ct[ic * 32 + rowc[e]];
            d.cp[e] = cp[ic * 32 + rowc[e]];
        }
        const double* im = simg + (size_t)ic * 6 * FP_IMG;
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int ks = 0; ks < FP_KS; ++ks) d.img[I][ks] = im[(I * FP_KS + ks) * 64];
    };
    double xv[FP_KS], xp[FP_KS];               // x0, x0_pre in B-operand layout (used by stages 0 and 1)
#pragma unroll
    for (int ks = 0; ks < FP_KS; ++ks) {
        const int k = 4 * ks + g;
        const double t0 = x0[k < FP_N ? k : 0];
        const double t1 = x0p ? x0p[k < FP_N ? k : 0] : 0.0;
        xv[ks] = k < FP_N ? t0 : 0.0; xp[ks] = k < FP_N ? t1 : 0.0;
    }
    double rp2 = 0.0;
    auto comp = [&](int i, const FpS1& d) {
        if (i >= nb) return;
        d4 bx[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        if (i == 0 || (i == 1 && var2)) {                 // the prediction A1 x0 + A2 x0_pre enters b_0, b_1
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                if (i == 0) {
                    bx[I] = fp_mm_g(P->aimg + FP_AIMG_A1 * FP_IMG, I, lane, xv, bx[I]);
                    if (var2) bx[I] = fp_mm_g(P->aimg + FP_AIMG_A2 * FP_IMG, I, lane, xp, bx[I]);
                } else {
                    bx[I] = fp_mm_g(P->aimg + FP_AIMG_A2 * FP_IMG, I, lane, xv, bx[I]);
                }
            }
        }
        double v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool rok = 16 * (e >> 2) + 4 * (e & 3) + g < FP_N;
            const double b = (i < T ? d.w[e] : 0.0) + bx[e >> 2][e & 3];
            const double rp = d.cp[e] - b;
            if (rok) rp2 += rp * rp;
            v[e] = rok ? d.ct[e] - b : 0.0;
        }
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            d4 o = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < FP_KS; ++ks) o = MFMA64(d.img[I][ks], v[ks], o);
            fp_store_d(Y, i, I, g, c16, o);
        }
    };
    FpS1 A, B;
    ld(wv, A);
    for (int i = wv; i < nb; i += 2 * FP_WAVES) {
        ld(i + FP_WAVES, B);
        comp(i, A);
        ld(i + 2 * FP_WAVES, A);
        comp(i + FP_WAVES, B);
    }
    rp2 = fp_sum_g(rp2);
    if (g == 0) red[wv * FP_NP + c16] = rp2;
}

// ------------------------------------------------------------------------------------------------
// S2 / S4: the serial sweeps.  Wave q < 4: row block I = q & 1, term = q >> 1.
//   forward  (BWD = 0): step s = 1..nb-1:  term 0: Y[s]   += -W1_s   Y[s-1] ;  term 1: Y[s+1] += -W2_{s+1} Y[s-1]
//   backward (BWD = 1): step s = nb-2..0:  term 0: Y[s]   += -V1_s   Y[s+1] ;  term 1: Y[s-1] += -V2_{s-1} Y[s+1]
// One barrier per step; every wave of the workgroup calls this.  The worker loop is branch-free (the
// compiler's s_waitcnt placement gives up on conditional loads): the images are prefetched two steps ahead
// into three rotating register sets, and steps without a target (the ends of the lag-2 chains, padding of the
// step count to a multiple of 3) add a ZERO image (stage slot nb) into a stage nobody else updates.
template <int BWD>
FP_FN void fp_sweep(FpKP Pin, double* lds_g) {
    const FpKP P = fp_uniform(Pin);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c16 = lane & 15;
    const int nb = P->nb;
    const int nsteps = nb - 1, nsteps3 = (nsteps + 2) / 3 * 3;
    if (wv >= 4) {
        for (int q = 0; q < nsteps3; ++q) __syncthreads();
        return;
    }
    const FpLds L = fp_lds_layout(nb, P->mp);
    const fp_lds_t Y = (fp_lds_t)lds_g + L.Y;
    const int I = wv & 1, term = wv >> 1;
    const double* simg = P->simg + (size_t)((BWD ? FP_SIMG_V1 : FP_SIMG_W1) + term) * FP_IMG + (size_t)I * FP_KS * 64 + lane;
    const int dummy = BWD ? nb - 1 : 0;
    auto target = [&](int q) {
        const int s = BWD ? nb - 2 - q : 1 + q;
        const int t = BWD ? s - term : s + term;
        return (q < nsteps && t >= 0 && t < nb) ? t : -1;
    };
    auto load_img = [&](int q, double a[FP_KS]) {
        const int t = target(q);
        const double* s = simg + (size_t)(t < 0 ? nb : t) * 6 * FP_IMG;
#pragma unroll
        for (int ks = 0; ks < FP_KS; ++ks) a[ks] = s[ks * 64];
    };
    auto step = [&](int q, const double a[FP_KS]) {
        const int t0 = target(q);
        const int t = t0 < 0 ? dummy : t0;
        const int s = BWD ? nb - 2 - q : 1 + q;
        const int src = t0 < 0 ? dummy : (BWD ? s + 1 : s - 1);
        double v[FP_KS];
        fp_load_b(Y, src, g, c16, v);
        d4 acc = fp_load_d(Y, t, I, g, c16);
#pragma unroll
        for (int ks = 0; ks < FP_KS; ++ks) acc = MFMA64(a[ks], v[ks], acc);
        fp_store_d(Y, t, I, g, c16, acc);
        __syncthreads();
    };
    double a0[FP_KS], a1[FP_KS], a2[FP_KS];
    load_img(0, a0); load_img(1, a1);
    for (int q = 0; q < nsteps3; q += 3) {
        load_img(q + 2, a2);
        step(q, a0);
        load_img(q + 3, a0);
        step(q + 1, a1);
        load_img(q + 4, a1);
        step(q + 2, a2);
    }
}

// ------------------------------------------------------------------------------------------------
// S3: y~_i = Linv_i' y_i   (stage-parallel, in place; next stage's image prefetched)
FP_FN void fp_s3(FpKP Pin, double* lds_g) {
    const FpKP P = fp_uniform(Pin);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c16 = lane & 15;
    const int nb = P->nb;
    const FpLds L = fp_lds_layout(nb, P->mp);
    const fp_lds_t Y = (fp_lds_t)lds_g + L.Y;
    const double* simg = P->simg + FP_SIMG_LINVT * FP_IMG + lane;
    auto ld = [&](int i, double a[2][FP_KS]) {
        const double* im = simg + (size_t)(i < nb ? i : nb - 1) * 6 * FP_IMG;
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int ks = 0; ks < FP_KS; ++ks) a[I][ks] = im[(I * FP_KS + ks) * 64];
    };
    auto comp = [&](int i, const double a[2][FP_KS]) {
        if (i >= nb) return;
        double v[FP_KS];
        fp_load_b(Y, i, g, c16, v);
        d4 o0 = {0, 0, 0, 0}, o1 = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < FP_KS; ++ks) { o0 = MFMA64(a[0][ks], v[ks], o0); o1 = MFMA64(a[1][ks], v[ks], o1); }
        fp_store_d(Y, i, 0, g, c16, o0);
        fp_store_d(Y, i, 1, g, c16, o1);
    };
    double A[2][FP_KS], B[2][FP_KS];
    ld(wv, A);
    for (int i = wv; i < nb; i += 2 * FP_WAVES) {
        ld(i + FP_WAVES, B);
        comp(i, A);
        ld(i + 2 * FP_WAVES, A);
        comp(i + FP_WAVES, B);
    }
}

// ------------------------------------------------------------------------------------------------
// S5: d_z from nu+ (panel), z = zbar + d_z and nu = nu+ written out, ||e||^2 per problem.
// The products of this phase are only consumed element-wise, so they are computed TRANSPOSED: the nu+ panel
// registers (B-operand layout) are also a valid A operand with the problems as rows, and the unpermuted
// images of B', A1', A2' serve as B operands with the entries as columns.  Result register r of lane
// (g, c) is then (problem 4 r + g, entry 16 J + c): each store instruction writes, for 4 problems, 16
// consecutive entries (128 contiguous bytes) and the per-entry constants are one LDS read per lane.
// No global LOAD in this phase (a load behind a store waits for the store: vmcnt is in order); lanes of
// problems beyond the batch write to the dump area instead of being predicated.
template <int HAS_NU>
FP_FN void fp_s5(FpKP Pin, double* lds_g, int panel) {
    const FpKP P = fp_uniform(Pin);
    panel = __builtin_amdgcn_readfirstlane(panel);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c16 = lane & 15;
    const int T = P->T, nb = P->nb, batch = P->batch, m = P->m, mp = P->mp, s = FP_N + m;
    const FpLds L = fp_lds_layout(nb, mp);
    const fp_clds_t Y = (fp_clds_t)lds_g + L.Y;
    const fp_clds_t BT = (fp_clds_t)lds_g + L.BT + lane;
    const fp_clds_t A1T = (fp_clds_t)lds_g + L.A1T + lane;
    const fp_clds_t A2T = (fp_clds_t)lds_g + L.A2T + lane;
    const fp_clds_t UC = (fp_clds_t)lds_g + L.UC + c16;     // [c1 | wc | hc | ubar], c1 = -wc cu
    const fp_clds_t XQ = (fp_clds_t)lds_g + L.XQ;
    const fp_lds_t red = (fp_lds_t)lds_g + L.RED + FP_WAVES * FP_NP;
    double* zq[4]; double* nq[4];                           // problem 4 r + g of the panel
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int p = panel * FP_NP + 4 * r + g;
        zq[r] = (p < batch ? P->zout + (size_t)p * T * s : P->dump) + c16;
        nq[r] = HAS_NU ? (p < batch ? P->nuout + (size_t)p * nb * FP_N : P->dump + (size_t)T * s) + c16 : nullptr;
    }
    const bool has_xf = P->has_xf != 0, var2 = P->var2 != 0;
    const int NJ = mp / 16, NJF = m / 16;                   // column blocks, full column blocks
    double eps2[4] = {0.0, 0.0, 0.0, 0.0};
    for (int j = wv; j < T; j += FP_WAVES) {
        double v0[FP_KS], v1[FP_KS], v2[FP_KS];
        fp_load_b(Y, j, g, c16, v0);
        const bool h1 = j + 1 < T, h2 = j + 2 < T && var2;
        fp_load_b(Y, h1 ? j + 1 : j, g, c16, v1);
        fp_load_b(Y, h2 ? j + 2 : j, g, c16, v2);
#pragma unroll
        for (int ks = 0; ks < FP_KS; ++ks) { v1[ks] = h1 ? v1[ks] : 0.0; v2[ks] = h2 ? v2[ks] : 0.0; }
        const bool last = j + 1 == T;
        const size_t zoff = (size_t)j * s;
        // ---- u entries: d_u = wc o (B' nu+_j - cu)
        auto mm = [&](int J) {
            d4 acc = {0, 0, 0, 0};
            const fp_clds_t im = BT + J * FP_KS * 64;
#pragma unroll
            for (int ks = 0; ks < FP_KS; ++ks) acc = MFMA64(v0[ks], im[ks * 64], acc);
            return acc;
        };
        auto epi = [&](int J, d4 acc) {
            const fp_clds_t uc = UC + 16 * J;
            const double c1 = uc[0], wc = uc[mp], hc = uc[2 * mp], ub = uc[3 * mp];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double du = fma(wc, acc[r], c1);
                const double e = hc * du;
                eps2[r] = fma(e, e, eps2[r]);
                zq[r][zoff + 16 * J] = ub + du;
            }
        };
        if (NJF > 0) {
            d4 a0 = mm(0);
            int J = 0;
            for (; J + 2 < NJF; J += 2) {
                const d4 a1 = mm(J + 1);
                epi(J, a0);
                a0 = mm(J + 2);
                epi(J + 1, a1);
            }
            if (J + 1 < NJF) {
                const d4 a1 = mm(J + 1);
                epi(J, a0);
                epi(J + 1, a1);
            } else {
                epi(J, a0);
            }
        }
        if (NJF < NJ) {                                   // partial last column block (m not a multiple of 16)
            const d4 acc = mm(NJF);
            const fp_clds_t uc = UC + 16 * NJF;
            const bool cok = 16 * NJF + c16 < m;
            const double c1 = uc[0], wc = uc[mp], hc = uc[2 * mp], ub = uc[3 * mp];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double du = fma(wc, acc[r], c1);
                const double e = hc * du;
                if (cok) { eps2[r] = fma(e, e, eps2[r]); zq[r][zoff + 16 * NJF] = ub + du; }
            }
        }
        // ---- x entries: d_x = (2Q)^-1 (-dx0 - nu+_j + A1' nu+_{j+1} + A2' nu+_{j+2} [- nu+_T])
        const fp_clds_t xcv = XQ + (last ? 32 : 0), iqv = XQ + 64 + (last ? 32 : 0);
        const bool xfl = last && has_xf;
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            d4 h = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < FP_KS; ++ks) h = MFMA64(v1[ks], A1T[(I * FP_KS + ks) * 64], h);
#pragma unroll
            for (int ks = 0; ks < FP_KS; ++ks) h = MFMA64(v2[ks], A2T[(I * FP_KS + ks) * 64], h);
            const int row = 16 * I + c16;
            const bool rok = row < FP_N;
            const int rc = rok ? row : 0;
            const double xc = xcv[rc], iq = iqv[rc];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double nj = Y[(j * FP_N + rc) * FP_NP + 4 * r + g];
                const double t = Y[((xfl ? T : j) * FP_N + rc) * FP_NP + 4 * r + g];
                const double nx = xfl ? t : 0.0;
                const double zx = xc + iq * (h[r] - nj - nx);
                if (rok) {                                 // rows 27..31 of the second row block do not exist
                    zq[r][zoff + m + 16 * I] = zx;
                    if (HAS_NU) {
                        nq[r][j * FP_N + 16 * I] = nj;
                        if (xfl) nq[r][T * FP_N + 16 * I] = nx;
                    }
                }
            }
        }
    }
    // per problem: sum over the 16 entry lanes; lane (g, 0) then holds problems 4 r + g
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double v = eps2[r];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        if (c16 == 0) red[wv * FP_NP + 4 * r + g] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// S6 (only when the cheap acceptance test fails somewhere in the panel and nu0 is given):
// ||r_d(nu0)||^2 per problem.  x entries element-wise; u entries as a 27-dimensional quadratic form
//   sum_j |cu - B' nu_j|^2 = T |cu|^2 - 2 (B cu)' sum_j nu_j + sum_j nu_j' (B B') nu_j
// red3 = ||r_d||^2 - T |cu|^2, red4 = the positive part of the quadratic form (cancellation guard).
FP_FN void fp_s6(FpKP Pin, double* lds_g, int panel) {
    const FpKP P = fp_uniform(Pin);
    panel = __builtin_amdgcn_readfirstlane(panel);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c16 = lane & 15;
    const int T = P->T, nb = P->nb, batch = P->batch;
    const FpLds L = fp_lds_layout(nb, P->mp);
    const fp_clds_t A1T = (fp_clds_t)lds_g + L.A1T;
    const fp_clds_t A2T = (fp_clds_t)lds_g + L.A2T;
    const fp_lds_t red3 = (fp_lds_t)lds_g + L.RED + 2 * FP_WAVES * FP_NP;
    const fp_lds_t red4 = red3 + FP_WAVES * FP_NP;
    const int p = panel * FP_NP + c16;
    const size_t pc = p < batch ? p : batch - 1;
    const double* nu = P->nu0 + pc * (size_t)nb * FP_N;
    const FpVec V = fp_vec_layout(nb, T);
    const double* dx0 = P->vec + V.dx0;
    const double* bcu = P->vec + V.bcu;
    const bool has_xf = P->has_xf != 0, var2 = P->var2 != 0;
    double rd2 = 0.0, pos = 0.0;
    auto load_nu = [&](int i, bool on, double v[FP_KS]) {
#pragma unroll
        for (int ks = 0; ks < FP_KS; ++ks) {
            const int k = 4 * ks + g;
            const double t = nu[(on ? i : 0) * FP_N + (k < FP_N ? k : 0)];
            v[ks] = (on && k < FP_N) ? t : 0.0;
        }
    };
    for (int j = wv; j < T; j += FP_WAVES) {
        double v0[FP_KS], v1[FP_KS], v2[FP_KS], vx[FP_KS];
        const bool last = j + 1 == T;
        load_nu(j, true, v0);
        load_nu(j + 1, j + 1 < T, v1);
        load_nu(j + 2, j + 2 < T && var2, v2);
        load_nu(T, last && has_xf, vx);
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            d4 h = fp_mm_l(A1T, I, lane, v1, (d4){0, 0, 0, 0});
            h = fp_mm_l(A2T, I, lane, v2, h);
            const d4 qd = fp_mm_g(P->aimg + FP_AIMG_BBT * FP_IMG, I, lane, v0, (d4){0, 0, 0, 0});
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * I + 4 * r + g;
                const bool rok = row < FP_N;
                const int rc = rok ? row : 0;
                const double nuj = v0[4 * I + r];
                const double x = dx0[j * 32 + rc] + nuj - h[r] + vx[4 * I + r];
                if (rok) {
                    const double qf = nuj * qd[r];
                    pos += qf;
                    rd2 += x * x + (qf - 2.0 * bcu[rc] * nuj);
                }
            }
        }
    }
    rd2 = fp_sum_g(rd2);
    pos = fp_sum_g(pos);
    if (g == 0) { red3[wv * FP_NP + c16] = rd2; red4[wv * FP_NP + c16] = pos; }
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(FP_THREADS, 2) fmpc_cold_panel(FpParams Pv) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const FpKP P = (FpKP)__builtin_amdgcn_kernarg_segment_ptr();
    const int tid = threadIdx.x;
    const int nb = P->nb, mp = P->mp, batch = P->batch;
    const FpLds L = fp_lds_layout(nb, mp);
    // ---- shared images and constants; a finite panel (pad rows are read, multiplied by zero columns)
    for (int i = tid; i < (nb * FP_N + 1) * FP_NP; i += FP_THREADS) lds[L.Y + i] = 0.0;
    for (int i = tid; i < (mp / 16) * FP_KS * 64; i += FP_THREADS) lds[L.BT + i] = P->btimg[i];
    for (int i = tid; i < FP_IMG; i += FP_THREADS) {
        lds[L.A1T + i] = P->aimg[FP_AIMG_A1T * FP_IMG + i];
        lds[L.A2T + i] = P->aimg[FP_AIMG_A2T * FP_IMG + i];
    }
    for (int i = tid; i < 4 * mp; i += FP_THREADS)               // [c1 | wc | hc | ubar], c1 = -wc cu
        lds[L.UC + i] = i < mp ? -P->ucon[mp + i] * P->ucon[i] : P->ucon[i];
    if (tid < 128) {
        const FpVec V = fp_vec_layout(nb, P->T);
        const int a = tid >> 5, r = tid & 31;                       // xc, xc(last), iq, iq(last)
        lds[L.XQ + tid] = P->vec[(a < 2 ? V.xc : V.iq) + ((a & 1) ? P->T - 1 : 0) * 32 + r];
    }
    __syncthreads();
#ifdef FW_TIMING
    unsigned long long _k0 = __builtin_readcyclecounter(), _k1, _ka[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define FP_TICK(k) do { _k1 = __builtin_readcyclecounter(); _ka[k] += _k1 - _k0; _k0 = _k1; } while (0)
#else
#define FP_TICK(k)
#endif
    const double* red = lds + L.RED;
    int* flag = (int*)(lds + L.FLAG);
    for (int panel = blockIdx.x; panel < P->npanels; panel += gridDim.x) {
        fp_s1(P, lds, panel);
        __syncthreads();
        FP_TICK(0);
        fp_sweep<0>(P, lds);
        FP_TICK(1);
        fp_s3(P, lds);
        __syncthreads();
        FP_TICK(2);
        fp_sweep<1>(P, lds);
        FP_TICK(3);
        if (P->nuout) fp_s5<1>(P, lds, panel); else fp_s5<0>(P, lds, panel);
        __syncthreads();
        FP_TICK(4);
        // ---- acceptance: threads 0..15, one problem each
        const int p = panel * FP_NP + tid;
        double rp2 = 0.0, eps2 = 0.0;
        bool undecided = false;
        if (tid < FP_NP) {
            for (int w = 0; w < FP_WAVES; ++w) { rp2 += red[w * FP_NP + tid]; eps2 += red[(FP_WAVES + w) * FP_NP + tid]; }
            const bool fin = rp2 < 1e300 && eps2 < 1e300;
            const bool cheap = fin && rp2 > 4e-16 && eps2 <= 0.5 * rp2;
            undecided = p < batch && !cheap;
        }
        if (tid == 0) *flag = 0;
        __syncthreads();
        if (undecided) *flag = 1;
        __syncthreads();
        const bool any = *flag != 0;                       // uniform
        const bool have_nu = P->nu0 != nullptr;
        if (any && have_nu) {
            fp_s6(P, lds, panel);
            __syncthreads();
        }
        if (tid < FP_NP && p < batch) {
            bool clear = !undecided;
            if (undecided) {
                double rd2 = P->rd2_0, pos = P->rd2_0;
                if (have_nu) {
                    rd2 = P->T * P->sa_cu; pos = rd2;
                    for (int w = 0; w < FP_WAVES; ++w) { rd2 += red[(2 * FP_WAVES + w) * FP_NP + tid]; pos += red[(3 * FP_WAVES + w) * FP_NP + tid]; }
                }
                const double rho2 = rd2 + rp2;
                const bool fin = rp2 < 1e300 && eps2 < 1e300 && rd2 < 1e300 && pos < 1e300;
                clear = fin && rd2 >= 1e-6 * pos && (rp2 > 4e-16 || rho2 > 4e-12) && eps2 <= 0.5 * rho2;
            }
            if (clear) {
                if (P->status) P->status[p] = FMPC_OK;
                if (P->iters) P->iters[p] = 1;
                if (P->step) {
                    for (int q = 0; q < P->step_ld; ++q) P->step[(size_t)p * P->step_ld + q] = q == 0 ? 1.0 : -1.0;
                }
            } else {
                const int idx = atomicAdd(P->sel_count, 1);
                P->sel[idx] = p;
            }
        }
        __syncthreads();
        FP_TICK(5);
    }
#ifdef FW_TIMING
    if ((tid & 63) == 0) for (int q = 0; q < 8; ++q) atomicAdd(&fp_timing[q], _ka[q]);
#endif
}

// ---------------------------------------------------------------- host side
size_t fmpc_panel_lds_bytes(int nb, int mp) { return (size_t)fp_lds_layout(nb, mp).total * sizeof(double); }

hipError_t fmpc_panel_prepare(size_t lds_bytes) {
    return hipFuncSetAttribute((const void*)fmpc_cold_panel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

hipError_t fmpc_launch_panel(const FpParams& P, int grid, size_t lds_bytes, hipStream_t stream) {
    hipLaunchKernelGGL(fmpc_cold_panel, dim3(grid), dim3(FP_THREADS), lds_bytes, stream, P);
    return hipGetLastError();
}
