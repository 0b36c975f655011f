// CDNA4 fastMPC, cold-start Newton step on PANELS of 16 problems (n = 27): the dual solve.
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
// THIS kernel (fmpc_cold_panel) produces nu+ (the new dual variable at the full step t = 1) and, per problem,
// ||r_p||^2 and a lower bound of rho^2 = ||r_d||^2 + ||r_p||^2.  It is latency-bound: two serial sweeps over
// the horizon on one workgroup per panel.  The throughput-bound rest of the step,
//     d_u_j  = wc o (B' nu+_j - cu) ,  d_x_j = (2Q_j)^-1 (-dx0_j - nu+_j + A1' nu+_{j+1} + A2' nu+_{j+2} [- nu+_T])
//     z = zbar + d_z ,  ||e||^2 ,  the step-length decision,
// is fmpc_cold_dz (fmpc_kernel_dz.hip), one independent task per (panel, stage) over the whole chip.
//
// One 512-thread workgroup per panel, one workgroup per CU.  LDS: the rhs -> y -> nu+ panel
// ((27 nb + 1) x 16 doubles, 107 KB).  Phases:
//   S1  stage-parallel: Y[i] = Linv_i rhs_i                         (8 waves; no product at all without w)
//   S2  forward sweep: per step up to 4 edges  Y[t] += IMG Y[s]  (a wave pair = the two row blocks of a target,
//       7 MFMAs per wave), one LDS-only barrier per step
//   S3  stage-parallel: Y[i] = Linv_i' y_i                          (8 waves)
//   S4  backward sweep like S2; then nu+ leaves for HBM in panel layout (full-rate coalesced stores)
// S2 and S4 execute a host-built SCHEDULE (fmpc_upload_panel in fmpc_api.hip): the factor of Y is computed on the
// host in a twisted elimination order (two independent chains that meet in the middle of the horizon), so the
// device only sees operator images and a list of edges per step.
#include <hip/hip_runtime.h>
#include <math.h>
#include "fmpc_device.h"
#include "fmpc_panel.h"
#include "../../include/fastmpc.h"

#define FP_WAVES 8
#define FP_THREADS (FP_WAVES * 64)
#define FP_FN __device__ __forceinline__

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
typedef __attribute__((address_space(3))) double* fp_lds_t;
typedef const __attribute__((address_space(3))) double* fp_clds_t;
typedef const FpParams* FpKP;          // points at a LOCAL copy of the kernel argument: every field is loaded from the
                                        // kernarg segment once, at kernel entry (scattered s_load + s_waitcnt pairs cost ~6k cycles per panel)

#ifdef FW_TIMING
__device__ unsigned long long fp_timing[16];

extern "C" int fmpc_debug_panel_timing(unsigned long long* out) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fp_timing), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(fp_timing), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL
// access (vmcnt(0)), which would expose the latency of the image prefetches and of the z stores at every
// pipeline step; the steps communicate through LDS alone.
__device__ __forceinline__ void fp_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LDS map (doubles)
struct FpLds { int Y, RED, SCH, total; };
__host__ __device__ static inline FpLds fp_lds_layout(int nb, int mp) {
    (void)mp;
    FpLds L; int o = 0;
    L.Y = o;   o += (nb * FP_N + 1) * FP_NP;
    L.RED = o; o += (FP_WAVES + 1) * FP_NP;         // per wave and problem: ||r_p||^2; then the lower bound of ||r_d||^2
    L.SCH = o; o += FP_MAX_STEPS(nb) * FP_STEP_INTS; // both sweep schedules (ints; forward rows, then backward rows)
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
__device__ __forceinline__ double fp_sum_g(double v) {         // sum over the 4 lane groups (same problem)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
// S1, task (stage i, row block I) = wave-strided over 2 nb tasks:
//     Y[i][I] = rt_i[I] - Linv_i[I] w_i        rt_i = Linv_i ct_i (host)
//               [+ i = 0: -Linv_0 (A1 x0 + A2 x0_pre), i = 1: -Linv_1 A2 x0, as precomputed product images]
// and ||r_p||^2 per problem: tasks with I = 0 add stage i >= 2 (a host constant without w); the tasks of
// stages 0 and 1 add the rows of their block (they need A1 x0 + A2 x0_pre itself).  *rdlb (last wave only): a
// lower bound of ||r_d(nu0)||^2, per lane.
template <int HAS_W>
FP_FN double fp_s1(FpKP P, double* lds_g, int panel, double* rdlb, unsigned long long* tk) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c16 = lane & 15;
    const int T = P->T, nb = P->nb, batch = P->batch;
    const FpLds L = fp_lds_layout(nb, P->mp);
    const fp_lds_t Y = (fp_lds_t)lds_g + L.Y;
    const int p = panel * FP_NP + c16;
    const size_t pc = p < batch ? p : batch - 1;
    const double* w = HAS_W ? P->w + pc * (size_t)T * FP_N : nullptr;
    const double* x0 = P->x0 + pc * FP_N;
    const double* x0p = P->x0p ? P->x0p + pc * FP_N : nullptr;
    const FpVec V = fp_vec_layout(nb, T);
    const double* cp = P->vec + V.cp;
    const double* rt = P->vec + V.rt;
    const bool var2 = P->var2 != 0;
    double rp2 = 0.0;
    // (the code of this phase runs once per panel: it is kept small, cold instruction fetch is what it costs)
#ifdef FW_TIMING
    unsigned long long _s0 = __builtin_readcyclecounter(), _s1;
#define FP_STICK(k) do { _s1 = __builtin_readcyclecounter(); tk[k] += _s1 - _s0; _s0 = _s1; } while (0)
#else
#define FP_STICK(k)
#endif
    if (wv == FP_WAVES - 1 && P->nu0) {
        // ||r_d(nu0)||^2 >= its x entries of the last stage, dx0_{T-1} + nu0_{T-1} [+ nu0_T]: no product needed
        const double* nu = P->nu0 + pc * (size_t)nb * FP_N;
        const double* dx0 = P->vec + V.dx0 + (T - 1) * 32;
        double a = 0.0;
#pragma unroll
        for (int e = 0; e < FP_KS; ++e) {
            const bool rok = 4 * e + g < FP_N;
            const int rc = rok ? 4 * e + g : 0;
            double x = dx0[rc] + nu[(T - 1) * FP_N + rc];
            if (P->has_xf) x += nu[T * FP_N + rc];
            if (rok) a = fma(x, x, a);
        }
        *rdlb = a;
    }
    FP_STICK(0);
    if (wv < 4 && (wv >> 1) < nb) {
        // ---- stages 0 and 1: one (stage, row block) task per wave
        const int i = wv >> 1, I = wv & 1;
        d4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + 4 * r + g;
            acc[r] = rt[i * 32 + (row < FP_N ? row : 0)];
        }
        {
            // ---- the prediction terms, and ||r_p||^2 of this stage restricted to the rows of block I
            double xv[FP_KS], xp[FP_KS];               // x0, x0_pre in B-operand layout
#pragma unroll
            for (int ks = 0; ks < FP_KS; ++ks) {
                const int k2 = 4 * ks + g;
                const double t0 = x0[k2 < FP_N ? k2 : 0];
                const double t1 = x0p ? x0p[k2 < FP_N ? k2 : 0] : 0.0;
                xv[ks] = k2 < FP_N ? t0 : 0.0; xp[ks] = k2 < FP_N ? t1 : 0.0;
            }
            const double* ximg = P->simg + (size_t)nb * FP_IMG + (size_t)I * FP_KS * 64 + lane;   // -Linv_0 A1, -Linv_0 A2, -Linv_1 A2
            const double* aim = P->aimg + (size_t)I * FP_KS * 64 + lane;
            // all four images first (one memory latency), then the products
            double m0[FP_KS], m1[FP_KS], m2[FP_KS], m3[FP_KS];
            const bool st0 = i == 0, st1 = i == 1 && i < T;
#pragma unroll
            for (int ks = 0; ks < FP_KS; ++ks) {
                m0[ks] = ximg[(st0 ? 0 : 2 * FP_IMG) + ks * 64];            // acts on x0
                m1[ks] = ximg[FP_IMG + ks * 64];                             // acts on x0_pre (stage 0)
                m2[ks] = aim[(st0 ? FP_AIMG_A1 : FP_AIMG_A2) * FP_IMG + ks * 64];   // b: A1 x0 (stage 0) | A2 x0 (stage 1)
                m3[ks] = aim[FP_AIMG_A2 * FP_IMG + ks * 64];                // b: A2 x0_pre (stage 0)
            }
            d4 bx = {0, 0, 0, 0};
            if (st0 || st1) {
#pragma unroll
                for (int ks = 0; ks < FP_KS; ++ks) acc = MFMA64(m0[ks], xv[ks], acc);
                if (st0 || var2) {
#pragma unroll
                    for (int ks = 0; ks < FP_KS; ++ks) bx = MFMA64(m2[ks], xv[ks], bx);
                }
            }
            if (st0) {
#pragma unroll
                for (int ks = 0; ks < FP_KS; ++ks) acc = MFMA64(m1[ks], xp[ks], acc);
                if (var2) {
#pragma unroll
                    for (int ks = 0; ks < FP_KS; ++ks) bx = MFMA64(m3[ks], xp[ks], bx);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * I + 4 * r + g;
                const bool rok = row < FP_N;
                const int rc = rok ? row : 0;
                const double wi = (HAS_W && i < T) ? w[i * FP_N + rc] : 0.0;
                const double rp = cp[i * 32 + rc] - (wi + bx[r]);
                if (rok) rp2 = fma(rp, rp, rp2);
            }
        }
        if (HAS_W && i < T) {
            double v[FP_KS];
#pragma unroll
            for (int e = 0; e < FP_KS; ++e) {
                const bool rok = 4 * e + g < FP_N;
                const int rc = rok ? 4 * e + g : 0;
                const double wi = w[i * FP_N + rc];
                if (rok && i >= 2 && I == 0) { const double rp = cp[i * 32 + rc] - wi; rp2 = fma(rp, rp, rp2); }
                v[e] = rok ? -wi : 0.0;
            }
            acc = fp_mm_g(P->simg + (size_t)i * FP_IMG, I, lane, v, acc);
        } else if (false) {
#pragma unroll
            for (int e = 0; e < FP_KS; ++e)
                if (4 * e + g < FP_N) { const double c = cp[i * 32 + 4 * e + g]; rp2 = fma(c, c, rp2); }
        }
        fp_store_d(Y, i, I, g, c16, acc);
    }
    FP_STICK(1);
    if (!HAS_W) {
        // ---- stages >= 2 without w: Y[i][row][problem] = rt_i[row], a broadcast fill.  One load per stage
        // (lane r holds rt_i[r]), all of a wave's stages requested before the first use: one memory latency.
        double rv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 2 + wv + k * FP_WAVES;
            rv[k] = rt[(i < nb ? i : 0) * 32 + (lane & 31)];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 2 + wv + k * FP_WAVES;
            if (i < nb) {
#pragma unroll
                for (int q = 0; q < FP_KS; ++q) {            // element q*64 + lane of the stage: row 4 q + g, problem c16
                    const double val = __shfl(rv[k], 4 * q + g, 64);
                    if (4 * q + g < FP_N) Y[(i * FP_N + 4 * q + g) * FP_NP + c16] = val;
                }
            }
        }
        for (int i = 2 + wv + 4 * FP_WAVES; i < nb; i += FP_WAVES) {     // horizons beyond 33 stages
            const double r0 = rt[i * 32 + (lane & 31)];
#pragma unroll
            for (int q = 0; q < FP_KS; ++q) {
                const double val = __shfl(r0, 4 * q + g, 64);
                if (4 * q + g < FP_N) Y[(i * FP_N + 4 * q + g) * FP_NP + c16] = val;
            }
        }
    } else {
        // ---- stages >= 2 with w: tasks (stage, row block) dealt to the waves.  Everything a task reads from memory (rt,
        // w, the image of Linv_i) is requested one task ahead: a wave has ~7 tasks and a load costs 1-2 k cycles here.
        struct Tk { double rtv[4], wv[FP_KS], im[FP_KS], cpv[FP_KS]; };
        auto tload = [&](int task, Tk& t) {
            const int tc = task < 2 * nb ? task : 2 * nb - 1;            // (past the end: a valid task, loaded and dropped)
            const int i = tc >> 1, I = tc & 1;
            const int iw = i < T ? i : T - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * I + 4 * r + g;
                t.rtv[r] = rt[i * 32 + (row < FP_N ? row : 0)];
            }
            const double* img = P->simg + (size_t)iw * FP_IMG + (size_t)(I * FP_KS) * 64 + lane;
#pragma unroll
            for (int e = 0; e < FP_KS; ++e) {
                t.wv[e] = w[iw * FP_N + (4 * e + g < FP_N ? 4 * e + g : 0)];
                t.cpv[e] = cp[i * 32 + (4 * e + g < FP_N ? 4 * e + g : 0)];
                t.im[e] = img[e * 64];
            }
        };
        auto tdo = [&](int task, const Tk& t) {
            if (task >= 2 * nb) return;
            const int i = task >> 1, I = task & 1;
            d4 acc = {t.rtv[0], t.rtv[1], t.rtv[2], t.rtv[3]};
            if (i < T) {
                double v[FP_KS];
#pragma unroll
                for (int e = 0; e < FP_KS; ++e) {
                    const bool rok = 4 * e + g < FP_N;
                    if (rok && I == 0) { const double rp = t.cpv[e] - t.wv[e]; rp2 = fma(rp, rp, rp2); }
                    v[e] = rok ? -t.wv[e] : 0.0;
                }
#pragma unroll
                for (int ks = 0; ks < FP_KS; ++ks) acc = MFMA64(t.im[ks], v[ks], acc);
            } else if (I == 0) {                             // the xf row: b_T = 0
#pragma unroll
                for (int e = 0; e < FP_KS; ++e)
                    if (4 * e + g < FP_N) rp2 = fma(t.cpv[e], t.cpv[e], rp2);
            }
            fp_store_d(Y, i, I, g, c16, acc);
        };
        Tk ta, tb;
        tload(4 + wv, ta);
        for (int task = 4 + wv; task < 2 * nb; task += 2 * FP_WAVES) {
            tload(task + FP_WAVES, tb);
            tdo(task, ta);
            tload(task + 2 * FP_WAVES, ta);
            tdo(task + FP_WAVES, tb);
        }
    }
    FP_STICK(2);
    return rp2;
}

// ------------------------------------------------------------------------------------------------
// S2 / S4: the sweeps, executed from a host-built schedule (fmpc_panel.h; copied to LDS at kernel start): per
// step every wave has at most one edge  Y[tgt][I] += IMG Y[src]  (wave 2p + I: row block I of the step's p-th
// target); one LDS-only barrier per step.
// A global load takes 1-2 k cycles here, more than a step should: nothing a step needs may have been requested
// less than TWO steps earlier.  The images are therefore prefetched two steps ahead into three rotating register
// sets, with unconditional loads (id nb is a zero image; the compiler's s_waitcnt placement gives up on
// conditional ones) in a straight-line loop body (the step count is padded to a multiple of 3 with empty
// steps), so that the sets rotate by position and are never copied.
FP_FN void fp_sweep(FpKP P, double* lds_g, int row0, int nsteps) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c16 = lane & 15;
    const int nb = P->nb;
    const FpLds L = fp_lds_layout(nb, P->mp);
    const fp_lds_t Y = (fp_lds_t)lds_g + L.Y;
    const int I = wv & 1;
    const double* limg = P->limg + (size_t)(I * 64 + lane) * 8;
    const __attribute__((address_space(3))) int* mine = (const __attribute__((address_space(3))) int*)((fp_lds_t)lds_g + L.SCH) + row0 * FP_STEP_INTS + wv * 3;
    struct Ent { int t, s, i; };
    auto entry = [&](int q) {                               // wave-uniform
        Ent e = {0, 0, -1};
        if (q < nsteps) {
            const __attribute__((address_space(3))) int* r = mine + q * FP_STEP_INTS;
            e.t = __builtin_amdgcn_readfirstlane(r[0]); e.s = __builtin_amdgcn_readfirstlane(r[1]);
            e.i = __builtin_amdgcn_readfirstlane(r[2]);
        }
        return e;
    };
    auto load_img = [&](int id, double a[FP_KS]) {
        const d2* s = (const d2*)(limg + (size_t)(id < 0 ? nb : id) * FP_IMGL);
        const d2 q0 = s[0], q1 = s[1], q2 = s[2], q3 = s[3];            // 4 x 16 bytes: the lane's 7 values
        a[0] = q0[0]; a[1] = q0[1]; a[2] = q1[0]; a[3] = q1[1]; a[4] = q2[0]; a[5] = q2[1]; a[6] = q3[0];
    };
    // executes entry e with image a; requests the image of the entry two steps later into a2
    auto step = [&](const Ent& e, const double a[FP_KS], const Ent& e2, double a2[FP_KS]) {
        if (e.i >= 0) {
            double v[FP_KS];
            fp_load_b(Y, e.s, g, c16, v);
            d4 acc = fp_load_d(Y, e.t, I, g, c16);
            __builtin_amdgcn_sched_barrier(0);
            load_img(e2.i, a2);                            // issued in the shadow of the LDS latency
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < FP_KS; ++ks) acc = MFMA64(a[ks], v[ks], acc);
            fp_store_d(Y, e.t, I, g, c16, acc);
        } else {
            load_img(e2.i, a2);
        }
        fp_barrier();
    };
    double a0[FP_KS], a1[FP_KS], a2[FP_KS];
    Ent e0 = entry(0), e1 = entry(1), e2 = entry(2);
    load_img(e0.i, a0); load_img(e1.i, a1);
    for (int q = 0; q < nsteps; q += 3) {
        step(e0, a0, e2, a2);
        e0 = entry(q + 3);
        step(e1, a1, e0, a0);
        e1 = entry(q + 4);
        step(e2, a2, e1, a1);
        e2 = entry(q + 5);
    }
}

// ------------------------------------------------------------------------------------------------
// S3: y~_i = Linv_i' y_i   (stage-parallel, in place; next stage's image prefetched)
FP_FN void fp_s3(FpKP P, double* lds_g) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c16 = lane & 15;
    const int nb = P->nb;
    const FpLds L = fp_lds_layout(nb, P->mp);
    const fp_lds_t Y = (fp_lds_t)lds_g + L.Y;
    const double* limg = P->limg + (size_t)lane * 8;              // image id = stage: Linv_i'
    auto ld = [&](int i, double a[2][FP_KS]) {
        const double* im = limg + (size_t)(i < nb ? i : nb - 1) * FP_IMGL;
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            const d2* s = (const d2*)(im + I * 64 * 8);
            const d2 q0 = s[0], q1 = s[1], q2 = s[2], q3 = s[3];
            a[I][0] = q0[0]; a[I][1] = q0[1]; a[I][2] = q1[0]; a[I][3] = q1[1]; a[I][4] = q2[0]; a[I][5] = q2[1]; a[I][6] = q3[0];
        }
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
__global__ void __launch_bounds__(FP_THREADS, 2) fmpc_cold_panel(FpParams Pv) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const FpParams Q = Pv;
    const FpKP P = &Q;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, c16 = lane & 15;
    const int nb = P->nb, batch = P->batch;
    const FpLds L = fp_lds_layout(nb, P->mp);
    if (blockIdx.x == 0 && tid == 0 && P->handed) { P->handed[0] = 0; P->handed[1] = 0; }   // counters of the exact-path launches
    // S1 writes every row of every stage; only the pad row behind the last stage (read by the k-step that holds row 26,
    // multiplied by a zero image column) has to be made finite here
    if (tid < FP_NP) lds[L.Y + nb * FP_N * FP_NP + tid] = 0.0;
    {   // both sweep schedules, packed: forward rows 0 .. nsf-1, backward rows nsf .. nsf+nsb-1
        int* sch = (int*)(lds + L.SCH);
        for (int i = tid; i < P->nsf * FP_STEP_INTS; i += FP_THREADS) sch[i] = P->sched_f[i];
        for (int i = tid; i < P->nsb * FP_STEP_INTS; i += FP_THREADS) sch[P->nsf * FP_STEP_INTS + i] = P->sched_b[i];
    }
    __syncthreads();
#ifdef FW_TIMING
    unsigned long long _k0 = __builtin_readcyclecounter(), _k1, _ka[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define FP_TICK(k) do { _k1 = __builtin_readcyclecounter(); _ka[k] += _k1 - _k0; _k0 = _k1; } while (0)
#else
#define FP_TICK(k)
#endif
    double* red = lds + L.RED;
    const bool has_w = P->w != nullptr;
    for (int panel = blockIdx.x; panel < P->npanels; panel += gridDim.x) {
        double rdl = 0.0;
        unsigned long long tk[3] = {0, 0, 0};
        double rp2 = has_w ? fp_s1<1>(P, lds, panel, &rdl, tk) : fp_s1<0>(P, lds, panel, &rdl, tk);
#ifdef FW_TIMING
        if (lane == 0) { atomicAdd(&fp_timing[8], tk[0]); atomicAdd(&fp_timing[9], tk[1]); atomicAdd(&fp_timing[10], tk[2]); }
#endif
        FP_TICK(4);
        rp2 = fp_sum_g(rp2);
        rdl = fp_sum_g(rdl);
        if (g == 0) { red[wv * FP_NP + c16] = rp2; if (wv == FP_WAVES - 1) red[FP_WAVES * FP_NP + c16] = rdl; }
        __syncthreads();
        FP_TICK(0);
        // ---- per problem: ||r_p||^2 and a lower bound of rho^2 (exact without nu0) for the step-length decision
        if (tid < FP_NP) {
            const int p = panel * FP_NP + tid;
            double s = has_w ? 0.0 : P->rp2c;                  // stages >= 2 without w: r_p,i = cp_i
            for (int w = 0; w < FP_WAVES; ++w) s += red[w * FP_NP + tid];
            const double rd = P->nu0 ? red[FP_WAVES * FP_NP + tid] : P->rd2_0;
            if (p < batch) { P->gate[2 * p] = s; P->gate[2 * p + 1] = s + rd; }
        }
        fp_sweep(P, lds, 0, P->nsf);
        FP_TICK(1);
        fp_s3(P, lds);
        fp_barrier();
        FP_TICK(2);
        fp_sweep(P, lds, P->nsf, P->nsb);
        FP_TICK(3);
        {   // ---- nu+ to HBM in PANEL layout ([stage row][16 problems], as it sits in LDS): full-rate, fully
            // coalesced stores.  The d_z kernel reads it back and produces the per-problem nu_out if asked to.
            const int nrow = nb * FP_N * FP_NP;
            double* dst = P->nuws + (size_t)panel * nrow;
            const double* src = lds + L.Y;
            for (int e = tid; e < nrow; e += FP_THREADS) dst[e] = src[e];
        }
        fp_barrier();                                          // the panel may be reused
        FP_TICK(5);
    }
#ifdef FW_TIMING
    if ((tid & 63) == 0) for (int q = 0; q < 8; ++q) atomicAdd(&fp_timing[q], _ka[q]);
#endif
}

// ---------------------------------------------------------------- host side
size_t fmpc_panel_lds_bytes(int nb, int mp) { return (size_t)fp_lds_layout(nb, mp).total * sizeof(double); }
// what a launch with nsteps (= forward + backward) schedule rows really needs: the schedule region is the last one
size_t fmpc_panel_lds_used(int nb, int mp, int nsteps) {
    return (size_t)fp_lds_layout(nb, mp).SCH * sizeof(double) + (size_t)nsteps * FP_STEP_INTS * sizeof(int);
}

hipError_t fmpc_panel_prepare(size_t lds_bytes) {
    return hipFuncSetAttribute((const void*)fmpc_cold_panel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

hipError_t fmpc_launch_panel(const FpParams& P, int grid, size_t lds_bytes, hipStream_t stream) {
    hipLaunchKernelGGL(fmpc_cold_panel, dim3(grid), dim3(FP_THREADS), lds_bytes, stream, P);
    return hipGetLastError();
}
