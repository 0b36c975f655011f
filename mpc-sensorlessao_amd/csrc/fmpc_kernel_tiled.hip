// Tiled fastMPC Newton kernel for gfx950: ONE WORKGROUP PER PROBLEM, any n <= 79, factorisation in fp64 or fp32.
//
// Runs the whole `inf_newton_solver` loop of a problem (reference: Fast_MPC/VAR_2/inf_newton_solver.m:10-41):
//   P1  residuals r_d, r_p with the barrier terms        inf_newton_solver.m:11-17, inf_newton_KKT_H.m:3-13
//   P2  rhs = r_p - C Phi^-1 r_d                          inf_newton_solver.m:28-29
//   P3  banded Cholesky of Y = C Phi^-1 C' + forward sweep  inf_newton_solver.m:27,30-31
//   P4  backward sweep -> d_nu                            inf_newton_solver.m:32
//   P5  d_z = Phi^-1(-r_d - C' d_nu), line search, update inf_newton_solver.m:34-38, backtracking_inf_newton.m:2-11
//
// P3 is a left-looking block Cholesky of the block-penta-diagonal Y in the "R" form Y = R'R (R upper triangular,
// R_ii = L_ii').  A stage i owns the block row [R_i | U1_i | U2_i] with U1_i = R_i^-T (Y_{i,i+1} - Ua' Ub),
// U2_i = R_i^-T Y_{i,i+2}, S_i = Y_ii + B W_i B' - Ua' Ua - Uc' Uc = R_i' R_i  (Ua = U1_{i-1}, Ub = U2_{i-1},
// Uc = U2_{i-2}).  Every n x n block is NB x NB tiles of 16 x 16; every product is an X'Z
// (contraction over the ROW index of both operands), so tiles stored row-major in LDS are read as MFMA operands with
// 64 consecutive elements per wave (no bank conflicts), on v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32:
//   phase A (all waves, independent tiles): P = Y_const (+ rhs in column n) + B'(W B) - Ua'Ua - Uc'Uc | Y1 - Ua'Ub | Y2
//   phase B (per 16-row block kb of the stage): P(kb,.) -= sum_{j<kb} R(j,kb)' Rwide(j,.) ; the owner of the diagonal
//           tile factors it (16 rank-1 MFMA updates on [S | I] -> R(kb,kb) and W = R(kb,kb)^-T); everybody scales its
//           tiles of the row, Rwide(kb,.) = W P(kb,.) (4 MFMAs per tile), into LDS (operands of the following rows and
//           stages) and into the factor stream in HBM (read back once by P4).
// Column n of the blocks carries rhs_i -> y_i, so the forward substitution needs no instruction of its own.
// REAL = float: Y, the factor and both sweeps in fp32; r_d, r_p, the rhs, the line search and z, nu stay fp64
// (the Newton iteration itself is the fp64 residual refinement of the fp32 KKT solves).
#include <hip/hip_runtime.h>
#include <math.h>
#include "fmpc_tiled.h"
#include "../../include/fastmpc.h"

#define FT_MAX_HALVINGS 64

// diagnostic build (make timing): per-phase cycle counters of workgroup 0, thread 0 (scripts/tiled_phases.py)
#ifdef FW_TIMING
__device__ unsigned long long ft_timing[16];
extern "C" int fmpc_debug_tiled_timing(unsigned long long* out, int reset) {
    if (reset) { unsigned long long z[16] = {0}; return hipMemcpyToSymbol(HIP_SYMBOL(ft_timing), z, sizeof(z)) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ft_timing), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#define FT_T0() unsigned long long _t0 = __builtin_readcyclecounter(), _t1
#define FT_TICK(k) do { _t1 = __builtin_readcyclecounter(); if (blockIdx.x == 0 && threadIdx.x == 0) ft_timing[k] += _t1 - _t0; _t0 = _t1; } while (0)
#else
#define FT_T0()
#define FT_TICK(k)
#endif

typedef double ft_d4 __attribute__((ext_vector_type(4)));
typedef float ft_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ft_lds_barrier() {       // orders LDS traffic only; global loads/stores stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <typename R> struct FtT;
template <> struct FtT<double> {
    typedef ft_d4 v4;
    static __device__ __forceinline__ v4 mfma(double a, double b, v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ v4 mfma_sub(double a, double b, v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 1); }   // c - a b (neg:[1,0,0])
    // accumulator layout (measured, scripts/mfma_f64_probe.hip): register r of lane (c, g) is row g + 4 r, column c
    static __device__ __forceinline__ int row(int g, int r) { return g + 4 * r; }
    static constexpr int kg(int k) { return k & 3; }
    static constexpr int kr(int k) { return k >> 2; }
    static __device__ __forceinline__ double readlane(double v, int l) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_readlane(lo, l); hi = __builtin_amdgcn_readlane(hi, l);
        return __hiloint2double(hi, lo);
    }
    static __device__ __forceinline__ double rsqrt(double d) {
        double y = __builtin_amdgcn_rsq(d);
        const double h = 0.5 * d;
        y = y * fma(-h * y, y, 1.5);
        y = y * fma(-h * y, y, 1.5);
        return y;
    }
};
template <> struct FtT<float> {
    typedef ft_f4 v4;
    static __device__ __forceinline__ v4 mfma(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ v4 mfma_sub(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(-a, b, c, 0, 0, 0); }
    // register r of lane (c, g) is row 4 g + r, column c
    static __device__ __forceinline__ int row(int g, int r) { return 4 * g + r; }
    static constexpr int kg(int k) { return k >> 2; }
    static constexpr int kr(int k) { return k & 3; }
    static __device__ __forceinline__ float readlane(float v, int l) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
    }
    static __device__ __forceinline__ float rsqrt(float d) {
        float y = __builtin_amdgcn_rsqf(d);
        y = y * fmaf(-0.5f * d * y, y, 1.5f);
        return y;
    }
};

// 1/x in fp64: hardware estimate + two Newton steps (<= 1 ulp; the parity tolerance is 1e-9)
__device__ __forceinline__ double ft_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}

// sum over the 16 lanes of a DPP row, result in every lane (row_ror 8, 4, 2, 1)
template <int CTRL> __device__ __forceinline__ double ft_dpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ float ft_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <typename R> __device__ __forceinline__ R ft_row16_sum(R v) {
    v += ft_dpp<0x128>(v); v += ft_dpp<0x124>(v); v += ft_dpp<0x122>(v); v += ft_dpp<0x121>(v);
    return v;
}

__device__ __forceinline__ double ft_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the workgroup, result to every thread; fixed order -> bitwise reproducible
template <int NW> __device__ __forceinline__ double ft_block_sum(double v, double* red) {
    v = ft_wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}

// One 16 x 16 output tile of a stage-batched product of the residual phases, always on the fp64 matrix cores:
// acc += X'Z over k = 0..K-1 (K a multiple of 4).  xf(k) is X[k][a], zf(k) is Z[k][b] for this lane's a = b = lane & 15;
// the lane group g = lane >> 4 takes k = k0 + g.  The T horizon stages are the row dimension of the output.
template <class XF, class ZF>
__device__ __forceinline__ void ft_vec_gemm(ft_d4& acc, int K, int g, XF xf, ZF zf) {
#pragma unroll 4
    for (int k0 = 0; k0 < K; k0 += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xf(k0 + g), zf(k0 + g), acc, 0, 0, 0);
}

// acc -= X' Z for two row-major 16 x 16 tiles in LDS (64 consecutive elements per operand read)
template <typename R> __device__ __forceinline__ void ft_xtz_sub(typename FtT<R>::v4& acc, const R* X, const R* Z, int lane) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = FtT<R>::mfma_sub(X[64 * r + lane], Z[64 * r + lane], acc);
}

// Cholesky of the leading cnt x cnt part of the symmetric tile P (accumulator layout) in the R form, P = R'R, by
// 16 rank-1 updates on the matrix cores: row k of the reduced tile sits in register kr(k) of lane group kg(k), which is
// the k-slot kg(k) of both MFMA operands, so t (x) t needs no data movement.  The same row operations applied to an
// identity give W = R^-T.  Columns >= cnt of P (the rhs column, padding) are right-hand sides and are transformed along.
template <typename R>
__device__ __forceinline__ bool ft_potrf16(const typename FtT<R>::v4& P, int cnt, int c, int g,
                                           typename FtT<R>::v4& Rout, typename FtT<R>::v4& Wout) {
    typedef FtT<R> TT;
    typename TT::v4 acc = P, E;
#pragma unroll
    for (int r = 0; r < 4; ++r) { E[r] = TT::row(g, r) == c ? (R)1 : (R)0; Rout[r] = (R)0; Wout[r] = (R)0; }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k < cnt) {                                           // uniform
            const int gk = TT::kg(k), rk = TT::kr(k);          // constants after unrolling
            const R piv = TT::readlane(acc[rk], k + 16 * gk);
            ok = ok && (piv > (R)0) && (piv < (R)INFINITY);
            const R rinv = TT::rsqrt(piv);
            const bool sel = g == gk;
            const R t = sel ? acc[rk] * rinv : (R)0;
            const R te = sel ? E[rk] * rinv : (R)0;
            Rout[rk] = sel ? t : Rout[rk];
            Wout[rk] = sel ? te : Wout[rk];
            acc = TT::mfma_sub(t, t, acc);
            E = TT::mfma_sub(t, te, E);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (TT::row(g, r) > c) Rout[r] = (R)0;                   // rounding residue below the diagonal
    return ok;
}

// upper-triangular tile enumeration (row-major, I <= J)
__device__ __forceinline__ int ft_lt_index(int NB, int I, int J) { return I * NB - (I * (I - 1)) / 2 + (J - I); }

template <typename R, int NB, int NW>
__global__ void __launch_bounds__(NW * 64, 2) fmpc_newton_tiled(FtParams P) {
    typedef FtT<R> TT;
    typedef typename TT::v4 v4;
    constexpr int NT = NW * 64, NP = 16 * NB;
    constexpr int NS = NB * (NB + 1) / 2, NQ = NB * NB;
    constexpr int SS = (NS + NW - 1) / NW, MS = (NQ + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FmpcDevModel& M = P.M;
    const FtModel& V = P.V;
    const int n = M.n, m = M.m, T = M.T, nb = M.nb;
    const int s = n + m, Nz = T * s, nbn = nb * n;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const bool var2 = M.var2 != 0;
    const int mb = V.mb, cn = V.cn, nl = V.nl;

    const FtLds LL = ft_lds_layout(NB, mb, NW, (int)sizeof(R), nb);
    R* sBT = (R*)(smem + LL.bt);
    R* sSLOT = (R*)(smem + LL.slot);
    R* sLT = (R*)(smem + LL.lt);
    R* sWT = (R*)(smem + LL.wt);
    R* sWL = (R*)(smem + LL.wl);
    R* sYSH = (R*)(smem + LL.ysh);
    R* sXV = (R*)(smem + LL.xv);
    R* sPART = (R*)(smem + LL.part);
    double* red = (double*)(smem + LL.red);
    int* sflag = (int*)(smem + LL.flag);
    // residual phases: nu / d_nu as [stage][state] (leading dimension NP + 1: conflict-free transposed reads) in the
    // space of the U slots, which only the factor phase uses
    double* sNU = (double*)(smem + LL.slot);
    constexpr int LDN = 16 * NB + 1;
    const int TA = (nb + 15) / 16, NUROWS = 16 * TA + 2;

    {   // B' tiles stay in LDS for the whole launch
        const R* src = (const R*)V.btimg;
        for (int i = tid; i < mb * NB * FT_TILE; i += NT) sBT[i] = src[i];
    }

    const FtWs L = ft_ws_layout(n, m, T, nb, NB, (int)sizeof(R));
    double* wsp = P.ws + (size_t)blockIdx.x * P.ws_stride;
    double* b = wsp + L.b;
    double* nu = wsp + L.nu;
    double* hess = wsp + L.hess;
    double* winv = wsp + L.winv;
    double* rdu = wsp + L.rdu;
    double* rdx = wsp + L.rdx;
    double* phx = wsp + L.phx;
    double* rp = wsp + L.rp;
    double* yv = wsp + L.y;
    double* dnu = wsp + L.dnu;
    R* fac = (R*)(wsp + L.fac);
    constexpr int STAGE_TILES = NB + 3 * NB * NB;
    const R* yimg = (const R*)V.yimg;

    // tile ownership: S tile t (upper-triangular enumeration) -> wave t % NW; M1 tile q = I NB + J -> wave (NS + q) % NW;
    // M2 tile q -> wave (NS + NQ + q) % NW.  Slot sl of this wave is the tile first + sl NW of its kind.
    const int firstS = wv, firstM1 = ((wv - NS) % NW + NW) % NW, firstM2 = ((wv - NS - NQ) % NW + 2 * NW) % NW;
    int sI[SS], sJ[SS];
#pragma unroll
    for (int sl = 0; sl < SS; ++sl) {
        int t = firstS + sl * NW, I = 0;
        if (t < NS) { while (t >= NB - I) { t -= NB - I; ++I; } sI[sl] = I; sJ[sl] = I + t; }
        else { sI[sl] = -1; sJ[sl] = -1; }
    }

    for (int p = blockIdx.x; p < P.batch; p += gridDim.x) {
        double* zp = P.zout + (size_t)p * Nz;
        const double* x0v = P.x0 + (size_t)p * n;
        const double* x0pv = P.x0p ? P.x0p + (size_t)p * n : nullptr;
        __syncthreads();
        FT_T0();
        // ================= P0: start point, nu, b  (fast_mpc_init.m:12-27, fast_mpc_eq_const.m:39,44,47,68)
        for (int idx = tid; idx < Nz; idx += NT) {
            const int e = idx % s;
            zp[idx] = P.zinit ? P.zinit[(size_t)p * Nz + idx] : (e < m ? M.umid[e] : M.xmid[e - m]);
        }
        for (int idx = tid; idx < nbn; idx += NT) {
            nu[idx] = P.nu0 ? P.nu0[(size_t)p * nbn + idx] : 0.0;
            const int i = idx / n, r = idx - i * n;
            double v = (i < T && P.w) ? P.w[(size_t)p * T * n + idx] : 0.0;
            if (i == 0) {
                for (int q = 0; q < n; ++q) v += M.A1t[q * n + r] * x0v[q];
                if (var2 && x0pv)
                    for (int q = 0; q < n; ++q) v += M.A2t[q * n + r] * x0pv[q];
            } else if (i == 1 && i < T && var2) {
                for (int q = 0; q < n; ++q) v += M.A2t[q * n + r] * x0v[q];
            }
            if (i == T) v = M.xf[r];
            b[idx] = v;
        }
        if (P.step)
            for (int idx = tid; idx < P.step_ld; idx += NT) P.step[(size_t)p * P.step_ld + idx] = -1.0;
        __syncthreads();

        int st = FMPC_OK, nsteps = 0;
        FT_TICK(0);
        for (int it = 0; it < P.max_iter; ++it) {
            // ================= P1: residuals.  Every product is a GEMM with the horizon stages as one dimension
            // (out[stage][entry] = sum_k X[k][stage] Z[k][entry]) on the fp64 matrix cores; Z (B, A1, A2 and transposes)
            // is read from L2 with 128-byte rows, the epilogues read and write along the entries of a stage.
            double acc_d = 0.0, acc_p = 0.0;
            int bad = 0;
            for (int idx = tid; idx < NUROWS * LDN; idx += NT) {       // nu as [stage][state] in LDS, zero padded
                const int j = idx / LDN, r = idx - j * LDN;
                sNU[idx] = (j < nb && r < n) ? nu[j * n + r] : 0.0;
            }
            __syncthreads();
            {
                const int nC = NB * TA, nB_ = NB * TA, nA = mb * TA;   // items: r_p tiles, r_d[x] tiles, r_d[u] tiles (heaviest first)
                for (int item = wv; item < nC + nB_ + nA; item += NW) {
                    ft_d4 acc = {0, 0, 0, 0};
                    if (item < nC) {
                        // ---- r_p,i = x_{i+1} - b_i - B u_i - A1 x_i - A2 x_{i-1}      (terminal row: x_T - xf)
                        const int Jr = item / TA, A = item - Jr * TA;
                        const int i = 16 * A + c, r = 16 * Jr + c;      // as A-operand lane: stage i; as B-operand lane: state r
                        const bool rok = r < n;
                        ft_vec_gemm(acc, 16 * mb, g,
                                    [&](int k) { return (i < T && k < m) ? zp[i * s + k] : 0.0; },
                                    [&](int k) { return (rok && k < m) ? M.Bt[k * n + r] : 0.0; });
                        ft_vec_gemm(acc, NP, g,
                                    [&](int k) { return (i >= 1 && i < T && k < n) ? zp[(i - 1) * s + m + k] : 0.0; },
                                    [&](int k) { return (rok && k < n) ? M.A1t[k * n + r] : 0.0; });
                        if (var2)
                            ft_vec_gemm(acc, NP, g,
                                        [&](int k) { return (i >= 2 && i < T && k < n) ? zp[(i - 2) * s + m + k] : 0.0; },
                                        [&](int k) { return (rok && k < n) ? M.A2t[k * n + r] : 0.0; });
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int io = 16 * A + g + 4 * rr;
                            if (io < nb && rok) {
                                const double v = (io < T ? zp[io * s + m + r] - acc[rr] : zp[(T - 1) * s + m + r]) - b[io * n + r];
                                rp[io * n + r] = v;
                                acc_p += v * v;
                            }
                        }
                    } else if (item < nC + nB_) {
                        // ---- r_d on x_j (j = jj + 1): 2Q x + q + nu_{j-1} - A1' nu_j - A2' nu_{j+1}  (+ nu_T with xf)
                        const int it2 = item - nC, Jr = it2 / TA, A = it2 - Jr * TA;
                        const int jj = 16 * A + c, r = 16 * Jr + c;
                        const bool rok = r < n;
                        ft_vec_gemm(acc, NP, g,
                                    [&](int k) { return jj + 1 < T ? sNU[(jj + 1) * LDN + k] : 0.0; },
                                    [&](int k) { return (rok && k < n) ? M.A1[k * n + r] : 0.0; });
                        if (var2)
                            ft_vec_gemm(acc, NP, g,
                                        [&](int k) { return jj + 2 < T ? sNU[(jj + 2) * LDN + k] : 0.0; },
                                        [&](int k) { return (rok && k < n) ? M.A2[k * n + r] : 0.0; });
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int jo = 16 * A + g + 4 * rr;
                            if (jo < T && rok) {
                                const bool last = jo + 1 == T;
                                const double q2 = last ? M.Qf2[r] : M.Q2[r];
                                double v = q2 * zp[jo * s + m + r] + (last ? M.qfl[r] : M.ql[r]) + sNU[jo * LDN + r] - acc[rr];
                                if (last && M.has_xf) v += sNU[T * LDN + r];
                                rdx[jo * n + r] = v;
                                phx[jo * n + r] = v * ft_rcp(q2);              // Phi^-1 r_d on x_j
                                acc_d += v * v;
                            }
                        }
                    } else {
                        // ---- r_d on u_j: 2R u + r + k P'd - B' nu_j ; barrier Hessian and Phi^-1 on the way
                        const int it3 = item - nC - nB_, J = it3 / TA, A = it3 - J * TA;
                        const int j = 16 * A + c, q = 16 * J + c;
                        const bool qok = q < m;
                        ft_vec_gemm(acc, NP, g,
                                    [&](int k) { return sNU[j * LDN + k]; },
                                    [&](int k) { return (qok && k < n) ? V.Bm[(size_t)k * m + q] : 0.0; });
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int jo = 16 * A + g + 4 * rr;
                            if (jo < T && qok) {
                                const double u = zp[jo * s + q];
                                const double dp = ft_rcp(M.umax[q] - u), dm = ft_rcp(u - M.umin[q]);
                                const double hs = P.kbar * (dp * dp + dm * dm);
                                const double rt = M.R2[q] + hs;
                                if (!(rt > 0.0) || isinf(rt)) bad = 1;
                                const double rd = M.R2[q] * u + M.rl[q] + P.kbar * (dp - dm) - acc[rr];
                                hess[jo * m + q] = hs;
                                winv[jo * m + q] = ft_rcp(rt);
                                rdu[jo * m + q] = rd;
                                acc_d += rd * rd;
                            }
                        }
                    }
                }
            }
            const double rp2 = ft_block_sum<NW>(acc_p, red);
            const double rho2 = ft_block_sum<NW>(acc_d, red) + rp2;
            const double badsum = ft_block_sum<NW>((double)bad, red);
            // early exit, tested before the step (inf_newton_solver.m:19-22)
            if (sqrt(rho2) <= 1e-6 && sqrt(rp2) <= 1e-8) break;
            if (badsum > 0.0) { st = FMPC_E_NOT_PD_PHI; break; }
            FT_TICK(1);

            // ================= P2: rhs_i = r_p,i - (C Phi^-1 r_d)_i   (into yv)
            for (int item = wv; item < NB * TA; item += NW) {
                const int Jr = item / TA, A = item - Jr * TA;
                const int i = 16 * A + c, r = 16 * Jr + c;
                const bool rok = r < n;
                ft_d4 acc = {0, 0, 0, 0};
                ft_vec_gemm(acc, 16 * mb, g,
                            [&](int k) { return (i < T && k < m) ? rdu[i * m + k] * winv[i * m + k] : 0.0; },
                            [&](int k) { return (rok && k < m) ? M.Bt[k * n + r] : 0.0; });
                ft_vec_gemm(acc, NP, g,
                            [&](int k) { return (i >= 1 && i < T && k < n) ? phx[(i - 1) * n + k] : 0.0; },
                            [&](int k) { return (rok && k < n) ? M.A1t[k * n + r] : 0.0; });
                if (var2)
                    ft_vec_gemm(acc, NP, g,
                                [&](int k) { return (i >= 2 && i < T && k < n) ? phx[(i - 2) * n + k] : 0.0; },
                                [&](int k) { return (rok && k < n) ? M.A2t[k * n + r] : 0.0; });
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int io = 16 * A + g + 4 * rr;
                    if (io < nb && rok) {
                        const double cv = io < T ? phx[io * n + r] - acc[rr] : phx[(T - 1) * n + r];
                        yv[io * n + r] = rp[io * n + r] - cv;
                    }
                }
            }
            __syncthreads();                                           // (the staging area of nu is the U slots' space)
            // zero the three U slots: stages 0 and 1 then need no special cases
            for (int i = tid; i < 3 * NQ * FT_TILE; i += NT) sSLOT[i] = (R)0;
            if (tid == 0) sflag[0] = 0;
            __syncthreads();
            FT_TICK(2);

            // ================= P3: factor + forward sweep
            int ua = 0, ub = 1, uc = 2;                               // roles of the three LDS slots
            bool fail = false;
            for (int i = 0; i < nb; ++i) {
                const bool hasB = i < T;
                R* UA = sSLOT + (size_t)ua * NQ * FT_TILE;
                R* UB = sSLOT + (size_t)ub * NQ * FT_TILE;
                R* UC = sSLOT + (size_t)uc * NQ * FT_TILE;
                R* facs = fac + (size_t)i * STAGE_TILES * FT_TILE;
                const R* YD = yimg + (size_t)V.iD[i] * NQ * FT_TILE;
                const R* Y1 = yimg + (size_t)V.i1[i] * NQ * FT_TILE;
                const R* Y2 = yimg + (size_t)V.i2[i] * NQ * FT_TILE;
                if (hasB)
                    for (int q = tid; q < mb * 16; q += NT) sWL[q] = q < m ? (R)winv[i * m + q] : (R)0;
                ft_lds_barrier();
                // ---------------- phase A: all tiles of the stage, independent
                v4 aS[SS], aM1[MS], aM2[MS];
#pragma unroll
                for (int sl = 0; sl < SS; ++sl) {
                    const int I = sI[sl], J = sJ[sl];
                    if (I >= 0) {
                        v4 a;
                        const R* yt = YD + (size_t)(I * NB + J) * FT_TILE;
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[r] = yt[TT::row(g, r) * 16 + c];
                        if (J == cn && c == nl) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * I + TT::row(g, r);
                                a[r] = row < n ? (R)yv[i * n + row] : (R)0;
                            }
                        }
                        if (hasB) {                                    // + B'(W B): contraction over the actuators
                            for (int kb = 0; kb < mb; ++kb) {
                                const R* X = sBT + (size_t)(kb * NB + I) * FT_TILE;
                                const R* Z = sBT + (size_t)(kb * NB + J) * FT_TILE;
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    a = TT::mfma(X[64 * r + lane], Z[64 * r + lane] * sWL[16 * kb + 4 * r + g], a);
                            }
                        }
                        for (int j = 0; j < NB; ++j) {
                            ft_xtz_sub<R>(a, UA + (size_t)(j * NB + I) * FT_TILE, UA + (size_t)(j * NB + J) * FT_TILE, lane);
                            ft_xtz_sub<R>(a, UC + (size_t)(j * NB + I) * FT_TILE, UC + (size_t)(j * NB + J) * FT_TILE, lane);
                        }
                        aS[sl] = a;
                    }
                }
#pragma unroll
                for (int sl = 0; sl < MS; ++sl) {
                    const int q1 = firstM1 + sl * NW;
                    if (q1 < NQ) {
                        const int I = q1 / NB, J = q1 - I * NB;
                        v4 a;
                        const R* yt = Y1 + (size_t)q1 * FT_TILE;
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[r] = yt[TT::row(g, r) * 16 + c];
                        for (int j = 0; j < NB; ++j)
                            ft_xtz_sub<R>(a, UA + (size_t)(j * NB + I) * FT_TILE, UB + (size_t)(j * NB + J) * FT_TILE, lane);
                        aM1[sl] = a;
                    }
                    const int q2 = firstM2 + sl * NW;
                    if (q2 < NQ) {
                        v4 a;
                        const R* yt = Y2 + (size_t)q2 * FT_TILE;
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[r] = yt[TT::row(g, r) * 16 + c];
                        aM2[sl] = a;
                    }
                }
                ft_lds_barrier();                                      // Ua, Uc are dead from here: their slots take U1_i, U2_i
                FT_TICK(3);
                R* U1N = UA; R* U2N = UC;
                // ---------------- phase B: the 16-row blocks of the stage, in order
                for (int kb = 0; kb < NB; ++kb) {
                    int cnt = n - 16 * kb; cnt = cnt > 16 ? 16 : (cnt < 0 ? 0 : cnt);
                    // (1) products with the rows of this stage already done
#pragma unroll
                    for (int sl = 0; sl < SS; ++sl)
                        if (sI[sl] == kb)
                            for (int j = 0; j < kb; ++j)
                                ft_xtz_sub<R>(aS[sl], sLT + (size_t)ft_lt_index(NB, j, kb) * FT_TILE,
                                              sLT + (size_t)ft_lt_index(NB, j, sJ[sl]) * FT_TILE, lane);
#pragma unroll
                    for (int sl = 0; sl < MS; ++sl) {
                        const int q1 = firstM1 + sl * NW, q2 = firstM2 + sl * NW;
                        if (q1 < NQ && q1 / NB == kb)
                            for (int j = 0; j < kb; ++j)
                                ft_xtz_sub<R>(aM1[sl], sLT + (size_t)ft_lt_index(NB, j, kb) * FT_TILE,
                                              U1N + (size_t)(j * NB + q1 % NB) * FT_TILE, lane);
                        if (q2 < NQ && q2 / NB == kb)
                            for (int j = 0; j < kb; ++j)
                                ft_xtz_sub<R>(aM2[sl], sLT + (size_t)ft_lt_index(NB, j, kb) * FT_TILE,
                                              U2N + (size_t)(j * NB + q2 % NB) * FT_TILE, lane);
                    }
                    // (2) the rhs column of this row block, still unscaled: shared with the owners of M1(kb,cn), M2(kb,cn);
                    //     the owner of the diagonal tile factors it
#pragma unroll
                    for (int sl = 0; sl < SS; ++sl) {
                        if (sI[sl] == kb && sJ[sl] == cn && c == nl) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) sYSH[TT::row(g, r)] = aS[sl][r];
                        }
                        if (sI[sl] == kb && sJ[sl] == kb) {
                            v4 Ro, Wo;
                            const bool ok = ft_potrf16<R>(aS[sl], cnt, c, g, Ro, Wo);
                            if (!ok && lane == 0) sflag[0] = 1;
                            R* ri = facs + (size_t)kb * FT_TILE;       // R(kb,kb)^-1 = W' for the backward sweep
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                sWT[c * FT_WLD + TT::row(g, r)] = Wo[r];
                                ri[c * 16 + TT::row(g, r)] = Wo[r];
                            }
                            if (kb == cn && c == nl) {                 // y of this row block is the rhs column of the factored tile
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int row = 16 * kb + TT::row(g, r);
                                    if (row < n) yv[i * n + row] = (double)Ro[r];
                                }
                            }
                            aS[sl] = Ro;
                        }
                    }
                    ft_lds_barrier();
                    FT_TICK(4);
                    // (3) scale the tiles of the row: Rwide(kb, .) = W P(kb, .)
                    R wop[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) wop[r] = sWT[TT::row(g, r) * FT_WLD + c];
#pragma unroll
                    for (int sl = 0; sl < SS; ++sl) {
                        if (sI[sl] == kb && sJ[sl] > kb) {
                            v4 o = {0, 0, 0, 0};
#pragma unroll
                            for (int r = 0; r < 4; ++r) o = TT::mfma(wop[r], aS[sl][r], o);
                            R* dl = sLT + (size_t)ft_lt_index(NB, kb, sJ[sl]) * FT_TILE;
                            R* dg = facs + (size_t)(NB + kb * NB + sJ[sl]) * FT_TILE;
#pragma unroll
                            for (int r = 0; r < 4; ++r) { dl[TT::row(g, r) * 16 + c] = o[r]; dg[TT::row(g, r) * 16 + c] = o[r]; }
                            if (sJ[sl] == cn && c == nl) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int row = 16 * kb + TT::row(g, r);
                                    if (row < n) yv[i * n + row] = (double)o[r];
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int sl = 0; sl < MS; ++sl) {
                        const int q1 = firstM1 + sl * NW, q2 = firstM2 + sl * NW;
                        if (q1 < NQ && q1 / NB == kb) {
                            const int J = q1 % NB;
                            v4 pv = aM1[sl];
                            if (J == cn && c == nl) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) pv[r] = sYSH[TT::row(g, r)];
                            }
                            v4 o = {0, 0, 0, 0};
#pragma unroll
                            for (int r = 0; r < 4; ++r) o = TT::mfma(wop[r], pv[r], o);
                            R* dl = U1N + (size_t)q1 * FT_TILE;
                            R* dg = facs + (size_t)(NB + NQ + q1) * FT_TILE;
#pragma unroll
                            for (int r = 0; r < 4; ++r) { dl[TT::row(g, r) * 16 + c] = o[r]; dg[TT::row(g, r) * 16 + c] = o[r]; }
                        }
                        if (q2 < NQ && q2 / NB == kb) {
                            const int J = q2 % NB;
                            v4 pv = aM2[sl];
                            if (J == cn && c == nl) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) pv[r] = sYSH[TT::row(g, r)];
                            }
                            v4 o = {0, 0, 0, 0};
#pragma unroll
                            for (int r = 0; r < 4; ++r) o = TT::mfma(wop[r], pv[r], o);
                            R* dl = U2N + (size_t)q2 * FT_TILE;
                            R* dg = facs + (size_t)(NB + 2 * NQ + q2) * FT_TILE;
#pragma unroll
                            for (int r = 0; r < 4; ++r) { dl[TT::row(g, r) * 16 + c] = o[r]; dg[TT::row(g, r) * 16 + c] = o[r]; }
                        }
                    }
                    ft_lds_barrier();
                    FT_TICK(5);
                    if (sflag[0]) { fail = true; break; }              // uniform: read after the barrier
                }
                if (fail) break;
                // next stage: Ua <- U1_i (slot ua), Ub <- U2_i (slot uc), Uc <- old Ub (slot ub)
                const int t = ub; ub = uc; uc = t;
            }
            if (fail) { st = FMPC_E_NOT_PD_SCHUR; break; }
            __syncthreads();                                           // the factor stream and y are in HBM (same workgroup reads them)
            FT_TICK(6);

            // ================= P4: backward sweep, d_nu_i = R_i^-1 (y_i - U1_i d_nu_{i+1} - U2_i d_nu_{i+2})
            {
                constexpr int NG = NT / 256;                           // tile groups: 256 threads cover one tile
                const int tg = tid >> 8, ta = (tid >> 4) & 15, tb = tid & 15;
                for (int q = tid; q < 3 * NP; q += NT) sXV[q] = (R)0;
                int xc = 0, x1 = 1, x2 = 2;                            // roles of the three x vectors
                __syncthreads();
                for (int i = nb - 1; i >= 0; --i) {
                    const R* facs = fac + (size_t)i * STAGE_TILES * FT_TILE;
                    R* XC = sXV + xc * NP; const R* X1 = sXV + x1 * NP; const R* X2 = sXV + x2 * NP;
                    for (int kb = NB - 1; kb >= 0; --kb) {
                        const int nt = (NB - 1 - kb) + 2 * NB;         // tiles of this block row: R(kb, c > kb), U1(kb, .), U2(kb, .)
                        R accv = (R)0;
                        for (int t = tg; t < nt; t += NG) {
                            const R* tile; const R* xvv;
                            if (t < NB - 1 - kb) { const int cc = kb + 1 + t; tile = facs + (size_t)(NB + kb * NB + cc) * FT_TILE; xvv = XC + 16 * cc; }
                            else if (t < NB - 1 - kb + NB) { const int cc = t - (NB - 1 - kb); tile = facs + (size_t)(NB + NQ + kb * NB + cc) * FT_TILE; xvv = X1 + 16 * cc; }
                            else { const int cc = t - (NB - 1 - kb) - NB; tile = facs + (size_t)(NB + 2 * NQ + kb * NB + cc) * FT_TILE; xvv = X2 + 16 * cc; }
                            accv += tile[ta * 16 + tb] * xvv[tb];
                        }
                        accv = ft_row16_sum<R>(accv);
                        const R riv = tg == 0 ? facs[(size_t)kb * FT_TILE + ta * 16 + tb] : (R)0;
                        const int yrow = 16 * kb + tb;
                        const R yb = (tg == 0 && yrow < n) ? (R)yv[i * n + yrow] : (R)0;
                        if (tb == 0) sPART[tg * 16 + ta] = accv;
                        ft_lds_barrier();
                        if (tg == 0) {
                            R sb = yb;
#pragma unroll
                            for (int q = 0; q < NG; ++q) sb -= sPART[q * 16 + tb];
                            R xv = ft_row16_sum<R>(riv * sb);
                            if (tb == 0) {
                                XC[16 * kb + ta] = xv;
                                const int row = 16 * kb + ta;
                                if (row < n) dnu[i * n + row] = (double)xv;
                            }
                        }
                        ft_lds_barrier();
                    }
                    const int t = x2; x2 = x1; x1 = xc; xc = t;
                    for (int q = tid; q < NP; q += NT) sXV[xc * NP + q] = (R)0;
                }
            }
            __syncthreads();
            FT_TICK(7);

            // ================= P5: d_z, line-search scalars, update (the same stage-batched GEMMs with d_nu)
            double be = 0.0, e2 = 0.0;
            for (int idx = tid; idx < NUROWS * LDN; idx += NT) {
                const int j = idx / LDN, r = idx - j * LDN;
                sNU[idx] = (j < nb && r < n) ? dnu[j * n + r] : 0.0;
            }
            __syncthreads();
            for (int item = wv; item < NB * TA + mb * TA; item += NW) {
                ft_d4 acc = {0, 0, 0, 0};
                if (item < NB * TA) {
                    // ---- d_x_j = (2Q_j)^-1 (-r_d[x_j] - d_nu_{j-1} + A1' d_nu_j + A2' d_nu_{j+1}  [- d_nu_T])
                    const int Jr = item / TA, A = item - Jr * TA;
                    const int jj = 16 * A + c, r = 16 * Jr + c;
                    const bool rok = r < n;
                    ft_vec_gemm(acc, NP, g,
                                [&](int k) { return jj + 1 < T ? sNU[(jj + 1) * LDN + k] : 0.0; },
                                [&](int k) { return (rok && k < n) ? M.A1[k * n + r] : 0.0; });
                    if (var2)
                        ft_vec_gemm(acc, NP, g,
                                    [&](int k) { return jj + 2 < T ? sNU[(jj + 2) * LDN + k] : 0.0; },
                                    [&](int k) { return (rok && k < n) ? M.A2[k * n + r] : 0.0; });
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int jo = 16 * A + g + 4 * rr;
                        if (jo < T && rok) {
                            const bool last = jo + 1 == T;
                            double v = -rdx[jo * n + r] - sNU[jo * LDN + r] + acc[rr];
                            if (last && M.has_xf) v -= sNU[T * LDN + r];
                            rdx[jo * n + r] = v * ft_rcp(last ? M.Qf2[r] : M.Q2[r]);   // reuse as d_x
                        }
                    }
                } else {
                    // ---- d_u_j = Rt_j^-1 (B' d_nu_j - r_d[u_j]) ; e = k P'DP d_z for the line search
                    const int it3 = item - NB * TA, J = it3 / TA, A = it3 - J * TA;
                    const int j = 16 * A + c, q = 16 * J + c;
                    const bool qok = q < m;
                    ft_vec_gemm(acc, NP, g,
                                [&](int k) { return sNU[j * LDN + k]; },
                                [&](int k) { return (qok && k < n) ? V.Bm[(size_t)k * m + q] : 0.0; });
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int jo = 16 * A + g + 4 * rr;
                        if (jo < T && qok) {
                            const int idx = jo * m + q;
                            const double rd = rdu[idx];
                            const double du = (acc[rr] - rd) * winv[idx];
                            const double e = hess[idx] * du;
                            be += rd * e;
                            e2 += e * e;
                            rdu[idx] = du;                              // reuse as d_u
                        }
                    }
                }
            }
            const double beta_e = ft_block_sum<NW>(be, red);
            const double eps2 = ft_block_sum<NW>(e2, red);
            // closed form of backtracking_inf_newton.m:2-11 with the frozen barrier gradient:
            // ||r(t)||^2 - ((1-al t) rho)^2 = t * gq(t)
            double t = 1.0;
            {
                const double al = 1e-4;
                int halv = 0;
                while (true) {
                    const double gq = (t - 2.0 + 2.0 * al - al * al * t) * rho2 - 2.0 * (1.0 - t) * beta_e + t * eps2;
                    if (gq <= 0.0) break;
                    t *= 0.5;
                    if (++halv >= FT_MAX_HALVINGS) { t = 0.0; st = FMPC_W_LINESEARCH; break; }
                }
            }
            for (int idx = tid; idx < Nz; idx += NT) {
                const int j = idx / s, e = idx - j * s;
                zp[idx] += t * (e < m ? rdu[j * m + e] : rdx[j * n + e - m]);
            }
            for (int idx = tid; idx < nbn; idx += NT) nu[idx] += t * dnu[idx];
            if (P.step && tid == 0 && it < P.step_ld) P.step[(size_t)p * P.step_ld + it] = t;
            ++nsteps;
            __syncthreads();
            FT_TICK(8);
        }
        __syncthreads();
        if (P.nuout)
            for (int idx = tid; idx < nbn; idx += NT) P.nuout[(size_t)p * nbn + idx] = nu[idx];
        if (P.u0out)
            for (int idx = tid; idx < m; idx += NT) P.u0out[(size_t)p * m + idx] = zp[idx];
        if (tid == 0) {
            if (P.status) P.status[p] = st;
            if (P.iters) P.iters[p] = nsteps;
        }
    }
}

// ---------------------------------------------------------------- host side
template <typename R, int NB, int NW>
static hipError_t ft_launch(const FtParams& P, int grid, size_t lds, hipStream_t stream) {
    hipLaunchKernelGGL((fmpc_newton_tiled<R, NB, NW>), dim3(grid), dim3(NW * 64), lds, stream, P);
    return hipGetLastError();
}
template <typename R, int NB, int NW>
static hipError_t ft_prepare(size_t lds) {
    return hipFuncSetAttribute((const void*)fmpc_newton_tiled<R, NB, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

// instantiations: fp64 for n <= 47 (NB <= 3), fp32 for n <= 79 (NB <= 5)
#define FT_DISPATCH(fn, ...)                                                                   \
    if (!is_float) {                                                                           \
        if (NB == 1 && NW == 4) return fn<double, 1, 4>(__VA_ARGS__);                          \
        if (NB == 2 && NW == 4) return fn<double, 2, 4>(__VA_ARGS__);                          \
        if (NB == 3 && NW == 4) return fn<double, 3, 4>(__VA_ARGS__);                          \
    } else {                                                                                   \
        if (NB == 1 && NW == 4) return fn<float, 1, 4>(__VA_ARGS__);                           \
        if (NB == 2 && NW == 4) return fn<float, 2, 4>(__VA_ARGS__);                           \
        if (NB == 3 && NW == 4) return fn<float, 3, 4>(__VA_ARGS__);                           \
        if (NB == 4 && NW == 8) return fn<float, 4, 8>(__VA_ARGS__);                           \
        if (NB == 5 && NW == 8) return fn<float, 5, 8>(__VA_ARGS__);                           \
    }                                                                                          \
    return hipErrorInvalidValue;

bool fmpc_tiled_supports(int n, int m, int nb, int is_float, int* NB_out, int* NW_out) {
    const int NB = n / 16 + 1;                                     // 16 NB >= n + 1
    if (NB > (is_float ? 5 : 3)) return false;
    const int NW = NB >= 4 ? 8 : 4;
    const int mb = (m + 15) / 16;
    if (ft_lds_layout(NB, mb, NW, is_float ? 4 : 8, nb).total > 160 * 1024) return false;
    if (NB_out) *NB_out = NB;
    if (NW_out) *NW_out = NW;
    return true;
}
size_t fmpc_tiled_lds_bytes(int NB, int mb, int NW, int is_float, int nb) { return ft_lds_layout(NB, mb, NW, is_float ? 4 : 8, nb).total; }
hipError_t fmpc_tiled_prepare(int NB, int NW, int is_float, size_t lds_bytes) { FT_DISPATCH(ft_prepare, lds_bytes) }
hipError_t fmpc_launch_tiled(const FtParams& P, int NB, int NW, int is_float, int grid, size_t lds_bytes, hipStream_t stream) {
    FT_DISPATCH(ft_launch, P, grid, lds_bytes, stream)
}
