// 16 x 16 tile operations on the CDNA4 matrix cores shared by the tiled Newton kernel (fmpc_kernel_tiled.hip) and the
// dense factorisation of the ramp kernel (fmpc_kernel_ramp.hip): accumulator layouts of v_mfma_f64_16x16x4_f64 /
// v_mfma_f32_16x16x4_f32, X'Z products from row-major tiles, the rank-1 Cholesky of a tile in the R form with
// W = R^-T alongside, DPP row sums.  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

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
    // accumulator layout (measured, scripts/probes/mfma_f64_probe.hip): register r of lane (c, g) is row g + 4 r, column c
    static __device__ __forceinline__ int row(int g, int r) { return g + 4 * r; }
    static constexpr int kg(int k) { return k & 3; }
    static constexpr int kr(int k) { return k >> 2; }
    static __device__ __forceinline__ double readlane(double v, int l) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_readlane(lo, l); hi = __builtin_amdgcn_readlane(hi, l);
        return __hiloint2double(hi, lo);
    }
    // A tile in LDS as the matrix cores take it: operand r of lane l is element 64 r + l of the row-major tile, which is
    // also where accumulator register r of that lane belongs (row g + 4 r, column c).
    static __device__ __forceinline__ v4 ld4(const double* t, int lane) { v4 v = {t[lane], t[64 + lane], t[128 + lane], t[192 + lane]}; return v; }
    static __device__ __forceinline__ void st4(double* t, int lane, v4 v) { t[lane] = v[0]; t[64 + lane] = v[1]; t[128 + lane] = v[2]; t[192 + lane] = v[3]; }
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
    // A tile in LDS is the REGISTER IMAGE of its accumulator: the four registers of lane l (rows 4 g .. 4 g + 3 of column c)
    // at 4 l .. 4 l + 3, one 16-byte access per lane either way.  As operands of X'Z the k-slot g of product r is then row
    // 4 g + r of both tiles: a permutation of the sum over rows, the same on both sides.
    static __device__ __forceinline__ v4 ld4(const float* t, int lane) { return *(const v4*)(t + 4 * lane); }
    static __device__ __forceinline__ void st4(float* t, int lane, v4 v) { *(v4*)(t + 4 * lane) = v; }
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
template <int UNR, class XF, class ZF>
__device__ __forceinline__ void ft_vec_gemm(ft_d4& acc, int K, int g, XF xf, ZF zf) {
    // UNR k-steps at a time: all their operand loads are issued before the first product (a load costs ~1 k cycles
    // from L2, a product 64: the loop is bound by how many loads are in flight)
    int k0 = 0;
    for (; k0 + 4 * UNR <= K; k0 += 4 * UNR) {
        double xv[UNR], zv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) { xv[u] = xf(k0 + 4 * u + g); zv[u] = zf(k0 + 4 * u + g); }
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[u], zv[u], acc, 0, 0, 0);
    }
    for (; k0 < K; k0 += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xf(k0 + g), zf(k0 + g), acc, 0, 0, 0);
}

// acc -= X' Z for two 16 x 16 tiles in LDS (layout: FtT<R>::ld4 / st4)
template <typename R> __device__ __forceinline__ void ft_xtz_sub(typename FtT<R>::v4& acc, typename FtT<R>::v4 x, typename FtT<R>::v4 z) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = FtT<R>::mfma_sub(x[r], z[r], acc);
}
template <typename R> __device__ __forceinline__ void ft_xtz_sub(typename FtT<R>::v4& acc, const R* X, const R* Z, int lane) {
    ft_xtz_sub<R>(acc, FtT<R>::ld4(X, lane), FtT<R>::ld4(Z, lane));
}
// The same over a chain of cnt products into one accumulator, xf(j), zf(j) the operand tiles of product j: the operands
// of product j + 1 are requested before the matrix cores take product j (an LDS read is ~130 cycles, a product 4 x 32).
template <typename R, class XF, class ZF>
__device__ __forceinline__ void ft_xtz_chain(typename FtT<R>::v4& acc, int cnt, int lane, XF xf, ZF zf) {
    typedef FtT<R> TT;
    if (cnt <= 0) return;
    typename TT::v4 x = TT::ld4(xf(0), lane), z = TT::ld4(zf(0), lane);
    for (int j = 0; j < cnt; ++j) {
        typename TT::v4 nx = x, nz = z;
        if (j + 1 < cnt) { nx = TT::ld4(xf(j + 1), lane); nz = TT::ld4(zf(j + 1), lane); }
        ft_xtz_sub<R>(acc, x, z);
        x = nx; z = nz;
    }
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
    // the pivot of step k + 1 is formed from the still unreduced tile and row k (two readlanes and an fma), so its
    // reciprocal square root is computed while the matrix cores apply the rank-1 update of step k
    R piv = TT::readlane(acc[TT::kr(0)], 16 * TT::kg(0));
    R rinv = TT::rsqrt(piv);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k < cnt) {                                           // uniform
            const int gk = TT::kg(k), rk = TT::kr(k);          // constants after unrolling
            ok = ok && (piv > (R)0) && (piv < (R)INFINITY);
            const bool sel = g == gk;
            const R t = sel ? acc[rk] * rinv : (R)0;
            const R te = sel ? E[rk] * rinv : (R)0;
            Rout[rk] = sel ? t : Rout[rk];
            Wout[rk] = sel ? te : Wout[rk];
            if (k + 1 < 16 && k + 1 < cnt) {
                const int g1 = TT::kg(k + 1), r1 = TT::kr(k + 1);
                const R aold = TT::readlane(acc[r1], (k + 1) + 16 * g1);
                const R tk = TT::readlane(t, (k + 1) + 16 * gk);
                piv = aold - tk * tk;
                rinv = TT::rsqrt(piv);
            }
            acc = TT::mfma_sub(t, t, acc);
            E = TT::mfma_sub(t, te, E);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (TT::row(g, r) > c) Rout[r] = (R)0;                   // rounding residue below the diagonal
    return ok;
}

// The same with the number of live rows known at compile time: straight-line code (the runtime version pays a uniform
// branch and the register copies of its merge per step).  The matrix-core updates of step k are issued first, the
// pivot of step k + 1 and its reciprocal square root follow in their shadow.
template <typename R, int CNT>
__device__ __forceinline__ bool ft_potrf16_ct(const typename FtT<R>::v4& P, int c, int g,
                                              typename FtT<R>::v4& Rout, typename FtT<R>::v4& Wout) {
    typedef FtT<R> TT;
    typename TT::v4 acc = P, E;
#pragma unroll
    for (int r = 0; r < 4; ++r) { E[r] = TT::row(g, r) == c ? (R)1 : (R)0; Rout[r] = (R)0; Wout[r] = (R)0; }
    bool ok = true;
    if (CNT > 0) {
        R piv = TT::readlane(acc[TT::kr(0)], 16 * TT::kg(0));
        R rinv = TT::rsqrt(piv);
#pragma unroll
        for (int k = 0; k < CNT; ++k) {
            const int gk = TT::kg(k), rk = TT::kr(k);
            ok = ok && (piv > (R)0) && (piv < (R)INFINITY);
            const bool sel = g == gk;
            const R t = sel ? acc[rk] * rinv : (R)0;
            const R te = sel ? E[rk] * rinv : (R)0;
            Rout[rk] = sel ? t : Rout[rk];
            Wout[rk] = sel ? te : Wout[rk];
            R aold = (R)0, tk = (R)0;
            if (k + 1 < CNT) {
                aold = TT::readlane(acc[TT::kr(k + 1)], (k + 1) + 16 * TT::kg(k + 1));
                tk = TT::readlane(t, (k + 1) + 16 * gk);
            }
            acc = TT::mfma_sub(t, t, acc);
            E = TT::mfma_sub(t, te, E);
            if (k + 1 < CNT) {
                piv = aold - tk * tk;
                rinv = TT::rsqrt(piv);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (TT::row(g, r) > c) Rout[r] = (R)0;
    return ok;
}

