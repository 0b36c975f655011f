// CDNA4 fastMPC, closed-loop step of MANY realisations with first moves only (n = 27): the first-move form as a product.
//
// fmpc_kernel_first.hip evaluates  u0 = u0c + K0 d,  ||e||^2 = d'E d + 2 e'd + e0,  ||r_p||^2 = d'Ep d - 2 ep'd + ep0  for up to 64
// realisations with one workgroup each (its rows of K0, E, Ep from L2: 340 KB per workgroup).  For more realisations the same
// three maps are products over the batch on the fp64 matrix cores: two workgroups of 8 wavefronts per 16 realisations,
//     U0 (16 x 144) = D (16 x 112) [K0 | u0c | 0]'      9 tiles of 28 k-steps                          (workgroup y = 0)
//     Q  (112 x 16) = E^ D' ,  Qp = Ep^ D'               then d_r Q_r summed per realisation, and the decision (workgroup y = 1)
// where E^ is E with its linear term in the column of the constant and only the block-upper triangle kept (the rest doubled):
// 28 products per wavefront instead of 49.
// with d = [x0 ; x0_pre ; B u1 ; B u2 ; 1 ; 0 0 0] read from what the loop-input kernel has just written (x0, x0_pre and the
// 2 n numbers w depends on).  The decision is fmpc_kernel_first.hip's (bounds of the forms with their rounding guard); a
// realisation that is not clear-cut is flagged in `need` and redone by the exact path (fmpc_newton_wave, flag mode).
// A closed-loop step of 512 realisations is then three launches -- loop inputs, this, flag mode -- instead of four with the
// dual solve and d_z over T stages (42 us).
#include <hip/hip_runtime.h>
#include <math.h>
#include "fmpc_device.h"
#include "fmpc_loopu0.h"
#include "../../include/fastmpc.h"

typedef double d4l __attribute__((ext_vector_type(4)));
#define FL_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define FL_THREADS 512

#ifdef FW_TIMING
// diagnostic build: per workgroup (thread 0) stamps of the constant 100 MHz clock
__device__ unsigned long long fl_trace[256 * 8];
extern "C" int fmpc_debug_loopu0_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fl_trace), sizeof(unsigned long long) * 256 * 8) == hipSuccess ? 0 : -1;
}
#define FL_TICK(k) do { if (tid == 0 && blockIdx.x < 128) fl_trace[(blockIdx.y * 128 + blockIdx.x) * 8 + (k)] = (unsigned long long)wall_clock64(); } while (0)
#else
#define FL_TICK(k) do { } while (0)
#endif

// The forms' share of wavefront J (0..3) of a form: row tiles J and 7 - J (J = 0: tile 0 alone) of the block-upper-triangular
// image (off-diagonal 16-blocks doubled, the lower ones dropped: E is symmetric), k-steps 4 t .. 27 of row tile t -- 28 products
// for every J.  Operand order (image, data): register r of lane (g, c) is row 16 t + 4 r + g of E d for realisation c, and the
// matching entry of d is the lane's own operand register D[4 t + r].  The linear term rides in column 4 n (d = 1 there).
template <int J>
__device__ __forceinline__ void fl_forms_share(const double (&A)[FL_KS], const double (&D)[FL_KS], int g, int nc, double& qsum, double& dsum) {
    constexpr int TA = J, TB = 7 - J, NA = FL_KS - 4 * TA, NB = J == 0 ? 0 : FL_KS - 4 * TB;      // NA + NB = 28 for every J
    d4l acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < NA; ++q) acc0 = FL_MFMA(A[q], D[4 * TA + q], acc0);
#pragma unroll
    for (int q = 0; q < NB; ++q) acc1 = FL_MFMA(A[NA + q], D[4 * TB + q], acc1);
    double qs = 0.0, ds = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double da = D[4 * TA + r];
        qs = fma(da, acc0[r], qs);
        ds = fma(da, da, ds);
        if (J > 0) {
            const double db = D[4 * TB + r];
            qs = fma(db, acc1[r], qs);
            ds = (16 * TB + 4 * r + g == nc) ? ds : fma(db, db, ds);          // (the constant 1 of column 4 n is not part of |d|^2)
        }
    }
    qsum = qs; dsum = ds;
}

__global__ void __launch_bounds__(FL_THREADS, 1) fmpc_loop_u0(FlParams P) {
    __shared__ double sD[4 * FL_KS * 17];                  // d of the 16 realisations, entry-major with a pad: sD[k * 17 + realisation]
    __shared__ double sF[8][2][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, c = lane & 15;
    const int n = P.n, m = P.m, nc = 4 * n;
    const int p0 = blockIdx.x * 16;
    const bool forms = blockIdx.y == 1;
    FL_TICK(0);
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && P.handed) *P.handed = 0;
    // ---- the images of this wavefront's products: requested before anything else (they do not depend on the data)
    const int mt = (m + 15) / 16;
    const int tA = wv, tB = wv + 8;                                        // (m <= 256: at most 16 row tiles)
    const bool hasB = tB < mt;
    double A0[FL_KS], A1[FL_KS];
    if (!forms) {
#pragma unroll
        for (int q = 0; q < FL_KS; ++q) A0[q] = P.imgU[((size_t)tA * FL_KS + q) * 64 + lane];
        if (hasB) {
#pragma unroll
            for (int q = 0; q < FL_KS; ++q) A1[q] = P.imgU[((size_t)tB * FL_KS + q) * 64 + lane];
        }
    } else {
        // forms: wavefront j = wv % 4 of a form has row tiles j (k-steps 4 j .. 27) and 7 - j (k-steps 28 - 4 j .. 27): 28 images
        const double* img = wv < 4 ? P.imgE : P.imgEp;
        const int j = wv & 3, na = FL_KS - 4 * j;
#pragma unroll
        for (int q = 0; q < FL_KS; ++q) {
            const int tl = q < na ? j : 7 - j, ks = q < na ? 4 * j + q : 4 * (7 - j) + (q - na);
            A0[q] = img[((size_t)tl * FL_KS + ks) * 64 + lane];
        }
    }
    // lower bound of ||r_d(nu0)||^2 (x entries of the last stage): 32 threads per realisation, requested before the products
    double xr = 0.0;
    if (forms && P.nu0 && (tid & 31) < n && p0 + (tid >> 5) < P.batch) {
        const int r = tid & 31;
        const double* nu = P.nu0 + (size_t)(p0 + (tid >> 5)) * P.nb * n;
        xr = P.dx0T[r] + nu[(P.T - 1) * n + r] + (P.has_xf ? nu[P.T * n + r] : 0.0);
    }
    // ---- d = [x0 ; x0_pre ; B u1 ; B u2 ; 1 ; 0 0 0] of the 16 realisations into LDS (consecutive threads, consecutive entries)
    for (int e = tid; e < 16 * 4 * FL_KS; e += FL_THREADS) {
        const int pl = e / (4 * FL_KS), k = e - pl * 4 * FL_KS;
        const int pp = p0 + pl < P.batch ? p0 + pl : P.batch - 1;
        const double* src = k < n ? P.x0 + (size_t)pp * n + k
                          : (k < 2 * n ? P.x0_pre + (size_t)pp * n + (k - n) : P.v + (size_t)pp * 2 * n + (k < nc ? k - 2 * n : 0));
        const double v = *src;
        sD[k * 17 + pl] = k < nc ? ((k >= n && k < 2 * n && !P.var2) ? 0.0 : v) : (k == nc ? 1.0 : 0.0);     // (VAR(1): the x0_pre block is zero)
    }
    __syncthreads();
    double D[FL_KS];                                        // lane (g, c): d[p0 + c][4 q + g]
#pragma unroll
    for (int q = 0; q < FL_KS; ++q) D[q] = sD[(4 * q + g) * 17 + c];
#ifdef FW_TIMING
    asm volatile("s_nop 0" :: "v"(D[0]), "v"(D[27]));
    FL_TICK(1);
#endif
    if (!forms) {
        // ---- first moves: operand order (data, image) -- register r of lane (g, c) is row 16 t + c of realisation 4 r + g,
        // 16 lanes store 128 contiguous bytes
        d4l acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < FL_KS; ++q) acc0 = FL_MFMA(D[q], A0[q], acc0);
        if (hasB) {
#pragma unroll
            for (int q = 0; q < FL_KS; ++q) acc1 = FL_MFMA(D[q], A1[q], acc1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pp = p0 + 4 * r + g;
            if (16 * tA + c < m && pp < P.batch) P.u0out[(size_t)pp * m + 16 * tA + c] = acc0[r];
            if (hasB && 16 * tB + c < m && pp < P.batch) P.u0out[(size_t)pp * m + 16 * tB + c] = acc1[r];
        }
        FL_TICK(3);
        return;
    }
    // ---- the two forms: wavefronts 0-3 the e-form (and |d|^2), 4-7 the p-form; 28 products each
    double qs, ds;
    {
        switch (wv & 3) {
            case 0: fl_forms_share<0>(A0, D, g, nc, qs, ds); break;
            case 1: fl_forms_share<1>(A0, D, g, nc, qs, ds); break;
            case 2: fl_forms_share<2>(A0, D, g, nc, qs, ds); break;
            default: fl_forms_share<3>(A0, D, g, nc, qs, ds); break;
        }
    }
    qs += __shfl_xor(qs, 16, 64); qs += __shfl_xor(qs, 32, 64);
    ds += __shfl_xor(ds, 16, 64); ds += __shfl_xor(ds, 32, 64);
    if (g == 0) { sF[wv][0][c] = qs; sF[wv][1][c] = ds; }
    FL_TICK(3);
    double rd = xr * xr;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) rd += __shfl_xor(rd, o, 64);
    __syncthreads();
    FL_TICK(4);
    if ((tid & 31) == 0 && p0 + (tid >> 5) < P.batch) {
        const int i = tid >> 5, pp = p0 + i;
        double se = 0.0, sp = 0.0, sn = 0.0;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) { se += sF[w4][0][i]; sn += sF[w4][1][i]; sp += sF[4 + w4][0][i]; }      // fixed order
        const double rdl = P.nu0 ? rd : P.rd2_0;
        const double dn = sqrt(sn);
        double e2 = se + P.e0, rp2 = sp + P.ep0;
        const double ce = 4096.0 * 2.220446049250313e-16;                     // rounding of the forms (fmpc_kernel_first.hip)
        const double de = ce * (sn * P.normE + 2.0 * P.norme * dn + fabs(P.e0));
        const double dp = ce * (sn * P.normEp + 2.0 * P.normep * dn + fabs(P.ep0));
        e2 += de;
        rp2 = rp2 - dp > 0.0 ? rp2 - dp : 0.0;
        const double rho2 = rp2 + rdl;
        const bool fin = rp2 < 1e300 && rho2 < 1e300 && e2 < 1e300 && e2 >= 0.0;
        const bool clear = fin && (rp2 > 4e-16 || rho2 > 4e-12) && e2 <= 0.5 * rho2;
        P.need[pp] = clear ? 0 : 1;
        if (clear) {
            if (P.status) P.status[pp] = FMPC_OK;
            if (P.iters) P.iters[pp] = 1;
            if (P.step) for (int q = 0; q < P.step_ld; ++q) P.step[(size_t)pp * P.step_ld + q] = q == 0 ? 1.0 : -1.0;
        }
    }
    FL_TICK(5);
}

hipError_t fmpc_launch_loop_u0(const FlParams& P, hipStream_t stream) {
    if (P.n != 27 || 4 * P.n + 1 > 4 * FL_KS || P.m > 16 * 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fmpc_loop_u0, dim3((P.batch + 15) / 16, 2), dim3(FL_THREADS), 0, stream, P);
    return hipGetLastError();
}
