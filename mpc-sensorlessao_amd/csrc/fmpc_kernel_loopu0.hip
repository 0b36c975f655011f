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
__device__ unsigned long long fl_trace[512 * 8];
extern "C" int fmpc_debug_loopu0_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fl_trace), sizeof(unsigned long long) * 512 * 8) == hipSuccess ? 0 : -1;
}
#define FL_TICK(k) do { if (tid == 0 && blockIdx.x < 128) fl_trace[(blockIdx.y * 128 + blockIdx.x) * 8 + (k)] = (unsigned long long)wall_clock64(); } while (0)
#define FS_TICK(k) do { if (tid == 0 && blockIdx.x < 128 && (blockIdx.y == 0 || (int)blockIdx.y >= rs)) fl_trace[((blockIdx.y == 0 ? 0 : blockIdx.y - rs + 1) * 128 + blockIdx.x) * 8 + (k)] = (unsigned long long)wall_clock64(); } while (0)
#else
#define FL_TICK(k) do { } while (0)
#define FS_TICK(k) do { } while (0)
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

// the step-length / exit decision of one realisation from the two forms and |d|^2 (fmpc_kernel_first.hip's rule, with the
// forms' own rounding guard); a realisation that is not clear-cut is flagged for the exact path
__device__ __forceinline__ void fl_decide(const FlParams& P, int pp, double se, double sp, double sn, double rdl) {
    const double dn = sqrt(sn);
    double e2 = se + P.e0, rp2 = sp + P.ep0;
    const double ce = 4096.0 * 2.220446049250313e-16;
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
        fl_decide(P, pp, se, sp, sn, rdl);
    }
    FL_TICK(5);
}

hipError_t fmpc_launch_loop_u0(const FlParams& P, hipStream_t stream) {
    if (P.n != 27 || 4 * P.n + 1 > 4 * FL_KS || P.m > 16 * 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fmpc_loop_u0, dim3((P.batch + 15) / 16, 2), dim3(FL_THREADS), 0, stream, P);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------
// The whole closed-loop step of many realisations with first moves only as ONE launch (n = 27, m <= 144): loop inputs, first
// moves, forms and decision.  256 threads; blockIdx.x = 16 realisations, blockIdx.y = role:
//     [0, rs)   a slice of <= 64 rows of w = -M1 (B u1) - M2 (B u2)   (y = 0 also writes x0 = a + B u1, x0_pre = x0_last)
//     rs, rs+1  the first moves: row tiles 0-4 / 5-8 of [K0 | u0c]
//     rs+2      the two forms and the decision
// Every workgroup computes B u1, B u2 of its 16 realisations itself, from operand images of B and u through LDS (coalesced
// loads), the four wavefronts splitting the actuators and adding their partial tiles through LDS.  The result
// tile of B u IS the data operand of everything that follows (register r of lane (lk, li) = entry 16 I + 4 r + lk of
// realisation li = k-step 4 I + r), so d = [x0 ; x0_pre ; B u1 ; B u2] is laid out in four blocks of 7 k-steps (27 entries + a
// zero), the constant 1 in the last pad slot (column 111), and the images of K0, E, Ep are permuted to that order on the host.
#define FS_THREADS 256
template <int J>
__device__ __forceinline__ void fs_forms_pair(const double (&AE)[FL_KS], const double (&AP)[FL_KS], const double (&D)[FL_KS], int g,
                                              double& qe, double& qp, double& dsum) {
    double de;
    fl_forms_share<J>(AE, D, g, 4 * FL_KS - 1, qe, dsum);
    fl_forms_share<J>(AP, D, g, 4 * FL_KS - 1, qp, de);
}

__global__ void __launch_bounds__(FS_THREADS, 2) fmpc_loop_step27(FlParams P, FlStepIn I) {
    constexpr int n = 27;
    // u1, u2 of the 16 realisations (rows padded to 145: the 16 realisations of a k-step read different banks); afterwards the
    // partial tiles of B u: [wavefront][tile][register][lane]
    __shared__ double su[2 * 16 * 145 > 4 * 4 * 4 * 64 ? 2 * 16 * 145 : 4 * 4 * 4 * 64];
    __shared__ double sF[4][3][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int m = P.m, p0 = blockIdx.x * 16, np = P.batch - p0 < 16 ? P.batch - p0 : 16;
    const int Tn = P.T * n, rs = I.rs;
    const int role = (int)blockIdx.y < rs ? 0 : ((int)blockIdx.y < rs + 2 ? 1 : 2);
    const int e0 = blockIdx.y * I.rows, e1 = e0 + I.rows < Tn ? e0 + I.rows : Tn;
    const int pl = p0 + li < P.batch ? p0 + li : P.batch - 1;                    // (a realisation beyond the batch repeats the last one; never stored)
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && P.handed) *P.handed = 0;
    FS_TICK(0);
    // ---- B u1, B u2: this wavefront's quarter of the actuators (<= 9 k-steps), images of B and gathered u as operands
    d4l bu[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
    double b0[9], b1[9], v1[9], v2[9];
    unsigned okm = 0;                                   // bit i: column 4 ks + lk of k-step i is an actuator of a realisation of the batch; 16 + i: k-step in range
    {
        const int ksteps = (m + 3) / 4, per = (ksteps + 3) / 4;                          // per <= 9 (checked by the launcher)
        const int k0 = wv * per;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int ks = k0 + i, c = 4 * ks + lk;
            const bool on = i < per && ks < ksteps;
            const bool cok = on && c < m && li < np;
            const size_t ob = (size_t)(on ? ks : 0) * 64 + lane;
            b0[i] = I.imgB[ob]; b1[i] = I.imgB[(size_t)ksteps * 64 + ob];
            okm |= (cok ? 1u : 0u) << i; okm |= (on ? 1u : 0u) << (16 + i);
        }
    }
    // u1, u2 of the tile: 2 x 16 m consecutive doubles, coalesced (a gather in operand order costs 16 cache lines per request and
    // the CU's load path is what this launch waits for), through LDS
    double tu[2][9];
    {
        const int len = np * m;
#pragma unroll
        for (int k9 = 0; k9 < 9; ++k9) {
            const int idx = k9 * FS_THREADS + tid;
            const size_t o = (size_t)p0 * m + (idx < len ? idx : 0);
            tu[0][k9] = I.u1 ? I.u1[o] : 0.0; tu[1][k9] = I.u2 ? I.u2[o] : 0.0;
        }
    }
    // u into LDS as soon as it is there; the images are requested behind it and arrive while B u is being formed
    {
        const int len = np * m;
#pragma unroll
        for (int k9 = 0; k9 < 9; ++k9) {
            const int idx = k9 * FS_THREADS + tid;
            if (idx < 16 * m) {
                const int pp = idx / m;
                su[pp * 145 + idx - pp * m] = idx < len ? tu[0][k9] : 0.0;
                su[(16 + pp) * 145 + idx - pp * m] = idx < len ? tu[1][k9] : 0.0;
            }
        }
    }
    asm volatile("" ::: "memory");
    __syncthreads();
    {
        const int ksteps = (m + 3) / 4, per = (ksteps + 3) / 4, k0 = wv * per;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int c = 4 * (k0 + i) + lk;
            const bool cok = (okm >> i) & 1u, on = (okm >> (16 + i)) & 1u;
            const double t1 = su[li * 145 + (c < m ? c : 0)], t2 = su[(16 + li) * 145 + (c < m ? c : 0)];
            v1[i] = cok ? t1 : 0.0; v2[i] = cok ? t2 : 0.0;
            b0[i] = on ? b0[i] : 0.0; b1[i] = on ? b1[i] : 0.0;
        }
    }
#ifdef FW_TIMING
    asm volatile("s_nop 0" :: "v"(b0[0]), "v"(b1[8]), "v"(v1[0]), "v"(v2[8]));
    FS_TICK(1);
#endif
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        bu[0][0] = FL_MFMA(b0[i], v1[i], bu[0][0]); bu[0][1] = FL_MFMA(b1[i], v1[i], bu[0][1]);
        bu[1][0] = FL_MFMA(b0[i], v2[i], bu[1][0]); bu[1][1] = FL_MFMA(b1[i], v2[i], bu[1][1]);
    }
    // ---- requests that do not depend on B u -- a and x0_last in operand order, this wavefront's images -- issued behind the
    // products of B u (a wavefront whose requests queue up in front of the barrier above holds its whole workgroup back)
    double A0[FL_KS], A1[FL_KS];
    bool hasB = false;
    int tA = 0, tB = 0;
    if (role == 0) {
        const int e = e0 + 16 * wv + li;                                        // first row tile of M1 | M2: A0[0..6] | A0[7..13]
        const bool eok = e < e1;
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            const int q = 4 * ks + lk;
            const bool ok = eok && q < n;
            const size_t off = (size_t)(eok ? e : e0) * n + (q < n ? q : 0);
            const double t1 = I.M1[off], t2 = I.M2[off];
            A0[ks] = ok ? t1 : 0.0; A0[7 + ks] = ok ? t2 : 0.0;
        }
    } else if (role == 1) {
        const int half = blockIdx.y - rs, mt = (m + 15) / 16, h0 = (mt + 1) / 2;       // tiles [0, h0) and [h0, mt): <= 8 each
        const int base = half ? h0 : 0, cnt = half ? mt - h0 : h0;
        tA = base + wv; tB = base + wv + 4;
        hasB = wv + 4 < cnt;
        if (wv >= cnt) tA = base;                                                      // (idle wavefront: repeats a tile, stores nothing)
#pragma unroll
        for (int q = 0; q < FL_KS; ++q) A0[q] = P.imgU[((size_t)tA * FL_KS + q) * 64 + lane];
        if (hasB) {
#pragma unroll
            for (int q = 0; q < FL_KS; ++q) A1[q] = P.imgU[((size_t)tB * FL_KS + q) * 64 + lane];
        }
        if (wv >= cnt) tA = -1;
    } else {
        const int j = wv, na = FL_KS - 4 * j;                                            // share j of both forms (fl_forms_share)
#pragma unroll
        for (int q = 0; q < FL_KS; ++q) {
            const int tl = q < na ? j : 7 - j, ks = q < na ? 4 * j + q : 4 * (7 - j) + (q - na);
            A0[q] = P.imgE[((size_t)tl * FL_KS + ks) * 64 + lane];
            A1[q] = P.imgEp[((size_t)tl * FL_KS + ks) * 64 + lane];
        }
    }
    double xa[7], xl[7];
    if (role != 0 || (blockIdx.y == 0 && wv == 0)) {
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            const int q = 4 * ks + lk;
            const size_t g = (size_t)pl * n + (q < n ? q : 0);
            const double ta = I.a[g], tl = I.x0_last ? I.x0_last[g] : 0.0;
            xa[ks] = q < n ? ta : 0.0; xl[ks] = q < n ? tl : 0.0;
        }
    }
    double xr = 0.0;
    if (role == 2 && P.nu0 && p0 + (tid >> 4) < P.batch) {                               // bound of ||r_d(nu0)||^2: 16 threads per realisation
        const double* nu = P.nu0 + (size_t)(p0 + (tid >> 4)) * P.nb * n;
        const int r = tid & 15, r2 = r + 16;
        const double x1 = P.dx0T[r] + nu[(P.T - 1) * n + r] + (P.has_xf ? nu[P.T * n + r] : 0.0);
        const double x2 = r2 < n ? P.dx0T[r2] + nu[(P.T - 1) * n + r2] + (P.has_xf ? nu[P.T * n + r2] : 0.0) : 0.0;
        xr = fma(x1, x1, x2 * x2);
    }
    __syncthreads();                                    // (u in LDS is dead: its space takes the partial tiles)
#pragma unroll
    for (int wh = 0; wh < 2; ++wh)
#pragma unroll
        for (int Ib = 0; Ib < 2; ++Ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) su[((wv * 4 + wh * 2 + Ib) * 4 + r) * 64 + lane] = bu[wh][Ib][r];
    __syncthreads();
#pragma unroll
    for (int wh = 0; wh < 2; ++wh)
#pragma unroll
        for (int Ib = 0; Ib < 2; ++Ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double sacc = 0.0;
#pragma unroll
                for (int v = 0; v < 4; ++v) sacc += su[((v * 4 + wh * 2 + Ib) * 4 + r) * 64 + lane];      // fixed order: the same on every workgroup
                bu[wh][Ib][r] = sacc;
            }
#ifdef FW_TIMING
    asm volatile("s_nop 0" :: "v"(bu[0][0][0]), "v"(bu[1][1][3]));
    FS_TICK(2);
#endif
    if (role == 0) {
        // ---- x0 = a + B u1, x0_pre = x0_last (one wavefront of the first slice), then the slice of w
        if (blockIdx.y == 0 && wv == 0 && li < np && !I.x0_given) {
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {
                const int q = 4 * ks + lk;
                if (q < n) {
                    const size_t g = (size_t)(p0 + li) * n + q;
                    P.x0w[g] = xa[ks] + bu[0][ks >> 2][ks & 3];
                    P.x0pw[g] = xl[ks];
                }
            }
        }
        for (int t = wv; 16 * t < e1 - e0; t += 4) {
            if (t != wv) {
                const int e = e0 + 16 * t + li;
                const bool eok = e < e1;
#pragma unroll
                for (int ks = 0; ks < 7; ++ks) {
                    const int q = 4 * ks + lk;
                    const bool ok = eok && q < n;
                    const size_t off = (size_t)(eok ? e : e0) * n + (q < n ? q : 0);
                    const double t1 = I.M1[off], t2 = I.M2[off];
                    A0[ks] = ok ? t1 : 0.0; A0[7 + ks] = ok ? t2 : 0.0;
                }
            }
            d4l acc = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {
                acc = FL_MFMA(A0[ks], bu[0][ks >> 2][ks & 3], acc);
                acc = FL_MFMA(A0[7 + ks], bu[1][ks >> 2][ks & 3], acc);
            }
            if (li < np) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = e0 + 16 * t + 4 * r + lk;
                    if (e < e1) I.w[(size_t)(p0 + li) * Tn + e] = -acc[r];
                }
            }
        }
        FS_TICK(3);
        return;
    }
    // ---- d in operand order: blocks of 7 k-steps (x0 | x0_pre | B u1 | B u2), the constant in the last pad slot
    double D[FL_KS];
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
        D[ks] = I.x0_given ? xa[ks] : xa[ks] + bu[0][ks >> 2][ks & 3];
        D[7 + ks] = P.var2 ? xl[ks] : 0.0;
        D[14 + ks] = bu[0][ks >> 2][ks & 3];
        D[21 + ks] = bu[1][ks >> 2][ks & 3];
    }
    D[FL_KS - 1] = lk == 3 ? 1.0 : D[FL_KS - 1];
    if (role == 1) {
        // first moves: (image, data): register r of lane (lk, li) is row 16 t + 4 r + lk of realisation li
        d4l acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < FL_KS; ++q) acc0 = FL_MFMA(A0[q], D[q], acc0);
        if (hasB) {
#pragma unroll
            for (int q = 0; q < FL_KS; ++q) acc1 = FL_MFMA(A1[q], D[q], acc1);
        }
        if (li < np) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (tA >= 0 && 16 * tA + 4 * r + lk < m) P.u0out[(size_t)(p0 + li) * m + 16 * tA + 4 * r + lk] = acc0[r];
                if (hasB && 16 * tB + 4 * r + lk < m) P.u0out[(size_t)(p0 + li) * m + 16 * tB + 4 * r + lk] = acc1[r];
            }
        }
        FS_TICK(3);
        return;
    }
    // ---- forms: wavefront j has share j of both (2 x 28 products)
    double qe, qp, ds;
    switch (wv) {
        case 0: fs_forms_pair<0>(A0, A1, D, lk, qe, qp, ds); break;
        case 1: fs_forms_pair<1>(A0, A1, D, lk, qe, qp, ds); break;
        case 2: fs_forms_pair<2>(A0, A1, D, lk, qe, qp, ds); break;
        default: fs_forms_pair<3>(A0, A1, D, lk, qe, qp, ds); break;
    }
    qe += __shfl_xor(qe, 16, 64); qe += __shfl_xor(qe, 32, 64);
    qp += __shfl_xor(qp, 16, 64); qp += __shfl_xor(qp, 32, 64);
    ds += __shfl_xor(ds, 16, 64); ds += __shfl_xor(ds, 32, 64);
    if (lk == 0) { sF[wv][0][li] = qe; sF[wv][1][li] = qp; sF[wv][2][li] = ds; }
    double rd = xr;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) rd += __shfl_xor(rd, o, 64);
    FS_TICK(4);
    __syncthreads();
    if ((tid & 15) == 0 && p0 + (tid >> 4) < P.batch) {
        const int i = tid >> 4;
        double se = 0.0, sp = 0.0, sn = 0.0;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) { se += sF[w4][0][i]; sp += sF[w4][1][i]; sn += sF[w4][2][i]; }
        fl_decide(P, p0 + i, se, sp, sn, P.nu0 ? rd : P.rd2_0);
    }
    FS_TICK(3);
}

hipError_t fmpc_launch_loop_step27(const FlParams& P, const FlStepIn& I, hipStream_t stream) {
    if (P.n != 27 || P.m > 144 || (((P.m + 3) / 4 + 3) / 4) > 9 || I.rs < 1 || I.rows > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fmpc_loop_step27, dim3((P.batch + 15) / 16, I.rs + 3), dim3(FS_THREADS), 0, stream, P, I);
    return hipGetLastError();
}
