// CDNA4 phase-diversity estimator (README.md:456-480): residual phase screen -> PSF windows -> Zernike coefficients.
//
// The reference forms, per timestep and diversity k (zd_list = -3, 0, 3 waves of defocus),
//     P = pupil .* exp(1i*(scrn + kW)) ;  I = fftshift(fft2(fftshift(P), res, res)) * dx^2 ;  im = abs(I).^2 ;
//     v_im(:,:,k) = im(range_min:range_max, range_min:range_max) * AU                                   README.md:461-471
// i.e. a 512 x 512 FFT of which it keeps a 31 x 31 window around the centre, then
//     ad_est = lsqminnorm(A_s'*A_s, A_s'*(Y_M - b_s))                                                   README.md:478
// Only the window is needed, and a window of a centred DFT is a PARTIAL DFT -- two small matrix products,
//     out (d x d) = F' P F ,    F[y][j] = exp(-2 pi i (first + j - len/2)(y - len/2) / len)   (len x d, d = 31 padded to 32)
// which is what the fp64 matrix cores do well (8 len^2 d real multiply-adds per diversity against 5 len^2 log2(len^2) for the
// full FFT, but as dense 16 x 16 x 4 products and without a len^2 complex intermediate in memory).  With P = E .* D_k,
// E = exp(1i*scrn) (one sincos per pixel, shared by the diversities) and D_k = pupil .* exp(1i*zd_k*W) constant per handle.
//
// fmpc_est_psf<NW>: one workgroup of NW wavefronts per (16 rows of the screen, realisation): 4 wavefronts for batches that fill
//   the chip (two workgroups per CU), 8 for a few screens (16 would leave 128 registers per lane: spills; a lone 512 x 512 screen is 32 workgroups: the sequential loop of the
//   reference estimates ONE screen per timestep, and its latency is what counts there).  Wavefront w takes the columns
//   [w len/NW, (w+1) len/NW):
//   per k-step (4 columns) a lane computes its pixel of E, the three P_k = E D_k, and issues 6 matrix instructions per
//   diversity for T_k (16 x 32 complex) += P_k (16 x 4) F (4 x 32) (three real products per complex one); the four partial T_k meet in LDS, and the workgroup
//   applies the other factor at once: O_k (32 x 32 complex) = F_blk' T_k (16 more instructions per wavefront), a PARTIAL sum
//   of the window over these 16 rows, written to the workspace.
// fmpc_est_finish: one workgroup per realisation sums the partial windows in a fixed order (deterministic), forms
//   Y_M = |O|^2 dx^4 AU (+ noise) in the reference's order (README.md:471: column-major per diversity) and
//   ad_est = G (Y_M - b_s) with G = pinv(A_s'A_s) A_s' built once on the host (fmpc_host_estimator_gain).
// Layouts are MATLAB's: scrn, D_k column-major len x len per realisation / diversity (element (row y, column x) at y + len x).
#include <hip/hip_runtime.h>
#include <math.h>
#include "fmpc_device.h"
#include "fmpc_estimator.h"

typedef double d4e __attribute__((ext_vector_type(4)));
#define FE_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define FE_TSTRIDE (2 * 16 * 33)         // doubles of one wavefront's partial T in LDS: (re, im) x 16 rows x 32 columns (+1 pad)

// sin and cos of a phase, for |x| < 1e6 without the library's large-argument machinery (its double-double reduction was 85 of
// the 130 fp64 vector instructions of a k-step, and fp64 vector instructions share the SIMD's units with the matrix instructions):
// Cody-Waite reduction by pi/2 in three parts (the first two have 33 bits: k * part is exact for |k| < 2^20), then the fdlibm
// kernels on [-pi/4, pi/4] (< 1 ulp each; 2.1e-16 absolute against long double over +-1e5, checked on the CPU).  Larger or
// non-finite arguments give NaN (keeping the library routine as a fallback costs the registers the accumulators need: spills).
__device__ __forceinline__ void fe_sincos(double x, double& sn, double& co) {
    x = fabs(x) < 1.0e6 ? x : __builtin_nan("");           // (a screen of a million radians is not a phase screen: NaN in, NaN out)
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050630396597660e-11, r);
    r = fma(-k, 2.02226624879595063154e-21, r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                           -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double s0 = fma(r * z, ps, r);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                           2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)k & 3;
    const double sa = (q & 1) ? c0 : s0, ca = (q & 1) ? s0 : c0;
    sn = (q & 2) ? -sa : sa;
    co = ((q + 1) & 2) ? -ca : ca;
}

template <int NW>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) fmpc_est_psf(FeParams P) {      // (second argument: workgroups per CU here = 2 wavefronts per SIMD either way)
    extern __shared__ double sTd[];                           // [NW][2][16][33]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, c = lane & 15;
    const int CS = gridDim.z, cs = blockIdx.z;                // column split over workgroups (few screens: a lone screen is 32 row blocks)
    const int len = P.len;
    // (the screen is the fastest grid index: the workgroups in flight at one time share a few row blocks, whose rows of the
    //  six D_k arrays -- 12.6 MB in all, more than an L2 -- then come from L2 instead of being streamed once per screen)
    const int r = blockIdx.x, blk = blockIdx.y, nblk = gridDim.y;
    const int y0 = 16 * blk;
    const size_t npx = (size_t)len * len;
    const double* scrn = P.scrn + (size_t)r * npx;
    // Complex products with THREE real ones (Gauss): A = sum Re P Re F, B = sum Im P Im F, C = sum (Re P + Im P)(Re F + Im F);
    // Re T = A - B, Im T = C - A - B.  18 matrix instructions per k-step instead of 24.
    d4e TA[FE_MAXDIV][2], TB[FE_MAXDIV][2], TC[FE_MAXDIV][2];
#pragma unroll
    for (int k = 0; k < FE_MAXDIV; ++k)
#pragma unroll
        for (int t = 0; t < 2; ++t) { TA[k][t] = (d4e){0, 0, 0, 0}; TB[k][t] = (d4e){0, 0, 0, 0}; TC[k][t] = (d4e){0, 0, 0, 0}; }
    // ---- T_k += P_k F over this wavefront's columns.  Lane (g, c): pixel (row y0 + c, column 4 Q + g) of the A operand,
    //      entry (k-row 4 Q + g, window column 16 t + c) of the B operand.
    //      Everything a k-step reads is requested one k-step ahead (11 loads per lane): a lone screen has nothing else to
    //      hide the memory latency behind.
    //      Only the k-steps between the first and the last one that see the pupil are visited (D_k = 0 for every k outside: the
    //      corners of the reference's pin-hole pupil are 21 % of the grid; a RANGE per row block, built at create time -- a list
    //      of k-steps put a dependent scalar load in front of every prefetch and cost more than it saved), dealt evenly to the
    //      wavefronts of the workgroup(s).
    const int qlo = P.qrange[2 * blk], cnt = P.qrange[2 * blk + 1] - qlo, W = NW * CS, wi = cs * NW + wv;
    const int i0 = (int)((long long)cnt * wi / W), i1 = (int)((long long)cnt * (wi + 1) / W), nq = i1 - i0;
    const int Q0 = qlo + i0;
    double ph_n, dr_n[FE_MAXDIV], di_n[FE_MAXDIV], f_n[4];
    {
        const size_t px = (size_t)(4 * Q0 + g) * len + (y0 + c);
        ph_n = scrn[px];
#pragma unroll
        for (int k = 0; k < FE_MAXDIV; ++k) { const size_t o = (size_t)(k < P.ndiv ? k : 0) * npx + px; dr_n[k] = P.Dre[o]; di_n[k] = P.Dim[o]; }
        const double* fp = P.Fimg + (size_t)Q0 * 256 + lane;
        f_n[0] = fp[0]; f_n[1] = fp[64]; f_n[2] = fp[128]; f_n[3] = fp[192];
    }
    for (int q = 0; q < nq; ++q) {
        const double ph = ph_n, fr0 = f_n[0], fi0 = f_n[1], fr1 = f_n[2], fi1 = f_n[3];
        double dr[FE_MAXDIV], di[FE_MAXDIV];
#pragma unroll
        for (int k = 0; k < FE_MAXDIV; ++k) { dr[k] = dr_n[k]; di[k] = di_n[k]; }
        {
            const int Qn = Q0 + (q + 1 < nq ? q + 1 : q);
            const size_t px = (size_t)(4 * Qn + g) * len + (y0 + c);
            ph_n = scrn[px];
#pragma unroll
            for (int k = 0; k < FE_MAXDIV; ++k) { const size_t o = (size_t)(k < P.ndiv ? k : 0) * npx + px; dr_n[k] = P.Dre[o]; di_n[k] = P.Dim[o]; }
            const double* fp = P.Fimg + (size_t)Qn * 256 + lane;
            f_n[0] = fp[0]; f_n[1] = fp[64]; f_n[2] = fp[128]; f_n[3] = fp[192];
        }
        double sn, co;
        fe_sincos(ph, sn, co);
        const double fs0 = fr0 + fi0, fs1 = fr1 + fi1;
#pragma unroll
        for (int k = 0; k < FE_MAXDIV; ++k) {
            if (k < P.ndiv) {                                    // (uniform)
                const double pr = co * dr[k] - sn * di[k], pi = co * di[k] + sn * dr[k], ps = pr + pi;
                TA[k][0] = FE_MFMA(pr, fr0, TA[k][0]); TB[k][0] = FE_MFMA(pi, fi0, TB[k][0]); TC[k][0] = FE_MFMA(ps, fs0, TC[k][0]);
                TA[k][1] = FE_MFMA(pr, fr1, TA[k][1]); TB[k][1] = FE_MFMA(pi, fi1, TB[k][1]); TC[k][1] = FE_MFMA(ps, fs1, TC[k][1]);
            }
        }
    }
    // ---- per diversity: the NW partial T meet in LDS; O = F_blk' T, wavefront w < 4 the tile (w / 2, w % 2) of the window
    const int tu = (wv >> 1) & 1, tv = wv & 1;
    double far[4], fai[4];                                   // A operand of the second product: F[y0 + 4 q2 + g][16 tu + c]
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {
        const double* fp = P.Fimg + (size_t)(y0 / 4 + q2) * 256 + tu * 128 + lane;
        far[q2] = fp[0]; fai[q2] = fp[64];
    }
    for (int k = 0; k < P.ndiv; ++k) {
        __syncthreads();                                     // sT free again
#pragma unroll
        for (int kk = 0; kk < FE_MAXDIV; ++kk) {
            if (kk == k) {                                   // (uniform; static register indices)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {          // register rr <-> row 4 rr + g, column 16 t + c
                        sTd[wv * FE_TSTRIDE + (4 * rr + g) * 33 + 16 * t + c] = TA[kk][t][rr] - TB[kk][t][rr];
                        sTd[wv * FE_TSTRIDE + 528 + (4 * rr + g) * 33 + 16 * t + c] = (TC[kk][t][rr] - TA[kk][t][rr]) - TB[kk][t][rr];
                    }
            }
        }
        __syncthreads();
        if (wv >= 4) continue;                               // (uniform per wavefront; the barriers above are reached by all)
        d4e Or = {0, 0, 0, 0}, Oi = {0, 0, 0, 0};
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
            // B operand: T[4 q2 + g][16 tv + c], summed over the wavefronts in a fixed order (groups of four)
            const int o = (4 * q2 + g) * 33 + 16 * tv + c;
            double tr = 0.0, ti = 0.0;
#pragma unroll
            for (int w4 = 0; w4 < NW; w4 += 4) {
                const double* b4 = sTd + w4 * FE_TSTRIDE + o;
                tr += (b4[0] + b4[FE_TSTRIDE]) + (b4[2 * FE_TSTRIDE] + b4[3 * FE_TSTRIDE]);
                ti += (b4[528] + b4[FE_TSTRIDE + 528]) + (b4[2 * FE_TSTRIDE + 528] + b4[3 * FE_TSTRIDE + 528]);
            }
            const double nfai = -fai[q2];
            Or = FE_MFMA(far[q2], tr, Or); Or = FE_MFMA(nfai, ti, Or);
            Oi = FE_MFMA(far[q2], ti, Oi); Oi = FE_MFMA(fai[q2], tr, Oi);
        }
        // partial window of these 16 rows: [r][k][blk][re, im][32][32], register rr <-> row 16 tu + 4 rr + g, column 16 tv + c
        double* dst = P.part + ((((size_t)r * P.ndiv + k) * nblk * CS + (blk * CS + cs)) * 2) * 1024;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int o = (16 * tu + 4 * rr + g) * 32 + 16 * tv + c;
            dst[o] = Or[rr]; dst[1024 + o] = Oi[rr];
        }
    }
}

__global__ void __launch_bounds__(1024) fmpc_est_finish(FeParams P) {
    extern __shared__ double sY[];                           // d^2 measurements of this diversity minus b_s, then nx x 16 partial sums
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = blockIdx.x, k = blockIdx.y, d = P.d, nblk = P.len / 16, dd = d * d, p = P.ndiv * dd;
    double* sRed = sY + ((dd + 1) & ~1);
    const int v = tid & 31, u = tid >> 5;                    // window column (x frequency: consecutive lanes read consecutive doubles
                                                             // of a partial window -- the other way round a wavefront touched 64 cache
                                                             // lines per load and the kernel took 104 us for one screen), row (y frequency)
    // One workgroup per (screen, diversity), which leaves its share of ad_est (fmpc_est_combine adds the shares).  Every load sits in a chain of dependent steps (a lone screen: the reference's loop): loads are
    // issued in batches of 16 (partial windows) and 9 (rows of G) so that a batch costs one memory round trip, not one each.
    if (u < d && v < d) {
        const double* src = P.part + (((size_t)r * P.ndiv + k) * nblk * 2) * 1024 + u * 32 + v;
        double orr = 0.0, oi = 0.0;                        // fixed order (nblk is a multiple of 4)
        int b = 0;
        for (; b + 8 <= nblk; b += 8) {
            double re[8], im[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { re[j] = src[(size_t)(b + j) * 2048]; im[j] = src[(size_t)(b + j) * 2048 + 1024]; }
            orr += ((re[0] + re[1]) + (re[2] + re[3])) + ((re[4] + re[5]) + (re[6] + re[7]));
            oi += ((im[0] + im[1]) + (im[2] + im[3])) + ((im[4] + im[5]) + (im[6] + im[7]));
        }
        for (; b < nblk; b += 4) {
            double re[4], im[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { re[j] = src[(size_t)(b + j) * 2048]; im[j] = src[(size_t)(b + j) * 2048 + 1024]; }
            orr += (re[0] + re[1]) + (re[2] + re[3]);
            oi += (im[0] + im[1]) + (im[2] + im[3]);
        }
        const int il = v * d + u, idx = k * dd + il;       // reshape(v_im(:,:,k), [], 1): column-major
        double y = (orr * orr + oi * oi) * P.scale;
        if (P.noise) y += P.noise[(size_t)r * p + idx];
        if (P.Yout) P.Yout[(size_t)r * p + idx] = y;
        sY[il] = y - P.bs[idx];
    }
    __syncthreads();
    // this diversity's share of ad_est = G (Y - b_s): every thread a strided share of every row (9 rows at a time: their loads in
    // one batch), wavefront sums, then a fixed-order sum of the 16 wavefronts
    constexpr int FE_ROWB = 9;
    for (int j0 = 0; j0 < P.nx; j0 += FE_ROWB) {
        double acc[FE_ROWB];
#pragma unroll
        for (int jj = 0; jj < FE_ROWB; ++jj) acc[jj] = 0.0;
        for (int i = tid; i < dd; i += 1024) {
            const double yv = sY[i];
            double gv[FE_ROWB];
#pragma unroll
            for (int jj = 0; jj < FE_ROWB; ++jj) gv[jj] = P.G[(size_t)(j0 + jj < P.nx ? j0 + jj : j0) * p + k * dd + i];
#pragma unroll
            for (int jj = 0; jj < FE_ROWB; ++jj) acc[jj] = fma(gv[jj], yv, acc[jj]);
        }
#pragma unroll
        for (int jj = 0; jj < FE_ROWB; ++jj) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) acc[jj] += __shfl_xor(acc[jj], o, 64);
            if (lane == 0 && j0 + jj < P.nx) sRed[(j0 + jj) * 16 + wv] = acc[jj];
        }
    }
    __syncthreads();
    double* share = P.shares + ((size_t)r * P.ndiv + k) * P.nx;
    if (tid < P.nx) {
        const double* q = sRed + tid * 16;
        share[tid] = (((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]))) +
                     (((q[8] + q[9]) + (q[10] + q[11])) + ((q[12] + q[13]) + (q[14] + q[15])));
    }
}

// fmpc_est_finish for a FEW screens (the reference's loop: one): the same sums by four workgroups of 256 threads per (screen,
// diversity) -- eight window rows each -- whose threads may hold 256 registers: the 64 loads of a pixel's 32 partial windows and
// its 27 entries of G are ALL requested before the first is used, one memory round trip where the 1024-thread form (128 registers
// per thread) has four for the partial windows and three for G.  Shares: [screen][diversity][quarter][nx].
#define FE_FQ 4
#ifndef FE_FEW_MAX
#define FE_FEW_MAX 64          // (screens x diversities) up to which the finish pass takes this form
#endif
template <int NCH>                                           // partial windows per pixel: 32 NCH (column split of the PSF kernel)
__global__ void __launch_bounds__(256) fmpc_est_finish_few(FeParams P) {
    __shared__ double sRed[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = blockIdx.x, k = blockIdx.y, qd = blockIdx.z, d = P.d, nblk = 32 * NCH, dd = d * d, p = P.ndiv * dd;
    const int v = tid & 31, u = 8 * qd + (tid >> 5);          // window column (consecutive lanes: consecutive doubles), row
    const bool own = u < d && v < d;
    const int il = v * d + u, idx = k * dd + il;              // reshape(v_im(:,:,k), [], 1): column-major
    constexpr int NB = 32, NG = 27;                           // (the launcher takes this form for nblk == 32 and nx <= 27 only)
    double re[NB], im[NB], gv[NG];
    const double* src = P.part + (((size_t)r * P.ndiv + k) * nblk * 2) * 1024 + (own ? u * 32 + v : 0);
#pragma unroll
    for (int b = 0; b < NB; ++b) { re[b] = src[(size_t)b * 2048]; im[b] = src[(size_t)b * 2048 + 1024]; }
#pragma unroll
    for (int j = 0; j < NG; ++j) gv[j] = P.G[(size_t)(j < P.nx ? j : 0) * p + (own ? idx : 0)];
    double orr = 0.0, oi = 0.0;                               // fixed order: groups of eight
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
        for (int b = 0; b < NB; b += 8) {
            orr += ((re[b] + re[b + 1]) + (re[b + 2] + re[b + 3])) + ((re[b + 4] + re[b + 5]) + (re[b + 6] + re[b + 7]));
            oi += ((im[b] + im[b + 1]) + (im[b + 2] + im[b + 3])) + ((im[b + 4] + im[b + 5]) + (im[b + 6] + im[b + 7]));
        }
        if (ch + 1 < NCH) {                                   // (a further 32 partial windows: one more round trip)
#pragma unroll
            for (int b = 0; b < NB; ++b) { re[b] = src[(size_t)((ch + 1) * NB + b) * 2048]; im[b] = src[(size_t)((ch + 1) * NB + b) * 2048 + 1024]; }
        }
    }
    double y = (orr * orr + oi * oi) * P.scale;
    if (own && P.noise) y += P.noise[(size_t)r * p + idx];
    if (own && P.Yout) P.Yout[(size_t)r * p + idx] = y;
    const double yv = own ? y - P.bs[idx] : 0.0;
    double* share = P.shares + (((size_t)r * P.ndiv + k) * FE_FQ + qd) * P.nx;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        double a = j < P.nx ? gv[j] * yv : 0.0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) sRed[wv][j] = a;
    }
    __syncthreads();
    if (tid < P.nx) share[tid] = (sRed[0][tid] + sRed[1][tid]) + (sRed[2][tid] + sRed[3][tid]);
}

// ad_est = sum over the diversities of their shares, in a fixed order.  (A launch of its own: the last-arriving workgroup of
// fmpc_est_finish doing it needs device-scope release fences, which write back an L2 full of partial windows: 35 % slower at
// 256 screens, measured.)
__global__ void __launch_bounds__(256) fmpc_est_combine(FeParams P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.batch * P.nx) return;
    const int r = i / P.nx, j = i - r * P.nx;
    const double* sh = P.shares + (size_t)r * P.ndiv * P.nshare * P.nx + j;
    double a = 0.0;
    for (int kk = 0; kk < P.ndiv * P.nshare; ++kk) a += sh[(size_t)kk * P.nx];
    P.ad_est[i] = a;
}

hipError_t fmpc_launch_estimator(const FeParams& P, hipStream_t stream) {
    if (P.len % 64 != 0 || P.len < 64 || P.d < 1 || P.d > 32 || P.ndiv < 1 || P.ndiv > FE_MAXDIV) return hipErrorInvalidValue;
    // few screens: 8 wavefronts per workgroup so that a lone screen is 256 wavefronts, not 128
    const bool wide = (size_t)P.batch * (P.len / 16) < 512 && P.len % 128 == 0;
    // very few screens of the reference's size: the columns of a row block split over two workgroups as well (a lone screen is
    // then 64 workgroups of 8 wavefronts), twice the partial windows for the finish pass
    int csplit = 1;
    if (P.len == 512 && P.nx <= 27 && P.shares_cap >= (size_t)P.batch * P.ndiv * FE_FQ * P.nx) {
        // (a split of four for ONE screen -- 128 workgroups, four round trips in the finish pass -- measured no faster: 39.6 against 38.1 us)
        if ((size_t)P.batch * P.ndiv <= 12 && P.part_cap >= (size_t)P.batch * P.ndiv * 64 * 2048) csplit = 2;
    }
    if (wide) {
        static bool prepared = false;
        const size_t lds8 = (size_t)8 * FE_TSTRIDE * sizeof(double);
        if (!prepared) {
            if (hipFuncSetAttribute((const void*)fmpc_est_psf<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8) != hipSuccess) return hipErrorInvalidValue;
            prepared = true;
        }
        hipLaunchKernelGGL(fmpc_est_psf<8>, dim3(P.batch, P.len / 16, csplit), dim3(512), lds8, stream, P);
    } else {
        hipLaunchKernelGGL(fmpc_est_psf<4>, dim3(P.batch, P.len / 16), dim3(256), (size_t)4 * FE_TSTRIDE * sizeof(double), stream, P);
    }
    FeParams Q = P;
    if ((size_t)P.batch * P.ndiv <= FE_FEW_MAX && P.len == 512 && P.nx <= 27 && P.shares_cap >= (size_t)P.batch * P.ndiv * FE_FQ * P.nx) {
        Q.nshare = FE_FQ;
        if (csplit == 2) hipLaunchKernelGGL(fmpc_est_finish_few<2>, dim3(P.batch, P.ndiv, FE_FQ), dim3(256), 0, stream, Q);
        else hipLaunchKernelGGL(fmpc_est_finish_few<1>, dim3(P.batch, P.ndiv, FE_FQ), dim3(256), 0, stream, Q);
    } else {
        Q.nshare = 1;
        const size_t lds = ((((size_t)P.d * P.d + 1) & ~(size_t)1) + (size_t)P.nx * 16) * sizeof(double);
        hipLaunchKernelGGL(fmpc_est_finish, dim3(P.batch, P.ndiv), dim3(1024), lds, stream, Q);
    }
    hipLaunchKernelGGL(fmpc_est_combine, dim3((P.batch * P.nx + 255) / 256), dim3(256), 0, stream, Q);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------
// Residual phase screen of a timestep (README.md:453 with :590-601):
//     ad_cor = B*u_prev ;  phase_cor = sum_j ad_cor(j) .* Zs(j+1,:,:) ;  phase_res = phase_valid(:,:,k) + phase_cor
// for a batch of screens: out[b][px] = phase[b][px] + sum_j (B u[b])_j Z[j][px].  A workgroup takes 256 pixels and walks over
// the batch in groups of FE_PB screens, so that the n mode maps (n x npx doubles: 57 MB at n = 27, len = 512) are read once per
// group, not once per screen.  u == NULL: out = phase (the first step of the loop, README.md:447).
#define FE_PB 8
__global__ void __launch_bounds__(256) fmpc_phase_residual(int batch, size_t npx, int n, int m, const double* Bt, const double* phase,
                                                           const double* u, const double* Z, double* out) {
    __shared__ double sp[8][FE_PB][32];                       // partial sums of ad_cor = B u: [eighth of the actuators][screen][mode]
    __shared__ double sc[FE_PB][32];                          // ad_cor of the group's screens (n <= 32)
    const int tid = threadIdx.x;
    const size_t px = (size_t)blockIdx.x * 256 + tid;
    for (int b0 = 0; b0 < batch; b0 += FE_PB) {
        __syncthreads();
        {
            // thread (mode j, eighth `part` of the actuators): a chain of m / 8 products per screen instead of m, the rows of B'
            // loaded once for the group's screens (every workgroup forms B u itself: a chain of 144 dependent loads and products
            // in front of the pass over the maps was most of this kernel's time for one screen)
            const int j = tid & 31, part = tid >> 5, per = (m + 7) / 8, c0 = part * per, c1 = c0 + per < m ? c0 + per : m;
            double a[FE_PB];
#pragma unroll
            for (int bb = 0; bb < FE_PB; ++bb) a[bb] = 0.0;
            if (u && j < n) {
                for (int cb = c0; cb < c1; cb += 6) {
                    double bt[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) bt[q] = cb + q < c1 ? Bt[(size_t)(cb + q) * n + j] : 0.0;      // Bt[c*n + r] = B[r][c]
#pragma unroll
                    for (int bb = 0; bb < FE_PB; ++bb) {
                        if (b0 + bb < batch) {
                            const double* ub = u + (size_t)(b0 + bb) * m;
#pragma unroll
                            for (int q = 0; q < 6; ++q) a[bb] = fma(bt[q], cb + q < c1 ? ub[cb + q] : 0.0, a[bb]);
                        }
                    }
                }
            }
#pragma unroll
            for (int bb = 0; bb < FE_PB; ++bb) sp[part][bb][j] = a[bb];
        }
        __syncthreads();
        {
            const int bb = tid >> 5, j = tid & 31;
            double a = 0.0;
#pragma unroll
            for (int part = 0; part < 8; ++part) a += sp[part][bb][j];
            sc[bb][j] = j < n ? a : 0.0;
        }
        __syncthreads();
        if (px < npx) {
            double acc[FE_PB];
#pragma unroll
            for (int bb = 0; bb < FE_PB; ++bb) acc[bb] = b0 + bb < batch ? phase[(size_t)(b0 + bb) * npx + px] : 0.0;
            if (u) {
                for (int j0 = 0; j0 < n; j0 += 9) {           // the maps nine at a time: their loads in flight together
                    double z[9];
#pragma unroll
                    for (int q = 0; q < 9; ++q) z[q] = Z[(size_t)(j0 + q < n ? j0 + q : 0) * npx + px];
#pragma unroll
                    for (int q = 0; q < 9; ++q) {
                        const int jc = j0 + q < 32 ? j0 + q : 31;      // (sc is zero from n on)
                        const double zz = j0 + q < n ? z[q] : 0.0;
#pragma unroll
                        for (int bb = 0; bb < FE_PB; ++bb) acc[bb] = fma(sc[bb][jc], zz, acc[bb]);
                    }
                }
            }
#pragma unroll
            for (int bb = 0; bb < FE_PB; ++bb) if (b0 + bb < batch) out[(size_t)(b0 + bb) * npx + px] = acc[bb];
        }
    }
}

hipError_t fmpc_launch_phase_residual(int batch, size_t npx, int n, int m, const double* Bt, const double* phase, const double* u,
                                      const double* Z, double* out, hipStream_t stream) {
    if (n > 32 || n < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fmpc_phase_residual, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, stream, batch, npx, n, m, Bt, phase, u, Z, out);
    return hipGetLastError();
}
