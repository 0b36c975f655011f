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
// fmpc_est_psf: one workgroup per (16 rows of the screen, realisation).  Wavefront w takes the columns [w len/4, (w+1) len/4):
//   per k-step (4 columns) a lane computes its pixel of E, the three P_k = E D_k, and issues 8 matrix instructions per
//   diversity for T_k (16 x 32 complex) += P_k (16 x 4) F (4 x 32); the four partial T_k meet in LDS, and the workgroup
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
#define FE_THREADS 256

__global__ void __launch_bounds__(FE_THREADS, 2) fmpc_est_psf(FeParams P) {
    __shared__ double sT[4][2][16][33];                       // per wavefront: partial T (re, im), 16 rows x 32 columns
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, c = lane & 15;
    const int len = P.len, nq = len / 16;                     // k-steps per wavefront
    const int blk = blockIdx.x, r = blockIdx.y;
    const int y0 = 16 * blk;
    const size_t npx = (size_t)len * len;
    const double* scrn = P.scrn + (size_t)r * npx;
    d4e Tr[FE_MAXDIV][2], Ti[FE_MAXDIV][2];
#pragma unroll
    for (int k = 0; k < FE_MAXDIV; ++k)
#pragma unroll
        for (int t = 0; t < 2; ++t) { Tr[k][t] = (d4e){0, 0, 0, 0}; Ti[k][t] = (d4e){0, 0, 0, 0}; }
    // ---- T_k += P_k F over this wavefront's columns.  Lane (g, c): pixel (row y0 + c, column 4 Q + g) of the A operand,
    //      entry (k-row 4 Q + g, window column 16 t + c) of the B operand.
    for (int q = 0; q < nq; ++q) {
        const int Q = wv * nq + q;
        const size_t px = (size_t)(4 * Q + g) * len + (y0 + c);
        const double ph = scrn[px];
        double dr[FE_MAXDIV], di[FE_MAXDIV];
#pragma unroll
        for (int k = 0; k < FE_MAXDIV; ++k) {
            const size_t o = (size_t)(k < P.ndiv ? k : 0) * npx + px;
            dr[k] = P.Dre[o]; di[k] = P.Dim[o];
        }
        const double* fp = P.Fimg + (size_t)Q * 256 + lane;
        const double fr0 = fp[0], fi0 = fp[64], fr1 = fp[128], fi1 = fp[192];
        double sn, cs;
        sincos(ph, &sn, &cs);
#pragma unroll
        for (int k = 0; k < FE_MAXDIV; ++k) {
            if (k < P.ndiv) {                                    // (uniform)
                const double pr = cs * dr[k] - sn * di[k], pi = cs * di[k] + sn * dr[k], npi = -pi;
                Tr[k][0] = FE_MFMA(pr, fr0, Tr[k][0]); Tr[k][0] = FE_MFMA(npi, fi0, Tr[k][0]);
                Ti[k][0] = FE_MFMA(pr, fi0, Ti[k][0]); Ti[k][0] = FE_MFMA(pi, fr0, Ti[k][0]);
                Tr[k][1] = FE_MFMA(pr, fr1, Tr[k][1]); Tr[k][1] = FE_MFMA(npi, fi1, Tr[k][1]);
                Ti[k][1] = FE_MFMA(pr, fi1, Ti[k][1]); Ti[k][1] = FE_MFMA(pi, fr1, Ti[k][1]);
            }
        }
    }
    // ---- per diversity: the four partial T meet in LDS; O = F_blk' T, wavefront w the tile (w / 2, w % 2) of the window
    const int tu = wv >> 1, tv = wv & 1;
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
                        sT[wv][0][4 * rr + g][16 * t + c] = Tr[kk][t][rr];
                        sT[wv][1][4 * rr + g][16 * t + c] = Ti[kk][t][rr];
                    }
            }
        }
        __syncthreads();
        d4e Or = {0, 0, 0, 0}, Oi = {0, 0, 0, 0};
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
            // B operand: T[4 q2 + g][16 tv + c], summed over the four wavefronts in a fixed order
            const int row = 4 * q2 + g, col = 16 * tv + c;
            const double tr = (sT[0][0][row][col] + sT[1][0][row][col]) + (sT[2][0][row][col] + sT[3][0][row][col]);
            const double ti = (sT[0][1][row][col] + sT[1][1][row][col]) + (sT[2][1][row][col] + sT[3][1][row][col]);
            const double nfai = -fai[q2];
            Or = FE_MFMA(far[q2], tr, Or); Or = FE_MFMA(nfai, ti, Or);
            Oi = FE_MFMA(far[q2], ti, Oi); Oi = FE_MFMA(fai[q2], tr, Oi);
        }
        // partial window of these 16 rows: [r][k][blk][re, im][32][32], register rr <-> row 16 tu + 4 rr + g, column 16 tv + c
        double* dst = P.part + ((((size_t)r * P.ndiv + k) * gridDim.x + blk) * 2) * 1024;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int o = (16 * tu + 4 * rr + g) * 32 + 16 * tv + c;
            dst[o] = Or[rr]; dst[1024 + o] = Oi[rr];
        }
    }
}

__global__ void __launch_bounds__(1024) fmpc_est_finish(FeParams P) {
    extern __shared__ double sY[];                           // p = ndiv d^2 measurements minus b_s
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = blockIdx.x, d = P.d, nblk = P.len / 16, p = P.ndiv * d * d;
    const int u = tid & 31, v = tid >> 5;                    // window row (y frequency), column (x frequency)
    for (int k = 0; k < P.ndiv; ++k) {
        if (u < d && v < d) {
            const double* src = P.part + (((size_t)r * P.ndiv + k) * nblk * 2) * 1024 + u * 32 + v;
            double orr = 0.0, oi = 0.0;
            for (int b = 0; b < nblk; ++b) { orr += src[(size_t)b * 2048]; oi += src[(size_t)b * 2048 + 1024]; }
            const int idx = k * d * d + v * d + u;           // reshape(v_im(:,:,k), [], 1): column-major
            double y = (orr * orr + oi * oi) * P.scale;
            if (P.noise) y += P.noise[(size_t)r * p + idx];
            if (P.Yout) P.Yout[(size_t)r * p + idx] = y;
            sY[idx] = y - P.bs[idx];
        }
    }
    __syncthreads();
    // ad_est = G (Y - b_s): row j by wavefront j mod 16, fixed-order sums
    for (int j = wv; j < P.nx; j += 16) {
        const double* gj = P.G + (size_t)j * p;
        double acc = 0.0;
        for (int i = lane; i < p; i += 64) acc = fma(gj[i], sY[i], acc);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) P.ad_est[(size_t)r * P.nx + j] = acc;
    }
}

hipError_t fmpc_launch_estimator(const FeParams& P, hipStream_t stream) {
    if (P.len % 64 != 0 || P.len < 64 || P.d < 1 || P.d > 32 || P.ndiv < 1 || P.ndiv > FE_MAXDIV) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fmpc_est_psf, dim3(P.len / 16, P.batch), dim3(FE_THREADS), 0, stream, P);
    const size_t lds = (size_t)P.ndiv * P.d * P.d * sizeof(double);
    hipLaunchKernelGGL(fmpc_est_finish, dim3(P.batch), dim3(1024), lds, stream, P);
    return hipGetLastError();
}
