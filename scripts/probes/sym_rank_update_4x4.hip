// Probe (round 5): S = B diag(w) B' (27 x 144 x 27, the rank-144 update of the wave kernel's factor phase) per wavefront, 8 wavefronts
// per CU, operands in LDS -- (0) as the kernel does it: three 16x16x4 tiles per k-step (S00, S01, S11: 192 matrix-pipe cycles per 4 k),
// (1) on v_mfma_f64_4x4x4_4b: the 28 lower 4x4 block pairs of the 28 x 28 result as 7 instructions per k-step, operands gathered from a
// [row][k] copy of B with 16-byte reads (two k-steps per read; the k order inside a step is free: A and B use the same permutation).
// Lane maps (mfma_f64_4x4.hip): A at lane i + 4b + 16k, B at j + 4b + 16k, D at j + 4b + 16i.
// Block pairs per slot b and instruction s: b0: (6,s) | b1: (5,0..5),(0,0) | b2: (4,0..4),(1,0),(1,1) | b3: (3,0..3),(2,0..2).
//   hipcc --offload-arch=gfx950 -O3 -o sym_rank_update_4x4 sym_rank_update_4x4.hip && ./sym_rank_update_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define NN 27
#define MM 144
#define LDB 33          // [k][row] layout of the kernel
#define LDK 146         // [row][k] layout of the probe (even: 16-byte aligned pairs)
#define IT 200
__device__ __forceinline__ void pair_of(int b, int s, int& I, int& J) {
    if (b == 0) { I = 6; J = s; }
    else if (b == 1) { if (s < 6) { I = 5; J = s; } else { I = 0; J = 0; } }
    else if (b == 2) { if (s < 5) { I = 4; J = s; } else { I = 1; J = s - 5; } }
    else { if (s < 4) { I = 3; J = s; } else { I = 2; J = s - 4; } }
}
template <int MODE>
__global__ void __launch_bounds__(512, 1) probe(const double* Bt, const double* w, double* out, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) double sB[MM * LDB];        // [k][row]
    __shared__ __attribute__((aligned(16))) double sBk[28 * LDK];       // [row][k]
    __shared__ __attribute__((aligned(16))) double sW[8][MM];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < MM * LDB; i += 512) { const int k = i / LDB, r = i - k * LDB; sB[i] = r < NN ? Bt[k * NN + r] : 0.0; }
    for (int i = tid; i < 28 * LDK; i += 512) { const int r = i / LDK, k = i - r * LDK; sBk[i] = (r < NN && k < MM) ? Bt[k * NN + r] : 0.0; }
    for (int k = lane; k < MM; k += 64) sW[wv][k] = w[wv * MM + k];
    __syncthreads();
    const int g = lane >> 4, c16 = lane & 15;
    double res[12];
    for (int q = 0; q < 12; ++q) res[q] = 0.0;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {
        d4 S00 = {0, 0, 0, 0}, S01 = S00, S11 = S00;
        for (int it = 0; it < IT; ++it) {
            const double* wi = sW[wv] + g;
            const double* bp = sB + g * LDB + c16;
            for (int kc = 0; kc < MM; kc += 16) {
                double b0[4], b1[4], wk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { wk[q] = wi[kc + 4 * q]; b0[q] = bp[(kc + 4 * q) * LDB]; b1[q] = bp[(kc + 4 * q) * LDB + 16]; }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double a0 = b0[q] * wk[q], a1 = b1[q] * wk[q];
                    S00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0[q], S00, 0, 0, 0);
                    S01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1[q], S01, 0, 0, 0);
                    S11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1[q], S11, 0, 0, 0);
                }
            }
        }
        for (int r = 0; r < 4; ++r) { res[r] = S00[r]; res[4 + r] = S01[r]; res[8 + r] = S11[r]; }
    } else {
        const int i4 = lane & 3, b = (lane >> 2) & 3, kk = lane >> 4;
        // per-lane row of the A operand (two row blocks per slot) and of the 7 B operands
        int I0, Jd, I1 = 0;
        pair_of(b, 0, I0, Jd);
        pair_of(b, 6, I1, Jd);
        const d2* pa0 = (const d2*)(sBk + (4 * I0 + i4) * LDK + 2 * kk);
        const d2* pa1 = (const d2*)(sBk + (4 * I1 + i4) * LDK + 2 * kk);
        const d2* pb[7];
        bool second[7];
#pragma unroll
        for (int s = 0; s < 7; ++s) { int I, J; pair_of(b, s, I, J); pb[s] = (const d2*)(sBk + (4 * J + i4) * LDK + 2 * kk); second[s] = I != I0; }
        const d2* pw = (const d2*)(sW[wv] + 2 * kk);
        double acc[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int it = 0; it < IT; ++it) {
#pragma unroll 2
            for (int q = 0; q < MM / 8; ++q) {                 // 8 k per pass: two k-steps of 4
                const d2 wv2 = pw[4 * q], a0 = pa0[4 * q], a1 = pa1[4 * q];
                d2 bb[7];
#pragma unroll
                for (int s = 0; s < 7; ++s) bb[s] = pb[s][4 * q];
                const double a00 = a0.x * wv2.x, a01 = a0.y * wv2.y, a10 = a1.x * wv2.x, a11 = a1.y * wv2.y;
#pragma unroll
                for (int s = 0; s < 7; ++s) {
                    acc[s] = __builtin_amdgcn_mfma_f64_4x4x4f64(second[s] ? a10 : a00, bb[s].x, acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f64_4x4x4f64(second[s] ? a11 : a01, bb[s].y, acc[s], 0, 0, 0);
                }
            }
        }
        for (int s = 0; s < 7; ++s) res[s] = acc[s];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    for (int q = 0; q < 12; ++q) out[((size_t)blockIdx.x * 512 + tid) * 12 + q] = res[q];
    if (lane == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}
int main() {
    std::vector<double> hB(MM * NN), hw(8 * MM);
    for (int i = 0; i < MM * NN; ++i) hB[i] = sin(0.37 * i) + 0.1;
    for (int i = 0; i < 8 * MM; ++i) hw[i] = 1.0 / (1.0 + (i % 17));
    double *dB, *dw, *dout; unsigned long long* dc;
    hipMalloc(&dB, hB.size() * 8); hipMalloc(&dw, hw.size() * 8); hipMalloc(&dout, (size_t)256 * 512 * 12 * 8); hipMalloc(&dc, 256 * 8 * 8);
    hipMemcpy(dB, hB.data(), hB.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 8, hipMemcpyHostToDevice);
    std::vector<double> o0((size_t)512 * 12), o1((size_t)512 * 12);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) probe<0><<<256, 512>>>(dB, dw, dout, dc); else probe<1><<<256, 512>>>(dB, dw, dout, dc);
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> hc(256 * 8);
        hipMemcpy(hc.data(), dc, hc.size() * 8, hipMemcpyDeviceToHost);
        double m = 0; for (auto v : hc) m += v; m /= hc.size();
        printf("mode %d (%s): %.0f cycles per wavefront and product (K = 144), 8 wavefronts per CU\n", mode, mode ? "4x4x4_4b, 28 block pairs" : "16x16x4, three tiles", m / IT);
        hipMemcpy((mode ? o1 : o0).data(), dout, (size_t)512 * 12 * 8, hipMemcpyDeviceToHost);
    }
    // compare: wave 0 of block 0.  S[r][c] for r >= c from both layouts (each product accumulated IT times)
    double worst = 0, ref = 0;
    for (int r = 0; r < NN; ++r)
        for (int c = 0; c <= r; ++c) {
            // exact
            double ex = 0; for (int k = 0; k < MM; ++k) ex += hB[k * NN + r] * hw[k] * hB[k * NN + c];
            ex *= IT;
            // mode 0: tile (r / 16, c / 16): register rr of lane (g, c16): row 4 rr + g ... S = A' B with A[k][i], result D[i][j] at lane j + 16 * (i / 4)?, reg i % 4
            // (16x16x4: D[i][j]: lane = j + 16 * (i % 4)?? -- use the kernel's convention: register r of lane (g, c16) holds (4 r + g, c16))
            const int I = r / 16, J = c / 16;
            double v0;
            { const int ri = r % 16, cj = c % 16; const int rr = ri / 4, gg = ri % 4; const int lane = gg * 16 + cj;
              const int base = I == 0 ? 0 : (J == 0 ? 4 : 8);
              if (I == 1 && J == 0) { // S01 holds rows 0..15 x cols 16..31: entry (c, r) by symmetry
                  const int ri2 = c % 16, cj2 = r % 16, rr2 = ri2 / 4, gg2 = ri2 % 4; v0 = o0[(size_t)(gg2 * 16 + cj2) * 12 + 4 + rr2];
              } else v0 = o0[(size_t)lane * 12 + base + rr]; }
            // mode 1: block pair (r / 4, c / 4): find slot / instruction
            double v1 = NAN;
            for (int b = 0; b < 4; ++b) for (int s = 0; s < 7; ++s) {
                int II, JJ;
                if (b == 0) { II = 6; JJ = s; } else if (b == 1) { if (s < 6) { II = 5; JJ = s; } else { II = 0; JJ = 0; } }
                else if (b == 2) { if (s < 5) { II = 4; JJ = s; } else { II = 1; JJ = s - 5; } } else { if (s < 4) { II = 3; JJ = s; } else { II = 2; JJ = s - 4; } }
                if (II == r / 4 && JJ == c / 4) v1 = o1[(size_t)((c % 4) + 4 * b + 16 * (r % 4)) * 12 + s];
            }
            worst = fmax(worst, fmax(fabs(v0 - ex), fabs(v1 - ex))); ref = fmax(ref, fabs(ex));
        }
    printf("max |S - exact| over the lower triangle, both layouts: %.3e (|S| <= %.3e)\n", worst, ref);
    return 0;
}
