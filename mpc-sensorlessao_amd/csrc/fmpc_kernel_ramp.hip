// fastMPC Newton kernel WITH ramp-rate rows (the VAR_1 variant of the reference) for gfx950: any (n <= 64, m, T).
//
// Reference: Fast_MPC/VAR_1/fast_mpc_ineq_const.m:58-76 appends, per stage j, the rows
//      u_j - u_{j-1} <= du_max ,  -(u_j - u_{j-1}) <= -du_min          (u_{-1} = u_prev, moved into h: :70-72)
// to the box rows; everything else is the solver of inf_newton_solver.m:10-41 / inf_newton_KKT_H.m:3-13 /
// backtracking_inf_newton.m:2-11 unchanged.  With D = diag(1/slack^2) over ALL rows:
//   * Phi = 2H + k P'DP couples u_j with u_{j+-1} through DIAGONAL m x m blocks: for every actuator c the u-part of
//     Phi is an independent T x T symmetric tridiagonal matrix
//         diag_j = 2R_cc + k(1/s+^2 + 1/s-^2)_j + er_j + er_{j+1} ,  off_{j,j+1} = -er_{j+1} ,
//         er_j = k(1/sr+_j^2 + 1/sr-_j^2)                                    (sr: the two ramp slacks of stage j)
//     (R diagonal, as on the other device paths); the x-part stays 2Q;
//   * Y = C Phi^-1 C' is therefore DENSE across stages (SURVEY.md §8 a6'):  Y_IJ = Yx_IJ + B diag(g^{IJ}) B' with
//     g^{IJ}_c = (Phi_u,c^-1)_{IJ} and Yx the iteration-invariant block-penta-diagonal part the handle already holds.
// One 256-thread workgroup owns one problem at a time:
//   P1  slacks, barrier terms, r_d, r_p, exit test
//   P2  per actuator: LDL' of its tridiagonal, Phi_u^-1 r_d, the explicit inverse g^{IJ} (O(T^2) recurrence);
//       rhs = r_p - C Phi^-1 r_d
//   P3  Y assembled block by block into the HBM workspace (lower block triangle)
//   P4  blocked left-looking Cholesky of Y (n x n tiles through LDS), forward and backward substitution -> d_nu
//   P5  d_z = Phi^-1(-r_d - C' d_nu) (tridiagonal solves), closed-form line search, update
// The two O(n^3)-class parts run on the matrix cores (v_mfma_f64_16x16x4_f64, generic 16 x 16 tiling with masked
// edges): the B diag(g) B' products of P3 (k = actuators) and the panel updates of the Cholesky factorisation (k = the
// columns already factored).  This is the path of BASELINE config 0 (VAR(1), T = 10); measured numbers in DESIGN.md §6.
#include <hip/hip_runtime.h>
#include <math.h>
#include "fmpc_device.h"
#include "fmpc_tile_ops.h"
#include "fmpc_rampcold.h"
#include "../../include/fastmpc.h"

#define FR_MAX_HALVINGS 64
#define FR_MAXKS 16                     // k-steps of 4 covering n <= 64
#define FR_OWN 3                        // tiles of a block row a wavefront keeps in registers (fr_tile_cholesky)

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ double fr_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// Sum over the workgroup, result to every thread; fixed order -> bitwise reproducible.
template <int NT>
__device__ __forceinline__ double fr_block_sum(double v, double* red) {
    v = fr_wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < (NT >> 6); ++i) s += red[i];
    return s;
}

#ifdef FW_TIMING
// per-phase time of workgroup 0 (100 MHz ticks): P1, P2, P3, P4 factor, P4 substitutions, P5
__device__ unsigned long long fr_timing[16];
extern "C" int fmpc_debug_ramp_timing(unsigned long long* out) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fr_timing), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(fr_timing), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#define FR_TICK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long _t = (unsigned long long)wall_clock64(); fr_timing[k] += _t - _t0; _t0 = _t; } } while (0)
#define FR_T0() unsigned long long _t0 = (unsigned long long)wall_clock64()
#else
#define FR_TICK(k)
#define FR_T0()
#endif

struct FrWsLayout { size_t b, nu, hs, er, gr, dg, lo, rdu, rdx, phx, phu, rp, y, dnu, G, Y, W, total; };
__host__ __device__ static inline FrWsLayout fr_ws_layout(int n, int m, int T, int nb) {
    FrWsLayout L; size_t o = 0;
    const size_t nbn = (size_t)nb * n, Tm = (size_t)T * m, Tn = (size_t)T * n;
    L.b = o; o += nbn;   L.nu = o; o += nbn;
    L.hs = o; o += Tm;   L.er = o; o += Tm;   L.gr = o; o += Tm;   L.dg = o; o += Tm;   L.lo = o; o += Tm;
    L.rdu = o; o += Tm;  L.rdx = o; o += Tn;  L.phx = o; o += Tn;  L.phu = o; o += Tm;
    L.rp = o; o += nbn;  L.y = o; o += nbn;   L.dnu = o; o += nbn;
    L.G = o; o += (size_t)T * (T + 1) / 2 * m;
    const size_t NTl = (nbn + 1 + 15) / 16;  // 16 x 16 tiles covering [Y | rhs] (the rhs is column nbn)
    L.Y = o; o += NTl * NTl * 256;          // dense Y / its factor R (Y = R'R), tile (I, J) at (I NTl + J) 256, upper triangle used
    L.W = o; o += NTl * 256;                // R(kb,kb)^-1 per diagonal tile
    L.total = (o + 15) & ~(size_t)15;
    return L;
}

// Solve the tridiagonal system of actuator c in place (LDL' factors dg = pivots, lo = multipliers), stride m.
__device__ __forceinline__ void fr_tri_solve(const double* dg, const double* lo, double* f, int T, int m, int c) {
    // (the running value is carried in a register: a load of what the previous step stored would wait for that store)
    double prev = f[c];
    for (int j = 1; j < T; ++j) { const double cur = f[j * m + c] - lo[(j - 1) * m + c] * prev; f[j * m + c] = cur; prev = cur; }
    prev = prev / dg[(T - 1) * m + c];
    f[(T - 1) * m + c] = prev;
    for (int j = T - 2; j >= 0; --j) { const double cur = f[j * m + c] / dg[j * m + c] - lo[j * m + c] * prev; f[j * m + c] = cur; prev = cur; }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Dense Cholesky of Y in 16 x 16 tiles on the matrix cores, in the R form Y = R'R of the tiled kernel (fmpc_tile_ops.h):
// Yt holds the upper tile triangle of [Y | rhs] (rhs = column nbn, so y = R^-T rhs appears in that column of the factor),
// row-major tiles in HBM/L2, read as MFMA operands 64 consecutive elements at a time (every product is an X'Z).
// Per 16-row block kb (left-looking):  P(kb,J) = Y(kb,J) - sum_{k<kb} R(k,kb)' R(k,J)  for the tiles J >= kb dealt to the
// wavefronts; the owner of the diagonal tile factors it by 16 rank-1 updates (ft_potrf16: R(kb,kb) and W = R(kb,kb)^-T);
// then R(kb,J) = W P(kb,J).  Two workgroup barriers per block row.  Afterwards the backward substitution R d_nu = y, one block
// row at a time from the bottom (tile x vector products, 16 lanes per tile row, DPP row sums).
// sh: 16 x 17 + 16 NTl + 16 NW + 16 doubles of LDS.  Returns 1 if a pivot is not positive.
__device__ __noinline__ int fr_tile_cholesky(double* Yt, int NTl, int nbn, double* RIt, double* dnu, double* sh) {
    typedef FtT<double> TT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, NW = blockDim.x >> 6, c = lane & 15, g = lane >> 4;
    double* sW = sh;                         // W' of the current diagonal tile, leading dimension 17
    double* xs = sW + 16 * 17;               // d_nu, padded to 16 NTl
    double* part = xs + 16 * NTl;            // [NW][16] partial sums of the backward substitution
    double* tsh = part + 16 * NW;            // [16]
    __shared__ int sfail;
    if (tid == 0) sfail = 0;
    __syncthreads();
    // A wavefront keeps up to FR_OWN tiles of a block row in registers between the two passes; with more tiles per wavefront
    // (long horizons, few wavefronts) the unscaled tiles wait in the workspace instead.
    const bool inreg = (NTl + NW - 1) / NW <= FR_OWN;
    for (int kb = 0; kb < NTl; ++kb) {
        const int cnt = nbn - 16 * kb < 16 ? nbn - 16 * kb : 16;          // live rows of this block row
        ft_d4 own[FR_OWN];
#ifdef FW_TIMING
        const unsigned long long _ta = (unsigned long long)wall_clock64();
#endif
        // ---- pass A: products with the block rows already done; the diagonal tile is factored, the others wait unscaled
        int slot = 0;
        for (int J = kb + wv; J < NTl; J += NW, ++slot) {
            double* tp = Yt + ((size_t)kb * NTl + J) * 256;
            ft_d4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = tp[TT::row(g, r) * 16 + c];
#pragma unroll 4
            for (int k = 0; k < kb; ++k) {
                const double* X = Yt + ((size_t)k * NTl + kb) * 256;
                const double* Z = Yt + ((size_t)k * NTl + J) * 256;
                double xv[4], zv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { xv[r] = X[64 * r + lane]; zv[r] = Z[64 * r + lane]; }
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = TT::mfma_sub(xv[r], zv[r], acc);
            }
            if (J == kb) {
                ft_d4 Ro, Wo;
                const bool ok = ft_potrf16<double>(acc, cnt, c, g, Ro, Wo);
                if (!ok && lane == 0) sfail = 1;
                double* ri = RIt + (size_t)kb * 256;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sW[c * 17 + TT::row(g, r)] = Wo[r];
                    ri[c * 16 + TT::row(g, r)] = Wo[r];
                    tp[TT::row(g, r) * 16 + c] = Ro[r];
                }
            } else if (inreg) {
#pragma unroll
                for (int q = 0; q < FR_OWN; ++q) if (q == slot) own[q] = acc;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) tp[TT::row(g, r) * 16 + c] = acc[r];
            }
        }
        __syncthreads();
#ifdef FW_TIMING
        if (blockIdx.x == 0 && tid == 0) fr_timing[6] += (unsigned long long)wall_clock64() - _ta;
#endif
        if (sfail) return 1;                                                // uniform
        // ---- pass B: R(kb, J) = W P(kb, J)
        {
            double wop[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) wop[r] = sW[TT::row(g, r) * 17 + c];
            slot = 0;
            for (int J = kb + wv; J < NTl; J += NW, ++slot) {
                if (J == kb) continue;
                double* tp = Yt + ((size_t)kb * NTl + J) * 256;
                ft_d4 pv, o = {0, 0, 0, 0};
                if (inreg) {
                    pv = own[0];
#pragma unroll
                    for (int q = 1; q < FR_OWN; ++q) if (q == slot) pv = own[q];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[r] = tp[TT::row(g, r) * 16 + c];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) o = TT::mfma(wop[r], pv[r], o);
#pragma unroll
                for (int r = 0; r < 4; ++r) tp[TT::row(g, r) * 16 + c] = o[r];
            }
        }
        __syncthreads();                                                    // (the tiles of this block row are read by every wave from here on)
    }
    // ---- backward substitution: x_kb = R(kb,kb)^-1 (y_kb - sum_{J>kb} R(kb,J) x_J), y = column nbn of the factor
    for (int i = tid; i < 16 * NTl; i += blockDim.x) xs[i] = 0.0;
    const int yc = nbn & 15, yt = nbn >> 4;
#ifdef FW_TIMING
    const unsigned long long _tb = (unsigned long long)wall_clock64();
#endif
    for (int kb = NTl - 1; kb >= 0; --kb) {
        // (everything this block row reads from memory is requested before the barrier that publishes x of the row below)
        double tv[FR_OWN][4];
        int nown = 0;
        for (int J = kb + 1 + wv; J < NTl && nown < FR_OWN; J += NW, ++nown) {
            const double* tp = Yt + ((size_t)kb * NTl + J) * 256;
#pragma unroll
            for (int q = 0; q < FR_OWN; ++q)
                if (q == nown) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) tv[q][r] = tp[64 * r + lane];
                }
        }
        const double riv = tid < 256 ? RIt[(size_t)kb * 256 + tid] : 0.0;
        const double yv0 = tid < 16 ? Yt[((size_t)kb * NTl + yt) * 256 + tid * 16 + yc] : 0.0;
        __syncthreads();
        double ps[4] = {0.0, 0.0, 0.0, 0.0};
        {
            int q = 0;
            for (int J = kb + 1 + wv; J < NTl; J += NW, ++q) {
                const double xv = xs[16 * J + c];
                if (q < FR_OWN) {
#pragma unroll
                    for (int u = 0; u < FR_OWN; ++u)
                        if (u == q) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) ps[r] = fma(tv[u][r], xv, ps[r]);
                        }
                } else {
                    const double* tp = Yt + ((size_t)kb * NTl + J) * 256;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ps[r] = fma(tp[64 * r + lane], xv, ps[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = ft_row16_sum<double>(ps[r]);
            if (c == 0) part[wv * 16 + 4 * r + g] = v;
        }
        __syncthreads();
        if (tid < 16) {
            double sacc = 0.0;
            for (int q = 0; q < NW; ++q) sacc += part[q * 16 + tid];
            tsh[tid] = yv0 - sacc;
        }
        __syncthreads();
        if (tid < 256) {
            const int row = tid >> 4;
            double v = riv * tsh[c];
            v = ft_row16_sum<double>(v);
            if (c == 0) {
                const int e = 16 * kb + row;
                xs[e] = e < nbn ? v : 0.0;
                if (e < nbn) dnu[e] = v;
            }
        }
    }
    __syncthreads();
#ifdef FW_TIMING
    if (blockIdx.x == 0 && tid == 0) fr_timing[7] += (unsigned long long)wall_clock64() - _tb;
#endif
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------------
// The same factorisation with the tiles IN LDS (the m x m system of the cold-start form, fmpc_ramp_cold): upper tile triangle of
// [M | rhs] packed, tile (I, J), I <= J < NT1, at FR_TIDX(I, J, NT1) * 256 doubles, row-major 16 x 16 (element 64 r + lane is
// accumulator register r of that lane); NTm block rows of M, the rhs in column 0 of tile column NTm = NT1 - 1.  Left-looking per
// block row kb: every wavefront takes the tiles kb + wv, kb + wv + NW, ... of the row, subtracts the products with the block rows
// already done (operands straight from LDS, the next pair requested before the matrix cores take the current one), the owner of
// the diagonal tile factors it (ft_potrf16: R(kb,kb), W = R(kb,kb)^-T), then R(kb,J) = W P(kb,J) from registers; the DIAGONAL
// tiles are updated right-looking in that pass, so a diagonal tile is complete when its block row starts.  Two
// LDS-only barriers per block row.  The diagonal inverses W go to RIt (global, read back one block row ahead in the backward
// substitution R x = y).  Returns 1 if a pivot is not positive.  sh: 16 x 17 + 16 NT1 + 16 NW + 16 doubles.
typedef __attribute__((address_space(3))) double* fr_lds_t;
#define FR_TIDX(I, J, NT1) ((I) * (NT1) - (I) * ((I) - 1) / 2 + ((J) - (I)))
#define FR_LOWN 3                       // tiles of a block row per wavefront kept in registers
__device__ __noinline__ int fr_tile_cholesky_lds(double* tiles_g, int NTm, int mm, double* RIt, double* xout, double* sh_g) {
    typedef FtT<double> TT;
    const fr_lds_t tiles = (fr_lds_t)tiles_g;
    const fr_lds_t sh = (fr_lds_t)sh_g;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, NW = blockDim.x >> 6, c = lane & 15, g = lane >> 4;
    const int NT1 = NTm + 1;
    const fr_lds_t sW = sh;                  // W' of the current diagonal tile, leading dimension 17
    const fr_lds_t xs = sW + 16 * 17;        // x, padded to 16 NT1
    const fr_lds_t part = xs + 16 * NT1;     // [NW][16] partial sums of the backward substitution
    const fr_lds_t tsh = part + 16 * NW;     // [16]
    __shared__ int sfail_l;
    if (tid == 0) sfail_l = 0;
    __syncthreads();
    for (int kb = 0; kb < NTm; ++kb) {
        const int cnt = mm - 16 * kb < 16 ? mm - 16 * kb : 16;             // live rows of this block row
        ft_d4 own[FR_LOWN];
        int slot = 0;
#ifdef FW_TIMING
        const unsigned long long _ta = (unsigned long long)wall_clock64();
#endif
        for (int J = kb + wv; J < NT1; J += NW, ++slot) {
            const fr_lds_t tp = tiles + FR_TIDX(kb, J, NT1) * 256;
            ft_d4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = tp[64 * r + lane];
            if (kb > 0 && J != kb) {                                        // (the diagonal tiles are kept up to date in pass B)
                ft_d4 x, z;
                {
                    const fr_lds_t X = tiles + FR_TIDX(0, kb, NT1) * 256, Z = tiles + FR_TIDX(0, J, NT1) * 256;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { x[r] = X[64 * r + lane]; z[r] = Z[64 * r + lane]; }
                }
                for (int k = 0; k < kb; ++k) {
                    ft_d4 nx = x, nz = z;
                    if (k + 1 < kb) {
                        const fr_lds_t X = tiles + FR_TIDX(k + 1, kb, NT1) * 256, Z = tiles + FR_TIDX(k + 1, J, NT1) * 256;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { nx[r] = X[64 * r + lane]; nz[r] = Z[64 * r + lane]; }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = TT::mfma_sub(x[r], z[r], acc);
                    x = nx; z = nz;
                }
            }
            if (J == kb) {
                ft_d4 Ro, Wo;
#ifdef FW_TIMING
                const unsigned long long _tp = (unsigned long long)wall_clock64();
#endif
                const bool ok = cnt == 16 ? ft_potrf16_ct<double, 16>(acc, c, g, Ro, Wo) : ft_potrf16<double>(acc, cnt, c, g, Ro, Wo);
#ifdef FW_TIMING
                if (blockIdx.x == 0 && lane == 0) fr_timing[7] += (unsigned long long)wall_clock64() - _tp;
#endif
                if (!ok && lane == 0) sfail_l = 1;
                double* ri = RIt + (size_t)kb * 256;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sW[c * 17 + TT::row(g, r)] = Wo[r];
                    ri[c * 16 + TT::row(g, r)] = Wo[r];
                    tp[64 * r + lane] = Ro[r];
                }
            } else {
#pragma unroll
                for (int q = 0; q < FR_LOWN; ++q) if (q == slot) own[q] = acc;
            }
        }
        ft_lds_barrier();
#ifdef FW_TIMING
        if (blockIdx.x == 0 && tid == 0) fr_timing[6] += (unsigned long long)wall_clock64() - _ta;
#endif
        if (sfail_l) return 1;                                              // uniform
        {
            double wop[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) wop[r] = sW[TT::row(g, r) * 17 + c];
            slot = 0;
            for (int J = kb + wv; J < NT1; J += NW, ++slot) {
                if (J == kb) continue;
                const fr_lds_t tp = tiles + FR_TIDX(kb, J, NT1) * 256;
                ft_d4 pv = own[0], o = {0, 0, 0, 0};
#pragma unroll
                for (int q = 1; q < FR_LOWN; ++q) if (q == slot) pv = own[q];
#pragma unroll
                for (int r = 0; r < 4; ++r) o = TT::mfma(wop[r], pv[r], o);
#pragma unroll
                for (int r = 0; r < 4; ++r) tp[64 * r + lane] = o[r];
                if (J < NTm) {
                    // right-looking for the DIAGONAL tiles only: M(J,J) -= R(kb,J)' R(kb,J) now, from the registers -- the owner of the
                    // next diagonal tile then starts its 16-step factorisation at once instead of first summing kb products (they
                    // were the critical path: every other wavefront waits for that tile).  Tile (kb, J) has one owner: no conflict.
                    const fr_lds_t dp = tiles + FR_TIDX(J, J, NT1) * 256;
                    ft_d4 dacc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dacc[r] = dp[64 * r + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) dacc = TT::mfma_sub(o[r], o[r], dacc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dp[64 * r + lane] = dacc[r];
                }
            }
        }
        ft_lds_barrier();                                                   // (the tiles of this block row are read by every wave from here on)
    }
    // ---- backward substitution: x_kb = R(kb,kb)^-1 (y_kb - sum_{J>kb} R(kb,J) x_J), y = column 0 of tile column NTm
    for (int i = tid; i < 16 * NT1; i += blockDim.x) xs[i] = 0.0;
    __syncthreads();                                                        // (also: the W tiles in RIt are visible to every wave)
    // (the inverse diagonal tiles come from memory: all of them requested here -- one round trip, not one per block row)
    constexpr int FR_RIVN = 12;
    double rivs[FR_RIVN];
#pragma unroll
    for (int q = 0; q < FR_RIVN; ++q) rivs[q] = (tid < 256 && q < NTm) ? RIt[(size_t)q * 256 + tid] : 0.0;
    for (int kb = NTm - 1; kb >= 0; --kb) {
        double riv = 0.0;
        if (kb < FR_RIVN) {
#pragma unroll
            for (int q = 0; q < FR_RIVN; ++q) if (q == kb) riv = rivs[q];
        } else if (tid < 256) riv = RIt[(size_t)kb * 256 + tid];
        double ps[4] = {0.0, 0.0, 0.0, 0.0};
        for (int J = kb + 1 + wv; J < NTm; J += NW) {
            const double xv = xs[16 * J + c];
            const fr_lds_t tp = tiles + FR_TIDX(kb, J, NT1) * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r) ps[r] = fma(tp[64 * r + lane], xv, ps[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = ft_row16_sum<double>(ps[r]);
            if (c == 0) part[wv * 16 + 4 * r + g] = v;
        }
        ft_lds_barrier();
        if (tid < 16) {
            double sacc = 0.0;
            for (int q = 0; q < NW; ++q) sacc += part[q * 16 + tid];
            tsh[tid] = tiles[FR_TIDX(kb, NTm, NT1) * 256 + tid * 16] - sacc;
        }
        ft_lds_barrier();
        if (tid < 256) {
            const int row = tid >> 4;
            double v = riv * tsh[c];
            v = ft_row16_sum<double>(v);
            if (c == 0) {
                const int e = 16 * kb + row;
                xs[e] = e < mm ? v : 0.0;
                if (e < mm) xout[e] = v;
            }
        }
        ft_lds_barrier();
    }
    return 0;
}

// NT threads per workgroup: 256 (3 workgroups per CU, throughput) or 512 (one problem spread over twice the waves, latency)
template <int NT>
__global__ void __launch_bounds__(NT, NT == 256 ? 3 : (NT == 512 ? 2 : 1))
fmpc_newton_ramp(FmpcDevModel M, const double* __restrict__ dumin, const double* __restrict__ dumax, int batch,
                 const double* __restrict__ x0, const double* __restrict__ x0p, const double* __restrict__ w,
                 const double* __restrict__ uprev, const double* zinit, const double* __restrict__ nu0,
                 int max_iter, double kbar, double* zout, double* __restrict__ nuout, int* __restrict__ status,
                 int* __restrict__ iters, double* __restrict__ step, int step_ld, double* __restrict__ ws,
                 size_t ws_stride, int it0) {
    // it0 = 1: CONTINUATION behind fmpc_ramp_cold, which has taken the first Newton step of every problem: zinit (= zout) and nu0
    // hold the iterate after that step, status / iters / step[0] its record; this launch runs the iterations 1 .. max_iter - 1 with
    // their exit tests (inf_newton_solver.m:19-22) and adds to the record.  A problem whose first step ended in an error, or that
    // left before stepping, is skipped.
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NW = NT / 64;
    const int n = M.n, m = M.m, T = M.T, nb = M.nb;
    const int s = n + m, Nz = T * s, nbn = nb * n, ldt = n + 1, tsz = n * ldt;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int ntile = (n + 15) >> 4;
    const bool var2 = M.var2 != 0;

    // ---- LDS carve
    double* sBt = lds;                    // m*n   Bt[c*n + r] = B[r][c]
    double* tA = sBt + (size_t)m * n;     // n*ldt: the diagonal block being factored
    double* tB = tA + tsz;                // n*ldt: its inverse factor
    double* sv = tB + tsz;                // n     vector of the backward substitution
    double* srs = sv + n;                 // n     1/sqrt(pivot)
    double* sred = srs + n;               // NW x 64 partial sums of the backward substitution
    double* red = sred + NW * 64;   // NW
    double* sCh = red + 16;         // scratch of fr_tile_cholesky: 16 x 17 + 16 NTl + 16 NW + 16

    for (int i = tid; i < m * n; i += NT) sBt[i] = M.Bt[i];

    const FrWsLayout L = fr_ws_layout(n, m, T, nb);
    double* wsp = ws + (size_t)blockIdx.x * ws_stride;
    double* b = wsp + L.b;     double* nu = wsp + L.nu;   double* hs = wsp + L.hs;   double* er = wsp + L.er;
    double* gr = wsp + L.gr;   double* dg = wsp + L.dg;   double* lo = wsp + L.lo;   double* rdu = wsp + L.rdu;
    double* rdx = wsp + L.rdx; double* phx = wsp + L.phx; double* phu = wsp + L.phu; double* rp = wsp + L.rp;
    double* y = wsp + L.y;     double* dnu = wsp + L.dnu; double* G = wsp + L.G;     double* Yd = wsp + L.Y;
    double* Wg = wsp + L.W;

    for (int p = blockIdx.x; p < batch; p += gridDim.x) {
        double* zp = zout + (size_t)p * Nz;
        const double* x0v = x0 + (size_t)p * n;
        const double* x0pv = x0p ? x0p + (size_t)p * n : nullptr;
        const double* upv = uprev + (size_t)p * m;
        __syncthreads();
        if (it0 > 0 && (status[p] < 0 || iters[p] < it0)) continue;      // (uniform: every thread reads the same two words)
        // ================= P0: start point, nu, b  (fast_mpc_init.m:12-27, fast_mpc_eq_const.m)
        for (int idx = tid; idx < Nz; idx += NT) {
            const int e = idx % s;
            zp[idx] = zinit ? zinit[(size_t)p * Nz + idx] : (e < m ? M.umid[e] : M.xmid[e - m]);
        }
        for (int idx = tid; idx < nbn; idx += NT) {
            nu[idx] = nu0 ? nu0[(size_t)p * nbn + idx] : 0.0;
            const int i = idx / n, r = idx - i * n;
            double v = (i < T && w) ? w[(size_t)p * T * n + idx] : 0.0;
            if (i == 0) {
                for (int c = 0; c < n; ++c) v += M.A1t[c * n + r] * x0v[c];
                if (var2 && x0pv)
                    for (int c = 0; c < n; ++c) v += M.A2t[c * n + r] * x0pv[c];
            } else if (i == 1 && i < T && var2) {
                for (int c = 0; c < n; ++c) v += M.A2t[c * n + r] * x0v[c];
            }
            if (i == T) v = M.xf[r];
            b[idx] = v;
        }
        if (step)
            for (int idx = tid + it0; idx < step_ld; idx += NT) step[(size_t)p * step_ld + idx] = -1.0;
        __syncthreads();

        int st = it0 > 0 ? status[p] : FMPC_OK, nsteps = it0;
        for (int it = it0; it < max_iter; ++it) {
            // ================= P1: slacks and residuals
            FR_T0();
            double acc_d = 0.0, acc_p = 0.0;
            for (int idx = tid; idx < T * m; idx += NT) {       // ramp terms of stage j (needed by j and j-1)
                const int j = idx / m, c = idx - j * m;
                const double dl = zp[j * s + c] - (j == 0 ? upv[c] : zp[(j - 1) * s + c]);
                const double rpv = 1.0 / (dumax[c] - dl), rmv = 1.0 / (dl - dumin[c]);
                er[idx] = kbar * (rpv * rpv + rmv * rmv);
                gr[idx] = kbar * (rpv - rmv);
            }
            __syncthreads();
            for (int idx = tid; idx < T * m; idx += NT) {
                const int j = idx / m, c = idx - j * m;
                const double u = zp[j * s + c];
                const double dp = 1.0 / (M.umax[c] - u), dm = 1.0 / (u - M.umin[c]);
                const double hb = kbar * (dp * dp + dm * dm);
                const bool nx = j + 1 < T;
                double dot = 0.0;
                const double* bt = sBt + c * n;
                const double* nj = nu + j * n;
                for (int r = 0; r < n; ++r) dot += bt[r] * nj[r];
                const double rd = M.R2[c] * u + M.rl[c] + kbar * (dp - dm) + gr[idx] - (nx ? gr[idx + m] : 0.0) - dot;
                hs[idx] = hb + er[idx] + (nx ? er[idx + m] : 0.0);          // diagonal of k P'DP
                rdu[idx] = rd;
                acc_d += rd * rd;
            }
            for (int idx = tid; idx < T * n; idx += NT) {
                const int jj = idx / n, r = idx - jj * n, j = jj + 1;   // x_j, j = 1..T
                const double x = zp[jj * s + m + r];
                double v = (j == T ? M.Qf2[r] * x + M.qfl[r] : M.Q2[r] * x + M.ql[r]) + nu[jj * n + r];
                if (j < T) {
                    const double* nj = nu + j * n;
                    for (int c = 0; c < n; ++c) v -= M.A1[c * n + r] * nj[c];
                }
                if (var2 && j + 1 < T) {
                    const double* nj = nu + (j + 1) * n;
                    for (int c = 0; c < n; ++c) v -= M.A2[c * n + r] * nj[c];
                }
                if (j == T && M.has_xf) v += nu[T * n + r];
                rdx[idx] = v;
                phx[idx] = v / (j == T ? M.Qf2[r] : M.Q2[r]);           // Phi^-1 r_d on x_j
                acc_d += v * v;
            }
            for (int idx = tid; idx < nbn; idx += NT) {
                const int i = idx / n, r = idx - i * n;
                double v;
                if (i < T) {
                    v = zp[i * s + m + r] - b[idx];
                    const double* ui = zp + i * s;
                    for (int c = 0; c < m; ++c) v -= sBt[c * n + r] * ui[c];
                    if (i >= 1) {
                        const double* xi = zp + (i - 1) * s + m;
                        for (int c = 0; c < n; ++c) v -= M.A1t[c * n + r] * xi[c];
                    }
                    if (var2 && i >= 2) {
                        const double* xi = zp + (i - 2) * s + m;
                        for (int c = 0; c < n; ++c) v -= M.A2t[c * n + r] * xi[c];
                    }
                } else {
                    v = zp[(T - 1) * s + m + r] - b[idx];
                }
                rp[idx] = v;
                acc_p += v * v;
            }
            const double rp2 = fr_block_sum<NT>(acc_p, red);
            const double rho2 = fr_block_sum<NT>(acc_d, red) + rp2;
            // early exit, tested before the step (inf_newton_solver.m:19-22)
            if (sqrt(rho2) <= 1e-6 && sqrt(rp2) <= 1e-8) break;
            FR_TICK(0);

            // ================= P2: per actuator LDL' of the tridiagonal u-part of Phi, Phi_u^-1 r_d, explicit inverse
            int bad = 0;
            for (int c = tid; c < m; c += NT) {
                double lprev = 0.0, oprev = 0.0;
                for (int j = 0; j < T; ++j) {
                    double d = M.R2[c] + hs[j * m + c];
                    if (j > 0) d -= lprev * oprev;
                    if (!(d > 0.0) || isinf(d)) bad = 1;
                    dg[j * m + c] = d;
                    if (j + 1 < T) {
                        oprev = -er[(j + 1) * m + c];
                        lprev = oprev / d;
                        lo[j * m + c] = lprev;
                    }
                }
                for (int j = 0; j < T; ++j) phu[j * m + c] = rdu[j * m + c];
                fr_tri_solve(dg, lo, phu, T, m, c);
                // inverse, pairs (i <= j) at index i*T - i(i-1)/2 + (j - i):  inv[j][j] = 1/d_j + l_j^2 inv[j+1][j+1] ,
                // inv[j][k] = -l_j inv[j+1][k] (k > j) = (-l_j)(-l_{j+1}) ... (-l_{k-1}) inv[k][k]: the diagonal as one chain carried in
                // a register, then every column k upwards from its diagonal entry -- no load of a value this thread has just stored
                {
                    double dprev = 0.0;
                    for (int j = T - 1; j >= 0; --j) {
                        const size_t rowj = (size_t)j * T - (size_t)j * (j - 1) / 2;
                        const double lj = j + 1 < T ? lo[j * m + c] : 0.0;
                        const double dv = 1.0 / dg[j * m + c] + lj * lj * dprev;
                        G[rowj * m + c] = dv;
                        double v = dv;
                        for (int i = j - 1; i >= 0; --i) {            // column j: rows i < j
                            v *= -lo[i * m + c];
                            G[((size_t)i * T - (size_t)i * (i - 1) / 2 + (j - i)) * m + c] = v;
                        }
                        dprev = dv;
                    }
                }
            }
            const double badsum = fr_block_sum<NT>((double)bad, red);
            if (badsum > 0.0) { st = FMPC_E_NOT_PD_PHI; break; }
            // rhs_i = r_p,i - (C Phi^-1 r_d)_i   (into y)
            for (int idx = tid; idx < nbn; idx += NT) {
                const int i = idx / n, r = idx - i * n;
                double cv;
                if (i < T) {
                    cv = phx[i * n + r];
                    const double* pu = phu + i * m;
                    for (int c = 0; c < m; ++c) cv -= sBt[c * n + r] * pu[c];
                    if (i >= 1) {
                        const double* px = phx + (i - 1) * n;
                        for (int c = 0; c < n; ++c) cv -= M.A1t[c * n + r] * px[c];
                    }
                    if (var2 && i >= 2) {
                        const double* px = phx + (i - 2) * n;
                        for (int c = 0; c < n; ++c) cv -= M.A2t[c * n + r] * px[c];
                    }
                } else {
                    cv = phx[(T - 1) * n + r];
                }
                y[idx] = rp[idx] - cv;
            }
            __syncthreads();
            FR_TICK(1);

            // ================= P3: [Y | rhs] into the workspace as 16 x 16 tiles (upper tile triangle; fr_tile_cholesky).
            // Y_IJ = Yx_IJ + B diag(g^{JI}) B' on the matrix cores: a wave per 16 x 16 piece of a block, k = 4 actuators per
            // MFMA; A operand B[a][c] g_c (row a = lane % 16, c = 4 ks + lane / 16), B operand B[b][c]; result register r of
            // lane (lk, li) is element (4 r + lk, li) of the piece.  The blocks (n x n) and the tiles (16 x 16) do not line up:
            // every element goes to its tile on its own; a diagonal tile receives both halves.
            const int NTl = (nbn + 1 + 15) >> 4;
            for (size_t idx = tid; idx < (size_t)NTl * NTl * 256; idx += NT) Yd[idx] = 0.0;
            __syncthreads();
            auto put = [&](int gr_, int gc_, double v) {                    // element (gr_, gc_), tile row <= tile column
                Yd[((size_t)(gr_ >> 4) * NTl + (gc_ >> 4)) * 256 + (gr_ & 15) * 16 + (gc_ & 15)] = v;
            };
            for (int idx = tid; idx < nbn; idx += NT) put(idx, nbn, y[idx]);
            // (no workgroup barrier in this phase: every wave walks its own blocks)
            if (ntile <= 2) {
                // n <= 32: a wave takes a whole block (I, J) with its (up to) four 16 x 16 pieces in registers: per k-step
                // (4 actuators) two LDS reads and two multiplications by g feed four products.
                const int nblk = nb * (nb + 1) / 2;
                const int ra0 = li < n ? li : n - 1, ra1 = 16 + li < n ? 16 + li : n - 1;
                for (int blk = wv; blk < nblk; blk += NW) {
                    int I = (int)((sqrt(8.0 * blk + 1.0) - 1.0) * 0.5);          // blk = I (I + 1) / 2 + J , J <= I
                    while (I * (I + 1) / 2 > blk) --I;
                    while ((I + 1) * (I + 2) / 2 <= blk) ++I;
                    const int J = blk - I * (I + 1) / 2;
                    const bool hasu = I < T;                         // (J <= I): both stages carry u
                    const double* Yc = nullptr; bool tr = false;
                    if (I == J) Yc = M.Yblk + (size_t)M.idxD[I] * n * n;
                    else if (I == J + 1 && M.idx1[J] >= 0) { Yc = M.Yblk + (size_t)M.idx1[J] * n * n; tr = true; }
                    else if (I == J + 2 && M.idx2[J] >= 0) { Yc = M.Yblk + (size_t)M.idx2[J] * n * n; tr = true; }
                    // everything the block reads from memory first: the constant part of its elements, then g
                    double yc4[2][2][4];
                    const double* ysrc = Yc ? Yc : M.Yblk;             // (no constant part: any valid address, times zero)
                    const double ycf = Yc ? 1.0 : 0.0;
#pragma unroll
                    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int arow = 16 * ta + 4 * r + lk, bcol = 16 * tb + li;
                                const int ar = arow < n ? arow : n - 1, bc = bcol < n ? bcol : n - 1;
                                yc4[ta][tb][r] = ysrc[tr ? bc * n + ar : ar * n + bc];
                            }
                    d4 acc[2][2];
#pragma unroll
                    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                        for (int tb = 0; tb < 2; ++tb) acc[ta][tb] = (d4){0, 0, 0, 0};
                    if (hasu) {
                        const double* gv = G + ((size_t)J * T - (size_t)J * (J - 1) / 2 + (I - J)) * m;
                        for (int c0 = 0; c0 < m; c0 += 144) {
                            double gq[36];
#pragma unroll
                            for (int q = 0; q < 36; ++q) { const int cq = c0 + 4 * q + lk; gq[q] = gv[cq < m ? cq : m - 1]; }
#pragma unroll
                            for (int q = 0; q < 36; ++q) {             // (straight-line: past m the A operands are zero)
                                const int cq = c0 + 4 * q + lk;
                                const int cc = cq < m ? cq : m - 1;
                                const double gm = cq < m ? gq[q] : 0.0;
                                const double x0 = sBt[cc * n + ra0], x1 = sBt[cc * n + ra1];
                                const double z0 = x0 * gm, z1 = x1 * gm;
                                acc[0][0] = MFMA64(z0, x0, acc[0][0]);
                                acc[1][0] = MFMA64(z1, x0, acc[1][0]);
                                acc[1][1] = MFMA64(z1, x1, acc[1][1]);
                                acc[0][1] = MFMA64(z0, x1, acc[0][1]);
                            }
                        }
                    }
#pragma unroll
                    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int arow = 16 * ta + 4 * r + lk, bcol = 16 * tb + li;
                                if (arow < n && bcol < n && (I != J || arow >= bcol)) {
                                    const double v = acc[ta][tb][r] + ycf * yc4[ta][tb][r];
                                    const int gr_ = I * n + arow, gc_ = J * n + bcol;      // gr_ >= gc_: the element of the lower triangle
                                    put(gc_, gr_, v);
                                    if ((gr_ >> 4) == (gc_ >> 4) && gr_ != gc_) put(gr_, gc_, v);
                                }
                            }
                }
            } else {
                const int nblk = nb * (nb + 1) / 2, tpb = ntile * ntile;
                for (int task = wv; task < nblk * tpb; task += NW) {
                    const int blk = task / tpb, tp = task - blk * tpb;
                    int I = (int)((sqrt(8.0 * blk + 1.0) - 1.0) * 0.5);          // blk = I (I + 1) / 2 + J , J <= I
                    while (I * (I + 1) / 2 > blk) --I;
                    while ((I + 1) * (I + 2) / 2 <= blk) ++I;
                    const int J = blk - I * (I + 1) / 2;
                    const int ta = tp / ntile, tb = tp - ta * ntile;
                    if (I == J && ta < tb) continue;                  // diagonal blocks: the lower pieces, mirrored below
                    const bool hasu = I < T;                         // (J <= I): both stages carry u
                    const double* Yc = nullptr; bool tr = false;
                    if (I == J) Yc = M.Yblk + (size_t)M.idxD[I] * n * n;
                    else if (I == J + 1 && M.idx1[J] >= 0) { Yc = M.Yblk + (size_t)M.idx1[J] * n * n; tr = true; }
                    else if (I == J + 2 && M.idx2[J] >= 0) { Yc = M.Yblk + (size_t)M.idx2[J] * n * n; tr = true; }
                    d4 acc = {0, 0, 0, 0};
                    const int bcol = 16 * tb + li;
                    // everything the task reads from memory first: the constant part of its four elements, then g
                    double yc4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int arow = 16 * ta + 4 * r + lk;
                        const int ar = arow < n ? arow : n - 1, bc = bcol < n ? bcol : n - 1;
                        const double* ysrc = Yc ? Yc : M.Yblk;                 // (no constant part: any valid address, times zero)
                        yc4[r] = ysrc[tr ? bc * n + ar : ar * n + bc];
                    }
                    const double ycf = Yc ? 1.0 : 0.0;
                    if (hasu) {
                        const double* gv = G + ((size_t)J * T - (size_t)J * (J - 1) / 2 + (I - J)) * m;
                        const int ra = 16 * ta + li < n ? 16 * ta + li : n - 1, rb = 16 * tb + li < n ? 16 * tb + li : n - 1;
                        // 36 k-steps (144 actuators) at a time: their g values are requested before the first product
                        for (int c0 = 0; c0 < m; c0 += 144) {
                            double gq[36];
#pragma unroll
                            for (int q = 0; q < 36; ++q) { const int cq = c0 + 4 * q + lk; gq[q] = gv[cq < m ? cq : m - 1]; }
#pragma unroll
                            for (int q = 0; q < 36; ++q) {             // (straight-line: past m the A operand is zero)
                                const int cq = c0 + 4 * q + lk;
                                const int cc = cq < m ? cq : m - 1;
                                const double msk = cq < m ? 1.0 : 0.0;
                                acc = MFMA64(sBt[cc * n + ra] * (gq[q] * msk), sBt[cc * n + rb], acc);
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int arow = 16 * ta + 4 * r + lk;
                        if (arow < n && bcol < n && (I != J || arow >= bcol)) {
                            const double v = acc[r] + ycf * yc4[r];
                            const int gr_ = I * n + arow, gc_ = J * n + bcol;      // gr_ >= gc_: the element of the lower triangle
                            put(gc_, gr_, v);
                            if ((gr_ >> 4) == (gc_ >> 4) && gr_ != gc_) put(gr_, gc_, v);
                        }
                    }
                }
            }
            __syncthreads();
            FR_TICK(2);

            // ================= P4: Cholesky of Y on 16 x 16 tiles (forward substitution in the rhs column), backward substitution
            if (fr_tile_cholesky(Yd, NTl, nbn, Wg, dnu, sCh)) { st = FMPC_E_NOT_PD_SCHUR; break; }
            FR_TICK(3);
            FR_TICK(4);
            // ================= P5: d_z, line-search scalars, update
            for (int idx = tid; idx < T * m; idx += NT) {           // rhs of Phi_u d_u = B' d_nu_j - r_d,u
                const int j = idx / m, c = idx - j * m;
                double dot = 0.0;
                const double* bt = sBt + c * n;
                const double* dj = dnu + j * n;
                for (int r = 0; r < n; ++r) dot += bt[r] * dj[r];
                phu[idx] = dot - rdu[idx];
            }
            __syncthreads();
            for (int c = tid; c < m; c += NT) fr_tri_solve(dg, lo, phu, T, m, c);     // phu = d_u
            __syncthreads();
            double be = 0.0, e2 = 0.0;
            for (int idx = tid; idx < T * m; idx += NT) {
                const int j = idx / m;
                double e = hs[idx] * phu[idx];                              // k P'DP d_z on u_j
                if (j > 0) e -= er[idx] * phu[idx - m];
                if (j + 1 < T) e -= er[idx + m] * phu[idx + m];
                be += rdu[idx] * e;
                e2 += e * e;
            }
            for (int idx = tid; idx < T * n; idx += NT) {
                const int jj = idx / n, r = idx - jj * n, j = jj + 1;
                double v = -rdx[idx] - dnu[jj * n + r];
                if (j < T) {
                    const double* dj = dnu + j * n;
                    for (int c = 0; c < n; ++c) v += M.A1[c * n + r] * dj[c];
                }
                if (var2 && j + 1 < T) {
                    const double* dj = dnu + (j + 1) * n;
                    for (int c = 0; c < n; ++c) v += M.A2[c * n + r] * dj[c];
                }
                if (j == T && M.has_xf) v -= dnu[T * n + r];
                rdx[idx] = v / (j == T ? M.Qf2[r] : M.Q2[r]);               // reuse as d_x
            }
            const double beta_e = fr_block_sum<NT>(be, red);
            const double eps2 = fr_block_sum<NT>(e2, red);
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
                    if (++halv >= FR_MAX_HALVINGS) { t = 0.0; st = FMPC_W_LINESEARCH; break; }
                }
            }
            for (int idx = tid; idx < Nz; idx += NT) {
                const int j = idx / s, e = idx - j * s;
                zp[idx] += t * (e < m ? phu[j * m + e] : rdx[j * n + e - m]);
            }
            for (int idx = tid; idx < nbn; idx += NT) nu[idx] += t * dnu[idx];
            if (step && tid == 0 && it < step_ld) step[(size_t)p * step_ld + it] = t;
            ++nsteps;
            __syncthreads();
            FR_TICK(5);
        }
        if (nuout)
            for (int idx = tid; idx < nbn; idx += NT) nuout[(size_t)p * nbn + idx] = nu[idx];
        if (tid == 0) {
            if (status) status[p] = st;
            if (iters) iters[p] = nsteps;
        }
    }
}

// =====================================================================================================================
// Cold-start Newton step WITH the ramp-rate rows in its Woodbury form (fmpc_rampcold.h, fmpc_host.h: FmpcRampColdOut).
// From the mid-box start only the ramp rows of stage 0 (u_0 - u_prev) depend on the problem: Phi = Phibar + E diag(delta) E', so
//     [d_z ; nu+] = Kbar^-1 (-[gbar + E s ; r_p]) ,   s = rho + q ,   (diag(1/delta) + G) q = y_u0 ,
//     y_u0 = y0c - G rho + Xi_u0 bhat                       (the u_0 rows of Kbar^-1 f, f = -[gbar + E rho ; r_p])
// with Kbar^-1 applied through the constant operators Phibar^-1 (per actuator its T x T inverse Gf), Ybar^-1 (explicit, nb n
// square) and the sparse C: ONE m x m Cholesky factorisation per problem (fr_tile_cholesky on the matrix cores) where the
// general path factors the dense (T n)^2 Schur complement and forms its T (T + 1) / 2 blocks B diag(g) B' first -- 0.8 instead
// of 18.7 MFLOP per problem at (27, 144, 10).  Same step as fmpc_newton_ramp takes from the cold start (inf_newton_solver.m:
// 10-41 with the rows of VAR_1/fast_mpc_ineq_const.m:58-76), different rounding; exit test, line search, status codes and the
// step record as there.  One workgroup per problem in flight.
// dst[0 .. cnt) = src[0 .. cnt) (global -> LDS) with eight loads per thread in flight (a load + store loop waits per element)
template <int NT>
__device__ __forceinline__ void fr_copy_in(double* dst, const double* __restrict__ src, int cnt, int tid) {
    for (int i0 = tid; i0 < cnt; i0 += 8 * NT) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * NT; v[u] = src[i < cnt ? i : i0]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * NT; if (i < cnt) dst[i] = v[u]; }
    }
}
// part[g][r] = sign * sum over the columns c = g, g + G, ... < ncols of At[c ld + r] x[c], for r < rows and the G = min(8, NT / pairs)
// column groups (pairs = ceil(rows / 2)): a thread owns a PAIR of rows (one 16-byte load per column: ld even, At 16-byte aligned)
// of one group, eight loads in flight, no cross-lane reduction; the caller adds the G partial sums per row.  Returns G.
// (Measured on the 270 x 270 product of the ramp cold form: a wavefront per row with shuffles to sum the lanes took 17 us however
// the loads were arranged -- the eight reductions per row block were the time -- against ~5 us this way.)
template <int NT>
__device__ __forceinline__ int fr_matvec_pairs(const double* __restrict__ At, int ld, int rows, int ncols, const double* x, double sign,
                                               double* part, int pld, int tid) {
    typedef double d2r __attribute__((ext_vector_type(2)));
    const int pairs = (rows + 1) >> 1;
    int G = NT / pairs; if (G > 8) G = 8; if (G < 1) G = 1;
    for (int t = tid; t < G * pairs; t += NT) {                       // (one pass unless pairs > NT)
        const int g = t / pairs, r = 2 * (t - g * pairs);
        double ax = 0.0, ay = 0.0;
        for (int c0 = g; c0 < ncols; c0 += 8 * G) {
            d2r v[8]; double xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u * G, cc = c < ncols ? c : c0;
                v[u] = *(const d2r*)(At + (size_t)cc * ld + r);
                xv[u] = c < ncols ? x[cc] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { ax = fma(v[u].x, xv[u], ax); ay = fma(v[u].y, xv[u], ay); }
        }
        part[(size_t)g * pld + r] = sign * ax;
        if (r + 1 < pld) part[(size_t)g * pld + r + 1] = sign * ay;
    }
    return G;
}
// sum_c a[c * as] * x[c] over LDS operands; NC > 0: the length at compile time (straight-line: every read requested up front)
template <int NC>
__device__ __forceinline__ double fr_ldsdot_n(const double* a, int as, const double* x, int cnt) {
    if (NC > 0) {
        double av[NC > 0 ? NC : 1], xv[NC > 0 ? NC : 1];
#pragma unroll
        for (int c = 0; c < NC; ++c) { av[c] = a[c * as]; xv[c] = x[c]; }
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if ((c & 3) == 0) a0 = fma(av[c], xv[c], a0); else if ((c & 3) == 1) a1 = fma(av[c], xv[c], a1);
            else if ((c & 3) == 2) a2 = fma(av[c], xv[c], a2); else a3 = fma(av[c], xv[c], a3);
        }
        return (a0 + a1) + (a2 + a3);
    }
    double a0 = 0.0, a1 = 0.0;
    int c = 0;
    for (; c + 1 < cnt; c += 2) { a0 += a[c * as] * x[c]; a1 += a[(c + 1) * as] * x[c + 1]; }
    if (c < cnt) a0 += a[c * as] * x[c];
    return a0 + a1;
}
// sum_c a[c * as] * x[c] over LDS operands, two accumulators
__device__ __forceinline__ double fr_ldsdot(const double* a, int as, const double* x, int cnt) {
    double a0 = 0.0, a1 = 0.0;
    int c = 0;
    for (; c + 1 < cnt; c += 2) { a0 += a[c * as] * x[c]; a1 += a[(c + 1) * as] * x[c + 1]; }
    if (c < cnt) a0 += a[c * as] * x[c];
    return a0 + a1;
}
// sum_c A[c * lda + r] * x[c], c < cnt, with four independent loads in flight (a loop of one load + one fma per iteration is one
// memory round trip per term)
__device__ __forceinline__ double fr_coldot(const double* __restrict__ A, int lda, int r, const double* x, int cnt, int xs = 1) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int c = 0;
    for (; c + 3 < cnt; c += 4) {
        const double v0 = A[(size_t)c * lda + r], v1 = A[(size_t)(c + 1) * lda + r], v2 = A[(size_t)(c + 2) * lda + r], v3 = A[(size_t)(c + 3) * lda + r];
        a0 += v0 * x[c * xs]; a1 += v1 * x[(c + 1) * xs]; a2 += v2 * x[(c + 2) * xs]; a3 += v3 * x[(c + 3) * xs];
    }
    for (; c < cnt; ++c) a0 += A[(size_t)c * lda + r] * x[c * xs];
    return (a0 + a1) + (a2 + a3);
}

// out[i][j] = sum_k X[i][k] Z[j][k] on the fp64 matrix cores, i < rows_x (horizon stages), j < rows_z, k < K:
//   X  in LDS, row-major with leading dimension ldx (rows beyond rows_x and k beyond K read as zero)
//   Z  as operand images in memory, [ceil(rows_z / 16)][ks][64] (fmpc_host_mfma_images of the rows_z x K matrix): 512-byte loads
// One (16 x 16) output tile per wavefront and turn, K in chunks of KCH k-steps whose operands are all requested before the first
// product.  emit(i, j, value) is called once per live element by the lane that holds it.
template <int KCH, class EMIT>
__device__ __forceinline__ void fr_stage_product(const double* X, int ldx, int rows_x, const double* __restrict__ Zimg, int rows_z, int K,
                                                 int wv, int NW, int lane, EMIT emit) {
    const int ks = (K + 3) >> 2, ti = (rows_x + 15) >> 4, tj = (rows_z + 15) >> 4, li = lane & 15, lk = lane >> 4;
    for (int task = wv; task < ti * tj; task += NW) {
        const int It = task / tj, Jt = task - It * tj;
        const double* zi = Zimg + (size_t)Jt * ks * 64 + lane;
        const int xi = 16 * It + li;
        const double* xr = X + (size_t)(xi < rows_x ? xi : 0) * ldx;
        const double xm = xi < rows_x ? 1.0 : 0.0;
        d4 acc = {0, 0, 0, 0};
        for (int q0 = 0; q0 < ks; q0 += KCH) {
            double zv[KCH], xv[KCH];
#pragma unroll
            for (int u = 0; u < KCH; ++u) {
                const int q = q0 + u < ks ? q0 + u : ks - 1, kk = 4 * q + lk;
                zv[u] = zi[(size_t)q * 64];
                xv[u] = (q0 + u < ks && kk < K) ? xr[kk < K ? kk : 0] * xm : 0.0;
            }
#pragma unroll
            for (int u = 0; u < KCH; ++u) acc = MFMA64(xv[u], zv[u], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * It + lk + 4 * r, j = 16 * Jt + li;
            if (i < rows_x && j < rows_z) emit(i, j, acc[r]);
        }
    }
}

// NC: n at compile time (27: the AO size -- the n-long dot products over LDS operands are then straight-line code, their reads
// requested together) or 0 = any n at run time.
template <int NT, int NC>
__global__ void __launch_bounds__(NT, NT == 512 ? 2 : 1) fmpc_ramp_cold(FrColdParams P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NW = NT / 64;
    const FmpcDevModel& M = P.M;
    const int n = NC ? NC : M.n, m = M.m, T = M.T, nb = M.nb;
    const int s = n + m, Nz = T * s, nbn = nb * n, Tm = T * m, Tn = T * n;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool var2 = M.var2 != 0;
    const double kbar = P.kbar;
    const int NTm = (m + 15) >> 4, NT1 = NTm + 1, ntiles = FR_TIDX(NTm - 1, NT1 - 1, NT1) + 1;
    // ---- LDS carve: what lives through the whole step, then a region used in turn by [phi | kappa] and by the tiles of the
    //      m x m factorisation.  B is not staged: the three products with it run on the matrix cores from operand images (L2)
    double* snu0 = lds;                   // nbn
    double* sbh = snu0 + nbn;             // nbn    bhat (data part of b), zero on the terminal rows
    double* sbe = sbh + nbn;              // nbn    beta
    double* snup = sbe + nbn;             // nbn    nu+
    double* srho = snup + nbn;            // m
    double* sdel = srho + m;              // m
    double* ss = sdel + m;                // m      y_u0, then q (solution), then s = rho + q
    double* srdu = ss + m;                // T m    r_d on the u entries
    double* red = srdu + Tm;              // 16
    double* spart = red + 16;             // NT     partial sums of the m-row products
    double* sCh = spart + NT;             // scratch of fr_tile_cholesky_lds: 16 x 17 + 16 NT1 + 16 NW + 16
    double* sA1 = sCh + (16 * 17 + 16 * NT1 + 16 * NW + 16);    // n n  A1 row-major (and A2 behind it): read by every problem, three times each
    double* sA2 = sA1 + (size_t)n * n;
    double* uni = sA2 + (var2 ? (size_t)n * n : 0);
    uni = (double*)(((size_t)uni + 15) & ~(size_t)15);
    double* sphi = uni;                   // T m    v = g0 s, then phi_u, then d_u
    double* skap = sphi + Tm;             // T m    kappa = B' nu+_j
    double* stl = uni;                    // ntiles x 256: upper tile triangle of [M | y_u0]
    double* Wg = P.ws + (size_t)blockIdx.x * P.ws_stride;      // inverse diagonal tiles (global: read back one block row ahead)
    for (int i = tid; i < n * n; i += NT) { sA1[i] = M.A1[i]; if (var2) sA2[i] = M.A2[i]; }

    for (int p = blockIdx.x; p < P.batch; p += gridDim.x) {
        const double* x0v = P.x0 + (size_t)p * n;
        const double* x0pv = P.x0p ? P.x0p + (size_t)p * n : nullptr;
        const double* upv = P.uprev + (size_t)p * m;
        __syncthreads();
        FR_T0();
        if (tid < 2 * n) spart[tid] = tid < n ? x0v[tid] : (x0pv ? x0pv[tid - n] : 0.0);      // x0, x0_pre through LDS
        __syncthreads();
        const double* sx0 = spart;
        // ================= P0: delta, rho of the stage-0 ramp rows; bhat; nu0
        int bad = 0;
        for (int c = tid; c < m; c += NT) {
            const double dl = M.umid[c] - upv[c];
            const double a = 1.0 / (P.dumax[c] - dl), b = 1.0 / (dl - P.dumin[c]);
            const double de = kbar * (a * a + b * b);
            sdel[c] = de; srho[c] = kbar * (a - b);
            if (!(de > 0.0) || isinf(de)) bad = 1;              // (the general path finds the same entry in its first pivot)
        }
        double acc_p = 0.0;
        for (int idx = tid; idx < nbn; idx += NT) {
            snu0[idx] = P.nu0 ? P.nu0[(size_t)p * nbn + idx] : 0.0;
            const int i = idx / n, r = idx - i * n;
            double v = (i < T && P.w) ? P.w[(size_t)p * Tn + idx] : 0.0;
            if (i == 0) {
                v += fr_ldsdot_n<NC>(sA1 + r * n, 1, sx0, n);
                if (var2 && x0pv) v += fr_ldsdot_n<NC>(sA2 + r * n, 1, sx0 + n, n);
            } else if (i == 1 && i < T && var2) {
                v += fr_ldsdot_n<NC>(sA2 + r * n, 1, sx0, n);
            }
            if (i >= T) v = 0.0;                                 // (x_T = xf is part of cpb)
            sbh[idx] = v;
            const double rp = P.cpb[idx] - v;                    // r_p = C z0 - b
            acc_p += rp * rp;
        }
        if (P.step)
            for (int idx = tid; idx < P.step_ld; idx += NT) P.step[(size_t)p * P.step_ld + idx] = -1.0;
        __syncthreads();
        // ================= P1: r_d = gbar + E rho + C' nu0 (u entries kept for the line search), exit test
        double acc_d = 0.0;
        fr_stage_product<8>(snu0, n, T, P.imgBk, m, n, wv, NW, lane, [&](int j, int c, double dot) {     // B' nu0_j on the matrix cores
            const int idx = j * m + c;
            const double rd = P.gbar_u[idx] + (j == 0 ? srho[c] : 0.0) - dot;
            srdu[idx] = rd;
            acc_d += rd * rd;
        });
        for (int idx = tid; idx < Tn; idx += NT) {
            const int jj = idx / n, r = idx - jj * n, j = jj + 1;   // x_j, j = 1..T
            double v = P.gbar_x[idx] + snu0[jj * n + r];
            if (j < T) v -= fr_ldsdot_n<NC>(sA1 + r, n, snu0 + j * n, n);
            if (var2 && j + 1 < T) v -= fr_ldsdot_n<NC>(sA2 + r, n, snu0 + (j + 1) * n, n);
            if (j == T && M.has_xf) v += snu0[T * n + r];
            acc_d += v * v;
        }
        const double rp2 = fr_block_sum<NT>(acc_p, red);
        const double rho2 = fr_block_sum<NT>(acc_d, red) + rp2;
        const double badsum = fr_block_sum<NT>((double)bad, red);
        int st = FMPC_OK, nsteps = 0;
        double t = 1.0;
        bool moved = false;
        FR_TICK(0);
        if (sqrt(rho2) <= 1e-6 && sqrt(rp2) <= 1e-8) {
            // early exit, tested before the step (inf_newton_solver.m:19-22): the start point is returned
        } else if (badsum > 0.0) {
            st = FMPC_E_NOT_PD_PHI;
        } else {
            // ================= P2: y_u0 = y0c - G rho + Xi_u0 bhat   (row pairs x column groups, summed in a fixed order)
            {
                const int ncb = P.w ? Tn : ((var2 ? 2 : 1) * n < Tn ? (var2 ? 2 : 1) * n : Tn);   // columns of bhat that can be non-zero
                const int ldg = (m + 1) & ~1;
                double* pg = uni;                          // [8][ldg] partial sums of -G rho, then [8][ldg] of Xi_u0 bhat (the region is free here)
                double* px = uni + 8 * (size_t)ldg;
                const int g1 = fr_matvec_pairs<NT>(P.G, ldg, m, m, srho, -1.0, pg, ldg, tid);
                const int g2 = fr_matvec_pairs<NT>(P.Xiu0t, ldg, m, ncb, sbh, 1.0, px, ldg, tid);
                __syncthreads();
                for (int r = tid; r < m; r += NT) {
                    double v = P.y0c[r];
                    for (int q = 0; q < g1; ++q) v += pg[(size_t)q * ldg + r];
                    for (int q = 0; q < g2; ++q) v += px[(size_t)q * ldg + r];
                    ss[r] = v;
                }
                __syncthreads();
            }
            FR_TICK(1);
            // ================= P3: [M | y_u0] as 16 x 16 tiles in LDS (upper tile triangle, over B' | phi | kappa), M = G + diag(1 / delta);
            //                   Cholesky; q
            fr_copy_in<NT>(stl, P.Gt, ntiles * 256, tid);                       // G in the tile layout (zero padding included)
            __syncthreads();
            for (int r = tid; r < m; r += NT) {
                stl[FR_TIDX(r >> 4, r >> 4, NT1) * 256 + (r & 15) * 17] += 1.0 / sdel[r];       // M = G + diag(1 / delta)
                stl[FR_TIDX(r >> 4, NTm, NT1) * 256 + (r & 15) * 16] = ss[r];                 // rhs: column 0 of tile column NTm
            }
            __syncthreads();
            FR_TICK(2);
            const int npd = fr_tile_cholesky_lds(stl, NTm, m, Wg, ss, sCh);
            __syncthreads();
            if (npd) {
                st = FMPC_E_NOT_PD_SCHUR;
            } else {
                FR_TICK(3);
                // ================= P4: s = rho + q; the pass through the constant operators  (timing build: ticks 8 .. 11 inside it)
                for (int c = tid; c < m; c += NT) ss[c] += srho[c];
                __syncthreads();
                for (int idx = tid; idx < Tm; idx += NT) { const int c = idx % m; sphi[idx] = P.g0[idx] * ss[c]; }      // v = Phibar^-1 E s
                __syncthreads();
                for (int idx = tid; idx < nbn; idx += NT) sbe[idx] = P.betab[idx] - sbh[idx];            // beta = C phi + r_p ...
                __syncthreads();
                fr_stage_product<12>(sphi, m, T, P.imgBb, n, m, wv, NW, lane, [&](int i, int r, double v) { sbe[i * n + r] += v; });   // ... + B v_i
                __syncthreads();
                FR_TICK(8);
                for (int idx = tid; idx < Tm; idx += NT) sphi[idx] = P.phib_u[idx] - sphi[idx];                          // phi_u
                // nu+ = Ybar^-1 beta (Yinv symmetric: its rows are its columns)
                {
                    const int ldy = (nbn + 1) & ~1;
                    double* py = skap + Tm;                                    // [8][ldy] behind phi | kappa (the tiles are done with)
                    const int gy = fr_matvec_pairs<NT>(P.Yinv, ldy, nbn, nbn, sbe, 1.0, py, ldy, tid);
                    __syncthreads();
                    for (int a1 = tid; a1 < nbn; a1 += NT) {
                        double v = 0.0;
                        for (int q = 0; q < gy; ++q) v += py[(size_t)q * ldy + a1];
                        snup[a1] = v;
                    }
                }
                __syncthreads();
                FR_TICK(9);
                fr_stage_product<8>(snup, n, T, P.imgBk, m, n, wv, NW, lane, [&](int j, int c, double v) { skap[j * m + c] = v; });   // kappa = B' nu+_j
                __syncthreads();
                FR_TICK(10);
                for (int idx = tid; idx < Tm; idx += NT) {                     // d_u = phi_u + Gf kappa   (eight loads in flight per output)
                    const int j = idx / m, c = idx - j * m;
                    const double* gp = P.Gf + (size_t)j * T * m + c;
                    double acc = 0.0;
                    for (int i0 = 0; i0 < T; i0 += 8) {
                        double gv[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) gv[q] = gp[(size_t)(i0 + q < T ? i0 + q : T - 1) * m];
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc += (i0 + q < T ? gv[q] : 0.0) * skap[(i0 + q < T ? i0 + q : T - 1) * m + c];
                    }
                    sphi[idx] += acc;
                }
                __syncthreads();
                FR_TICK(11);
                // ================= P5: line search (closed form of backtracking_inf_newton.m:2-11, frozen barrier gradient), update
                double be = 0.0, e2 = 0.0;
                for (int idx = tid; idx < Tm; idx += NT) {
                    const int j = idx / m, c = idx - j * m;
                    double e = (P.hd[idx] + (j == 0 ? sdel[c] : 0.0)) * sphi[idx];      // k P'DP d_z on u_j
                    if (j > 0) e -= P.erb[c] * sphi[idx - m];
                    if (j + 1 < T) e -= P.erb[c] * sphi[idx + m];
                    be += srdu[idx] * e;
                    e2 += e * e;
                }
                const double beta_e = fr_block_sum<NT>(be, red);
                const double eps2 = fr_block_sum<NT>(e2, red);
                {
                    const double al = 1e-4;
                    int halv = 0;
                    while (true) {
                        const double gq = (t - 2.0 + 2.0 * al - al * al * t) * rho2 - 2.0 * (1.0 - t) * beta_e + t * eps2;
                        if (gq <= 0.0) break;
                        t *= 0.5;
                        if (++halv >= FR_MAX_HALVINGS) { t = 0.0; st = FMPC_W_LINESEARCH; break; }
                    }
                }
                moved = true;
                nsteps = 1;
            }
        }
        // ---- outputs: z = z0 + t d_z, nu = nu0 + t (nu+ - nu0); a problem that did not step returns the start point
        if (P.zout) {
            double* zp = P.zout + (size_t)p * Nz;
            for (int idx = tid; idx < Nz; idx += NT) {
                const int j = idx / s, e = idx - j * s;
                double v = e < m ? M.umid[e] : M.xmid[e - m];
                if (moved) {
                    if (e < m) v += t * sphi[j * m + e];
                    else {
                        const int r = e - m, jx = j + 1;
                        double cv = snup[j * n + r];
                        if (jx < T) cv -= fr_ldsdot_n<NC>(sA1 + r, n, snup + jx * n, n);
                        if (var2 && jx + 1 < T) cv -= fr_ldsdot_n<NC>(sA2 + r, n, snup + (jx + 1) * n, n);
                        if (jx == T && M.has_xf) cv += snup[T * n + r];
                        v += t * (P.phib_x[j * n + r] - cv / (jx == T ? M.Qf2[r] : M.Q2[r]));
                    }
                }
                zp[idx] = v;
            }
        }
        if (P.u0out)
            for (int c = tid; c < m; c += NT) P.u0out[(size_t)p * m + c] = M.umid[c] + (moved ? t * sphi[c] : 0.0);
        if (P.nuout)
            for (int idx = tid; idx < nbn; idx += NT)
                P.nuout[(size_t)p * nbn + idx] = snu0[idx] + (moved ? t * (snup[idx] - snu0[idx]) : 0.0);
        if (moved && P.step && tid == 0 && P.step_ld > 0) P.step[(size_t)p * P.step_ld] = t;
        if (tid == 0) {
            if (P.status) P.status[p] = st;
            if (P.iters) P.iters[p] = nsteps;
        }
        FR_TICK(5);
    }
}

size_t fmpc_ramp_cold_lds_bytes(int n, int m, int T, int nb) {
    const size_t nbn = (size_t)nb * n, ntm = ((size_t)m + 15) / 16, nt1 = ntm + 1;
    const size_t ntiles = (ntm - 1) * nt1 - (ntm - 1) * (ntm - 2) / 2 + (nt1 - 1 - (ntm - 1)) + 1;      // FR_TIDX(ntm - 1, nt1 - 1, nt1) + 1
    const size_t keep = 4 * nbn + 3 * (size_t)m + (size_t)T * m + 16 + 512 + (16 * 17 + 16 * nt1 + 16 * 8 + 16) + 2 * (size_t)n * n + 2;
    const size_t ph4 = 2 * (size_t)T * m + 8 * ((nbn + 1) & ~(size_t)1);     // [phi | kappa | partial sums of the Yinv product]
    const size_t ph2 = 16 * (((size_t)m + 1) & ~(size_t)1);                    // partial sums of the two products of y_u0
    size_t uni = ph4 > ntiles * 256 ? ph4 : ntiles * 256;
    if (ph2 > uni) uni = ph2;
    return (keep + uni) * sizeof(double);
}
size_t fmpc_ramp_cold_ws_doubles(int m) { const size_t ntm = ((size_t)m + 15) / 16; return ntm * 256; }
hipError_t fmpc_ramp_cold_prepare(size_t lds_bytes) {
    const hipError_t e = hipFuncSetAttribute((const void*)fmpc_ramp_cold<512, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)fmpc_ramp_cold<512, 27>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}
hipError_t fmpc_launch_ramp_cold(const FrColdParams& P, int grid, hipStream_t stream) {
    const size_t lds = fmpc_ramp_cold_lds_bytes(P.M.n, P.M.m, P.M.T, P.M.nb);
    if (P.M.n == 27) hipLaunchKernelGGL((fmpc_ramp_cold<512, 27>), dim3(grid), dim3(512), lds, stream, P);
    else hipLaunchKernelGGL((fmpc_ramp_cold<512, 0>), dim3(grid), dim3(512), lds, stream, P);
    return hipGetLastError();
}

// ---------------------------------------------------------------- host side
size_t fmpc_ramp_lds_bytes(int n, int m, int nbn) {      // sized for the 1024-thread variant (16 waves)
    const size_t ntl = ((size_t)nbn + 1 + 15) / 16;
    const size_t d = (size_t)m * n + 2 * (size_t)n * (n + 1) + 2 * (size_t)n + 16 * 64 + 16 + (16 * 17 + 16 * ntl + 16 * 16 + 16);
    return d * sizeof(double);
}
size_t fmpc_ramp_ws_doubles(int n, int m, int T, int nb) { return fr_ws_layout(n, m, T, nb).total; }

hipError_t fmpc_ramp_prepare(size_t lds_bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)fmpc_newton_ramp<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)fmpc_newton_ramp<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)fmpc_newton_ramp<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

hipError_t fmpc_launch_ramp(const FmpcDevModel& M, const double* dumin, const double* dumax, int batch, int grid,
                            const double* x0, const double* x0p, const double* w, const double* uprev,
                            const double* zinit, const double* nu0, int max_iter, double kbar, double* zout,
                            double* nuout, int* status, int* iters, double* step, int step_ld, double* ws,
                            size_t ws_stride, int threads, hipStream_t stream, int it0) {
    const size_t lds = fmpc_ramp_lds_bytes(M.n, M.m, M.nb * M.n);
    if (threads == 1024)
        hipLaunchKernelGGL(fmpc_newton_ramp<1024>, dim3(grid), dim3(1024), lds, stream, M, dumin, dumax, batch, x0, x0p, w, uprev,
                           zinit, nu0, max_iter, kbar, zout, nuout, status, iters, step, step_ld, ws, ws_stride, it0);
    else if (threads == 512)
        hipLaunchKernelGGL(fmpc_newton_ramp<512>, dim3(grid), dim3(512), lds, stream, M, dumin, dumax, batch, x0, x0p, w, uprev,
                           zinit, nu0, max_iter, kbar, zout, nuout, status, iters, step, step_ld, ws, ws_stride, it0);
    else
        hipLaunchKernelGGL(fmpc_newton_ramp<256>, dim3(grid), dim3(256), lds, stream, M, dumin, dumax, batch, x0, x0p, w, uprev,
                           zinit, nu0, max_iter, kbar, zout, nuout, status, iters, step, step_ld, ws, ws_stride, it0);
    return hipGetLastError();
}
