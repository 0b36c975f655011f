// Shared between the tiled Newton kernel (fmpc_kernel_tiled.hip) and the host code that builds its constants
// (fmpc_api.hip).  Internal to the library.
//
// The tiled kernel is the per-problem-factor path for any n: ONE WORKGROUP PER PROBLEM, NW wavefronts, every n x n
// block of the Schur complement handled as NB x NB tiles of 16 x 16 (n_pad = 16 NB >= n + 1: column n of a stage's
// blocks carries the right-hand side, so the forward substitution rides along in the matrix products).  It is
// templated on the arithmetic type of the factorisation: double (parity path) or float ("fp32 mixed precision":
// fp32 factor and substitutions, all residuals, the line search and the iterate in fp64).
#pragma once
#include <stddef.h>
#include <hip/hip_runtime.h>
#include "fmpc_device.h"

#define FT_TILE 256                      // elements of a 16 x 16 tile, row-major, no padding
#define FT_MAX_NB 5                      // n <= 79
#define FT_WLD 17                        // leading dimension of the transposed inverse diagonal factor in LDS

// The factor stream (HBM workspace) is a sequence of fixed-size RECORDS, one per 16-row block (i, kb), 3 NB tiles each,
// in the order the backward sweep walks them (a record is one contiguous read with constant offsets):
//   tile 0                   RI(kb)    = R(kb,kb)^-1 (upper triangular): what the backward sweep needs of the diagonal
//   tile J, kb < J < NB      R(kb,J):   the off-diagonal tiles of R_i = L_i'   (tiles 1..kb are never written nor used)
//   tile NB + J              U1(kb,J)  = U_{i,i+1}
//   tile 2 NB + J            U2(kb,J)  = U_{i,i+2}
__host__ __device__ static inline int ft_stage_tiles(int NB) { return 3 * NB * NB; }

struct FtModel {
    int NB;                 // 16-blocks per stage
    int mb;                 // 16-blocks covering m
    int cn, nl;             // tile column / local column of the rhs column: n / 16, n % 16
    int nblk;               // unique constant blocks; block id nblk is an all-zero block
    const void* yimg;       // [nblk + 1][NB][NB][256] REAL: constant Y blocks as tiles, zero padded
    const void* btimg;      // [mb][NB][256] REAL: tile (kb, J) element [c % 16][r % 16] = B[r = 16 J + ..][c = 16 kb + ..]
    const int* iD;          // per block row: block id of the constant part of Y_ii
    const int* i1;          //                of Y_{i,i+1}   (zero block if none)
    const int* i2;          //                of Y_{i,i+2}   (zero block if none)
    // zero-padded fp64 copies for the stage-batched GEMMs of the residual phases (no masks, no conditional loads):
    const double* BtP;      // [16 mb][16 NB]: BtP[c*NP + r] = B[r][c]
    const double* BmP;      // [16 NB][16 mb]: BmP[r*MP + c] = B[r][c]
    const double* A1P; const double* A2P;       // [16 NB][16 NB] row-major A1, A2
    const double* A1tP; const double* A2tP;     // and their transposes
    int denseQ;             // Q or Qf not diagonal (fast_mpc_objective.m:52-55 takes any square Q, Qf): the state part of Phi and of
                            // Phi^-1 are then applied as products with these symmetric [16 NB][16 NB] images
    const double* Q2P; const double* Qf2P;      // 2Q, 2Qf
    const double* XP; const double* XfP;        // (2Q)^-1, (2Qf)^-1
    int denseR;             // R not diagonal (fast_mpc_objective.m:51-54 takes any square R): Rt_j = 2R + k diag(1/s+^2 + 1/s-^2) is a dense
                            // m x m matrix per stage, factored in LDS; [Rt_j^-1 B' | Rt_j^-1 r_d[u_j]] goes to the workspace (FtWs::zt)
    const double* R2P;      // [16 mb][16 mb]: 2R, zero padded
};

// Per-workgroup scratch in HBM.  Vectors in doubles, then the factor stream in REAL.
struct FtWs {
    size_t b, nu, hess, winv, rdu, rdx, phx, rp, y, dnu, zt, gt, fac, total;   // offsets in doubles
};
__host__ __device__ static inline FtWs ft_ws_layout(int n, int m, int T, int nb, int NB, int real_bytes, int denseR = 0) {
    FtWs L; size_t o = 0;
    const size_t nbn = (size_t)nb * n, Tm = (size_t)T * m, Tn = (size_t)T * n;
    auto take = [&](size_t cnt) { size_t r = o; o += (cnt + 1) & ~(size_t)1; return r; };
    L.b = take(nbn); L.nu = take(nbn); L.hess = take(Tm); L.winv = take(Tm); L.rdu = take(Tm);
    L.rdx = take(Tn); L.phx = take(Tn); L.rp = take(nbn); L.y = take(nbn); L.dnu = take(nbn);
    L.zt = take(denseR ? Tm * (size_t)(n + 1) : 0);             // dense R: [Rt_j^-1 B' | Rt_j^-1 r_d[u_j]], m x (n + 1) per stage
    o = (o + 31) & ~(size_t)31;
    L.gt = o;                                                    // Y_ii const + B W_i B' (+ rhs column) of every block row, REAL
    o += ((size_t)nb * (NB * (NB + 1) / 2) * FT_TILE * real_bytes + 7) / 8;
    o = (o + 31) & ~(size_t)31;
    L.fac = o;
    o += ((size_t)nb * ft_stage_tiles(NB) * FT_TILE * real_bytes + 7) / 8;
    L.total = (o + 31) & ~(size_t)31;
    return L;
}

// LDS map in bytes.  One region is used three ways, by phases separated by workgroup barriers:
//   residual phases   nu / d_nu as [stage][state]: (16 ceil(nb/16) + 2) rows of 16 NB + 1 doubles
//   G pre-pass        the B' tiles [mb][NB] (B W_i B' for all stages, ahead of the serial factorisation)
//   factor + sweeps   two or three U slots (NB^2 tiles each), the R tiles of the stage, W' of the diagonal tile, small vectors
struct FtLds { size_t bt, slot, lt, wt, ysh, xv, part, wl, red, flag, total; };
// pr_doubles: dense R only -- the packed lower triangle of Rt_j, the m x (n + 1) right-hand sides, m reciprocal pivots
__host__ __device__ static inline size_t ft_pr_doubles(int n, int m) { return (size_t)m * (m + 1) / 2 + (size_t)m * (n + 1) + m; }
// U slots of the factor phase: two (the U2'U2 term of S_i applied one stage ahead, on tiles in registers) except for the
// fp64 instances with 2 wavefronts, where the extra tile set costs registers the kernel does not have (measured: +9 %)
__host__ __device__ static inline int ft_u_slots(int real_bytes, int NW) { return (real_bytes == 8 && NW == 2) ? 3 : 2; }
__host__ __device__ static inline FtLds ft_lds_layout(int NB, int mb, int NW, int real_bytes, int nb, size_t pr_doubles = 0) {
    FtLds L; size_t o = 0;
    const size_t tile = (size_t)FT_TILE * real_bytes;
    L.bt = 0; L.slot = 0;
    o = (size_t)ft_u_slots(real_bytes, NW) * NB * NB * tile;      // the U slots (fmpc_kernel_tiled.hip, P3)
    L.lt = o;   o += (size_t)(NB * (NB + 1) / 2) * tile;
    L.wt = o;   o += (size_t)16 * FT_WLD * real_bytes;
    L.ysh = o;  o += 16 * real_bytes;
    const size_t bt_bytes = (size_t)mb * NB * tile;
    const size_t nu_bytes = (size_t)(16 * ((nb + 15) / 16) + 2) * (16 * NB + 1) * sizeof(double);
    if (bt_bytes > o) o = bt_bytes;
    if (nu_bytes > o) o = nu_bytes;
    if (pr_doubles * sizeof(double) > o) o = pr_doubles * sizeof(double);
    o = (o + 15) & ~(size_t)15;
    // outside the shared region: the x vectors of the backward sweep (live while d_nu is written to the staging area),
    // the Phi^-1 diagonal of NW stages during the S pre-pass
    L.xv = o;   o += 3 * (size_t)16 * NB * real_bytes;
    L.part = o; o += (size_t)(NW / 4 > 0 ? NW / 4 : 1) * 16 * real_bytes;
    L.wl = o;   o += (size_t)NW * mb * 16 * real_bytes;     // the pre-pass works on NW block rows (one per wave) at a time
    o = (o + 15) & ~(size_t)15;
    L.red = o;  o += 16 * sizeof(double);
    L.flag = o; o += 16;
    L.total = (o + 15) & ~(size_t)15;
    return L;
}

struct FtParams {
    FmpcDevModel M;
    FtModel V;
    int batch;
    const double* x0; const double* x0p; const double* w; const double* zinit; const double* nu0;
    int max_iter; double kbar;
    double* zout; double* nuout; int* status; int* iters; double* step; int step_ld;
    double* ws; size_t ws_stride;
    double* u0out;          // nullable: first move of every problem (fmpc_solve_u0_device)
    // Continuation of a Newton budget > 1 after the cold-start step of the panel path (fmpc_api.hip): the problems to work on come
    // from the compacted list the decision launch wrote (entry = problem | FT_LIST_HANDED: redo from the cold start; else z+ is in
    // zout and nu+ in the panel workspace, the first step is done).  list == NULL: all problems 0 .. batch-1 from the start.
    const int* list; const int* nlist; const double* nuws;
};
#define FT_LIST_HANDED (1 << 30)       // (= FW_LIST_HANDED of fmpc_kernel_wave.hip)
#define FT_LIST_GENERAL (1 << 29)      // (= FW_LIST_GENERAL) continue behind ONE step of the one-wavefront kernel from an explicit start: z in
                                       // zout, nu in nuout (per problem, plain), status / step record of that step kept

// supported (type, NB) pairs
bool fmpc_tiled_supports(int n, int m, int nb, int is_float, int* NB_out, int* NW_out, int denseR = 0);
size_t fmpc_tiled_lds_bytes(int NB, int mb, int NW, int is_float, int nb, size_t pr_doubles = 0);
hipError_t fmpc_tiled_prepare(int n, int NB, int NW, int is_float, size_t lds_bytes, int denseR = 0);
hipError_t fmpc_launch_tiled(const FtParams& P, int NB, int NW, int is_float, int grid, size_t lds_bytes, hipStream_t stream);
