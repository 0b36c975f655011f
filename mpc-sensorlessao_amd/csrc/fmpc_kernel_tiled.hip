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
#include <stdlib.h>
#include "fmpc_tiled.h"
#include "fmpc_tile_ops.h"
#include "fmpc_dense_r.h"
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
// (accumulated in LDS by an atomic without return and copied out once at the end: an update in global memory would
// make every tick wait for all of the wave's loads and stores in flight, which is what the factor phase must not do)
__shared__ unsigned long long ft_tl[16];
#define FT_TICK(k) do { _t1 = __builtin_readcyclecounter(); if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&ft_tl[k], _t1 - _t0); _t0 = _t1; } while (0)
#define FT_TL_BEGIN() do { if (threadIdx.x < 16) ft_tl[threadIdx.x] = 0; __syncthreads(); } while (0)
#define FT_TL_END() do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x < 16) ft_timing[threadIdx.x] += ft_tl[threadIdx.x]; } while (0)
#else
#define FT_T0()
#define FT_TICK(k)
#define FT_TL_BEGIN()
#define FT_TL_END()
#endif

// upper-triangular tile enumeration (row-major, I <= J)
__device__ __forceinline__ int ft_lt_index(int NB, int I, int J) { return I * NB - (I * (I - 1)) / 2 + (J - I); }

// P4 as a function of its own (not inlined): the backward sweep needs few registers, but inside the kernel body it
// inherits the register pressure of the factor phase and its loads get spilled addresses with full waits in front.
template <typename R, int NB, int NW>
__device__ __noinline__ void ft_backward(const R* fac_, const double* yv_, R* sXV_, R* sPART_, double* sNU_, int n, int nb, int NUROWS) {
    // The arguments arrive as generic pointers: left so, every access below is a FLAT one -- counted on both memory
    // counters, possibly out of order, so the compiler waits for everything in flight before each use and no request can
    // stay ahead of the step it is for.  Said with their address spaces they are global loads and LDS accesses.
    typedef const R __attribute__((address_space(1))) * GR;
    typedef const double __attribute__((address_space(1))) * GD;
    typedef R __attribute__((address_space(3))) * LR;
    typedef const R __attribute__((address_space(3))) * LCR;
    typedef double __attribute__((address_space(3))) * LD;
    const GR fac = (GR)fac_;
    const GD yv = (GD)yv_;
    const LR sXV = (LR)sXV_, sPART = (LR)sPART_;
    const LD sNU = (LD)sNU_;
    constexpr int NT = NW * 64, NP = 16 * NB, NQ = NB * NB, REC_TILES = 3 * NB, LDN = 16 * NB + 1;
    (void)NQ;
    const int tid = threadIdx.x;
    constexpr int NG = NT >= 256 ? NT / 256 : 1;           // tile groups working on different tiles
    constexpr int RPT = NT >= 256 ? 1 : 256 / NT;          // tile rows per thread (128 threads: rows ta, ta + 8)
    constexpr int RSTEP = 16 / RPT;
    constexpr int MAXT = (REC_TILES - 1 + NG - 1) / NG;    // tiles 1 .. 3 NB - 1 of a record, dealt to the groups
    const int tg = tid >> 8, ta = (tid >> 4) & (RSTEP - 1), tb = tid & 15;
    for (int q = tid; q < 3 * NP; q += NT) sXV[q] = (R)0;
    for (int q = tid; q < NUROWS * LDN; q += NT) sNU[q] = 0.0;   // d_nu as [stage][state] for P5 (the U slots are dead)
    __syncthreads();
    if (NB > 2) {
        // One record (block row) per step, the records of the NEXT G steps requested a group ahead: the factor stream of a
        // problem (4.6 MB at n = 65) comes from HBM, a round trip of a few thousand cycles against ~700 for the step itself.
        // (Every load unconditional at a constant offset from an opaque record offset: see the grouped loop below.)
        constexpr int G = 3;
        R tvA[G][MAXT][RPT], rivA[G][RPT];
        double ybA[G];
        auto request = [&](int base, R (&tv)[G][MAXT][RPT], R (&riv)[G][RPT], double (&yb)[G]) {
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int blk = base - u < 0 ? 0 : base - u;       // (past the top of the horizon: a harmless re-read)
                unsigned off = (unsigned)blk * (unsigned)(REC_TILES * FT_TILE) + (unsigned)(ta * 16 + tb);
                asm volatile("" : "+v"(off));
                const GR rec = fac + off;
#pragma unroll
                for (int q = 0; q < MAXT; ++q) {
                    const int t = 1 + tg + q * NG;
                    const int tc = t < REC_TILES ? t : REC_TILES - 1;
#pragma unroll
                    for (int h = 0; h < RPT; ++h) tv[u][q][h] = rec[tc * FT_TILE + RSTEP * h * 16];
                }
#pragma unroll
                for (int h = 0; h < RPT; ++h) riv[u][h] = rec[RSTEP * h * 16];
                const int i = blk / NB, kb = blk - i * NB, yrow = 16 * kb + tb;
                yb[u] = yv[i * n + (yrow < n ? yrow : n - 1)];
            }
        };
        auto step = [&](int blk, R (&tv)[MAXT][RPT], R (&riv)[RPT], double yb) {
            const int i = blk / NB, kb = blk - i * NB;
            const int yrow = 16 * kb + tb;
            const LR XC = sXV + (i % 3) * NP; const LCR X1 = sXV + ((i + 1) % 3) * NP; const LCR X2 = sXV + ((i + 2) % 3) * NP;
            R accv[RPT];
#pragma unroll
            for (int h = 0; h < RPT; ++h) accv[h] = (R)0;
#pragma unroll
            for (int q = 0; q < MAXT; ++q) {
                const int t = 1 + tg + q * NG;
                const int tc = t < REC_TILES ? t : REC_TILES - 1;
                const LCR xvv = tc < NB ? (LCR)XC + 16 * tc : (tc < 2 * NB ? X1 + 16 * (tc - NB) : X2 + 16 * (tc - 2 * NB));
                const bool use = t < REC_TILES && (t >= NB || t > kb);
                const R xraw = xvv[tb];
                const R xb = use ? xraw : (R)0;
#pragma unroll
                for (int h = 0; h < RPT; ++h) accv[h] += (use ? tv[q][h] : (R)0) * xb;
            }
#pragma unroll
            for (int h = 0; h < RPT; ++h) {
                accv[h] = ft_row16_sum<R>(accv[h]);
                if (tb == 0) sPART[tg * 16 + ta + RSTEP * h] = accv[h];
            }
            ft_lds_barrier();
            if (tg == 0) {
                R sb = yrow < n ? (R)yb : (R)0;
#pragma unroll
                for (int q = 0; q < NG; ++q) sb -= sPART[q * 16 + tb];
#pragma unroll
                for (int h = 0; h < RPT; ++h) {
                    const R xv = ft_row16_sum<R>(riv[h] * sb);
                    if (tb == 0) {
                        const int lr = ta + RSTEP * h;
                        XC[16 * kb + lr] = xv;
                        if (16 * kb + lr < n) sNU[i * LDN + 16 * kb + lr] = (double)xv;
                    }
                }
            }
            ft_lds_barrier();
        };
        // the records that do not fill a group first (no branch inside the pipelined loop: behind one the compiler
        // cannot count the requests in flight and waits for all of them)
        int base = nb * NB - 1;
        for (int left = (nb * NB) % G; left > 0; --left, --base) {
            request(base, tvA, rivA, ybA);
            step(base, tvA[0], rivA[0], ybA[0]);
        }
        request(base, tvA, rivA, ybA);
        for (; base >= 0; base -= G) {
            R tvB[G][MAXT][RPT], rivB[G][RPT];
            double ybB[G];
            request(base - G, tvB, rivB, ybB);
#pragma unroll
            for (int u = 0; u < G; ++u) step(base - u, tvA[u], rivA[u], ybA[u]);
#pragma unroll
            for (int u = 0; u < G; ++u) {
#pragma unroll
                for (int q = 0; q < MAXT; ++q)
#pragma unroll
                    for (int h = 0; h < RPT; ++h) tvA[u][q][h] = tvB[u][q][h];
#pragma unroll
                for (int h = 0; h < RPT; ++h) rivA[u][h] = rivB[u][h];
                ybA[u] = ybB[u];
            }
        }
    } else {
    // KBB records (block rows) are requested together: the whole stage for small blocks (one memory round trip per
    // stage instead of one per block row), one at a time where a stage's tiles would not fit the registers
    constexpr int KBB = NB <= 2 ? NB : 1;
    for (int grp = nb * (NB / KBB) - 1; grp >= 0; --grp) {
        const int i = grp / (NB / KBB), kb0 = (grp - i * (NB / KBB)) * KBB;
        // every load unconditional, at a constant offset from the record: tile 1 + tg + q NG (clamped), RI = tile 0
        R tv[KBB][MAXT][RPT], riv[KBB][RPT], ybv[KBB];
#pragma unroll
        for (int kk = 0; kk < KBB; ++kk) {
            const int kb = kb0 + kk;
            // (the element offset is made opaque: left to itself the compiler strength-reduces one 64-bit address per
            //  load into loop-carried registers, spills them, and reloads each -- with a full wait -- before its load)
            unsigned off = (unsigned)(i * NB + kb) * (unsigned)(REC_TILES * FT_TILE) + (unsigned)(ta * 16 + tb);
            asm volatile("" : "+v"(off));
            const GR rec = fac + off;
            const GR rtg = rec + tg * FT_TILE;               // this thread group's first tile is 1 + tg: constant strides from here
            constexpr bool EXACT = (REC_TILES - 1) % NG == 0;   // no group runs past the record: no clamping
#pragma unroll
            for (int q = 0; q < MAXT; ++q) {
                if (EXACT) {
#pragma unroll
                    for (int h = 0; h < RPT; ++h) tv[kk][q][h] = rtg[(1 + q * NG) * FT_TILE + RSTEP * h * 16];
                } else {
                    const int t = 1 + tg + q * NG;
                    const int tc = t < REC_TILES ? t : REC_TILES - 1;
#pragma unroll
                    for (int h = 0; h < RPT; ++h) tv[kk][q][h] = rec[tc * FT_TILE + RSTEP * h * 16];
                }
            }
#pragma unroll
            for (int h = 0; h < RPT; ++h) riv[kk][h] = rec[RSTEP * h * 16];
            const int yrow = 16 * kb + tb;
            ybv[kk] = (R)yv[i * n + (yrow < n ? yrow : n - 1)];
        }
        const LR XC = sXV + (i % 3) * NP; const LCR X1 = sXV + ((i + 1) % 3) * NP; const LCR X2 = sXV + ((i + 2) % 3) * NP;
#pragma unroll
        for (int kk = KBB - 1; kk >= 0; --kk) {
            const int kb = kb0 + kk;
            R accv[RPT];
#pragma unroll
            for (int h = 0; h < RPT; ++h) accv[h] = (R)0;
#pragma unroll
            for (int q = 0; q < MAXT; ++q) {
                const int t = 1 + tg + q * NG;         // tile of the record: R(kb, t) | U1(kb, t - NB) | U2(kb, t - 2 NB)
                const int tc = t < REC_TILES ? t : REC_TILES - 1;
                const LCR xvv = tc < NB ? (LCR)XC + 16 * tc : (tc < 2 * NB ? X1 + 16 * (tc - NB) : X2 + 16 * (tc - 2 * NB));
                const bool use = t < REC_TILES && (t >= NB || t > kb);     // (tiles 1..kb of a record do not exist)
                const R xraw = xvv[tb];                // (unconditional: always a valid LDS address)
                const R xb = use ? xraw : (R)0;
#pragma unroll
                for (int h = 0; h < RPT; ++h) accv[h] += (use ? tv[kk][q][h] : (R)0) * xb;
            }
#pragma unroll
            for (int h = 0; h < RPT; ++h) {
                accv[h] = ft_row16_sum<R>(accv[h]);
                if (tb == 0) sPART[tg * 16 + ta + RSTEP * h] = accv[h];
            }
            ft_lds_barrier();
            if (tg == 0) {
                R sb = 16 * kb + tb < n ? ybv[kk] : (R)0;
#pragma unroll
                for (int q = 0; q < NG; ++q) sb -= sPART[q * 16 + tb];
#pragma unroll
                for (int h = 0; h < RPT; ++h) {
                    const R xv = ft_row16_sum<R>(riv[kk][h] * sb);
                    if (tb == 0) {
                        const int lr = ta + RSTEP * h;
                        XC[16 * kb + lr] = xv;
                        if (16 * kb + lr < n) sNU[i * LDN + 16 * kb + lr] = (double)xv;
                    }
                }
            }
            ft_lds_barrier();
        }
    }
    }

}

// (dense R: ft_dense_r lives in fmpc_dense_r.h, shared with the generic kernel's workspace instance)
// ---------------------------------------------------------------------------------------------------------------------------
// P3 of the kernel below (factor + forward sweep) as a function of its own, with a register allocation of its own.  It takes
// nothing but the kernel's parameter block: every pointer and size is rebuilt from it (scalar arithmetic), the LDS map from
// the same layout function, the tile ownership from the wavefront number.  Returns true when a pivot was not positive.
typedef const FtParams __attribute__((address_space(4))) * FtKP;
// The tile-index tables of the constant images (written by the host before the launch, uniform subscripts): read through
// the constant address space they are scalar loads.  As ordinary global loads each one makes the wave wait for ALL its
// vector memory traffic in flight (one in-order counter), the tiles requested a stage ahead included.
typedef const int __attribute__((address_space(4))) * ft_cidx;
__device__ __forceinline__ FtKP ft_params() { return (FtKP)__builtin_amdgcn_kernarg_segment_ptr(); }
__device__ __forceinline__ FtKP ft_uniform(FtKP P) {             // a function argument arrives in VGPRs: make it scalar again
    const unsigned long long a = (unsigned long long)P;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (FtKP)(((unsigned long long)hi << 32) | lo);
}
template <typename R, int NB, int NW, int NL, bool DR>
__device__ __noinline__ bool ft_phase_factor(FtKP Pin) {
    typedef FtT<R> TT;
    typedef typename TT::v4 v4;
    constexpr int NS = NB * (NB + 1) / 2, NQ = NB * NB;
    constexpr int SS = (NS + NW - 1) / NW, MS = (NQ + NW - 1) / NW;
    constexpr int STAGE_TILES = 3 * NB * NB, REC_TILES = 3 * NB;
    constexpr bool TS = !(sizeof(R) == 8 && NW == 2);
    constexpr bool PIPE = sizeof(R) == 4;                   // software-pipelined operand reads (see phase A)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FtKP P = ft_uniform(Pin);
    const int n = P->M.n, m = P->M.m, T = P->M.T, nb = P->M.nb;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int mb = P->V.mb, cn = P->V.cn, nl = P->V.nl;
    const FtLds LL = ft_lds_layout(NB, mb, NW, (int)sizeof(R), nb, DR ? ft_pr_doubles(n, m) : 0);
    R* sSLOT = (R*)(smem + LL.slot);
    R* sLT = (R*)(smem + LL.lt);
    R* sWT = (R*)(smem + LL.wt);
    R* sYSH = (R*)(smem + LL.ysh);
    int* sflag = (int*)(smem + LL.flag);
    const FtWs L = ft_ws_layout(n, m, T, nb, NB, (int)sizeof(R), DR ? 1 : 0);
    double* wsp = P->ws + (size_t)blockIdx.x * P->ws_stride;
    double* yv = wsp + L.y;
    R* fac = (R*)(wsp + L.fac);
    R* gws = (R*)(wsp + L.gt);
    const R* yimg = (const R*)P->V.yimg;
    const ft_cidx Vi1 = (ft_cidx)P->V.i1, Vi2 = (ft_cidx)P->V.i2;
    const int firstS = wv, firstM1 = ((wv - NS) % NW + NW) % NW, firstM2 = ((wv - NS - NQ) % NW + 2 * NW) % NW;
    int sI[SS], sJ[SS];
#pragma unroll
    for (int sl = 0; sl < SS; ++sl) {
        int t = firstS + sl * NW, I = 0;
        if (t < NS) { while (t >= NB - I) { t -= NB - I; ++I; } sI[sl] = I; sJ[sl] = I + t; }
        else { sI[sl] = -1; sJ[sl] = -1; }
    }
    // block row / column of every M1, M2 slot of this wave (-1: no tile), once: the tests in the loop over kb below are
    // then a compare each.  (That loop visits every slot of the wave per block row to find the three or four that are in
    // it; the tests, not the products, were a fifth of its time.)
    int rM1[MS], cM1[MS], rM2[MS], cM2[MS];
#pragma unroll
    for (int sl = 0; sl < MS; ++sl) {
        const int q1 = firstM1 + sl * NW, q2 = firstM2 + sl * NW;
        rM1[sl] = q1 < NQ ? q1 / NB : -1; cM1[sl] = q1 % NB;
        rM2[sl] = q2 < NQ ? q2 / NB : -1; cM2[sl] = q2 % NB;
    }
#define FT_UNLIKELY(x) __builtin_expect(!!(x), 0)
    FT_T0();
    bool fail = false;
    int ub = 1, uc = 2;                                     // !TS: roles of the slots 1 and 2
    // Loads queue behind the stores a wave has issued (vmcnt is in order), so everything stage i + 1 needs from
    // memory -- its S0 tiles and the constant Y_{i,i+1} tiles -- is requested at the top of stage i's phase B,
    // before that stage's factor tiles are stored.
    v4 nS[SS], fS[SS], nM1[MS];                          // S0 tiles of stage i + 1 (loaded, updated in phase A) and i + 2 (in flight)
    // (every request is unconditional -- a slot without a tile re-reads the last one and nothing uses the result -- so that
    // the compiler can COUNT the requests between a load and its use: behind a branch it has to assume none were made and
    // waits for the whole queue instead)
    auto requestS = [&](int i, v4 (&dst)[SS]) {
#pragma unroll
        for (int sl = 0; sl < SS; ++sl) {
            const int t = firstS + sl * NW < NS ? firstS + sl * NW : NS - 1;
            const R* gt = gws + ((size_t)i * NS + t) * FT_TILE;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[sl][r] = gt[TT::row(g, r) * 16 + c];
        }
    };
    auto requestM = [&](int i) {
        const R* Y1 = yimg + (size_t)Vi1[i] * NQ * FT_TILE;
#pragma unroll
        for (int sl = 0; sl < MS; ++sl) {
            const int q1 = firstM1 + sl * NW < NQ ? firstM1 + sl * NW : NQ - 1;
            const R* yt = Y1 + (size_t)q1 * FT_TILE;
#pragma unroll
            for (int r = 0; r < 4; ++r) nM1[sl][r] = yt[TT::row(g, r) * 16 + c];
        }
    };
    v4 cS[SS];                                             // S tiles of the current stage (S0_i - U2_{i-2}' U2_{i-2})
    if (TS) { requestS(0, cS); requestS(nb > 1 ? 1 : 0, nS); }
    else requestS(0, nS);                                  // (!TS: no third set, nS is the current stage's at the top of a stage)
    requestM(0);
    for (int i = 0; i < nb; ++i) {
        R* UA = sSLOT;
        R* UB = sSLOT + (size_t)(TS ? 1 : ub) * NQ * FT_TILE;
        R* UC = sSLOT + (size_t)(TS ? 1 : uc) * NQ * FT_TILE;
        R* facs = fac + (size_t)i * STAGE_TILES * FT_TILE;
        const R* Y2 = yimg + (size_t)Vi2[i] * NQ * FT_TILE;
        // the constant Y_{i,i+2} tiles of this stage (no products in phase A: they arrive in its shadow)
        v4 aM2[MS];
#pragma unroll
        for (int sl = 0; sl < MS; ++sl) {
            const int q2 = firstM2 + sl * NW < NQ ? firstM2 + sl * NW : NQ - 1;
            const R* yt = Y2 + (size_t)q2 * FT_TILE;
#pragma unroll
            for (int r = 0; r < 4; ++r) aM2[sl][r] = yt[TT::row(g, r) * 16 + c];
        }
        // ---------------- phase A: all tiles of the stage, independent
        v4 aS[SS], aM1[MS];
#pragma unroll
        for (int sl = 0; sl < SS; ++sl) aS[sl] = TS ? cS[sl] : nS[sl];
#pragma unroll
        for (int sl = 0; sl < MS; ++sl) aM1[sl] = nM1[sl];
        if constexpr (PIPE) {
            // fp32: the operands of product j + 1 are in flight while the matrix cores take product j (two accumulators per S
            // tile, two M1 tiles side by side: no product waits for the one before it either).
            // The last 16-row block of a stage has NL live rows; the others are zero in every U tile (the rows of W = R^-T
            // past the block's size are), and product r of a tile pair covers the rows r, 4 + r, 8 + r, 12 + r: with NL < 4
            // known at compile time only the first NL products of the last block are made (n = 65: one of four).
            constexpr int KL = (NL > 0 && NL < 4) ? NL : 4;
            const R* UB2 = TS ? UB : UC;
#pragma unroll
            for (int sl = 0; sl < SS; ++sl) {
                const int I = sI[sl], J = sJ[sl];
                if (I >= 0) {
                    const R* pa = UA + (size_t)I * FT_TILE; const R* qa = UA + (size_t)J * FT_TILE;
                    const R* pb = UB2 + (size_t)I * FT_TILE; const R* qb = UB2 + (size_t)J * FT_TILE;
                    v4 xa = TT::ld4(pa, lane), za = TT::ld4(qa, lane), xb = TT::ld4(pb, lane), zb = TT::ld4(qb, lane);
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        v4 nxa = xa, nza = za, nxb = xb, nzb = zb;
                        if (j + 1 < NB) {
                            const int o = (j + 1) * NB * FT_TILE;
                            nxa = TT::ld4(pa + o, lane); nza = TT::ld4(qa + o, lane); nxb = TT::ld4(pb + o, lane); nzb = TT::ld4(qb + o, lane);
                        }
#pragma unroll
                        for (int r = 0; r < (j == NB - 1 ? KL : 4); ++r) {
                            aS[sl] = TT::mfma_sub(xa[r], za[r], aS[sl]);
                            if (TS) nS[sl] = TT::mfma_sub(xb[r], zb[r], nS[sl]);       // for S_{i+1}
                        }
                        if (!TS) ft_xtz_sub<R>(aS[sl], xb, zb);
                        xa = nxa; za = nza; xb = nxb; zb = nzb;
                    }
                }
            }
            auto m1chain2 = [&](v4& a0, v4& a1, int q0, int q1) {
                const R* p0 = UA + (size_t)(q0 / NB) * FT_TILE; const R* z0 = UB + (size_t)(q0 % NB) * FT_TILE;
                const R* p1 = UA + (size_t)(q1 / NB) * FT_TILE; const R* z1 = UB + (size_t)(q1 % NB) * FT_TILE;
                v4 x0 = TT::ld4(p0, lane), y0 = TT::ld4(z0, lane), x1 = TT::ld4(p1, lane), y1 = TT::ld4(z1, lane);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    v4 nx0 = x0, ny0 = y0, nx1 = x1, ny1 = y1;
                    if (j + 1 < NB) {
                        const int o = (j + 1) * NB * FT_TILE;
                        nx0 = TT::ld4(p0 + o, lane); ny0 = TT::ld4(z0 + o, lane); nx1 = TT::ld4(p1 + o, lane); ny1 = TT::ld4(z1 + o, lane);
                    }
#pragma unroll
                    for (int r = 0; r < (j == NB - 1 ? KL : 4); ++r) { a0 = TT::mfma_sub(x0[r], y0[r], a0); a1 = TT::mfma_sub(x1[r], y1[r], a1); }
                    x0 = nx0; y0 = ny0; x1 = nx1; y1 = ny1;
                }
            };
#pragma unroll
            for (int sl = 0; sl < MS; sl += 2) {
                const int q0 = firstM1 + sl * NW, q1 = firstM1 + (sl + 1) * NW;
                if (sl + 1 < MS && q1 < NQ) m1chain2(aM1[sl], aM1[sl + 1 < MS ? sl + 1 : sl], q0, q1);
                else if (q0 < NQ)
                    ft_xtz_chain<R>(aM1[sl], NB, lane, [&](int j) { return UA + (size_t)(j * NB + q0 / NB) * FT_TILE; },
                                    [&](int j) { return UB + (size_t)(j * NB + q0 % NB) * FT_TILE; });
            }
        } else {
#pragma unroll
        for (int sl = 0; sl < SS; ++sl) {
            const int I = sI[sl], J = sJ[sl];
            if (I >= 0) {
#pragma unroll 1
                for (int j = 0; j < NB; ++j) {
                    ft_xtz_sub<R>(aS[sl], UA + (size_t)(j * NB + I) * FT_TILE, UA + (size_t)(j * NB + J) * FT_TILE, lane);
                    if (TS) ft_xtz_sub<R>(nS[sl], UB + (size_t)(j * NB + I) * FT_TILE, UB + (size_t)(j * NB + J) * FT_TILE, lane);   // for S_{i+1}
                    else ft_xtz_sub<R>(aS[sl], UC + (size_t)(j * NB + I) * FT_TILE, UC + (size_t)(j * NB + J) * FT_TILE, lane);
                }
            }
        }
#pragma unroll
        for (int sl = 0; sl < MS; ++sl) {
            const int q1 = firstM1 + sl * NW;
            if (q1 < NQ) {
                const int I = q1 / NB, J = q1 - I * NB;
#pragma unroll 1
                for (int j = 0; j < NB; ++j)
                    ft_xtz_sub<R>(aM1[sl], UA + (size_t)(j * NB + I) * FT_TILE, UB + (size_t)(j * NB + J) * FT_TILE, lane);
            }
        }
        }
        ft_lds_barrier();                                      // Ua, Ub are dead from here: their slots take U1_i, U2_i
        FT_TICK(3);
        R* U1N = UA; R* U2N = TS ? UB : UC;
        if (!TS) requestS(i + 1 < nb ? i + 1 : nb - 1, nS);    // (harmless re-reads at the end of the horizon)
        requestM(i + 1 < nb ? i + 1 : i);
        if (TS) requestS(i + 2 < nb ? i + 2 : nb - 1, fS);
        // aM2 (requested at the top of the stage) has to be there now.  Said here, with the count of the requests just made,
        // because the loop over kb has stores and no load: the compiler then drains the memory counter once in front of
        // such a loop if anything used inside is still in flight -- every tile requested above would be waited for here
        // instead of a stage later.  (An empty statement the compiler cannot see through: it waits for exactly these.)
#pragma unroll
        for (int sl = 0; sl < MS; ++sl)
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(aM2[sl][r]));
        FT_TICK(11);
        // ---------------- phase B: the 16-row blocks of the stage, in order
        for (int kb = 0; kb < NB; ++kb) {
            int cnt = n - 16 * kb; cnt = cnt > 16 ? 16 : (cnt < 0 ? 0 : cnt);
            // (1) products with the rows of this stage already done
            if constexpr (PIPE) {
#pragma unroll
                for (int sl = 0; sl < SS; ++sl)
                    if (FT_UNLIKELY(sI[sl] == kb)) {
                        const int J = sJ[sl];
                        ft_xtz_chain<R>(aS[sl], kb, lane, [&](int j) { return sLT + (size_t)ft_lt_index(NB, j, kb) * FT_TILE; },
                                        [&](int j) { return sLT + (size_t)ft_lt_index(NB, j, J) * FT_TILE; });
                    }
#pragma unroll
                for (int sl = 0; sl < MS; ++sl) {
                    if (FT_UNLIKELY(rM1[sl] == kb)) {
                        const int J = cM1[sl];
                        ft_xtz_chain<R>(aM1[sl], kb, lane, [&](int j) { return sLT + (size_t)ft_lt_index(NB, j, kb) * FT_TILE; },
                                        [&](int j) { return U1N + (size_t)(j * NB + J) * FT_TILE; });
                    }
                    if (FT_UNLIKELY(rM2[sl] == kb)) {
                        const int J = cM2[sl];
                        ft_xtz_chain<R>(aM2[sl], kb, lane, [&](int j) { return sLT + (size_t)ft_lt_index(NB, j, kb) * FT_TILE; },
                                        [&](int j) { return U2N + (size_t)(j * NB + J) * FT_TILE; });
                    }
                }
            } else {
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
                if (q2 < NQ && q2 / NB == kb) {
                    for (int j = 0; j < kb; ++j)
                        ft_xtz_sub<R>(aM2[sl], sLT + (size_t)ft_lt_index(NB, j, kb) * FT_TILE,
                                      U2N + (size_t)(j * NB + q2 % NB) * FT_TILE, lane);
                }
            }
            }
            FT_TICK(9);
            // (2) the rhs column of this row block, still unscaled: shared with the owners of M1(kb,cn), M2(kb,cn);
            //     the owner of the diagonal tile factors it
#pragma unroll
            for (int sl = 0; sl < SS; ++sl) {
                if (FT_UNLIKELY(sI[sl] == kb) && sJ[sl] == cn && c == nl) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sYSH[TT::row(g, r)] = aS[sl][r];
                }
                if (FT_UNLIKELY(sI[sl] == kb && sJ[sl] == kb)) {
                    v4 Ro, Wo;
                    FT_TICK(14);
                    __builtin_amdgcn_s_setprio(3);             // a chain of dependent steps: issue ahead of the SIMD's other wave
                    bool ok;
                    if (kb < NB - 1) ok = ft_potrf16_ct<R, 16>(aS[sl], c, g, Ro, Wo);        // (all blocks but the last are full)
                    else if (NL >= 0) ok = ft_potrf16_ct<R, (NL >= 0 ? NL : 0)>(aS[sl], c, g, Ro, Wo);
                    else ok = ft_potrf16<R>(aS[sl], cnt, c, g, Ro, Wo);
                    __builtin_amdgcn_s_setprio(0);
                    FT_TICK(15);
                    if (!ok && lane == 0) sflag[0] = 1;
                    R* ri = facs + (size_t)(kb * REC_TILES) * FT_TILE;   // R(kb,kb)^-1 = W' for the backward sweep
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
            FT_TICK(10);
            ft_lds_barrier();
            FT_TICK(4);
            // (3) scale the tiles of the row: Rwide(kb, .) = W P(kb, .)
            R wop[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) wop[r] = sWT[TT::row(g, r) * FT_WLD + c];
#pragma unroll
            for (int sl = 0; sl < SS; ++sl) {
                if (FT_UNLIKELY(sI[sl] == kb && sJ[sl] > kb)) {
                    v4 o = {0, 0, 0, 0};
#pragma unroll
                    for (int r = 0; r < 4; ++r) o = TT::mfma(wop[r], aS[sl][r], o);
                    R* dl = sLT + (size_t)ft_lt_index(NB, kb, sJ[sl]) * FT_TILE;
                    R* dg = facs + (size_t)(kb * REC_TILES + sJ[sl]) * FT_TILE;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dg[TT::row(g, r) * 16 + c] = o[r];
                    TT::st4(dl, lane, o);
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
                if (FT_UNLIKELY(rM1[sl] == kb)) {
                    const int J = cM1[sl];
                    v4 pv = aM1[sl];
                    if (J == cn && c == nl) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) pv[r] = sYSH[TT::row(g, r)];
                    }
                    v4 o = {0, 0, 0, 0};
#pragma unroll
                    for (int r = 0; r < 4; ++r) o = TT::mfma(wop[r], pv[r], o);
                    R* dl = U1N + (size_t)q1 * FT_TILE;
                    R* dg = facs + (size_t)(kb * REC_TILES + NB + J) * FT_TILE;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dg[TT::row(g, r) * 16 + c] = o[r];
                    TT::st4(dl, lane, o);
                }
                if (FT_UNLIKELY(rM2[sl] == kb)) {
                    const int J = cM2[sl];
                    v4 pv = aM2[sl];
                    if (J == cn && c == nl) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) pv[r] = sYSH[TT::row(g, r)];
                    }
                    v4 o = {0, 0, 0, 0};
#pragma unroll
                    for (int r = 0; r < 4; ++r) o = TT::mfma(wop[r], pv[r], o);
                    R* dl = U2N + (size_t)q2 * FT_TILE;
                    R* dg = facs + (size_t)(kb * REC_TILES + 2 * NB + J) * FT_TILE;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dg[TT::row(g, r) * 16 + c] = o[r];
                    TT::st4(dl, lane, o);
                }
            }
            ft_lds_barrier();
            FT_TICK(5);
            if (sflag[0]) { fail = true; break; }              // uniform: read after the barrier
        }
        if (fail) break;
        // next stage: its S tiles are the updated set, the set in flight becomes the next one
#pragma unroll
        for (int sl = 0; sl < SS; ++sl) { if (TS) { cS[sl] = nS[sl]; nS[sl] = fS[sl]; } }
        if (!TS) { const int t = ub; ub = uc; uc = t; }        // Ub <- U2_i (slot uc), Uc <- old Ub
    }
    return fail;
}

// The S pre-pass of the kernel below (S0_i = Y_ii const + B W_i B' of every block row) as a function of its own: see ft_phase_factor.
// Workgroup-collective (barriers inside): every thread of the workgroup calls it.
template <typename R, int NB, int NW, bool DR>
__device__ __noinline__ void ft_phase_spre(FtKP Pin) {
    typedef FtT<R> TT;
    typedef typename TT::v4 v4;
    constexpr int NT = NW * 64;
    constexpr int NS = NB * (NB + 1) / 2, NQ = NB * NB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FtKP P = ft_uniform(Pin);
    const int n = P->M.n, m = P->M.m, T = P->M.T, nb = P->M.nb;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int mb = P->V.mb, cn = P->V.cn, nl = P->V.nl;
    const FtLds LL = ft_lds_layout(NB, mb, NW, (int)sizeof(R), nb, DR ? ft_pr_doubles(n, m) : 0);
    R* sBT = (R*)(smem + LL.bt);
    R* sWL = (R*)(smem + LL.wl);
    const FtWs L = ft_ws_layout(n, m, T, nb, NB, (int)sizeof(R), DR ? 1 : 0);
    double* wsp = P->ws + (size_t)blockIdx.x * P->ws_stride;
    double* ztw = wsp + L.zt;
    const int ZLD = n + 1;
    double* winv = wsp + L.winv;
    double* yv = wsp + L.y;
    R* gws = (R*)(wsp + L.gt);
    const R* yimg = (const R*)P->V.yimg;
    const ft_cidx ViD = (ft_cidx)P->V.iD;
    {
        const R* src = (const R*)P->V.btimg;
        for (int q = tid; q < mb * NB * FT_TILE; q += NT) sBT[q] = src[q];
    }
    {
        // One block row (all its NS upper-triangular tiles) per wave and chunk: the B' tiles and Phi^-1 read from LDS
        // feed NS products (one LDS read per MFMA instead of three; the LDS port is what bounds this phase).
        constexpr int WPT = (NW * 16 * 16 + NT - 1) / NT;      // Phi^-1 entries per thread and chunk (mb <= 16)
        R wreg[WPT];
        auto wload = [&](int i0) {
#pragma unroll
            for (int e = 0; e < WPT; ++e) {
                const int q = tid + e * NT;
                const int ii = q / (mb * 16), k = q - ii * (mb * 16);
                const bool ok = q < NW * mb * 16 && i0 + ii < T && k < m;
                const R v = (R)winv[ok ? (size_t)(i0 + ii) * m + k : 0];
                wreg[e] = ok ? v : (R)0;
            }
        };
        wload(0);
        for (int i0 = 0; i0 < nb; i0 += NW) {
            __syncthreads();                                   // (the previous chunk's products are done with sWL)
#pragma unroll
            for (int e = 0; e < WPT; ++e) {
                const int q = tid + e * NT;
                if (q < NW * mb * 16) sWL[q] = wreg[e];
            }
            __syncthreads();
            const int i = i0 + wv;
            if (i0 + NW < nb) wload(i0 + NW);
            if (i < nb) {                                      // uniform per wave
                v4 a[NS];
                const R* yd = yimg + (size_t)ViD[i] * NQ * FT_TILE;
#pragma unroll
                for (int t = 0, I = 0, J = 0; t < NS; ++t) {  // every load of the block row first ...
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[t][r] = yd[(size_t)(I * NB + J) * FT_TILE + TT::row(g, r) * 16 + c];
                    if (J == cn) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * I + TT::row(g, r);
                            const R yr = (R)yv[i * n + (row < n ? row : n - 1)];
                            a[t][r] = c == nl ? (row < n ? yr : (R)0) : a[t][r];
                        }
                    }
                    if (++J == NB) { ++I; J = I; }
                }
                if (i < T) {
                    const R* wl = sWL + wv * mb * 16;
#pragma unroll 3
                    for (int kb = 0; kb < mb; ++kb) {
                        R x[NB][4], zw[NB][4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const R wk = wl[16 * kb + 4 * r + g];
                            const int qz = 16 * kb + 4 * r + g < m ? 16 * kb + 4 * r + g : m - 1;   // dense R: row of Z_i = Rt_i^-1 B'
#pragma unroll
                            for (int J = 0; J < NB; ++J) {
                                x[J][r] = sBT[(size_t)(kb * NB + J) * FT_TILE + 64 * r + lane];
                                if (DR) {
                                    const int cz = 16 * J + c;
                                    const double zv = ztw[((size_t)i * m + qz) * ZLD + (cz < n ? cz : n - 1)];
                                    zw[J][r] = cz < n ? (R)zv : (R)0;                 // (column n of a tile row is the rhs)
                                } else {
                                    zw[J][r] = x[J][r] * wk;
                                }
                            }
                        }
#pragma unroll
                        for (int t = 0, I = 0, J = 0; t < NS; ++t) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) a[t] = TT::mfma(x[I][r], zw[J][r], a[t]);
                            if (++J == NB) { ++I; J = I; }
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < NS; ++t) {                 // ... the stores last
                    R* dst = gws + ((size_t)i * NS + t) * FT_TILE;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[TT::row(g, r) * 16 + c] = a[t][r];
                }
            }
        }
    }
}

// Everything a phase function needs, rebuilt from the kernel's parameter block (scalar arithmetic; unused names cost nothing)
#define FT_VIEW(R, NB, NW, DR)                                                                                                      \
    constexpr int NT = NW * 64, NP = 16 * NB;                                                                                       \
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                                                            \
    const FtKP P = ft_uniform(Pin);                                                                                                 \
    [[maybe_unused]] const int n = P->M.n, m = P->M.m, T = P->M.T, nb = P->M.nb;                                                     \
    [[maybe_unused]] const int s = n + m, Nz = T * s, nbn = nb * n;                                                                 \
    [[maybe_unused]] const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);                   \
    [[maybe_unused]] const int c = lane & 15, g = lane >> 4;                                                                        \
    [[maybe_unused]] const bool var2 = P->M.var2 != 0;                                                                              \
    [[maybe_unused]] const int mb = P->V.mb;                                                                                        \
    const FtLds LL = ft_lds_layout(NB, mb, NW, (int)sizeof(R), nb, DR ? ft_pr_doubles(n, m) : 0);                                   \
    [[maybe_unused]] double* red = (double*)(smem + LL.red);                                                                        \
    [[maybe_unused]] double* sNU = (double*)(smem + LL.slot);                                                                       \
    [[maybe_unused]] constexpr int LDN = 16 * NB + 1;                                                                               \
    [[maybe_unused]] const int TA = (nb + 15) / 16, NUROWS = 16 * TA + 2, MP = 16 * mb;                                             \
    const FtWs L = ft_ws_layout(n, m, T, nb, NB, (int)sizeof(R), DR ? 1 : 0);                                                       \
    double* wsp = P->ws + (size_t)blockIdx.x * P->ws_stride;                                                                        \
    [[maybe_unused]] double* ztw = wsp + L.zt;                                                                                      \
    [[maybe_unused]] const int ZLD = n + 1;                                                                                         \
    [[maybe_unused]] double* b = wsp + L.b;                                                                                         \
    [[maybe_unused]] double* nu = wsp + L.nu;                                                                                       \
    [[maybe_unused]] double* hess = wsp + L.hess;                                                                                   \
    [[maybe_unused]] double* winv = wsp + L.winv;                                                                                   \
    [[maybe_unused]] double* rdu = wsp + L.rdu;                                                                                     \
    [[maybe_unused]] double* rdx = wsp + L.rdx;                                                                                     \
    [[maybe_unused]] double* phx = wsp + L.phx;                                                                                     \
    [[maybe_unused]] double* rp = wsp + L.rp;                                                                                       \
    [[maybe_unused]] double* yv = wsp + L.y

struct FtResid { double rp2, rho2, bad; };
struct FtStep { double t; int collapsed; };

// P1 of the kernel below (residuals of problem p, their norms) as a function of its own: see ft_phase_factor.  Workgroup-collective.
template <typename R, int NB, int NW, bool DR>
__device__ __noinline__ FtResid ft_phase_resid(FtKP Pin, int p) {
    FT_VIEW(R, NB, NW, DR);
    p = __builtin_amdgcn_readfirstlane(p);
    double* zp = P->zout + (size_t)p * Nz;
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
                // Z comes from zero-padded images and X needs no zeros where Z has them, so every load is
                // unconditional (a conditional load is a branch with a wait behind it); stage 0 / 1 terms that
                // do not exist are switched off by a factor
                const double* zi0 = zp + (size_t)(i < T ? i : T - 1) * s;
                const double* zi1 = zp + (size_t)((i >= 1 && i < T) ? i - 1 : 0) * s + m;
                const double* zi2 = zp + (size_t)((i >= 2 && i < T) ? i - 2 : 0) * s + m;
                const double f1 = (i >= 1 && i < T) ? 1.0 : 0.0, f2 = (i >= 2 && i < T) ? 1.0 : 0.0;
                ft_vec_gemm<12>(acc, MP, g,
                            [&](int k) { return zi0[k < m ? k : m - 1]; },
                            [&](int k) { return P->V.BtP[(size_t)k * NP + r]; });
                ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(acc, NP, g,
                            [&](int k) { return zi1[k < n ? k : n - 1] * f1; },
                            [&](int k) { return P->V.A1tP[k * NP + r]; });
                if (var2)
                    ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(acc, NP, g,
                                [&](int k) { return zi2[k < n ? k : n - 1] * f2; },
                                [&](int k) { return P->V.A2tP[k * NP + r]; });
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
                const double f1 = jj + 1 < T ? 1.0 : 0.0, f2 = jj + 2 < T ? 1.0 : 0.0;
                ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(acc, NP, g,
                            [&](int k) { return sNU[(jj + 1) * LDN + k] * f1; },
                            [&](int k) { return P->V.A1P[k * NP + r]; });
                if (var2)
                    ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(acc, NP, g,
                                [&](int k) { return sNU[(jj + 2) * LDN + k] * f2; },
                                [&](int k) { return P->V.A2P[k * NP + r]; });
                ft_d4 accq = {0, 0, 0, 0};                       // dense state weights: 2Q_j x_j as a product (Qf at the last stage)
                if (P->V.denseQ) {
                    const double* xj = zp + (size_t)(jj < T ? jj : T - 1) * s + m;
                    const double fq = jj + 1 < T ? 1.0 : 0.0, fqf = jj + 1 == T ? 1.0 : 0.0;
                    ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(accq, NP, g,
                                [&](int k) { return xj[k < n ? k : n - 1] * fq; },
                                [&](int k) { return P->V.Q2P[k * NP + r]; });
                    ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(accq, NP, g,
                                [&](int k) { return xj[k < n ? k : n - 1] * fqf; },
                                [&](int k) { return P->V.Qf2P[k * NP + r]; });
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int jo = 16 * A + g + 4 * rr;
                    if (jo < T && rok) {
                        const bool last = jo + 1 == T;
                        const double q2 = last ? P->M.Qf2[r] : P->M.Q2[r];
                        const double qx = P->V.denseQ ? accq[rr] : q2 * zp[jo * s + m + r];
                        double v = qx + (last ? P->M.qfl[r] : P->M.ql[r]) + sNU[jo * LDN + r] - acc[rr];
                        if (last && P->M.has_xf) v += sNU[T * LDN + r];
                        rdx[jo * n + r] = v;
                        phx[jo * n + r] = v * ft_rcp(q2);              // Phi^-1 r_d on x_j (dense weights: redone below)
                        acc_d += v * v;
                    }
                }
            } else {
                // ---- r_d on u_j: 2R u + r + k P'd - B' nu_j ; barrier Hessian and Phi^-1 on the way
                const int it3 = item - nC - nB_, J = it3 / TA, A = it3 - J * TA;
                const int j = 16 * A + c, q = 16 * J + c;
                const bool qok = q < m;
                ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(acc, NP, g,
                            [&](int k) { return sNU[j * LDN + k]; },
                            [&](int k) { return P->V.BmP[(size_t)k * MP + q]; });
                ft_d4 accr = {0, 0, 0, 0};
                if (DR) {                                           // dense R: (2R u_j)[q], one more stage-batched product
                    const double* uj = zp + (size_t)(j < T ? j : T - 1) * s;
                    ft_vec_gemm<12>(accr, MP, g,
                                [&](int k) { return uj[k < m ? k : m - 1]; },
                                [&](int k) { return P->V.R2P[(size_t)k * MP + q]; });
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int jo = 16 * A + g + 4 * rr;
                    if (jo < T && qok) {
                        const double u = zp[jo * s + q];
                        const double dp = ft_rcp(P->M.umax[q] - u), dm = ft_rcp(u - P->M.umin[q]);
                        const double hs = P->kbar * (dp * dp + dm * dm);
                        const double rt = P->M.R2[q] + hs;
                        if (!DR && (!(rt > 0.0) || isinf(rt))) bad = 1;     // (dense R: the factorisation below finds a bad pivot)
                        if (DR && !(hs >= 0.0 && !isinf(hs))) bad = 1;
                        const double rd = (DR ? accr[rr] : P->M.R2[q] * u) + P->M.rl[q] + P->kbar * (dp - dm) - acc[rr];
                        hess[jo * m + q] = hs;
                        winv[jo * m + q] = ft_rcp(rt);
                        rdu[jo * m + q] = rd;
                        acc_d += rd * rd;
                    }
                }
            }
        }
    }
    FtResid out;
    out.rp2 = ft_block_sum<NW>(acc_p, red);
    out.rho2 = ft_block_sum<NW>(acc_d, red) + out.rp2;
    out.bad = ft_block_sum<NW>((double)bad, red);
    return out;
}

// P5 of the kernel below (d_z, line search, update of z and nu of problem p) as a function of its own.  Workgroup-collective.
template <typename R, int NB, int NW, bool DR>
__device__ __noinline__ FtStep ft_phase_update(FtKP Pin, int p, double rho2) {
    FT_VIEW(R, NB, NW, DR);
    p = __builtin_amdgcn_readfirstlane(p);
    double* zp = P->zout + (size_t)p * Nz;
    double be = 0.0, e2 = 0.0;
    for (int item = wv; item < NB * TA + (DR ? 0 : mb * TA); item += NW) {   // (d_nu is in the staging area: written by P4)
        ft_d4 acc = {0, 0, 0, 0};
        if (item < NB * TA) {
            // ---- d_x_j = (2Q_j)^-1 (-r_d[x_j] - d_nu_{j-1} + A1' d_nu_j + A2' d_nu_{j+1}  [- d_nu_T])
            const int Jr = item / TA, A = item - Jr * TA;
            const int jj = 16 * A + c, r = 16 * Jr + c;
            const bool rok = r < n;
            const double f1 = jj + 1 < T ? 1.0 : 0.0, f2 = jj + 2 < T ? 1.0 : 0.0;
            ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(acc, NP, g,
                        [&](int k) { return sNU[(jj + 1) * LDN + k] * f1; },
                        [&](int k) { return P->V.A1P[k * NP + r]; });
            if (var2)
                ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(acc, NP, g,
                            [&](int k) { return sNU[(jj + 2) * LDN + k] * f2; },
                            [&](int k) { return P->V.A2P[k * NP + r]; });
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int jo = 16 * A + g + 4 * rr;
                if (jo < T && rok) {
                    const bool last = jo + 1 == T;
                    double v = -rdx[jo * n + r] - sNU[jo * LDN + r] + acc[rr];
                    if (last && P->M.has_xf) v -= sNU[T * LDN + r];
                    if (P->V.denseQ) phx[jo * n + r] = v;                                  // dense weights: d_x = (2Q_j)^-1 v below
                    else rdx[jo * n + r] = v * ft_rcp(last ? P->M.Qf2[r] : P->M.Q2[r]);       // reuse as d_x
                }
            }
        } else {
            // ---- d_u_j = Rt_j^-1 (B' d_nu_j - r_d[u_j]) ; e = k P'DP d_z for the line search
            const int it3 = item - NB * TA, J = it3 / TA, A = it3 - J * TA;
            const int j = 16 * A + c, q = 16 * J + c;
            const bool qok = q < m;
            ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(acc, NP, g,
                        [&](int k) { return sNU[j * LDN + k]; },
                        [&](int k) { return P->V.BmP[(size_t)k * MP + q]; });
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
    if (DR) {
        // ---- dense R: d_u_j = Rt_j^-1 (B' d_nu_j - r_d[u_j]) = Z_j d_nu_j - t_j
        for (int idx = tid; idx < T * m; idx += NT) {
            const int j = idx / m;
            const double* zr = ztw + (size_t)idx * ZLD;
            double du = -zr[n];
            for (int r = 0; r < n; ++r) du = fma(zr[r], sNU[j * LDN + r], du);
            const double rd = rdu[idx];
            const double e = hess[idx] * du;
            be += rd * e;
            e2 += e * e;
            rdu[idx] = du;                                      // reuse as d_u
        }
    }
    if (P->V.denseQ) {
        __syncthreads();
        for (int item = wv; item < NB * TA; item += NW) {
            const int Jr = item / TA, A = item - Jr * TA;
            const int jj = 16 * A + c, r = 16 * Jr + c;
            const double* vj = phx + (size_t)(jj < T ? jj : T - 1) * n;
            const double fq = jj + 1 < T ? 1.0 : 0.0, fqf = jj + 1 == T ? 1.0 : 0.0;
            ft_d4 acc = {0, 0, 0, 0};
            ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(acc, NP, g,
                        [&](int k) { return vj[k < n ? k : n - 1] * fq; },
                        [&](int k) { return P->V.XP[k * NP + r]; });
            ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(acc, NP, g,
                        [&](int k) { return vj[k < n ? k : n - 1] * fqf; },
                        [&](int k) { return P->V.XfP[k * NP + r]; });
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int jo = 16 * A + g + 4 * rr;
                if (jo < T && r < n) rdx[jo * n + r] = acc[rr];                       // d_x
            }
        }
    }
    const double beta_e = ft_block_sum<NW>(be, red);
    const double eps2 = ft_block_sum<NW>(e2, red);
    // closed form of backtracking_inf_newton.m:2-11 with the frozen barrier gradient:
    // ||r(t)||^2 - ((1-al t) rho)^2 = t * gq(t)
    double t = 1.0;
    int collapsed = 0;
    {
        const double al = 1e-4;
        int halv = 0;
        while (true) {
            const double gq = (t - 2.0 + 2.0 * al - al * al * t) * rho2 - 2.0 * (1.0 - t) * beta_e + t * eps2;
            if (gq <= 0.0) break;
            t *= 0.5;
            if (++halv >= FT_MAX_HALVINGS) { t = 0.0; collapsed = 1; break; }
        }
    }
#pragma unroll 8
    for (int idx = tid; idx < Nz; idx += NT) {
        const int j = idx / s, e = idx - j * s;
        const double dv = e < m ? rdu[j * m + e] : rdx[j * n + (e < m ? 0 : e - m)];
        zp[idx] += t * dv;
    }
    for (int idx = tid; idx < nbn; idx += NT) { const int j = idx / n; nu[idx] += t * sNU[j * LDN + idx - j * n]; }
    FtStep out; out.t = t; out.collapsed = collapsed;
    return out;
}

// NL: live rows of the last 16-row block of a stage, n - 16 (NB - 1), when known at compile time (the AO sizes), else -1
template <typename R, int NB, int NW, int NL, bool DR = false>
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
    FT_TL_BEGIN();
    const bool var2 = M.var2 != 0;
    const int mb = V.mb, cn = V.cn, nl = V.nl;

    const FtLds LL = ft_lds_layout(NB, mb, NW, (int)sizeof(R), nb, DR ? ft_pr_doubles(n, m) : 0);
    R* sBT = (R*)(smem + LL.bt);
    R* sSLOT = (R*)(smem + LL.slot);
    R* sLT = (R*)(smem + LL.lt);
    R* sWT = (R*)(smem + LL.wt);
    R* sYSH = (R*)(smem + LL.ysh);
    R* sWL = (R*)(smem + LL.wl);
    R* sXV = (R*)(smem + LL.xv);
    R* sPART = (R*)(smem + LL.part);
    double* red = (double*)(smem + LL.red);
    int* sflag = (int*)(smem + LL.flag);
    // residual phases: nu / d_nu as [stage][state] (leading dimension NP + 1: conflict-free transposed reads) in the
    // space of the U slots, which only the factor phase uses
    double* sNU = (double*)(smem + LL.slot);
    constexpr int LDN = 16 * NB + 1;
    const int TA = (nb + 15) / 16, NUROWS = 16 * TA + 2, MP = 16 * mb;

    const FtWs L = ft_ws_layout(n, m, T, nb, NB, (int)sizeof(R), DR ? 1 : 0);
    double* wsp = P.ws + (size_t)blockIdx.x * P.ws_stride;
    double* ztw = wsp + L.zt;                                       // dense R: [Rt_j^-1 B' | Rt_j^-1 r_d[u_j]] per stage
    const int ZLD = n + 1;
    double* b = wsp + L.b;
    double* nu = wsp + L.nu;
    double* hess = wsp + L.hess;
    double* winv = wsp + L.winv;
    double* rdu = wsp + L.rdu;
    double* rdx = wsp + L.rdx;
    double* phx = wsp + L.phx;
    double* rp = wsp + L.rp;
    double* yv = wsp + L.y;
    R* fac = (R*)(wsp + L.fac);
    R* gws = (R*)(wsp + L.gt);
    constexpr int STAGE_TILES = 3 * NB * NB, REC_TILES = 3 * NB;       // factor stream: one record per 16-row block
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

    const int nitems = P.list ? *P.nlist : P.batch;
    for (int item_p = blockIdx.x; item_p < nitems; item_p += gridDim.x) {
        int p = item_p;
        bool cont = false;                                           // continue behind the panel path's first step
        bool gen = false;                                            // ... behind one step of the one-wavefront kernel (FT_LIST_GENERAL)
        if (P.list) { const int e = P.list[item_p]; p = e & (FT_LIST_GENERAL - 1); cont = !(e & FT_LIST_HANDED); gen = (e & FT_LIST_GENERAL) != 0; }
        double* zp = P.zout + (size_t)p * Nz;
        const double* x0v = P.x0 + (size_t)p * n;
        const double* x0pv = P.x0p ? P.x0p + (size_t)p * n : nullptr;
        __syncthreads();
        FT_T0();
        // ================= P0: start point, nu, b  (fast_mpc_init.m:12-27, fast_mpc_eq_const.m:39,44,47,68)
        if (!cont) {
#pragma unroll 8
            for (int idx = tid; idx < Nz; idx += NT) {
                const int e = idx % s;
                zp[idx] = P.zinit ? P.zinit[(size_t)p * Nz + idx] : (e < m ? M.umid[e] : M.xmid[e - m]);
            }
        }
        for (int idx = tid; idx < nbn; idx += NT) {
            // (continuation: nu+ of the first step sits in the panel workspace, [panel][row][16 problems])
            nu[idx] = gen ? P.nuout[(size_t)p * nbn + idx]
                          : cont ? P.nuws[((size_t)(p >> 4) * nbn + idx) * 16 + (p & 15)] : (P.nu0 ? P.nu0[(size_t)p * nbn + idx] : 0.0);
            const int i = idx / n, r = idx - i * n;
            double v = (i < T && P.w) ? P.w[(size_t)p * T * n + idx] : 0.0;
            if (i == 0) {
#pragma unroll 9
                for (int q = 0; q < n; ++q) v += M.A1t[q * n + r] * x0v[q];
                if (var2 && x0pv) {
#pragma unroll 9
                    for (int q = 0; q < n; ++q) v += M.A2t[q * n + r] * x0pv[q];
                }
            } else if (i == 1 && i < T && var2) {
#pragma unroll 9
                for (int q = 0; q < n; ++q) v += M.A2t[q * n + r] * x0v[q];
            }
            if (i == T) v = M.xf[r];
            b[idx] = v;
        }
        if (P.step && !gen)                                          // (gen: the record of the first step stands)
            for (int idx = tid; idx < P.step_ld; idx += NT) P.step[(size_t)p * P.step_ld + idx] = (cont && idx == 0) ? 1.0 : -1.0;
        __syncthreads();

        int st = (gen && P.status) ? P.status[p] : FMPC_OK, nsteps = cont ? 1 : 0;     // (gen: a line-search warning of the first step is kept)
        FT_TICK(0);
        for (int it = cont ? 1 : 0; it < P.max_iter; ++it) {
            // ================= P1: residuals.  Every product is a GEMM with the horizon stages as one dimension
            // (out[stage][entry] = sum_k X[k][stage] Z[k][entry]) on the fp64 matrix cores; Z (B, A1, A2 and transposes)
            // is read from L2 with 128-byte rows, the epilogues read and write along the entries of a stage.
            const FtResid rs_ = ft_phase_resid<R, NB, NW, DR>(ft_params(), p);      // (not inlined: see ft_phase_factor)
            const double rp2 = rs_.rp2, rho2 = rs_.rho2, badsum = rs_.bad;
            // early exit, tested before the step (inf_newton_solver.m:19-22)
            if (sqrt(rho2) <= 1e-6 && sqrt(rp2) <= 1e-8) break;
            if (badsum > 0.0) { st = FMPC_E_NOT_PD_PHI; break; }
            if (V.denseQ) {
                // Phi^-1 r_d on the states with dense weights: phx_j = (2Q_j)^-1 r_d[x_j], one more stage-batched product
                for (int item = wv; item < NB * TA; item += NW) {
                    const int Jr = item / TA, A = item - Jr * TA;
                    const int jj = 16 * A + c, r = 16 * Jr + c;
                    const double* rj = rdx + (size_t)(jj < T ? jj : T - 1) * n;
                    const double fq = jj + 1 < T ? 1.0 : 0.0, fqf = jj + 1 == T ? 1.0 : 0.0;
                    ft_d4 acc = {0, 0, 0, 0};
                    ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(acc, NP, g,
                                [&](int k) { return rj[k < n ? k : n - 1] * fq; },
                                [&](int k) { return V.XP[k * NP + r]; });
                    ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(acc, NP, g,
                                [&](int k) { return rj[k < n ? k : n - 1] * fqf; },
                                [&](int k) { return V.XfP[k * NP + r]; });
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int jo = 16 * A + g + 4 * rr;
                        if (jo < T && r < n) phx[jo * n + r] = acc[rr];
                    }
                }
                __syncthreads();
            }
            if (DR) {
                // ================= dense R: factor Rt_j, [Rt_j^-1 B' | Rt_j^-1 r_d[u_j]] of every stage (the LDS copy of nu is overwritten:
                // nobody reads it before P4 writes d_nu there)
                __syncthreads();
                const int badr = ft_dense_r((double*)smem, V.R2P, MP, M.Bt, hess, rdu, ztw, n, m, T);
                if (badr) { st = FMPC_E_NOT_PD_PHI; break; }
            }
            FT_TICK(1);

            // ================= P2: rhs_i = r_p,i - (C Phi^-1 r_d)_i   (into yv)
            for (int item = wv; item < NB * TA; item += NW) {
                const int Jr = item / TA, A = item - Jr * TA;
                const int i = 16 * A + c, r = 16 * Jr + c;
                const bool rok = r < n;
                ft_d4 acc = {0, 0, 0, 0};
                const size_t iu = (size_t)(i < T ? i : T - 1) * m;
                const double* ph1 = phx + (size_t)((i >= 1 && i < T) ? i - 1 : 0) * n;
                const double* ph2 = phx + (size_t)((i >= 2 && i < T) ? i - 2 : 0) * n;
                const double f1 = (i >= 1 && i < T) ? 1.0 : 0.0, f2 = (i >= 2 && i < T) ? 1.0 : 0.0;
                if (DR)
                    ft_vec_gemm<12>(acc, MP, g,
                                [&](int k) { const int kc = k < m ? k : m - 1; return ztw[(iu + kc) * ZLD + n]; },
                                [&](int k) { return V.BtP[(size_t)k * NP + r]; });
                else
                    ft_vec_gemm<12>(acc, MP, g,
                                [&](int k) { const int kc = k < m ? k : m - 1; return rdu[iu + kc] * winv[iu + kc]; },
                                [&](int k) { return V.BtP[(size_t)k * NP + r]; });
                ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(acc, NP, g,
                            [&](int k) { return ph1[k < n ? k : n - 1] * f1; },
                            [&](int k) { return V.A1tP[k * NP + r]; });
                if (var2)
                    ft_vec_gemm<(NP / 4 < 12 ? NP / 4 : 10)>(acc, NP, g,
                                [&](int k) { return ph2[k < n ? k : n - 1] * f2; },
                                [&](int k) { return V.A2tP[k * NP + r]; });
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
            // ================= S pre-pass: the initial diagonal blocks S0_i = Y_ii const + B W_i B' (upper-triangular tiles,
            // rhs_i in column n) of EVERY block row, all independent, ahead of the serial factorisation.  The B' tiles are
            // in LDS only for this (the factor phase reuses the space); Phi^-1 of FT_GCH stages at a time in LDS.
            FT_TICK(12);
            ft_phase_spre<R, NB, NW, DR>(ft_params());                 // (not inlined: see ft_phase_factor)
            __syncthreads();
            FT_TICK(13);
            // zero the three U slots: stages 0 and 1 then need no special cases
            constexpr bool TS = !(sizeof(R) == 8 && NW == 2);       // two U slots (ft_u_slots): see below
            for (int q = tid; q < (TS ? 2 : 3) * NQ * FT_TILE; q += NT) sSLOT[q] = (R)0;
            if (tid == 0) sflag[0] = 0;
            __syncthreads();
            FT_TICK(2);

            // ================= P3: factor + forward sweep
            // TWO U slots: Ua = U1_{i-1}, Ub = U2_{i-1}.  The third term of S_i, U2_{i-2}' U2_{i-2}, is applied ONE STAGE AHEAD:
            // stage i - 1 holds U2_{i-2} as its Ub and subtracts Ub'Ub from the S0 tiles of stage i, which are already in
            // registers by then (requested two stages ahead).  After phase A both slots are dead and take U1_i, U2_i in place --
            // no rotation, and a third of the factor phase's LDS is gone (95 -> 70 KB at n = 65: two workgroups per CU).
            // !TS (fp64 with 2 wavefronts, short of registers for the extra tile set): three slots, Uc = U2_{i-2} kept in LDS
            // and subtracted in the stage itself.
            // (a function of its own, not inlined: inside the kernel body the factor phase's ~150 registers of tiles compete with
            // everything else that is live there -- the n = 65 instance spilled 928 bytes per lane, ~300 scratch accesses per stage)
            const bool fail = ft_phase_factor<R, NB, NW, NL, DR>(ft_params());
            if (fail) { st = FMPC_E_NOT_PD_SCHUR; break; }
            __syncthreads();                                           // the factor stream and y are in HBM (same workgroup reads them)
            FT_TICK(6);

            // ================= P4: backward sweep, d_nu_i = R_i^-1 (y_i - U1_i d_nu_{i+1} - U2_i d_nu_{i+2}), one 16-row
            // block at a time from the bottom: x_kb = RI(kb) (y_kb - sum of tile x vector products).  16 consecutive threads
            // read a tile row (coalesced) and sum over it by DPP; the tiles of the NEXT block row are requested before the
            // current one is reduced.  x of stage i lives in buffer i % 3.
            ft_backward<R, NB, NW>(fac, yv, sXV, sPART, sNU, n, nb, NUROWS);
            __syncthreads();
            FT_TICK(7);

            // ================= P5: d_z, line-search scalars, update (the same stage-batched GEMMs with d_nu)
            double t;
            {
                const FtStep sp_ = ft_phase_update<R, NB, NW, DR>(ft_params(), p, rho2);   // (not inlined: see ft_phase_factor)
                t = sp_.t;
                if (sp_.collapsed) st = FMPC_W_LINESEARCH;
            }
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
    FT_TL_END();
}

// ---------------------------------------------------------------- host side
template <typename R, int NB, int NW, int NL = -1, bool DR = false>
static hipError_t ft_launch(const FtParams& P, int grid, size_t lds, hipStream_t stream) {
    hipLaunchKernelGGL((fmpc_newton_tiled<R, NB, NW, NL, DR>), dim3(grid), dim3(NW * 64), lds, stream, P);
    return hipGetLastError();
}
template <typename R, int NB, int NW, int NL = -1, bool DR = false>
static hipError_t ft_prepare(size_t lds) {
    return hipFuncSetAttribute((const void*)fmpc_newton_tiled<R, NB, NW, NL, DR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

// instantiations: fp64 for n <= 79 (NB <= 5; 4 and 5 on request), fp32 for n <= 111 (NB <= 7; 6 and 7 on request); NW wavefronts per problem
#define FT_DISPATCH(fn, ...)                                                                   \
    /* dense R: fp64, eight wavefronts (the per-stage m x m factorisation is workgroup-wide vector work) */                  \
    if (denseR) {                                                                              \
        if (is_float || NW != 8) return hipErrorInvalidValue;                                  \
        if (NB == 1) return fn<double, 1, 8, -1, true>(__VA_ARGS__);                           \
        if (NB == 2) return fn<double, 2, 8, -1, true>(__VA_ARGS__);                           \
        if (NB == 3) return fn<double, 3, 8, -1, true>(__VA_ARGS__);                           \
        return hipErrorInvalidValue;                                                           \
    }                                                                                          \
    /* the AO sizes (n = 27 Zernike modes; n = 65: radial order 10) with their block structure at compile time, for the   \
       default wavefront counts only; other wavefront counts use the run-time form.  (An instance <double, 2, 4, 11> existed  \
       until round 5: miscompiled by round 2's monolithic build, never root-caused, bitwise equal to the run-time form since  \
       the phases are separate functions, 3-4 % faster -- removed rather than kept as a switch; docs/DESIGN_HISTORY.md.)    */ \
    if (!is_float && nlast == 11 && NB == 2 && NW == 2) return fn<double, 2, 2, 11>(__VA_ARGS__);  \
    if (is_float && nlast == 11 && NB == 2 && NW == 4) return fn<float, 2, 4, 11>(__VA_ARGS__);    \
    if (is_float && nlast == 1 && NB == 5 && NW == 8) return fn<float, 5, 8, 1>(__VA_ARGS__);      \
    if (is_float && nlast == 1 && NB == 5 && NW == 4) return fn<float, 5, 4, 1>(__VA_ARGS__);      \
    if (!is_float) {                                                                           \
        if (NB == 1 && NW == 2) return fn<double, 1, 2>(__VA_ARGS__);                          \
        if (NB == 2 && NW == 2) return fn<double, 2, 2>(__VA_ARGS__);                          \
        if (NB == 3 && NW == 2) return fn<double, 3, 2>(__VA_ARGS__);                          \
        if (NB == 1 && NW == 4) return fn<double, 1, 4>(__VA_ARGS__);                          \
        if (NB == 2 && NW == 4) return fn<double, 2, 4>(__VA_ARGS__);                          \
        if (NB == 3 && NW == 4) return fn<double, 3, 4>(__VA_ARGS__);                          \
        /* fp64 on request where the fp32 factor is the default (47 < n <= 79, fmpc_set_precision): one workgroup of 8 per CU */ \
        if (NB == 4 && NW == 8) return fn<double, 4, 8>(__VA_ARGS__);                          \
        if (NB == 5 && NW == 8) return fn<double, 5, 8>(__VA_ARGS__);                          \
    } else {                                                                                   \
        if (NB == 1 && NW == 2) return fn<float, 1, 2>(__VA_ARGS__);                           \
        if (NB == 2 && NW == 2) return fn<float, 2, 2>(__VA_ARGS__);                           \
        if (NB == 2 && NW == 4) return fn<float, 2, 4>(__VA_ARGS__);                           \
        if (NB == 3 && NW == 4) return fn<float, 3, 4>(__VA_ARGS__);                           \
        if (NB == 4 && NW == 4) return fn<float, 4, 4>(__VA_ARGS__);                           \
        if (NB == 4 && NW == 8) return fn<float, 4, 8>(__VA_ARGS__);                           \
        if (NB == 5 && NW == 4) return fn<float, 5, 4>(__VA_ARGS__);                           \
        if (NB == 5 && NW == 8) return fn<float, 5, 8>(__VA_ARGS__);                           \
        /* 79 < n <= 111 with the fp32 factor, on request (round 5): one workgroup of 8 per CU */ \
        if (NB == 6 && NW == 8) return fn<float, 6, 8>(__VA_ARGS__);                           \
        if (NB == 7 && NW == 8) return fn<float, 7, 8>(__VA_ARGS__);                           \
    }                                                                                          \
    return hipErrorInvalidValue;

// Wavefronts per problem.  Few waves per problem = many problems per CU (the factorisation is a chain of dependent
// steps that only other problems can hide) at two waves per SIMD, i.e. the full 256-register budget per lane.
static int ft_default_nw(int NB, int is_float) {
    // (n = 65, NB = 5: since the factor phase keeps two U slots instead of three its LDS is 70 KB, so TWO workgroups of 4
    // wavefronts share a CU -- one problem's pivot chains and barriers overlap the other's products: 7.98 ms per Newton step
    // of BASELINE configs[4] against 9.11 ms with one workgroup of 8)
    int NW = is_float ? (NB == 5 ? 4 : (NB >= 4 ? 8 : (NB >= 2 ? 4 : 2))) : (NB >= 4 ? 8 : (NB >= 3 ? 4 : 2));
    const char* e = getenv("FMPC_TILED_NW");                      // experiments
    if (e && (e[0] == '2' || e[0] == '4' || e[0] == '8')) {
        const int w = e[0] - '0';
        const bool ok = is_float ? ((NB <= 2 && w == 2) || (NB >= 2 && NB <= 5 && w == 4) || (NB >= 4 && w == 8))        /* (NB = 6, 7: 8 only) */
                                 : (NB <= 3 && (w == 2 || w == 4));
        if (ok) NW = w;
    }
    return NW;
}

bool fmpc_tiled_supports(int n, int m, int nb, int is_float, int* NB_out, int* NW_out, int denseR) {
    const int NB = n / 16 + 1;                                     // 16 NB >= n + 1
    if (NB > (is_float ? 7 : 5)) return false;
    if (denseR && (is_float || NB > 3)) return false;
    const int NW = denseR ? 8 : ft_default_nw(NB, is_float);
    const int mb = (m + 15) / 16;
    if (ft_lds_layout(NB, mb, NW, is_float ? 4 : 8, nb, denseR ? ft_pr_doubles(n, m) : 0).total > 160 * 1024) return false;
    if (NB_out) *NB_out = NB;
    if (NW_out) *NW_out = NW;
    return true;
}
size_t fmpc_tiled_lds_bytes(int NB, int mb, int NW, int is_float, int nb, size_t pr_doubles) { return ft_lds_layout(NB, mb, NW, is_float ? 4 : 8, nb, pr_doubles).total; }
static int ft_nlast(int n, int NB) {
    const char* e = getenv("FMPC_TILED_GENERIC");                 // experiments: the instance with the block structure at run time
    return (e && e[0] == '1') ? -1 : n - 16 * (NB - 1);
}
hipError_t fmpc_tiled_prepare(int n, int NB, int NW, int is_float, size_t lds_bytes, int denseR) {
    const int nlast = ft_nlast(n, NB);
    FT_DISPATCH(ft_prepare, lds_bytes)
}
hipError_t fmpc_launch_tiled(const FtParams& P, int NB, int NW, int is_float, int grid, size_t lds_bytes, hipStream_t stream) {
    const int nlast = ft_nlast(P.M.n, NB);
    const int denseR = P.V.denseR;
    FT_DISPATCH(ft_launch, P, grid, lds_bytes, stream)
}
