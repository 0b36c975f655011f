// CDNA4 fastMPC Newton kernel, one WAVEFRONT per problem, for 16 < n < 28 (n = 27 Zernike modes).
//
// Same algorithm and reference correspondence as fmpc_kernel_generic.hip (inf_newton_solver.m:10-41);
// what changes is the mapping to the machine:
//   * every matrix product runs on the fp64 matrix cores (v_mfma_f64_16x16x4_f64):
//       - the block products of the block-penta-diagonal Cholesky, kept in the transposed form
//           U_{i,i+1} = L_ii^-1 (Y_{i,i+1} - U_{i-1,i}' U_{i-1,i+1}),  U_{i,i+2} = L_ii^-1 Y_{i,i+2},
//           S_ii = Y_ii + B Rt_i^-1 B' - U_{i-1,i}' U_{i-1,i} - U_{i-2,i}' U_{i-2,i}
//         so that every product is X'Z (contraction over the ROW index of both operands): an MFMA
//         accumulator tile (row = 4*reg + lane/16, col = lane%16) is then directly the A and the B
//         operand of the next product -- no LDS round trip between products;
//       - the applications of C and C' to the stacked vectors (residuals r_d, r_p, the right-hand
//         side, d_z): the T horizon stages are the N dimension of a GEMM (32 stage columns/pass).
//   * tiles are 32 x 32 (2 x 2 subtiles of 16 x 16); column n of the U tiles carries y_i, so the
//     forward substitution of the right-hand side rides in the padding of the MFMA tiles.
//   * potrf(S_ii) and the triangular solves L^-1 [M1 | rhs | Y2 | rhs] are ONE fused column loop on
//     the VALU: lane r holds row r of S (right-looking potrf) and lane c holds column c of the
//     right-hand sides; each L[c][k] is broadcast once with v_readlane and feeds both updates.
//   * the factor (L, U1, U2 per stage) is streamed to HBM in the forward sweep and read back once,
//     in reverse, by the backward sweep.
// 8 waves (= 8 problems) per workgroup share B' in LDS; each wave owns two 32 x 29 LDS tiles used
// only to change layout between MFMA tiles and the row/column-per-lane VALU layouts.  Each phase is
// a separate non-inlined function (own register allocation); parameters come from the kernarg
// segment through scalar loads.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include "fmpc_device.h"
#include "../../include/fastmpc.h"

#ifndef FW_WAVES
#define FW_WAVES 8
#endif
#define FW_THREADS (FW_WAVES * 64)
#define FW_MAX_HALVINGS 64
#define FW_LDB 33                       // leading dimension of B' in LDS (odd)
#define FW_KCH 6                        // k-steps per prefetch chunk of the K = m products
// cold path: LDS region behind B' = [2 stage buffers of the shared sweeps][8 per-wave vectors rhs->y->d_nu]
#define FW_VEC_STRIDE 840               // doubles per wave (nb*n <= 840 checked on the host)
// ... followed by [cu | hc | wc | ubar] (4*mp doubles) for the cold step's epilogue: LDS reads do not queue behind
// the global stores of the previous column block (vmcnt counts stores too)
#define FW_MODE_NORMAL 0                // every problem factors its own Y
#define FW_MODE_SHARED 1                // first Newton step from a cold start uses the handle's shared factor
#define FW_MODE_EXPORT 2                // compute that shared factor (batch 1) and publish it

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
typedef __attribute__((address_space(3))) double* fw_lds_t;
typedef const __attribute__((address_space(3))) double* fw_clds_t;
typedef double d2v __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) d2v* fw_c2lds_t;
#define FW_FN __device__ __noinline__
#define FW_IN __device__ __forceinline__          // sub-phase of a merged phase function (below)

// Diagnostic build only (-DFW_TIMING): per-phase cycle totals, never in the shipped library.
#ifdef FW_TIMING
__device__ unsigned long long fw_timing[16];
#define FW_T0() unsigned long long _t0 = __builtin_readcyclecounter(), _t1, _a0 = 0, _a1 = 0, _a2 = 0, _a3 = 0
#define FW_TICK(k) do { _t1 = __builtin_readcyclecounter(); _a##k += _t1 - _t0; _t0 = _t1; } while (0)
#define FW_TFLUSH(base) do { if ((threadIdx.x & 63) == 0) { atomicAdd(&fw_timing[base], _a0); atomicAdd(&fw_timing[base + 1], _a1); atomicAdd(&fw_timing[base + 2], _a2); atomicAdd(&fw_timing[base + 3], _a3); } } while (0)
extern "C" int fmpc_debug_timing(unsigned long long* out) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fw_timing), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(fw_timing), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#else
#define FW_T0()
#define FW_TICK(k)
#define FW_TFLUSH(base)
#endif

template <int N>
struct FwCfg {
    static_assert(N > 16 && N < 28, "wave kernel: 16 < n < 28");
    static constexpr int RC = N;                  // tile column that carries the rhs / y_i
    static constexpr int LD = 29;                 // LDS tile leading dimension (odd); col 28 = dump
    static constexpr int LDG = (N + 2) & ~1;      // HBM factor tile leading dimension (even)
    // Factor stream of a stage (per-problem factor): L as its lower triangle packed by columns -- column j = L[j..N-1][j] at
    // LOFF(j), N (N + 1) / 2 doubles instead of a N x LDG tile (the upper half was never read: 0.38 GB of 3.2 GB per 2048-problem
    // launch, written and read back) -- followed by the U1 tile.  The exported shared factor keeps full tiles.
    static constexpr int LPACK = (N * (N + 1) / 2 + 1) & ~1;
    __host__ __device__ static constexpr int LOFF(int j) { return j * N - j * (j - 1) / 2; }
    static constexpr int TILE = 32 * LD;
    // Factor stream of a stage: the wave's LDS tile U1 | y as [row j][column c], leading dimension LD, and the strict lower triangle
    // of L FOLDED into (N + 1) / 2 rows of LDF doubles, both dumped LINEARLY, 16 bytes per lane per instruction: 7 + 3 instructions
    // of 1 KB per stage (10 KB; round 4 dumped the full L tile: 7 + 7; round 3's packed form took 27 + 27 instructions of 216
    // bytes).  A CU's load / store path takes a wave instruction every 12-17 cycles whatever its width
    // (scripts/probes/vmem_issue_rate.hip).  DUMP doubles: N rows rounded up to whole wave instructions of 128 doubles.
    static constexpr int DUMP = ((N * LD + 127) / 128) * 128;
    // The fold: rows r and N - 1 - r share a folded row of N - 1 entries (row min(r, N-1-r): the shorter of the two first).
    // Written without selects or per-entry predicates (fw_phase_factor: every lane writes its whole row in DESCENDING order, what
    // falls on another row's slots is overwritten by their owner afterwards); read back with per-lane bases and compile-time
    // offsets (LBASE).  -4 % launch time, -0.5 GB of traffic per 2000-problem launch against the full tile (round 5)
    static constexpr int LDF = N;                                   // (odd)
    static constexpr int LFOLD = (((N + 1) / 2 * LDF) + 127) / 128 * 128;
    __host__ __device__ static constexpr int LBASE(int r) { return (r <= (N - 1) / 2 ? r * LDF : (N - 1 - r) * LDF + (N - 1 - r)); }
    static constexpr int FST = DUMP + LFOLD;
    static_assert(DUMP <= TILE && LFOLD <= TILE, "the dumps must stay inside a tile");
    static constexpr int PER_WAVE = 2 * TILE;                 // tA, tB
    static constexpr int IMG_D = 3 * 4 * 64;                  // subtiles (0,0),(0,1),(1,1)
    static constexpr int IMG_1 = 4 * 4 * 64;                  // full tile
    static constexpr int IMG_2 = N * 32;                      // [row j][col c] for column-per-lane loads
    // (round 4) the same subtile images with the registers of a lane in PAIRS, [subtile][r / 2][lane][r % 2]: one 16-byte load per
    // lane and pair -- 14 wave instructions per stage for the constant tiles instead of 28
    static constexpr int IMG_P = IMG_D + IMG_1 + IMG_2;             // offset of the paired copy of IMG_D, then of IMG_1
    static constexpr int IMG_STRIDE = 2 * (IMG_D + IMG_1) + IMG_2;
    static constexpr int Y2LDS = (N * N + 1) & ~1;                  // LDS copy of the block Y_{i,i+2} shared by the workgroup (doubles)
};

// per-wave workspace in HBM (doubles)
struct FwWs { size_t b, nu, rdu, rdx, rp, rhs, y, dnu, phx, rs, fac, total; };
__host__ __device__ static inline FwWs fw_ws_layout(int N, int m, int mp, int T, int nb, int LDG, int FST = 0) {
    FwWs L; size_t o = 0;
    const size_t nbn = ((size_t)nb * N + 1) & ~(size_t)1;
    L.b = o; o += nbn;  L.nu = o; o += nbn;
    L.rdu = o; o += (size_t)T * m;   L.rdx = o; o += nbn;
    L.rp = o; o += nbn;  L.rhs = o; o += nbn;  L.y = o; o += nbn;  L.dnu = o; o += nbn;
    L.phx = o; o += nbn;
    L.rs = o; o += (size_t)nb * 32;
    o = (o + 1) & ~(size_t)1;
    L.fac = o; o += (size_t)nb * (FST > 2 * N * LDG ? FST : 2 * N * LDG);    // L and U1 per stage (U2 = L^-1 Y2 is not streamed: see fw_phase_backward)
    L.total = (o + 15) & ~(size_t)15;
    return L;
}

// Cold-start constants (first Newton step from the mid-box start u = ubar, x = xbar; all k-dependent
// pieces are rebuilt by the host when k changes).  Offsets in doubles into FwParams::cold.
//   cu = 2R ubar + r + k P'd      hc = k diag(P'DP)      wc = 1/(2R + hc)      G = B diag(wc) B'
//   cbu = B (wc o cu)             cp_i = xbar - B ubar - [i>=1] A1 xbar - [i>=2] A2 xbar
//   a = hc o wc:  Ma = B diag(a) B', Ma2 = B diag(a^2) B', va = B (a o cu), va2 = B (a^2 o cu),
//   sa = sum a cu^2, sa2 = sum a^2 cu^2      (line-search dots as 27-dimensional quadratic forms)
struct FwCold {
    int cu, hc, wc, G, cbu, cp0, cp1, cp2, Ma, Ma2, va, va2, sa, total;
};
__host__ __device__ static inline FwCold fw_cold_layout(int N, int mp) {
    FwCold c; int o = 0;
    c.cu = o; o += mp; c.hc = o; o += mp; c.wc = o; o += mp;
    c.G = o; o += N * N + (N * N & 1);
    c.cbu = o; o += 32; c.cp0 = o; o += 32; c.cp1 = o; o += 32; c.cp2 = o; o += 32;
    c.Ma = o; o += N * N + (N * N & 1); c.Ma2 = o; o += N * N + (N * N & 1);
    c.va = o; o += 32; c.va2 = o; o += 32; c.sa = o; o += 2;      // sa, sa2
    c.total = o;
    return c;
}

// The kernel's ONLY parameter: phases re-read it from the kernarg segment (scalar loads).
struct FwParams {
    FmpcDevModel M;
    FwModel V;
    int batch, max_iter, step_ld;
    int mode;                       // FW_MODE_*
    double kbar;
    const double* x0; const double* x0p; const double* w; const double* zinit; const double* nu0;
    double* zout; double* nuout; int* status; int* iters; double* step;
    int zld;                        // doubles between the z rows of consecutive problems (T (n + m) unless fmpc_set_z_ld: flag mode only)
    double* ws; size_t ws_stride;
    double* sh_fac; double* sh_rs; int* sh_ok;     // shared (cold-start) factor owned by the handle
    const double* cold;                             // cold-start constants (FwCold layout), k-dependent
    // panel path (fmpc_kernel_panel.hip + fmpc_kernel_dz.hip ran before this launch): per problem ||r_p||^2 and a
    // lower bound of rho^2 (gate), per (panel, stage, problem) the partial ||e||^2 (epsp).  Non-null: decide the
    // step length of every problem first and solve only those whose decision is not clear-cut.
    const double* gate; const double* epsp; int* handed;
    const double* nuws;             // nu+ of the panel kernels, panel layout [panel][stage row][16]
    double* u0out;                  // optional: the first move u0 = z(1:m) of every problem (README.md:589), written here too
    // Newton budgets > 1 on the panel path run in two launches so that the few problems that go on are COMPACTED:
    // pphase 1 decides every problem (step length of the panel step, then the exit test of the next iteration from
    // rnp) and appends those that need this kernel to `list`; pphase 2 works through the list.  pphase 0: one launch
    // (budget 1: decide, and redo the handed-over problems right away).
    int pphase;
    const double* rnp;              // per (panel, stage, problem): partial ||r_d||^2 at the new point (fmpc_cold_dz<true>)
    int* list;                      // problem index, bit 30 set = handed over (to be redone from scratch)
    int flags;                      // experiment switches (environment FMPC_WAVE_FLAGS); 0 in production
    int* nflag;                             // flag mode behind the affine kernel: nflag[0] = running count of the problems the affine kernel has
                                    // flagged since the handle exists, nflag[1] = the count the last flag-mode launch has dealt with,
                                    // nflag[2] = its ticket.  Equal counts: nothing new is flagged, leave at once (two scalar loads, no store).
                                    // Nothing depends on the order of host calls, so a recorded graph replays it as it stands
    int u0_done;                    // panel path, first moves only: fmpc_cold_dz has written u0out itself (zout is a scratch
                                    // array that only the problems redone here touch)
};
#define FW_LIST_HANDED (1 << 30)
#define FW_LIST_GENERAL (1 << 29)     // (= FT_LIST_GENERAL of fmpc_tiled.h)

typedef const FwParams __attribute__((address_space(4))) * FwKP;
__device__ __forceinline__ FwKP fw_params() {
    return (FwKP)__builtin_amdgcn_kernarg_segment_ptr();
}

// A function argument arrives in VGPRs; make the kernarg pointer wave-uniform again so that
// every P->field is a scalar load.
__device__ __forceinline__ FwKP fw_uniform(FwKP P) {
    // (round 4: taking the pointer from __builtin_amdgcn_kernarg_segment_ptr() inside the non-inlined phase functions instead --
    //  it would need no register -- FAULTED on the device: in a callee the builtin is not the calling kernel's segment.  Not used.)
    const unsigned long long a = (unsigned long long)P;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (FwKP)(((unsigned long long)hi << 32) | lo);
}
// Element access as (wave-uniform base) + (32-bit BYTE offset): the form `global_load v, v_offset, s[base:base+1]` -- one offset register
// per access.  With an element index the compiler extends to 64 bits before scaling and keeps a 64-bit address per access.
__device__ __forceinline__ double fw_ldb(const double* base, unsigned boff) { return *(const double*)((const char*)base + boff); }
__device__ __forceinline__ void fw_stb(double* base, unsigned boff, double v) { *(double*)((char*)base + boff) = v; }
// 1/x: hardware estimate + two Newton steps (<= 1 ulp; the parity tolerance is 1e-9)
__device__ __forceinline__ double fw_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}

__device__ __forceinline__ double fw_readlane(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fw_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;                            // every lane holds the sum (fixed order: reproducible)
}
// 1/sqrt(d): hardware estimate + two Newton steps (full fp64 accuracy, no division)
__device__ __forceinline__ double fw_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}
__device__ __forceinline__ void fw_wave_fence() {          // order LDS traffic inside the wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void fw_mem_fence() {           // this wave's HBM writes -> its other lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Per-wave pointers, rebuilt in each phase from the kernarg parameters.
template <int N>
struct FwView {
    int m, mp, T, nb, s, has_xf, var2, fstride;
    double *zp, *b, *nu, *rdu, *rdx, *rp, *rhs, *yv, *dnu, *phx, *rsg, *fac;
    const double* zs;       // where the current iterate is read from: the caller's z_init until the first update has
                            // written z_out (`first`), z_out afterwards -- the start point is never copied
    __device__ __forceinline__ FwView(FwKP P, int p, int first = 0) {
        m = P->M.m; mp = P->V.mp; T = P->M.T; nb = P->M.nb; s = N + m; has_xf = P->M.has_xf; var2 = P->M.var2;
        const FwWs L = fw_ws_layout(N, m, mp, T, nb, FwCfg<N>::LDG, FwCfg<N>::FST);
        const int wave_g = blockIdx.x * FW_WAVES + (threadIdx.x >> 6);
        double* wsp = P->ws + (size_t)wave_g * P->ws_stride;
        zp = P->zout + (size_t)p * P->zld;
        zs = (first && P->zinit) ? P->zinit + (size_t)p * T * s : zp;
        b = wsp + L.b; nu = wsp + L.nu; rdu = wsp + L.rdu;
        rdx = wsp + L.rdx; rp = wsp + L.rp; rhs = wsp + L.rhs; yv = wsp + L.y; dnu = wsp + L.dnu;
        phx = wsp + L.phx; rsg = wsp + L.rs; fac = wsp + L.fac;
        fstride = FwCfg<N>::FST;                            // the two tile dumps of a stage (FwCfg::DUMP)
        if (P->mode == FW_MODE_EXPORT) { fac = P->sh_fac; rsg = P->sh_rs; fstride = 6 * N * FwCfg<N>::LDG; }   // the factor IS the product
    }
};

// ------------------------------------------------------------------------------------------------
// P0: start point, nu, b   (fast_mpc_init.m:12-27, fast_mpc_eq_const.m:39,44,47,68)
template <int N>
FW_IN void fw_phase_init(FwKP Pin, int p, int write_z) {
    const FwKP P = fw_uniform(Pin);
    p = __builtin_amdgcn_readfirstlane(p);
    const FwView<N> W(P, p);
    const int lane = threadIdx.x & 63;
    const int Nz = W.T * W.s, nbn = W.nb * N, m = W.m;
    const double* zinit = P->zinit;
    write_z = __builtin_amdgcn_readfirstlane(write_z);
    if (write_z)
        for (int idx = lane; idx < Nz; idx += 64) {
            const int e = idx % W.s;
            W.zp[idx] = zinit ? zinit[(size_t)p * Nz + idx] : (e < m ? P->M.umid[e] : P->M.xmid[e - m]);
        }
    const double* x0v = P->x0 + (size_t)p * N;
    const double* x0pv = P->x0p ? P->x0p + (size_t)p * N : nullptr;
    const double* w = P->w;
    const double* nu0 = P->nu0;
    // The prediction terms of b (fast_mpc_eq_const.m:39,44): A1 x0 + A2 x0_pre for block row 0 on lanes 0..N-1, A2 x0 for block
    // row 1 on lanes N..2N-1.  Every lane takes part with clamped indices and the operands are requested nine columns at a
    // time BEFORE they are used: as three loops of N dependent load + multiply-add pairs under a divergent condition this was
    // 3 % of a Newton iteration (66 k cycles per wavefront, round 4 timing build).
    double pred = 0.0;
    {
        const int blk = lane >= N ? 1 : 0, r = lane - blk * N, rc = (r < N) ? r : 0;
        const double* Aa = blk ? P->M.A2t : P->M.A1t;          // block row 0: A1 x0 ; block row 1: A2 x0
        const bool two = W.var2 && x0pv != nullptr;
        const bool on_a = blk == 0 || W.var2, on_b = blk == 0 && two;
#pragma unroll
        for (int c0 = 0; c0 < N; c0 += 9) {
            double aa[9], ab[9], xa[9], xb[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int c = c0 + q < N ? c0 + q : N - 1;
                aa[q] = Aa[c * N + rc]; ab[q] = P->M.A2t[c * N + rc]; xa[q] = x0v[c]; xb[q] = two ? x0pv[c] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 9; ++q)
                if (c0 + q < N) { pred = fma(on_a ? aa[q] : 0.0, xa[q], pred); pred = fma(on_b ? ab[q] : 0.0, xb[q], pred); }
        }
    }
    // every load before the first store: a load issued behind a global store waits for it (vmcnt is in order)
    const double* xfp = W.has_xf ? P->M.xf : P->M.xmid;           // (any readable n-vector when there is no terminal row)
    const double* nup = nu0 ? nu0 + (size_t)p * nbn : W.nu;        // (nu0 = NULL: zeros -- the loads then read this wave's workspace and a factor drops them)
    const double nuf = nu0 ? 1.0 : 0.0;
    const double* wp = w ? w + (size_t)p * W.T * N : nup;          // (w = NULL: zeros, by the same device)
    const int tn = W.T * N;
    for (int base = 0; base < nbn; base += 64 * 14) {
        double nv[14], bv[14], xfv[14], wv[14];
        bool isxf[14], wok[14];
#pragma unroll
        for (int q = 0; q < 14; ++q) {
            const int idx = base + lane + 64 * q;
            const bool ok = idx < nbn;
            const int ic = ok ? idx : 0;
            nv[q] = nup[ic];
            const int i = ic / N, r = ic - i * N;
            wv[q] = wp[ic < tn ? ic : 0];
            wok[q] = w != nullptr && ic < tn;
            double v = 0.0;
            if (base == 0 && q == 0 && ok && i < 2 && i < W.T) v = pred;        // (2 N <= 64: both block rows lie in the first pass)
            xfv[q] = xfp[r];                                          // (unconditional: a load under a condition is a branch with a wait
            isxf[q] = i == W.T;                                       //  behind it -- fourteen of them serialised the loads of this loop)
            bv[q] = v;
        }
#pragma unroll
        for (int q = 0; q < 14; ++q) {
            const int idx = base + lane + 64 * q;
            if (idx < nbn) { W.nu[idx] = nuf != 0.0 ? nv[q] : 0.0; W.b[idx] = isxf[q] ? xfv[q] : (wok[q] ? wv[q] + bv[q] : bv[q]); }
        }
    }
    if (write_z != 2 && P->step)
        for (int idx = lane; idx < P->step_ld; idx += 64) P->step[(size_t)p * P->step_ld + idx] = -1.0;
    fw_mem_fence();
}

// ------------------------------------------------------------------------------------------------
// C' applied to a stacked dual vector v (nu or d_nu), all stages at once on the matrix cores.
// Output tiles are (16 stages) x (16 consecutive entries of u_j or x_j): the epilogue then reads and
// writes along contiguous elements of a stage, and the per-column constants are loaded once per tile.
// Rt_j = 2R + k diag(1/s+^2 + 1/s-^2) is recomputed from z wherever it is needed (here twice, once per stage in the
// factorisation): it never makes a round trip through HBM.
//   MODE 0 (P1): r_d = 2Hz + g + kP'd + C'nu; Phi^-1 r_d is what is kept (u entries: rdu, x entries: phx; r_d on x: rdx);
//                out3 = { sum(r_d^2), -, "Phi not PD" flag }.
//   MODE 1 (P5): d_z = Phi^-1(-r_d - C'd_nu) written over Phi^-1 r_d, AND the new iterate z + d_z written to z_out for
//                the step length t = 1 (the usual outcome; fw_phase_zfix corrects it otherwise);
//                out3 = { <r_d,e>, ||e||^2, - } with e = k P'DP d_z (line search); k P'DP = Rt - 2R.
template <int N, int MODE>
FW_IN void fw_phase_CT(FwKP Pin, int p, double* lds_g, double* out3_g, int first) {
    const FwKP P = fw_uniform(Pin);
    p = __builtin_amdgcn_readfirstlane(p);
    first = __builtin_amdgcn_readfirstlane(first);
    const FwView<N> W(P, p, first);
    const fw_clds_t sBt = (fw_clds_t)lds_g;
    const fw_lds_t out3 = (fw_lds_t)out3_g;
    const int lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    const int m = W.m, T = W.T, s = W.s;
    const double* vec = MODE == 0 ? W.nu : W.dnu;
    const double* zs = W.zs;
    const double kbar = P->kbar;
    const double* umaxp = P->M.umax; const double* uminp = P->M.umin;
    const double* R2p = P->M.R2; const double* rlp = P->M.rl;
    const double* A1p = P->M.A1; const double* A2p = P->M.A2;
    const double* Q2p = P->M.Q2; const double* Qf2p = P->M.Qf2;
    const double* qlp = P->M.ql; const double* qflp = P->M.qfl;
    const bool has_xf = W.has_xf != 0, var2 = W.var2 != 0;
    double acc0 = 0.0, acc1 = 0.0;
    int bad = 0;
    for (int j0 = 0; j0 < T; j0 += 32) {
        // stage of each of the 8 accumulator elements of this lane (e = 4*I + r): j0 + 16I + 4r + g
        bool sok[8]; int sj[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = j0 + 16 * (e >> 2) + 4 * (e & 3) + g;
            sok[e] = j < T; sj[e] = sok[e] ? j : 0;
        }
        // A fragments (stage on the row index): v_j[k] for stage j = j0+16I+c16.  (v_{j+1}, v_{j+2} are only needed by the x
        // entries: loaded behind the u entries -- kept live across them, the 42 fragment values were spilled to scratch and
        // every reload waited for all outstanding loads and stores.)
        // (loads below: a per-lane row base chosen once + a constant per k-step, switched off by factors afterwards -- an index
        //  chosen per load gives every load an address register pair of its own, which the compiler spills; see fw_phase_C2)
        const int k6 = (24 + g < N ? 24 + g : N - 1) - 24;                          // last k-step: k = 24 + g clamped to N - 1
        const double f6 = 24 + g < N ? 1.0 : 0.0;
        double an[2][7];
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            const int j = j0 + 16 * I + c16;
            const double* vp = vec + (size_t)(j < T ? j : 0) * N;
            const double fj = j < T ? 1.0 : 0.0;
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) an[I][ks] = vp[ks < 6 ? 4 * ks + g : 24 + k6];
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) an[I][ks] *= ks < 6 ? fj : fj * f6;
        }
        // ---- u entries: G[j][c] = sum_k v_j[k] B[k][c].  The inputs of column block J + 1 are requested before the
        //      results of block J are stored (a load issued behind a store waits for it: vmcnt is in order).
        const int NJ = (m + 15) >> 4;
        // Element offsets of this lane's 8 values in z (stride s) and in the m-wide workspace array (stride m), as plain ints:
        // column block J adds the constant 16 J, which ends up in the instruction's offset field.  The columns of a partial
        // last block (c >= m) are read as they lie -- z has the x entries of the stage there, the workspace its next row --
        // and masked afterwards: a select on the index gave every block its own 64-bit addresses, which were spilled and
        // reloaded around every load.
        unsigned zo[8], ro[8];               // (unsigned: scalar array base + a 32-bit offset register per load / store, no 64-bit address per element)
#pragma unroll
        for (int e = 0; e < 8; ++e) { zo[e] = 8u * (unsigned)(sj[e] * s + c16); ro[e] = 8u * (unsigned)(sj[e] * m + c16); }   // BYTE offsets
        double zu[8], in0[8], zun[8], in0n[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            zu[e] = fw_ldb(zs, zo[e]);
            in0[e] = MODE == 1 ? fw_ldb(W.rdu, ro[e]) : 0.0;
        }
        auto utile = [&](const int J) {
            const int c = 16 * J + c16;
            const bool cok = c < m;
            d4 g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0};
            const fw_clds_t br = sBt + c * FW_LDB + g;            // rows >= m of B' are zero padding
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {
                const double bb = br[4 * ks];                        // B[k][c], zero for k >= n
                g0 = MFMA64(an[0][ks], bb, g0);
                g1 = MFMA64(an[1][ks], bb, g1);
            }
            // (per-column constants by the same rule: lane offset + constant; beyond m they read the pool's next array, masked)
            const double cmax = umaxp[c16 + 16 * J], cmin = uminp[c16 + 16 * J], cr2 = R2p[c16 + 16 * J], crl = MODE == 0 ? rlp[c16 + 16 * J] : 0.0;
            {
                const unsigned jn = 128u * (unsigned)(J + 1 < NJ ? J + 1 : J);            // bytes
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    zun[e] = fw_ldb(zs, zo[e] + jn);
                    in0n[e] = MODE == 1 ? fw_ldb(W.rdu, ro[e] + jn) : 0.0;
                }
            }
            double o0[8], o1[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double G = (e >> 2) == 0 ? g0[e & 3] : g1[e & 3];
                const bool ok = cok && sok[e];
                const double u = zu[e];
                const double dp = fw_rcp(cmax - u), dm = fw_rcp(u - cmin);
                const double hb = kbar * (dp * dp + dm * dm);        // k P'DP on this entry
                const double rt = cr2 + hb;
                const double wv = fw_rcp(rt);
                if (MODE == 0) {
                    if (ok && (!(rt > 0.0) || isinf(rt))) bad = 1;
                    const double rd = cr2 * u + crl + kbar * (dp - dm) - G;
                    o0[e] = wv * rd;
                    if (ok) acc0 += rd * rd;
                } else {
                    const double rd = in0[e] * rt;
                    const double du = (G - rd) * wv;
                    const double ee = hb * du;
                    o0[e] = du; o1[e] = u + du;
                    if (ok) { acc0 += rd * ee; acc1 += ee * ee; }
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (cok && sok[e]) {
                    fw_stb(W.rdu, ro[e] + 128u * (unsigned)J, o0[e]);
                    if (MODE == 1) fw_stb(W.zp, zo[e] + 128u * (unsigned)J, o1[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { zu[e] = zun[e]; in0[e] = in0n[e]; }
        };
        // The first 9 column blocks (m <= 144: all of them) are straight-line code: across a loop's back-edge the compiler waits
        // for vmcnt(0), i.e. for the 16 stores of the previous block, before it touches the values requested for this one.
#ifdef FW_CT_UNROLL
#pragma unroll
        for (int J = 0; J < 9; ++J) {
            if (J < NJ) utile(J);
        }
        for (int J = 9; J < NJ; ++J) utile(J);
#else
        for (int J = 0; J < NJ; ++J) utile(J);
#endif
        // ---- x entries (x_jx, jx = j+1 for stage column j): H[j][r] = sum_k v_{j+1}[k] A1[k][r] + v_{j+2}[k] A2[k][r]
        double a1[2][7], a2[2][7];
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            const int j = j0 + 16 * I + c16;
            const double* v1p = vec + (size_t)(j + 1 < T ? j + 1 : 0) * N;
            const double* v2p = vec + (size_t)(j + 2 < T ? j + 2 : 0) * N;
            const double f1 = j + 1 < T ? 1.0 : 0.0, f2 = (j + 2 < T && var2) ? 1.0 : 0.0;
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) { const int kq = ks < 6 ? 4 * ks + g : 24 + k6; a1[I][ks] = v1p[kq]; a2[I][ks] = v2p[kq]; }
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) { a1[I][ks] *= ks < 6 ? f1 : f1 * f6; a2[I][ks] *= ks < 6 ? f2 : f2 * f6; }
        }
        d4 h[2][2];
        double cq2[2], cqf2[2], cql[2], cqfl[2], vprev[2][8], vxf[2][8], xin[2][8], zx[2][8];
#pragma unroll
        for (int J = 0; J < 2; ++J) {
            const int rr = 16 * J + c16;
            const bool rok = rr < N;
            const int rc = rok ? rr : 0;
            h[J][0] = (d4){0, 0, 0, 0}; h[J][1] = (d4){0, 0, 0, 0};
            {
                const double* p1 = A1p + rc;                           // A1[k][rr] at A1p[k N + rr]
                const double* p2 = A2p + rc;
                const double frr = rok ? 1.0 : 0.0;
                double b1[7], b2[7];
#pragma unroll
                for (int ks = 0; ks < 7; ++ks) { const int kq = ks < 6 ? 4 * ks + g : 24 + k6; b1[ks] = p1[kq * N]; b2[ks] = p2[kq * N]; }
#pragma unroll
                for (int ks = 0; ks < 7; ++ks) {
                    const double f = ks < 6 ? frr : frr * f6;
                    const double c1 = b1[ks] * f, c2 = b2[ks] * f;
                    h[J][0] = MFMA64(a1[0][ks], c1, h[J][0]);
                    h[J][1] = MFMA64(a1[1][ks], c1, h[J][1]);
                    h[J][0] = MFMA64(a2[0][ks], c2, h[J][0]);
                    h[J][1] = MFMA64(a2[1][ks], c2, h[J][1]);
                }
            }
            cq2[J] = Q2p[rc]; cqf2[J] = Qf2p[rc];
            cql[J] = MODE == 0 ? qlp[rc] : 0.0; cqfl[J] = MODE == 0 ? qflp[rc] : 0.0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool last = sj[e] + 1 == T;
                // (unsigned element offsets from the wave-uniform array bases: scalar base + one 32-bit offset register per load)
                const unsigned ov = 8u * (unsigned)(sj[e] * N + rc), ox = 8u * (unsigned)((last && has_xf ? T : sj[e]) * N + rc);
                vprev[J][e] = fw_ldb(vec, ov);
                vxf[J][e] = fw_ldb(vec, ox);
                zx[J][e] = fw_ldb(zs, 8u * (unsigned)(sj[e] * s + m + rc));
                xin[J][e] = MODE == 0 ? 0.0 : fw_ldb(W.rdx, ov);
            }
        }
        // all loads are issued: only now the stores
#pragma unroll
        for (int J = 0; J < 2; ++J) {
            const int rr = 16 * J + c16;
            const bool rok = rr < N;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double H = (e >> 2) == 0 ? h[J][0][e & 3] : h[J][1][e & 3];
                const bool last = sj[e] + 1 == T;
                const double q2 = last ? cqf2[J] : cq2[J];
                const double iq = fw_rcp(q2);
                if (MODE == 0) {
                    double v = q2 * zx[J][e] + (last ? cqfl[J] : cql[J]) + vprev[J][e] - H;
                    if (last && has_xf) v += vxf[J][e];
                    if (rok && sok[e]) {
                        W.rdx[sj[e] * N + rr] = v;                    // r_d on x_j
                        W.phx[sj[e] * N + rr] = v * iq;               // Phi^-1 r_d on x_j
                        acc0 += v * v;
                    }
                } else {
                    double v = -xin[J][e] - vprev[J][e] + H;
                    if (last && has_xf) v -= vxf[J][e];
                    const double dx = v * iq;
                    if (rok && sok[e]) {
                        W.rdx[sj[e] * N + rr] = dx;                   // d_x
                        W.zp[sj[e] * s + m + rr] = zx[J][e] + dx;
                    }
                }
            }
        }
    }
    acc0 = fw_wave_sum(acc0);
    acc1 = fw_wave_sum(acc1);
    const bool anybad = __ballot(bad) != 0ull;
    if (lane == 0) { out3[0] = acc0; out3[1] = acc1; out3[2] = anybad ? 1.0 : 0.0; }
    fw_mem_fence();
}

// ------------------------------------------------------------------------------------------------
// C applied to two stacked primal vectors in one pass, all stages at once on the matrix cores; output tiles are
// (16 block rows i) x (16 state entries), so the epilogue runs along contiguous entries.
//   P1: r_p = C z - b, out1 = sum(r_p^2)      P2: rhs = r_p - C Phi^-1 r_d   (inf_newton_solver.m:28-29)
// Both products share the operand tiles of B (LDS), A1, A2; the exit test that sits between them in the reference
// (inf_newton_solver.m:19-22) is taken by the caller from out1 -- rhs is then simply not used.
template <int N>
FW_IN void fw_phase_C2(FwKP Pin, int p, double* lds_g, double* out1_g, int first) {
    const FwKP P = fw_uniform(Pin);
    p = __builtin_amdgcn_readfirstlane(p);
    first = __builtin_amdgcn_readfirstlane(first);
    const FwView<N> W(P, p, first);
    const fw_clds_t sBt = (fw_clds_t)lds_g;
    const fw_lds_t out1 = (fw_lds_t)out1_g;
    const int lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    const int m = W.m, mp = W.mp, T = W.T, s = W.s, nb = W.nb;
    const double* A1tp = P->M.A1t; const double* A2tp = P->M.A2t;
    const double* zs = W.zs;
    const bool var2 = W.var2 != 0;
    double acc = 0.0;
    for (int j0 = 0; j0 < nb; j0 += 32) {
        d4 a[2][2], c[2][2];                     // a: C z, c: C Phi^-1 r_d;  [I][J]: block rows 16I.., entries 16J..
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int J = 0; J < 2; ++J) { a[I][J] = (d4){0, 0, 0, 0}; c[I][J] = a[I][J]; }
        // ---- B u_i (K = m): A operand = u_i[c] / (Phi^-1 r_d)[u_i][c] with the stage on the row index
        const int i0 = j0 + c16, i1 = j0 + 16 + c16;
        const bool ok0 = i0 < T, ok1 = i1 < T;
        const double* u0p = zs + (size_t)(ok0 ? i0 : 0) * s;
        const double* u1p = zs + (size_t)(ok1 ? i1 : 0) * s;
        const double* w0p = W.rdu + (size_t)(ok0 ? i0 : 0) * m;
        const double* w1p = W.rdu + (size_t)(ok1 ? i1 : 0) * m;
        double v0[FW_KCH], v1[FW_KCH], y0[FW_KCH], y1[FW_KCH];
#pragma unroll
        for (int q = 0; q < FW_KCH; ++q) {
            const int k = 4 * q + g;
            const int kk = k < m ? k : 0;
            v0[q] = u0p[kk]; v1[q] = u1p[kk]; y0[q] = w0p[kk]; y1[q] = w1p[kk];
        }
        for (int kc = 0; kc < mp; kc += 4 * FW_KCH) {
            double x0[FW_KCH], x1[FW_KCH], v0n[FW_KCH], v1n[FW_KCH], y0n[FW_KCH], y1n[FW_KCH];
            const int kn = kc + 4 * FW_KCH < mp ? kc + 4 * FW_KCH : kc;
#pragma unroll
            for (int q = 0; q < FW_KCH; ++q) {
                const int k = kc + 4 * q + g;
                x0[q] = sBt[k * FW_LDB + c16];                    // B[r = c16][c = k]
                x1[q] = sBt[k * FW_LDB + 16 + c16];
                const int k2 = kn + 4 * q + g;
                const int kk = k2 < m ? k2 : 0;
                v0n[q] = u0p[kk]; v1n[q] = u1p[kk]; y0n[q] = w0p[kk]; y1n[q] = w1p[kk];
            }
#pragma unroll
            for (int q = 0; q < FW_KCH; ++q) {
                const int k = kc + 4 * q + g;
                const double f0 = (ok0 && k < m) ? 1.0 : 0.0, f1 = (ok1 && k < m) ? 1.0 : 0.0;
                const double p0 = v0[q] * f0, p1 = v1[q] * f1, q0 = y0[q] * f0, q1 = y1[q] * f1;
                a[0][0] = MFMA64(p0, x0[q], a[0][0]);
                a[0][1] = MFMA64(p0, x1[q], a[0][1]);
                a[1][0] = MFMA64(p1, x0[q], a[1][0]);
                a[1][1] = MFMA64(p1, x1[q], a[1][1]);
                c[0][0] = MFMA64(q0, x0[q], c[0][0]);
                c[0][1] = MFMA64(q0, x1[q], c[0][1]);
                c[1][0] = MFMA64(q1, x0[q], c[1][0]);
                c[1][1] = MFMA64(q1, x1[q], c[1][1]);
            }
#pragma unroll
            for (int q = 0; q < FW_KCH; ++q) { v0[q] = v0n[q]; v1[q] = v1n[q]; y0[q] = y0n[q]; y1[q] = y1n[q]; }
        }
        // ---- A1 x_i + A2 x_{i-1}  (K = n): x_j at zs[(j-1)*s + m + k], (Phi^-1 r_d)[x_j] at phx[(j-1)*N + k]
        // Every load here is (a per-lane ROW BASE chosen once) + (a constant per k-step); what must not count is switched off by a
        // factor 0 / 1 afterwards.  An index chosen per load (`cond ? row * s + k : 0`) gave each of the 84 loads an address of its
        // own: the compiler spilled them, and every reload waits for all loads in flight (vmcnt(0)) in front of the load it feeds --
        // one memory round trip per load (round 4: 55 such reloads in this section, 160 k cycles per wavefront and iteration).
        {
            const double *zaP[2], *zbP[2], *paP[2], *pbP[2], *a1P[2], *a2P[2];
            double fa[2], fb[2], fr[2];
            const int k6 = (24 + g < N ? 24 + g : N - 1) - 24;                      // last k-step: k = 24 + g clamped to N - 1
            const double f6 = 24 + g < N ? 1.0 : 0.0;
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                const int i = j0 + 16 * I + c16;
                const bool ia = i >= 1 && i < T, ib = i >= 2 && i < T && var2;
                zaP[I] = zs + (ia ? (size_t)(i - 1) * s : 0) + m; zbP[I] = zs + (ib ? (size_t)(i - 2) * s : 0) + m;
                paP[I] = W.phx + (ia ? (size_t)(i - 1) * N : 0); pbP[I] = W.phx + (ib ? (size_t)(i - 2) * N : 0);
                fa[I] = ia ? 1.0 : 0.0; fb[I] = ib ? 1.0 : 0.0;
            }
#pragma unroll
            for (int J = 0; J < 2; ++J) {
                const int rr = 16 * J + c16;
                a1P[J] = A1tp + (rr < N ? rr : 0); a2P[J] = A2tp + (rr < N ? rr : 0);   // A1[rr][k] at A1t[k N + rr]
                fr[J] = rr < N ? 1.0 : 0.0;
            }
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {
                const int kq = ks < 6 ? 4 * ks + g : 24 + k6;                       // (k < N for ks < 6 since N > 24)
                double xa[2], xb[2], za[2], zb[2], pa[2], pb[2];
#pragma unroll
                for (int J = 0; J < 2; ++J) { xa[J] = a1P[J][kq * N]; xb[J] = a2P[J][kq * N]; }
#pragma unroll
                for (int I = 0; I < 2; ++I) { za[I] = zaP[I][kq]; zb[I] = zbP[I][kq]; pa[I] = paP[I][kq]; pb[I] = pbP[I][kq]; }
#pragma unroll
                for (int J = 0; J < 2; ++J) { const double f = ks < 6 ? fr[J] : fr[J] * f6; xa[J] *= f; xb[J] *= f; }
#pragma unroll
                for (int I = 0; I < 2; ++I) { za[I] *= fa[I]; zb[I] *= fb[I]; pa[I] *= fa[I]; pb[I] *= fb[I]; }
#pragma unroll
                for (int I = 0; I < 2; ++I)
#pragma unroll
                    for (int J = 0; J < 2; ++J) {
                        a[I][J] = MFMA64(za[I], xa[J], a[I][J]);
                        a[I][J] = MFMA64(zb[I], xb[J], a[I][J]);
                        c[I][J] = MFMA64(pa[I], xa[J], c[I][J]);
                        c[I][J] = MFMA64(pb[I], xb[J], c[I][J]);
                    }
            }
        }
        {   // epilogue: element e = (I, J, r): block row i = j0+16I+4r+g, entry 16J+c16
            bool ok[16]; int ir[16], ii[16]; double in0[16], in1[16], in2[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int I = e >> 3, J = (e >> 2) & 1, r = e & 3;
                const int i = j0 + 16 * I + 4 * r + g, row = 16 * J + c16;
                ok[e] = row < N && i < nb;
                ir[e] = ok[e] ? row : 0; ii[e] = ok[e] ? i : 0;
                const int jx = ii[e] < T ? ii[e] : T - 1;           // x_{i+1}; the xf row uses x_T
                in0[e] = zs[jx * s + m + ir[e]];
                in1[e] = W.b[ii[e] * N + ir[e]];
                in2[e] = W.phx[jx * N + ir[e]];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int I = e >> 3, J = (e >> 2) & 1, r = e & 3;
                const double cz = ii[e] < T ? a[I][J][r] : 0.0;
                const double cw = ii[e] < T ? c[I][J][r] : 0.0;
                const double rp = in0[e] - in1[e] - cz;
                if (ok[e]) {
                    W.rp[ii[e] * N + ir[e]] = rp; acc += rp * rp;
                    W.rhs[ii[e] * N + ir[e]] = rp - (in2[e] - cw);
                }
            }
        }
    }
    acc = fw_wave_sum(acc);
    if (lane == 0) out1[0] = acc;
    fw_mem_fence();
}


template <int N>
__device__ __forceinline__ fw_lds_t fw_cold_vec(double* lds_g, int mp) {      // this wave's rhs/y/d_nu vector
    return (fw_lds_t)lds_g + mp * FW_LDB + 2 * 3 * N * FwCfg<N>::LDG + (threadIdx.x >> 6) * FW_VEC_STRIDE;
}

template <int N>
__device__ __forceinline__ fw_lds_t fw_cold_consts(double* lds_g, int mp) {   // [cu | hc | wc | ubar], 4*mp doubles
    return (fw_lds_t)lds_g + mp * FW_LDB + 2 * 3 * N * FwCfg<N>::LDG + FW_WAVES * FW_VEC_STRIDE;
}

// ================================================================================================
// Cold-start variants of the vector phases (first Newton step from u = ubar, x = xbar; SURVEY §7.2a
// regime (ii)).  With a constant primal start the m-wide quantities collapse:
//     r_d[u_j] = cu - B'nu_j          B Phi^-1 r_d[u_i] = cbu - G nu_i          r_p,i = cp_i - b_i
// so no m-wide array (z, Rt^-1, r_d) makes a round trip through HBM and z is written exactly once.
// ================================================================================================

// residual norms, r_d on the x entries (+ Phi^-1 of it) and r_p.   out3 = { sum r_d^2, sum r_p^2, - }
template <int N>
FW_FN void fw_cold_resid(FwKP Pin, int p, double* lds_g, double* out3_g) {
    const FwKP P = fw_uniform(Pin);
    p = __builtin_amdgcn_readfirstlane(p);
    const FwView<N> W(P, p);
    const fw_clds_t sBt = (fw_clds_t)lds_g;
    const fw_lds_t out3 = (fw_lds_t)out3_g;
    const int lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    const int m = W.m, T = W.T, nb = W.nb;
    const FwCold CL = fw_cold_layout(N, W.mp);
    const double* cold = P->cold;
    const double* vec = W.nu;
    const double* A1p = P->M.A1; const double* A2p = P->M.A2;
    const double* Q2p = P->M.Q2; const double* Qf2p = P->M.Qf2;
    const double* qlp = P->M.ql; const double* qflp = P->M.qfl; const double* xmid = P->M.xmid;
    const bool has_xf = W.has_xf != 0, var2 = W.var2 != 0;
    double acc_d = 0.0, acc_p = 0.0;
    for (int j0 = 0; j0 < T; j0 += 32) {
        bool sok[8]; int sj[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = j0 + 16 * (e >> 2) + 4 * (e & 3) + g;
            sok[e] = j < T; sj[e] = sok[e] ? j : 0;
        }
        double an[2][7], a1[2][7], a2[2][7];
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {
                const int j = j0 + 16 * I + c16, k = 4 * ks + g;
                const bool kk = k < N;
                const double t0 = vec[(kk && j < T ? j : 0) * N + (kk ? k : 0)];
                const double t1 = vec[(kk && j + 1 < T ? j + 1 : 0) * N + (kk ? k : 0)];
                const double t2 = vec[(kk && j + 2 < T ? j + 2 : 0) * N + (kk ? k : 0)];
                an[I][ks] = (kk && j < T) ? t0 : 0.0;
                a1[I][ks] = (kk && j + 1 < T) ? t1 : 0.0;
                a2[I][ks] = (kk && j + 2 < T && var2) ? t2 : 0.0;
            }
        // ---- u entries: only ||cu - B'nu_j||^2 is needed
        for (int J = 0; J * 16 < m; ++J) {
            const int c = 16 * J + c16;
            const bool cok = c < m;
            d4 g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0};
            const fw_clds_t br = sBt + c * FW_LDB + g;
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {
                const double bb = br[4 * ks];
                g0 = MFMA64(an[0][ks], bb, g0);
                g1 = MFMA64(an[1][ks], bb, g1);
            }
            const double cuc = cold[CL.cu + (cok ? c : 0)];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double rd = cuc - ((e >> 2) == 0 ? g0[e & 3] : g1[e & 3]);
                if (cok && sok[e]) acc_d += rd * rd;
            }
        }
        // ---- x entries: r_d[x_j] = 2Q_j xbar + q_j + nu_{j-1} - A1'nu_j - A2'nu_{j+1} (+ nu_T)
#pragma unroll
        for (int J = 0; J < 2; ++J) {
            const int rr = 16 * J + c16;
            const bool rok = rr < N;
            const int rc = rok ? rr : 0;
            d4 h0 = {0, 0, 0, 0}, h1 = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {
                const int k = 4 * ks + g;
                const bool ok = k < N && rok;
                const int off = ok ? k * N + rr : 0;
                const double t1 = A1p[off], t2 = A2p[off];
                const double b1 = ok ? t1 : 0.0, b2 = ok ? t2 : 0.0;
                h0 = MFMA64(a1[0][ks], b1, h0);
                h1 = MFMA64(a1[1][ks], b1, h1);
                h0 = MFMA64(a2[0][ks], b2, h0);
                h1 = MFMA64(a2[1][ks], b2, h1);
            }
            const double cq2 = Q2p[rc], cqf2 = Qf2p[rc], cql = qlp[rc], cqfl = qflp[rc], xb = xmid[rc];
            double vprev[8], vxf[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool last = sj[e] + 1 == T;
                vprev[e] = vec[sj[e] * N + rc];
                vxf[e] = vec[(last && has_xf ? T : sj[e]) * N + rc];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double H = (e >> 2) == 0 ? h0[e & 3] : h1[e & 3];
                const bool last = sj[e] + 1 == T;
                const double q2 = last ? cqf2 : cq2;
                double v = q2 * xb + (last ? cqfl : cql) + vprev[e] - H;
                if (last && has_xf) v += vxf[e];
                if (rok && sok[e]) {
                    W.rdx[sj[e] * N + rr] = v;
                    W.phx[sj[e] * N + rr] = v * fw_rcp(q2);
                    acc_d += v * v;
                }
            }
        }
    }
    // ---- r_p,i = cp_i - b_i  (xf row: xbar - xf)
    for (int idx = lane; idx < nb * N; idx += 64) {
        const int i = idx / N, r = idx - i * N;
        const double c = i >= T ? xmid[r] : cold[(i == 0 ? CL.cp0 : i == 1 ? CL.cp1 : CL.cp2) + r];
        const double v = c - W.b[idx];
        W.rp[idx] = v;
        acc_p += v * v;
    }
    acc_d = fw_wave_sum(acc_d);
    acc_p = fw_wave_sum(acc_p);
    if (lane == 0) { out3[0] = acc_d; out3[1] = acc_p; }
    fw_mem_fence();
}

// rhs_i = r_p,i - (Phi^-1 r_d[x_{i+1}] - (cbu - G nu_i) - A1 Phi^-1 r_d[x_i] - A2 Phi^-1 r_d[x_{i-1}])
template <int N>
FW_FN void fw_cold_rhs(FwKP Pin, int p, double* lds_g) {
    const FwKP P = fw_uniform(Pin);
    p = __builtin_amdgcn_readfirstlane(p);
    const FwView<N> W(P, p);
    const fw_lds_t vl = fw_cold_vec<N>(lds_g, W.mp);
    const int lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    const int T = W.T, nb = W.nb;
    const FwCold CL = fw_cold_layout(N, W.mp);
    const double* cold = P->cold;
    const double* A1tp = P->M.A1t; const double* A2tp = P->M.A2t;
    const bool var2 = W.var2 != 0;
    for (int j0 = 0; j0 < nb; j0 += 32) {
        d4 a[2][2];
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int J = 0; J < 2; ++J) a[I][J] = (d4){0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            const int k = 4 * ks + g;
            double xg[2], xa[2], xb[2], zn[2], za[2], zb[2];
#pragma unroll
            for (int J = 0; J < 2; ++J) {
                const int rr = 16 * J + c16;
                const bool ok = k < N && rr < N;
                const int off = ok ? k * N + rr : 0;
                const double tg = cold[CL.G + off], t1 = A1tp[off], t2 = A2tp[off];
                xg[J] = ok ? -tg : 0.0;              // -G[k][r]  (G symmetric)
                xa[J] = ok ? t1 : 0.0;               // A1[r][k]
                xb[J] = ok ? t2 : 0.0;
            }
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                const int i = j0 + 16 * I + c16;
                const bool okn = k < N && i < T;
                const bool oka = k < N && i >= 1 && i < T;
                const bool okb = k < N && i >= 2 && i < T && var2;
                const double tn = W.nu[okn ? i * N + k : 0];
                const double ta = W.phx[oka ? (i - 1) * N + k : 0];
                const double tb = W.phx[okb ? (i - 2) * N + k : 0];
                zn[I] = okn ? tn : 0.0; za[I] = oka ? ta : 0.0; zb[I] = okb ? tb : 0.0;
            }
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int J = 0; J < 2; ++J) {
                    a[I][J] = MFMA64(zn[I], xg[J], a[I][J]);
                    a[I][J] = MFMA64(za[I], xa[J], a[I][J]);
                    a[I][J] = MFMA64(zb[I], xb[J], a[I][J]);
                }
        }
        bool ok[16]; int ir[16], ii[16]; double in0[16], in1[16], cb[2];
#pragma unroll
        for (int J = 0; J < 2; ++J) cb[J] = cold[CL.cbu + 16 * J + c16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int I = e >> 3, J = (e >> 2) & 1, r = e & 3;
            const int i = j0 + 16 * I + 4 * r + g, row = 16 * J + c16;
            ok[e] = row < N && i < nb;
            ir[e] = ok[e] ? row : 0; ii[e] = ok[e] ? i : 0;
            const int jx = ii[e] < T ? ii[e] : T - 1;
            in0[e] = W.phx[jx * N + ir[e]];
            in1[e] = W.rp[ii[e] * N + ir[e]];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int I = e >> 3, J = (e >> 2) & 1, r = e & 3;
            const double cz = ii[e] < T ? cb[J] + a[I][J][r] : 0.0;
            if (ok[e]) vl[ii[e] * N + ir[e]] = in1[e] - (in0[e] - cz);
        }
    }
    fw_wave_fence();
    fw_mem_fence();
}

// Line-search dots of the cold step as 27-dimensional quadratic forms (mu_j = nu_j + d_nu_j):
//   <r_d,e> = sum_j [ va'(mu_j + nu_j) - sa - nu_j' Ma mu_j ],   ||e||^2 = sum_j [ mu_j' Ma2 mu_j - 2 va2' mu_j + sa2 ]
// out3 = { <r_d,e>, ||e||^2 }.  Exact algebra, different rounding: the caller uses it only when the step
// length decision has a wide margin and falls back to the element-wise evaluation otherwise.
template <int N>
FW_FN void fw_cold_dots(FwKP Pin, int p, double* lds_g, double* out3_g) {
    const FwKP P = fw_uniform(Pin);
    p = __builtin_amdgcn_readfirstlane(p);
    const FwView<N> W(P, p);
    const fw_lds_t out3 = (fw_lds_t)out3_g;
    const fw_clds_t dn = fw_cold_vec<N>(lds_g, W.mp);
    const int lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    const int T = W.T;
    const FwCold CL = fw_cold_layout(N, W.mp);
    const double* cold = P->cold;
    double acc0 = 0.0, acc1 = 0.0;
    for (int j0 = 0; j0 < T; j0 += 32) {
        d4 a[2][2], b[2][2];
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int J = 0; J < 2; ++J) { a[I][J] = (d4){0, 0, 0, 0}; b[I][J] = a[I][J]; }
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            const int k = 4 * ks + g;
            double x1[2], x2[2], zm[2];
#pragma unroll
            for (int J = 0; J < 2; ++J) {
                const int rr = 16 * J + c16;
                const bool ok = k < N && rr < N;
                const int off = ok ? k * N + rr : 0;
                const double t1 = cold[CL.Ma + off], t2 = cold[CL.Ma2 + off];
                x1[J] = ok ? t1 : 0.0; x2[J] = ok ? t2 : 0.0;
            }
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                const int j = j0 + 16 * I + c16;
                const bool ok = k < N && j < T;
                const double t0 = W.nu[ok ? j * N + k : 0], t1 = dn[ok ? j * N + k : 0];
                zm[I] = ok ? t0 + t1 : 0.0;
            }
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int J = 0; J < 2; ++J) {
                    a[I][J] = MFMA64(zm[I], x1[J], a[I][J]);      // (Ma mu_j)[r]
                    b[I][J] = MFMA64(zm[I], x2[J], b[I][J]);      // (Ma2 mu_j)[r]
                }
        }
        double va[2], va2[2];
#pragma unroll
        for (int J = 0; J < 2; ++J) { va[J] = cold[CL.va + 16 * J + c16]; va2[J] = cold[CL.va2 + 16 * J + c16]; }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int I = e >> 3, J = (e >> 2) & 1, r = e & 3;
            const int j = j0 + 16 * I + 4 * r + g, row = 16 * J + c16;
            const bool ok = row < N && j < T;
            const double nuv = W.nu[ok ? j * N + row : 0];
            const double mu = nuv + dn[ok ? j * N + row : 0];
            if (ok) {
                acc0 += va[J] * (mu + nuv) - nuv * a[I][J][r];
                acc1 += mu * b[I][J][r] - 2.0 * va2[J] * mu;
            }
        }
    }
    acc0 = fw_wave_sum(acc0) - T * cold[CL.sa];
    acc1 = fw_wave_sum(acc1) + T * cold[CL.sa + 1];
    if (lane == 0) { out3[0] = acc0; out3[1] = acc1; }
    fw_mem_fence();
}

// d_z from d_nu (read from this wave's LDS vector).  pass 0: line-search dots out3 = { <r_d,e>, ||e||^2 }
// AND z = zbar + d_z written speculatively for t = 1 (the usual outcome); pass 1 (only if t != 1):
// z = zbar + t d_z rewritten.  nu += t d_nu is done by the caller's last pass.   d_u_j = wc o (B'(d_nu_j) - r_d[u_j]),  r_d[u_j] = cu - B'nu_j.
template <int N>
FW_FN void fw_cold_step(FwKP Pin, int p, double* lds_g, double* out3_g, int pass, double t) {
    const FwKP P = fw_uniform(Pin);
    p = __builtin_amdgcn_readfirstlane(p);
    pass = __builtin_amdgcn_readfirstlane(pass);
    const FwView<N> W(P, p);
    const fw_clds_t sBt = (fw_clds_t)lds_g;
    const fw_lds_t out3 = (fw_lds_t)out3_g;
    const fw_clds_t dn = fw_cold_vec<N>(lds_g, W.mp);
    const fw_clds_t sC = fw_cold_consts<N>(lds_g, W.mp);
    const int lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    const int m = W.m, mp = W.mp, T = W.T, s = W.s;
    if (pass == 0) t = 1.0;
    const double* A1p = P->M.A1; const double* A2p = P->M.A2;
    const double* Q2p = P->M.Q2; const double* Qf2p = P->M.Qf2;
    const double* xmid = P->M.xmid;
    const bool has_xf = W.has_xf != 0, var2 = W.var2 != 0;
    double acc0 = 0.0, acc1 = 0.0;
    for (int j0 = 0; j0 < T; j0 += 32) {
        bool sok[8]; int sj[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = j0 + 16 * (e >> 2) + 4 * (e & 3) + g;
            sok[e] = j < T; sj[e] = sok[e] ? j : 0;
        }
        double an[2][7], ad[2][7];
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {
                const int j = j0 + 16 * I + c16, k = 4 * ks + g;
                const bool ok = k < N && j < T;
                const double t0 = W.nu[ok ? j * N + k : 0], t1 = dn[ok ? j * N + k : 0];
                an[I][ks] = ok ? t0 : 0.0; ad[I][ks] = ok ? (pass == 2 ? t0 + t1 : t1) : 0.0;     // pass 2: mu = nu + d_nu
            }
        for (int J = 0; J * 16 < m; ++J) {
            const int c = 16 * J + c16;
            const bool cok = c < m;
            const int cc = cok ? c : 0;
            d4 g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0}, q0 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0};
            const fw_clds_t br = sBt + c * FW_LDB + g;
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {
                const double bb = br[4 * ks];
                if (pass != 2) {
                    g0 = MFMA64(an[0][ks], bb, g0);
                    g1 = MFMA64(an[1][ks], bb, g1);
                }
                q0 = MFMA64(ad[0][ks], bb, q0);
                q1 = MFMA64(ad[1][ks], bb, q1);
            }
            const double cuc = sC[cc], hcc = sC[mp + cc], wcc = sC[2 * mp + cc], ub = sC[3 * mp + cc];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // pass 0/1: q = B'd_nu, rd = cu - B'nu;  pass 2: q = B'(nu + d_nu), so q - cu = B'd_nu - rd
                const double rd = pass == 2 ? cuc : cuc - ((e >> 2) == 0 ? g0[e & 3] : g1[e & 3]);
                const double du = wcc * (((e >> 2) == 0 ? q0[e & 3] : q1[e & 3]) - rd);
                if (cok && sok[e]) {
                    if (pass == 0) { const double ee = hcc * du; acc0 += rd * ee; acc1 += ee * ee; }
                    W.zp[sj[e] * s + c] = ub + t * du;
                }
            }
        }
        {
            // ---- x entries: d_x_j = (2Q_j)^-1 (-r_d[x_j] - d_nu_{j-1} + A1'd_nu_j + A2'd_nu_{j+1} (- d_nu_T))
            double a1[2][7], a2[2][7];
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int ks = 0; ks < 7; ++ks) {
                    const int j = j0 + 16 * I + c16, k = 4 * ks + g;
                    const bool o1 = k < N && j + 1 < T, o2 = k < N && j + 2 < T && var2;
                    const double t1 = dn[o1 ? (j + 1) * N + k : 0], t2 = dn[o2 ? (j + 2) * N + k : 0];
                    a1[I][ks] = o1 ? t1 : 0.0; a2[I][ks] = o2 ? t2 : 0.0;
                }
            d4 h[2][2];
            double cq2[2], cqf2[2], xbv[2], vprev[2][8], vxf[2][8], rdx[2][8];
#pragma unroll
            for (int J = 0; J < 2; ++J) {
                const int rr = 16 * J + c16;
                const bool rok = rr < N;
                const int rc = rok ? rr : 0;
                h[J][0] = (d4){0, 0, 0, 0}; h[J][1] = (d4){0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < 7; ++ks) {
                    const int k = 4 * ks + g;
                    const bool ok = k < N && rok;
                    const int off = ok ? k * N + rr : 0;
                    const double t1 = A1p[off], t2 = A2p[off];
                    const double b1 = ok ? t1 : 0.0, b2 = ok ? t2 : 0.0;
                    h[J][0] = MFMA64(a1[0][ks], b1, h[J][0]);
                    h[J][1] = MFMA64(a1[1][ks], b1, h[J][1]);
                    h[J][0] = MFMA64(a2[0][ks], b2, h[J][0]);
                    h[J][1] = MFMA64(a2[1][ks], b2, h[J][1]);
                }
                cq2[J] = Q2p[rc]; cqf2[J] = Qf2p[rc]; xbv[J] = xmid[rc];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool last = sj[e] + 1 == T;
                    vprev[J][e] = dn[sj[e] * N + rc];
                    vxf[J][e] = dn[(last && has_xf ? T : sj[e]) * N + rc];
                    rdx[J][e] = W.rdx[sj[e] * N + rc];
                }
            }
            // all loads are issued: only now the stores (a later load would queue behind them)
#pragma unroll
            for (int J = 0; J < 2; ++J) {
                const int rr = 16 * J + c16;
                const bool rok = rr < N;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const double H = (e >> 2) == 0 ? h[J][0][e & 3] : h[J][1][e & 3];
                    const bool last = sj[e] + 1 == T;
                    double v = -rdx[J][e] - vprev[J][e] + H;
                    if (last && has_xf) v -= vxf[J][e];
                    if (rok && sok[e]) W.zp[sj[e] * s + m + rr] = xbv[J] + t * (v * fw_rcp(last ? cqf2[J] : cq2[J]));
                }
            }
        }
    }
    if (pass == 0) {
        acc0 = fw_wave_sum(acc0);
        acc1 = fw_wave_sum(acc1);
        if (lane == 0) { out3[0] = acc0; out3[1] = acc1; }
    }
    fw_mem_fence();
}

// nu += t d_nu (d_nu in this wave's LDS vector)
template <int N>
FW_FN void fw_cold_nu_update(FwKP Pin, int p, double* lds_g, double t) {
    const FwKP P = fw_uniform(Pin);
    p = __builtin_amdgcn_readfirstlane(p);
    const FwView<N> W(P, p);
    const fw_clds_t dn = fw_cold_vec<N>(lds_g, W.mp);
    const int lane = threadIdx.x & 63;
    const int nbn = W.nb * N;
    double v[14];                                  // nb*n <= 840 < 14*64: every load before the first store
#pragma unroll
    for (int q = 0; q < 14; ++q) { const int idx = lane + 64 * q; v[q] = W.nu[idx < nbn ? idx : 0]; }
#pragma unroll
    for (int q = 0; q < 14; ++q) { const int idx = lane + 64 * q; if (idx < nbn) W.nu[idx] = v[q] + t * dn[idx]; }
    fw_mem_fence();
}

// ------------------------------------------------------------------------------------------------
// P3: block-penta-diagonal Cholesky of Y fused with the forward sweep (inf_newton_solver.m:27,30-31)
template <int N, bool EX>
FW_IN int fw_phase_factor(FwKP Pin, int p, double* lds_g, int first) {
    using C = FwCfg<N>;
    constexpr int LD = C::LD, LDG = C::LDG, RC = C::RC;
    const FwKP P = fw_uniform(Pin);
    p = __builtin_amdgcn_readfirstlane(p);
    first = __builtin_amdgcn_readfirstlane(first);
    const FwView<N> W(P, p, first);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const int m = W.m, mp = W.mp, T = W.T, nb = W.nb;
    const double kbar = P->kbar;
    const bool var2 = W.var2 != 0;
    const fw_lds_t lds = (fw_lds_t)lds_g;
    const fw_clds_t sBt = lds;
    const fw_lds_t tA = lds + mp * FW_LDB + wv * C::PER_WAVE;
    const fw_lds_t tB = tA + C::TILE;
    const fw_clds_t sY2 = lds + mp * FW_LDB + FW_WAVES * C::PER_WAVE + FW_WAVES * 4;       // [row j][column c], leading dimension N
    const double* imgs = P->V.img;
    // experiment (FMPC_WAVE_FLAGS bits 0-7): the second half of the workgroup's wavefronts -- the SIMD partners of the first
    // half -- enters the factorisation that many units of 1024 cycles late, so that a SIMD's two problems are half a stage apart
    if ((P->flags & 0xff) && wv >= FW_WAVES / 2)
        for (int q = 0; q < (P->flags & 0xff); ++q) __builtin_amdgcn_s_sleep(16);
    FW_T0();

    d4 Ua[2][2], Ub[2][2], Uc[2][2];
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J) { Ua[I][J] = (d4){0, 0, 0, 0}; Ub[I][J] = Ua[I][J]; Uc[I][J] = Ua[I][J]; }
    int notpd = 0;
    // Rt_i^-1 = 1 / (2R + k (1/s+^2 + 1/s-^2)) of stage i from u_i: three entries per lane (lane, lane + 64, lane + 128),
    // handed to the k-steps of the product below through this wave's LDS tile tA (free at that point).  u_{i+1} is
    // requested a stage ahead.
    constexpr int NWQ = 3;
    double un[NWQ];
#pragma unroll
    for (int q = 0; q < NWQ; ++q) { const int c = lane + 64 * q; un[q] = W.zs[c < m ? c : 0]; }
    for (int i = 0; i < nb; ++i) {
        const double* img = imgs + (size_t)P->V.iD[i] * C::IMG_STRIDE + lane;
        const double* img1 = imgs + (size_t)P->V.i1[i] * C::IMG_STRIDE + C::IMG_D + lane;
        const double* img2 = imgs + (size_t)P->V.i2[i] * C::IMG_STRIDE + C::IMG_D + C::IMG_1;
        const bool y2lds = P->V.i2[i] == P->V.i2[0];                // (uniform) the block the workgroup keeps in LDS
        // Everything this stage reads from memory is REQUESTED here and CONSUMED behind the 108 products below (3.7 us): the
        // constant tiles and the rhs column are added to S afterwards.  (A load waits for every older store of the wave --
        // vmcnt is in order -- and the previous stage has just issued 54 factor stores: consumed here, each stage would
        // stand still until those are written.)
        d4 I00, I01, I11;
        d4 M00, M01, M10, M11;
        {
            const d2v* ip = (const d2v*)(img - lane + C::IMG_P) + lane;                       // paired copies (FwCfg::IMG_P)
            const d2v* mp2 = (const d2v*)(img1 - lane - C::IMG_D + C::IMG_P + C::IMG_D) + lane;
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
                const d2v a = ip[(0 * 2 + rp) * 64], b = ip[(1 * 2 + rp) * 64], c = ip[(2 * 2 + rp) * 64];
                I00[2 * rp] = a.x; I00[2 * rp + 1] = a.y; I01[2 * rp] = b.x; I01[2 * rp + 1] = b.y; I11[2 * rp] = c.x; I11[2 * rp + 1] = c.y;
            }
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
                const d2v a = mp2[(0 * 2 + rp) * 64], b = mp2[(1 * 2 + rp) * 64], c = mp2[(2 * 2 + rp) * 64], d = mp2[(3 * 2 + rp) * 64];
                M00[2 * rp] = a.x; M00[2 * rp + 1] = a.y; M01[2 * rp] = b.x; M01[2 * rp + 1] = b.y;
                M10[2 * rp] = c.x; M10[2 * rp + 1] = c.y; M11[2 * rp] = d.x; M11[2 * rp + 1] = d.y;
            }
        }
        double rh0[4], rh1[4];                   // rhs_i rides in column n of the tile
        {
            const double* rh = W.rhs + i * N;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rh0[r] = rh[4 * r + g];
                const int row1 = 16 + 4 * r + g;
                rh1[r] = rh[row1 < N ? row1 : N - 1];
            }
        }
        d4 S00 = {0, 0, 0, 0}, S01 = {0, 0, 0, 0}, S11 = {0, 0, 0, 0};
        // ---- S += B Rt_i^-1 B'  (K = m; three subtiles by symmetry; chunks of FW_KCH k-steps).
        //      (No software pipelining against the VALU loop below: on gfx950 the fp64 MFMA and
        //      VALU instructions of a SIMD do not execute concurrently -- measured, see DESIGN.md.)
        if (i < T) {
            {
                const double* umaxp = P->M.umax; const double* uminp = P->M.umin; const double* R2p = P->M.R2;
                double cmx[NWQ], cmn[NWQ], cr2[NWQ], uc[NWQ];
#pragma unroll
                for (int q = 0; q < NWQ; ++q) {
                    const int c = lane + 64 * q, cc = c < m ? c : 0;
                    cmx[q] = umaxp[cc]; cmn[q] = uminp[cc]; cr2[q] = R2p[cc]; uc[q] = un[q];
                }
                const double* znext = W.zs + (size_t)(i + 1 < T ? i + 1 : i) * (N + m);
#pragma unroll
                for (int q = 0; q < NWQ; ++q) { const int c = lane + 64 * q; un[q] = znext[c < m ? c : 0]; }
#pragma unroll
                for (int q = 0; q < NWQ; ++q) {
                    const int c = lane + 64 * q;
                    const double dp = fw_rcp(cmx[q] - uc[q]), dm = fw_rcp(uc[q] - cmn[q]);
                    const double wv = fw_rcp(cr2[q] + kbar * (dp * dp + dm * dm));
                    if (c < mp) tA[c] = c < m ? wv : 0.0;              // zero weight on the k-step padding
                }
            }
            fw_wave_fence();
            const fw_clds_t wi = tA + g;
            const fw_clds_t bp = sBt + g * FW_LDB + c16;
            for (int kc = 0; kc < mp; kc += 4 * FW_KCH) {
                double b0[FW_KCH], b1[FW_KCH], wk[FW_KCH];
#pragma unroll
                for (int q = 0; q < FW_KCH; ++q) {
                    wk[q] = wi[kc + 4 * q];
                    b0[q] = bp[(kc + 4 * q) * FW_LDB];
                    b1[q] = bp[(kc + 4 * q) * FW_LDB + 16];
                }
#pragma unroll
                for (int q = 0; q < FW_KCH; ++q) {
                    const double a0 = b0[q] * wk[q], a1 = b1[q] * wk[q];
                    S00 = MFMA64(a0, b0[q], S00);
                    S01 = MFMA64(a0, b1[q], S01);
                    S11 = MFMA64(a1, b1[q], S11);
                }
            }
        }
        // ---- S += constant part (column n: the rhs; B has no row n, so the products left that column zero)
        {
            const bool isr = c16 == RC - 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                S00[r] += I00[r];
                S01[r] = isr ? rh0[r] : S01[r] + I01[r];
                S11[r] = isr ? rh1[r] : S11[r] + I11[r];
            }
        }
        // ---- S -= Ua'Ua + Uc'Uc ;  M1 = Y_{i,i+1} - Ua'Ub
#pragma unroll
        for (int Ix = 0; Ix < 2; ++Ix)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * Ix + 4 * r >= N) continue;          // rows >= n of the U tiles are zero
                const double na0 = -Ua[Ix][0][r], na1 = -Ua[Ix][1][r];
                S00 = MFMA64(na0, Ua[Ix][0][r], S00);
                S01 = MFMA64(na0, Ua[Ix][1][r], S01);
                S11 = MFMA64(na1, Ua[Ix][1][r], S11);
                if (var2) {
                    M00 = MFMA64(na0, Ub[Ix][0][r], M00);
                    M01 = MFMA64(na0, Ub[Ix][1][r], M01);
                    M10 = MFMA64(na1, Ub[Ix][0][r], M10);
                    M11 = MFMA64(na1, Ub[Ix][1][r], M11);
                    const double nc0 = -Uc[Ix][0][r], nc1 = -Uc[Ix][1][r];
                    S00 = MFMA64(nc0, Uc[Ix][0][r], S00);
                    S01 = MFMA64(nc0, Uc[Ix][1][r], S01);
                    S11 = MFMA64(nc1, Uc[Ix][1][r], S11);
                }
            }
        FW_TICK(0);
        // ---- MFMA tiles -> LDS.  tA: S with its lower triangle complete; tB: M1 with the rhs in
        //      column n.  Tiles have 32 rows and a dump column (28): no predicates needed.
        {
            const int cA = 16 + c16 < 28 ? 16 + c16 : 28;                // clamped column for J = 1
            const fw_lds_t pa = tA + g * LD + c16;                        // (4r+g, c16)
            const fw_lds_t pt = tA + (16 + c16 < N ? 16 + c16 : 31) * LD + g;   // transposed (16+c16, 4r+g); pads -> row 31
            const fw_lds_t pa11 = tA + (16 + g) * LD + cA;
            const fw_lds_t pb0 = tB + g * LD;
            const bool isrhs = c16 == RC - 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pa[4 * r * LD] = S00[r];
                pt[4 * r] = S01[r];
                pa11[4 * r * LD] = S11[r];
                pb0[4 * r * LD + c16] = M00[r];
                pb0[4 * r * LD + cA] = isrhs ? S01[r] : M01[r];
                pb0[(16 + 4 * r) * LD + c16] = M10[r];
                pb0[(16 + 4 * r) * LD + cA] = isrhs ? S11[r] : M11[r];
            }
        }
        fw_wave_fence();
        // ---- row r of S on lane r; column c of [M1 | rhs | . | Y2 | rhs] on lane c
        double row[N], x[N];
        {
            const int rr = lane < N ? lane : N;                      // lanes >= n read a pad row
            const int cl = lane & 31;
            const fw_clds_t pr = tA + rr * LD;
#pragma unroll
            for (int j = 0; j < N; ++j) row[j] = pr[j];
            if (lane < 32 || cl == RC) {
                const fw_clds_t pc = tB + (cl < 28 ? cl : 28);
#pragma unroll
                for (int j = 0; j < N; ++j) x[j] = pc[j * LD];
            } else if (y2lds) {
                // the columns of Y_{i,i+2} from the workgroup's LDS copy (the block is the same for all stages but the last ones):
                // from memory these were 27 loads per stage consumed right away -- an L2 round trip in front of every factorisation
                const fw_clds_t pc = sY2 + (cl < N ? cl : 0);
#pragma unroll
                for (int j = 0; j < N; ++j) x[j] = pc[j * N];
            } else {
                const double* pc = img2 + cl;
#pragma unroll
                for (int j = 0; j < N; ++j) x[j] = pc[j * 32];
            }
        }
        fw_wave_fence();
        FW_TICK(1);
        // ---- fused potrf (lane = row) + forward substitution (lane = column)
        // Column k of L (one entry per lane) reaches all lanes THROUGH LDS: one ds_write of the column and
        // uniform-address reads (two entries per ds_read2) instead of two v_readlane + hazard wait states per entry --
        // the vector unit then issues little more than the 2 x 351 fma of the updates, and LDS instructions have their
        // own issue port.  Column k + 1 is final after the first update of step k: it is scaled and written out
        // before the rest of step k, whose fma cover the LDS round trip (and so does the SIMD's other wave).
        const fw_lds_t colb = tA;                    // 2 x 64 doubles (tA and tB are free until the results are written);
                                                     // every lane writes (no predicate: a conditional store is a branch)
        constexpr int CH = 8;                        // entries of a column in flight per request (registers)
        double myrs = 1.0;
        double lk, xk;
        {
            const double d = fw_readlane(row[0], 0);
            if (!(d > 0.0) || isinf(d)) notpd = 1;
            const double rs = fw_rsqrt(d);
            lk = row[0] * rs; xk = x[0] * rs;        // L[r][0] on lane r (r > 0)
            row[0] = lk; x[0] = xk;
            if (lane == 0) myrs = rs;
            colb[lane] = lk;
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            fw_wave_fence();
            const fw_clds_t cb = colb + (k & 1) * 64;                  // L[c][k]: one address for all lanes
            double lc[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) lc[q] = cb[k + 1 + q < N ? k + 1 + q : N - 1];
            double lkn = 0.0, xkn = 0.0;
            if (k + 1 < N) {
                // the entry the dependency chain runs through (pivot of step k + 1) comes by v_readlane: the chain
                // then holds no LDS round trip, which only feeds the updates that are off the critical path
                const double l1 = fw_readlane(lk, k + 1);
                row[k + 1] = fma(-lk, l1, row[k + 1]);
                x[k + 1] = fma(-l1, xk, x[k + 1]);
                const double d = fw_readlane(row[k + 1], k + 1);
                if (!(d > 0.0) || isinf(d)) notpd = 1;
                const double rs = fw_rsqrt(d);
                lkn = row[k + 1] * rs; xkn = x[k + 1] * rs;
                row[k + 1] = lkn; x[k + 1] = xkn;
                if (lane == k + 1) myrs = rs;
                colb[((k + 1) & 1) * 64 + lane] = lkn;
            }
            // (requesting the head of column k + 1 here, ahead of the rest of step k, was measured: 1 % slower)
#pragma unroll
            for (int c0 = k + 1; c0 < N; c0 += CH) {
                double ln[CH];
                if (c0 + CH < N) {
#pragma unroll
                    for (int q = 0; q < CH; ++q) ln[q] = cb[c0 + CH + q < N ? c0 + CH + q : N - 1];
                }
#pragma unroll
                for (int q = (c0 == k + 1 ? 1 : 0); q < CH; ++q) {
                    const int c = c0 + q;
                    if (c < N) { row[c] = fma(-lk, lc[q], row[c]); x[c] = fma(-lc[q], xk, x[c]); }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < CH; ++q) lc[q] = ln[q];
            }
            lk = lkn; xk = xkn;
        }
        fw_wave_fence();
        FW_TICK(2);
        // ---- results: U1|y -> tB, U2|y -> tA (layout change) and the factor to HBM
        {
            double* f = W.fac + (size_t)i * W.fstride;
            const int cl = lane & 31;
            const bool hi = lane >= 32;
            const fw_lds_t tdst = (hi ? tA : tB) + (cl < 28 ? cl : 28);
#pragma unroll
            for (int j = 0; j < N; ++j) tdst[j * LD] = x[j];
            constexpr bool ex = EX;                                  // (the export launch is an instance of its own)
            if (ex && cl < N) {
                // the exported shared factor: U1, U2 row-major tiles
                double* gdst = f + (hi ? 2 : 1) * N * LDG + cl;
#pragma unroll
                for (int j = 0; j < N; ++j) gdst[j * LDG] = x[j];
            }
            if (lane < N) {
                if (ex) {
                    // column j of L contiguous in r.  Shared (exported) tiles carry exact zeros on and above
                    // the diagonal so that the shared sweeps need no per-step masking
                    double* gl = f + lane;
#pragma unroll
                    for (int j = 0; j < N; ++j) gl[j * LDG] = (j >= lane) ? 0.0 : row[j];
                }
                W.rsg[i * 32 + lane] = myrs;
            }
            if (ex) {
                // transposed copies for the shared backward sweep: tile 3 = L as [r][j], 4/5 = U1'/U2' as [c][r]
                if (cl < N) {
                    double* gt = f + (hi ? 5 : 4) * N * LDG + cl * LDG;
#pragma unroll
                    for (int j = 0; j < N; ++j) gt[j] = x[j];
                }
                if (lane < N) {
                    double* gt = f + 3 * N * LDG + lane * LDG;
#pragma unroll
                    for (int j = 0; j < N; ++j) gt[j] = j < lane ? row[j] : 0.0;
                }
            }
        }
        fw_wave_fence();
        // ---- rotate and read the new U tiles back in MFMA layout (pad rows must be exact zeros)
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int J = 0; J < 2; ++J) Uc[I][J] = Ub[I][J];
        {
            const int cB = 16 + c16 < 28 ? 16 + c16 : 28;
            const fw_clds_t pa = tB + g * LD;
            const fw_clds_t pb = tA + g * LD;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Ua[0][0][r] = pa[4 * r * LD + c16];
                Ua[0][1][r] = pa[4 * r * LD + cB];
                Ub[0][0][r] = pb[4 * r * LD + c16];
                Ub[0][1][r] = pb[4 * r * LD + cB];
                if (16 + 4 * r < N) {
                    const bool ok = 16 + 4 * r + 3 < N || 16 + 4 * r + g < N;
                    const double v0 = pa[(16 + 4 * r) * LD + c16], v1 = pa[(16 + 4 * r) * LD + cB];
                    const double w0 = pb[(16 + 4 * r) * LD + c16], w1 = pb[(16 + 4 * r) * LD + cB];
                    Ua[1][0][r] = ok ? v0 : 0.0; Ua[1][1][r] = ok ? v1 : 0.0;
                    Ub[1][0][r] = ok ? w0 : 0.0; Ub[1][1][r] = ok ? w1 : 0.0;
                } else {
                    Ua[1][0][r] = 0.0; Ua[1][1][r] = 0.0; Ub[1][0][r] = 0.0; Ub[1][1][r] = 0.0;
                }
            }
        }
        if (lane < N) W.yv[i * N + lane] = tB[lane * LD + RC];
        if (!EX) {
            // ---- the factor to HBM (per-problem factor: U1 and L; U2 = L^-1 Y_{i,i+2} is not streamed, the backward sweep applies
            //      L^-1 to the constant block times d_nu_{i+2} instead): tB holds U1 | y as the backward sweep wants it, and the
            //      strict lower triangle of L goes, folded, into tA (its U2 tile has just been read back) -- both leave as linear
            //      dumps, 16 bytes per lane: 7 + 3 instructions, 10 KB per stage
            double* f = W.fac + (size_t)i * W.fstride;
            d2v* f2 = (d2v*)f + lane;
            const fw_c2lds_t tB2 = (fw_c2lds_t)tB + lane;
            d2v du[C::DUMP / 128];
#pragma unroll
            for (int q = 0; q < C::DUMP / 128; ++q) du[q] = tB2[q * 64];
            {
                // lane r: L[r][j], j < r, to its half of a folded row; entries on and above the diagonal, and the lanes beyond
                // N, go to a per-lane parking slot behind the folded rows (an address select: no branch, no store under a condition)
                // No select and no per-entry predicate: every lane 1 .. N-1 writes ALL its N - 1 entries at base + j, j DESCENDING.
                // What a lane writes beyond its own r entries lands on another row's slots, and always on one whose owner writes
                // it at a SMALLER j, i.e. later: a small row r (base r LDF) spills over the slots of its partner N-1-r, which writes
                // slot c at j = c - r < c; a big row rho (base (N-1-rho)(LDF + 1)) spills into column N-1 (nobody's) and the head of
                // the next folded row, written by small row N-rho at j - rho - 1 < j.  Row 0 has no entries: its lane stays out
                // (its spill would meet row N-1's slots in the SAME instruction).  A wavefront's LDS writes keep their order.
                if (lane >= 1 && lane < N) {
                    // (hand-issued: the compiler would pair the stores into ds_write2_b64 with the LOWER slot first, and is free to
                    //  reorder stores that cannot alias within a lane -- the order across lanes is what counts here)
                    const unsigned pr = (unsigned)(size_t)(tA + (lane <= (N - 1) / 2 ? lane * C::LDF : (N - 1 - lane) * C::LDF + (N - 1 - lane)));
#pragma unroll
                    for (int j = N - 2; j >= 0; --j)
                        asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(pr), "v"(row[j]), "n"(8 * j) : "memory");
                }
            }
#pragma unroll
            for (int q = 0; q < C::DUMP / 128; ++q) f2[q * 64] = du[q];
            fw_wave_fence();
            const fw_c2lds_t tA2 = (fw_c2lds_t)tA + lane;
            d2v dlf[C::LFOLD / 128];
#pragma unroll
            for (int q = 0; q < C::LFOLD / 128; ++q) dlf[q] = tA2[q * 64];
#pragma unroll
            for (int q = 0; q < C::LFOLD / 128; ++q) f2[(C::DUMP / 128 + q) * 64] = dlf[q];
        }
        fw_wave_fence();
        FW_TICK(3);
    }
    FW_TFLUSH(0);
    fw_mem_fence();
    return __ballot(notpd) != 0ull ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// P4: backward sweep, d_nu_i = L^-T (y_i - U1 d_nu_{i+1} - U2 d_nu_{i+2})   (inf_newton_solver.m:32)
// Only L_i and U1_i come back from HBM.  U2_i = L_i^-1 Y_{i,i+2} with a CONSTANT block Y_{i,i+2}, so
//     U2_i d_nu_{i+2} = L_i^-1 (Y_{i,i+2} d_nu_{i+2}):
// a product with a constant block (its rows stay in registers) and one more forward substitution with L_i, whose
// rows the lanes hold anyway -- a third less factor traffic in both directions for ~130 vector instructions per stage.
// The factor tiles are read with coalesced loads (one tile row across the lanes per instruction)
// and turned to the row-/column-per-lane layouts through this wave's LDS tile.
// Per-problem factor (round 4): the stream holds, per stage, the two LDS tiles of the factor phase as linear dumps (FwCfg::DUMP):
// 14 loads of 16 bytes per lane bring a stage back, they go into tA (U1 | y, [row][column]) and tB (L, [row r][column j]) as they
// are -- no transposition -- and the lanes read what they need: row lr of U1, row lr of L for the forward substitution,
// column lr of L for the backward one.  The loads of stage i - 1 are requested before stage i's substitutions.
template <int N>
FW_IN void fw_phase_backward_dump(FwKP Pin, int p, double* lds_g) {
    using C = FwCfg<N>;
    constexpr int LD = C::LD, NQ = C::DUMP / 128, NL = C::LFOLD / 128;
    const FwKP P = fw_uniform(Pin);
    p = __builtin_amdgcn_readfirstlane(p);
    const FwView<N> W(P, p);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const fw_lds_t lds = (fw_lds_t)lds_g;
    const fw_lds_t tA = lds + W.mp * FW_LDB + wv * C::PER_WAVE;
    const fw_lds_t tB = tA + C::TILE;
    typedef __attribute__((address_space(3))) d2v* fw_2lds_t;
    const fw_2lds_t tA2 = (fw_2lds_t)tA + lane, tB2 = (fw_2lds_t)tB + lane;
    const int lr = lane < N ? lane : N - 1;
    const int lbase = lr <= (N - 1) / 2 ? lr * C::LDF : (N - 1 - lr) * C::LDF + (N - 1 - lr);       // this lane's row of L in the folded tile
    const double* facp = W.fac;
    const double* rsp = W.rsg;
    const double* imgs = P->V.img + C::IMG_D + C::IMG_1;       // [block][row j][col c] copies of the constant blocks
    double x1 = 0.0, x2 = 0.0;        // lane j: d_nu_{i+1}[j], d_nu_{i+2}[j]
    // one stage in flight: (du, dl) = the dumps of the stage about to be used (two in flight spilled: docs/DESIGN_HISTORY.md)
    d2v du[NQ], dl[NL];
    double y2[N], yv_n, rs_n;
    int cur2 = -1;                    // block whose rows are in y2
    {
        const int i = W.nb - 1;
        const d2v* f2 = (const d2v*)(facp + (size_t)i * W.fstride) + lane;
#pragma unroll
        for (int q = 0; q < NQ; ++q) du[q] = f2[q * 64];
#pragma unroll
        for (int q = 0; q < NL; ++q) dl[q] = f2[(NQ + q) * 64];
        yv_n = W.yv[i * N + lr]; rs_n = rsp[i * 32 + lr];
    }
    for (int i = W.nb - 1; i >= 0; --i) {
        const int ip = i >= 1 ? i - 1 : 0;                     // stage to request now (a harmless re-read at the end)
        double v = yv_n;
        const double rsv = rs_n;
        const int b2 = P->V.i2[i];
        if (b2 != cur2) {                                      // (wave-uniform; changes once or twice per sweep)
            const double* yr = imgs + (size_t)b2 * C::IMG_STRIDE + lr * 32;
#pragma unroll
            for (int c = 0; c < N; ++c) y2[c] = yr[c];
            cur2 = b2;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) tA2[q * 64] = du[q];
#pragma unroll
        for (int q = 0; q < NL; ++q) tB2[q * 64] = dl[q];
        fw_wave_fence();
        // request the next stage into the registers that have just been written out
        {
            const d2v* f2 = (const d2v*)(facp + (size_t)ip * W.fstride) + lane;
#pragma unroll
            for (int q = 0; q < NQ; ++q) du[q] = f2[q * 64];
#pragma unroll
            for (int q = 0; q < NL; ++q) dl[q] = f2[(NQ + q) * 64];
        }
        yv_n = W.yv[ip * N + lr]; rs_n = rsp[ip * 32 + lr];
        double cc = 0.0;                                       // (Y_{i,i+2} d_nu_{i+2})[lr]
        {
            const fw_clds_t r1 = tA + lr * LD;
#pragma unroll
            for (int c = 0; c < N; ++c) {
                const double a = fw_readlane(x1, c), bb = fw_readlane(x2, c);
                v = fma(-r1[c], a, v);
                cc = fma(y2[c], bb, cc);
            }
        }
        // forward substitution  w = L^-1 cc  (lane r: row r of L from the folded tile; entries k >= r are other rows' data,
        // they only touch values nobody reads);  v -= w
        {
            const fw_clds_t gl = tB + lbase;
            double glr[N - 1];
#pragma unroll
            for (int k = 0; k < N - 1; ++k) glr[k] = gl[k];
            double wres = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const double wk = fw_readlane(cc * rsv, k);
                if (lane == k) wres = wk;
                if (k < N - 1) cc = fma(-glr[k], wk, cc);          // meaningful on lanes > k only
            }
            v -= wres;
        }
        double res = 0.0;
        {
            // column lr of L: L[r][lr] (r > lr) sits at LBASE(r) + lr of the folded tile
            double clr[N];
#pragma unroll
            for (int r = 1; r < N; ++r) clr[r] = tB[C::LBASE(r) + lr];
#pragma unroll
            for (int r = N - 1; r >= 0; --r) {
                const double xr = fw_readlane(v * rsv, r);
                if (lane == r) res = xr;
                if (r >= 1) v = fma(-clr[r], xr, v);        // meaningful on lanes < r only
            }
        }
        fw_wave_fence();
        if (lane < N) W.dnu[i * N + lane] = res;
        x2 = x1;
        x1 = res;
    }
    fw_mem_fence();
}

template <int N>
FW_IN void fw_phase_backward(FwKP Pin, int p, double* lds_g) {
    using C = FwCfg<N>;
    constexpr int LDG = C::LDG, LD = C::LD;
    const FwKP P = fw_uniform(Pin);
    p = __builtin_amdgcn_readfirstlane(p);
    const FwView<N> W(P, p);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const fw_lds_t lds = (fw_lds_t)lds_g;
    const fw_lds_t tA = lds + W.mp * FW_LDB + wv * C::PER_WAVE;
    const fw_lds_t tB = tA + C::TILE;
    const int lr = lane < N ? lane : N - 1;
    const int lc = lane < LDG ? lane : LDG - 1;       // tile rows are LDG doubles long in HBM
    const double* facp = W.fac;
    const bool ex = true;                                      // (only the export launch takes this form: full row-major tiles)
    const int u1o = (ex ? N * LDG : C::LPACK) + lc;
    const int lo = ex ? lc : lr;
    const double* rsp = W.rsg;
    const double* imgs = P->V.img + C::IMG_D + C::IMG_1;       // [block][row j][col c] copies of the constant blocks
    double x1 = 0.0, x2 = 0.0;        // lane j: d_nu_{i+1}[j], d_nu_{i+2}[j]
    // software pipeline: each register set is reloaded for stage i-1 as soon as stage i has consumed it
    double g1[N], gl[N], y2[N], yv_n, rs_n;
    int cur2 = -1;                    // block whose rows are in y2
    {
        const int i = W.nb - 1;
        const double* f = facp + (size_t)i * W.fstride;
#pragma unroll
        for (int j = 0; j < N; ++j) { g1[j] = f[u1o + j * LDG]; gl[j] = f[(ex ? j * LDG : C::LOFF(j) - j) + lo]; }
        yv_n = W.yv[i * N + lr]; rs_n = rsp[i * 32 + lr];
    }
    for (int i = W.nb - 1; i >= 0; --i) {
        const int ip = i > 0 ? i - 1 : 0;                      // stage to prefetch (harmless re-read at i = 0)
        const double* fp = facp + (size_t)ip * W.fstride;
        double v = yv_n;
        const double rsv = rs_n;
        const int b2 = P->V.i2[i];
        if (b2 != cur2) {                                      // (wave-uniform; changes once or twice per sweep)
            const double* yr = imgs + (size_t)b2 * C::IMG_STRIDE + lr * 32;
#pragma unroll
            for (int c = 0; c < N; ++c) y2[c] = yr[c];
            cur2 = b2;
        }
        // U1 -> tA (row-major tile), then row lr on lane lr
#pragma unroll
        for (int j = 0; j < N; ++j) tA[j * LD + lc] = g1[j];
        fw_wave_fence();
#pragma unroll
        for (int j = 0; j < N; ++j) g1[j] = fp[u1o + j * LDG];
        yv_n = W.yv[ip * N + lr]; rs_n = rsp[ip * 32 + lr];
        double cc = 0.0;                                       // (Y_{i,i+2} d_nu_{i+2})[lr]
        {
            const fw_clds_t r1 = tA + lr * LD;
#pragma unroll
            for (int c = 0; c < N; ++c) {
                const double a = fw_readlane(x1, c), bb = fw_readlane(x2, c);
                v = fma(-r1[c], a, v);
                cc = fma(y2[c], bb, cc);
            }
        }
        // forward substitution  w = L^-1 cc  (lane r holds row r of L in gl: gl[k] = L[r][k]);  v -= w
        {
            double wres = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const double wk = fw_readlane(cc * rsv, k);
                if (lane == k) wres = wk;
                cc = fma(-gl[k], wk, cc);          // meaningful on lanes > k only
            }
            v -= wres;
        }
        fw_wave_fence();
        // L -> tB as [column j][row r]; lane j then reads its column (row j of the LDS tile)
#pragma unroll
        for (int j = 0; j < N; ++j) tB[j * LD + lc] = gl[j];
        fw_wave_fence();
#pragma unroll
        for (int j = 0; j < N; ++j) gl[j] = fp[(ex ? j * LDG : C::LOFF(j) - j) + lo];      // (lanes above the diagonal read the tail of an earlier column: unused)
        double res = 0.0;
        {
            const fw_clds_t cl = tB + lr * LD;
#pragma unroll
            for (int r = N - 1; r >= 0; --r) {
                const double xr = fw_readlane(v * rsv, r);
                if (lane == r) res = xr;
                v = fma(-cl[r], xr, v);        // meaningful on lanes < r only
            }
        }
        if (lane < N) W.dnu[i * N + lane] = res;
        x2 = x1;
        x1 = res;
    }
    fw_mem_fence();
}

// ------------------------------------------------------------------------------------------------
// Sweeps against the handle's SHARED factor (first Newton step from a cold start: every problem has
// the same Phi, Y and Cholesky factor, only the right-hand side differs; SURVEY §7.2a regime (ii)).
//     forward   y_i    = L_ii^-1 (rhs_i - U_{i-1,i}' y_{i-1} - U_{i-2,i}' y_{i-2})
//     backward  d_nu_i = L_ii^-T (y_i  - U_{i,i+1} d_nu_{i+1} - U_{i,i+2} d_nu_{i+2})
// Six tiles per stage in the shared buffer: [0] L with f[j][r] = L[r][j], [1] U1, [2] U2 row-major,
// [3] L row-major, [4] U1', [5] U2'; the L tiles are exactly zero on and above the diagonal.
// WORKGROUP-COLLECTIVE: all 8 waves walk the stages together; the three tiles of a stage (18 KB) are
// staged ONCE per workgroup into LDS (double buffered in the per-wave tile region, idle here; one
// barrier per stage) and every wave reads its rows from LDS.  Every wave of the workgroup must call
// this function; `go` = 0 makes a wave take part in the staging only.
template <int N, int BWD>
FW_FN void fw_phase_sweep_shared(FwKP Pin, int p, int go, double* lds_g) {
    using C = FwCfg<N>;
    constexpr int LDG = C::LDG, TS = N * LDG, ST = 3 * TS, NPRE = (ST + FW_THREADS - 1) / FW_THREADS;
    static_assert(FW_WAVES != 8 || 2 * ST + FW_WAVES * FW_VEC_STRIDE + 4 * 192 <= FW_WAVES * C::PER_WAVE, "stage buffers + vectors must fit the tile region");   // (experiment builds with FW_WAVES != 8 run the explicit-start path only)
    const FwKP P = fw_uniform(Pin);
    p = __builtin_amdgcn_readfirstlane(p);
    go = __builtin_amdgcn_readfirstlane(go);
    const FwView<N> W(P, go ? p : 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int lr = lane < N ? lane : N - 1;
    const int nb = W.nb;
    const fw_lds_t buf = (fw_lds_t)lds_g + W.mp * FW_LDB;
    const double* fac = P->sh_fac;
    const double* rsp = P->sh_rs + lr;
    const fw_lds_t vl = fw_cold_vec<N>(lds_g, W.mp);          // rhs -> y (forward), y -> d_nu (backward), in place
    // tiles used by stage i:  forward: (i,0), (i-1,1), (i-2,2);  backward: (i,3), (i,4), (i,5)
    auto tile = [&](int i, int which) -> const double* {
        if (BWD) return fac + ((size_t)i * 6 + 3 + which) * TS;
        const int st = i - which;
        return fac + ((size_t)(st < 0 ? 0 : st) * 6 + which) * TS;
    };
    const int first = BWD ? nb - 1 : 0, step = BWD ? -1 : 1;
#pragma unroll
    for (int r = 0; r < NPRE; ++r) {
        const int idx = tid + FW_THREADS * r;
        if (idx < ST) buf[idx] = tile(first, idx / TS)[idx % TS];
    }
    __syncthreads();
    double ya = 0.0, yb = 0.0;                   // lane j: the two previously computed block entries
    for (int q = 0, i = first; q < nb; ++q, i += step) {
        const fw_clds_t cur = buf + (q & 1) * ST;
        const fw_lds_t nxt = buf + ((q + 1) & 1) * ST;
        const bool more = q + 1 < nb;
        double pre[NPRE];
        if (more) {
#pragma unroll
            for (int r = 0; r < NPRE; ++r) {
                const int idx = tid + FW_THREADS * r;
                pre[r] = idx < ST ? tile(i + step, idx / TS)[idx % TS] : 0.0;
            }
        }
        if (go) {
            double sv = vl[i * N + lr];
            const double rsv = rsp[i * 32];
            const bool has1 = BWD ? (i + 1 < nb) : (i >= 1);
            const bool has2 = BWD ? (i + 2 < nb) : (i >= 2);
            if (has1) {
#pragma unroll
                for (int j = 0; j < N; ++j) sv = fma(-cur[TS + j * LDG + lr], fw_readlane(ya, j), sv);
            }
            if (has2) {
#pragma unroll
                for (int j = 0; j < N; ++j) sv = fma(-cur[2 * TS + j * LDG + lr], fw_readlane(yb, j), sv);
            }
            // substitution: L is exactly zero on/above the diagonal, so lane j's sv is final after
            // step j-1 (forward) / j+1 (backward) and x_j = sv * rs for every lane at the end
            double lcol[N];
#pragma unroll
            for (int j = 0; j < N; ++j) lcol[j] = cur[j * LDG + lr];
            if (BWD) {
#pragma unroll
                for (int j = N - 1; j >= 1; --j) sv = fma(-lcol[j], fw_readlane(sv * rsv, j), sv);
            } else {
#pragma unroll
                for (int j = 0; j < N - 1; ++j) sv = fma(-lcol[j], fw_readlane(sv * rsv, j), sv);
            }
            const double res = sv * rsv;
            if (lane < N) vl[i * N + lane] = res;
            yb = ya;
            ya = res;
        }
        if (more) {
#pragma unroll
            for (int r = 0; r < NPRE; ++r) {
                const int idx = tid + FW_THREADS * r;
                if (idx < ST) nxt[idx] = pre[r];
            }
        }
        __syncthreads();
    }
    fw_mem_fence();
}

// ------------------------------------------------------------------------------------------------
// nu += t d_nu   (backtracking_inf_newton.m:11); every load before the first store
template <int N>
FW_IN void fw_phase_nu_update(FwKP Pin, int p, double t) {
    const FwKP P = fw_uniform(Pin);
    p = __builtin_amdgcn_readfirstlane(p);
    const FwView<N> W(P, p);
    const int lane = threadIdx.x & 63;
    const int nbn = W.nb * N;
    for (int base = 0; base < nbn; base += 64 * 14) {
        double a[14], d[14];
#pragma unroll
        for (int q = 0; q < 14; ++q) { const int idx = base + lane + 64 * q; const int ic = idx < nbn ? idx : 0; a[q] = W.nu[ic]; d[q] = W.dnu[ic]; }
#pragma unroll
        for (int q = 0; q < 14; ++q) { const int idx = base + lane + 64 * q; if (idx < nbn) W.nu[idx] = a[q] + t * d[q]; }
    }
    fw_mem_fence();
}

// The rare case of a step length t != 1 (backtracking_inf_newton.m:10): fw_phase_CT<1> has written z + d_z and left
// d_z in the workspace; z + t d_z = (z + d_z) + (t - 1) d_z.  Loads in groups of 8 ahead of the stores.
template <int N>
FW_FN void fw_phase_zfix(FwKP Pin, int p, double t) {
    const FwKP P = fw_uniform(Pin);
    p = __builtin_amdgcn_readfirstlane(p);
    const FwView<N> W(P, p);
    const int lane = threadIdx.x & 63;
    const int Nz = W.T * W.s, m = W.m, s = W.s;
    const double tm1 = t - 1.0;
    for (int base = 0; base < Nz; base += 64 * 8) {
        double zv[8], dv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = base + lane + 64 * q, ic = idx < Nz ? idx : 0;
            const int j = ic / s, e = ic - j * s;
            zv[q] = W.zp[ic];
            dv[q] = e < m ? W.rdu[j * m + e] : W.rdx[j * N + e - m];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int idx = base + lane + 64 * q; if (idx < Nz) W.zp[idx] = zv[q] + tm1 * dv[q]; }
    }
    fw_mem_fence();
}

// z_out = z_init for a problem that left before its first step (exit test met at the start point, or an error)
template <int N>
FW_FN void fw_phase_zcopy(FwKP Pin, int p) {
    const FwKP P = fw_uniform(Pin);
    p = __builtin_amdgcn_readfirstlane(p);
    const FwView<N> W(P, p, 1);
    const int lane = threadIdx.x & 63;
    const int Nz = W.T * W.s;
    for (int base = 0; base < Nz; base += 64 * 16) {
        double zv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int idx = base + lane + 64 * q; zv[q] = W.zs[idx < Nz ? idx : 0]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int idx = base + lane + 64 * q; if (idx < Nz) W.zp[idx] = zv[q]; }
    }
    fw_mem_fence();
}

// ------------------------------------------------------------------------------------------------
// The phases of a Newton iteration as THREE non-inlined functions (own register allocation each).  A non-inlined function
// that uses all 256 registers saves and restores the ~112 callee-saved ones on every call: 57 KB of scratch traffic per
// call and problem, which is HBM traffic at 2048 resident problems -- seven calls per iteration were 0.4 MB per problem,
// a quarter of everything the kernel moved (rocprofv3 FETCH_SIZE / WRITE_SIZE against the byte count of the algorithm).
//   pre : [start point,] r_d, r_p, norms, right-hand side          out: red[0] = |r_d|^2, red[1] = |r_p|^2, red[2] = Phi not PD
//   mid : factorisation + forward sweep, backward sweep             returns "Y not PD"
//   post: d_z with the new iterate for t = 1, line-search dots, nu += t d_nu (the caller passes t back in a second call
//         only in the rare case t != 1: fw_phase_zfix)
// One non-inlined function per Newton ITERATION (fw_iteration) instead of one per phase group: a function that uses all 256
// registers saves and restores the ~112 callee-saved ones on every call (57 KB of scratch per call and problem, HBM traffic at
// 2048 resident problems): 2.84 -> 2.62 GB per launch, same time (-DFW_THREE_CALLS: the three-call variant, for A/B runs).
#ifndef FW_THREE_CALLS
#define FW_ONE_CALL
#endif
#ifdef FW_ONE_CALL
#define FW_PH FW_IN
#else
#define FW_PH FW_FN
#endif
template <int N>
FW_PH void fw_phase_pre(FwKP Pin, int p, double* lds_g, double* red_g, int first, int do_init) {
    do_init = __builtin_amdgcn_readfirstlane(do_init);
    if (do_init >= 0) fw_phase_init<N>(Pin, p, do_init);
#ifdef FW_TIMING
    const unsigned long long tp0 = __builtin_readcyclecounter();
#endif
    fw_phase_CT<N, 0>(Pin, p, lds_g, red_g, first);
#ifdef FW_TIMING
    const unsigned long long tp1 = __builtin_readcyclecounter();
#endif
    fw_phase_C2<N>(Pin, p, lds_g, red_g + 1, first);
#ifdef FW_TIMING
    if ((threadIdx.x & 63) == 0) { atomicAdd(&fw_timing[13], tp1 - tp0); atomicAdd(&fw_timing[14], __builtin_readcyclecounter() - tp1); }
#endif
}
template <int N, bool EX>
FW_PH int fw_phase_mid(FwKP Pin, int p, double* lds_g, int first) {
    const int npd = fw_phase_factor<N, EX>(Pin, p, lds_g, first);
    const FwKP P = fw_uniform(Pin);
    if (P->mode == FW_MODE_EXPORT && (threadIdx.x & 63) == 0) *P->sh_ok = npd ? 0 : 1;
    if (npd) return 1;
#ifdef FW_TIMING
    const unsigned long long tb0 = __builtin_readcyclecounter();
#endif
    if (EX) fw_phase_backward<N>(Pin, p, lds_g); else fw_phase_backward_dump<N>(Pin, p, lds_g);
#ifdef FW_TIMING
    if ((threadIdx.x & 63) == 0) atomicAdd(&fw_timing[12], __builtin_readcyclecounter() - tb0);
#endif
    return 0;
}
template <int N> FW_FN void fw_phase_zfix(FwKP Pin, int p, double t);
// out: red[0] = accepted step length t, red[1] = 1 if the search collapsed (FMPC_W_LINESEARCH)
template <int N>
FW_PH void fw_phase_post(FwKP Pin, int p, double* lds_g, double* red_g, int first, double rho2) {
    fw_phase_CT<N, 1>(Pin, p, lds_g, red_g, first);
    const fw_lds_t red = (fw_lds_t)red_g;
    fw_wave_fence();
    const double beta_e = red[0], eps2 = red[1];
    fw_wave_fence();
    double t = 1.0, collapsed = 0.0;
    {
        const double al = 1e-4;
        int halv = 0;
        while (true) {      // closed form of backtracking_inf_newton.m:2-11 (frozen d)
            const double gq = (t - 2.0 + 2.0 * al - al * al * t) * rho2 - 2.0 * (1.0 - t) * beta_e + t * eps2;
            if (gq <= 0.0) break;
            t *= 0.5;
            if (++halv >= FW_MAX_HALVINGS) { t = 0.0; collapsed = 1.0; break; }
        }
    }
    if (t != 1.0) fw_phase_zfix<N>(Pin, p, t);
    fw_phase_nu_update<N>(Pin, p, t);
    if ((threadIdx.x & 63) == 0) { red[0] = t; red[1] = collapsed; }
    fw_wave_fence();
}
#ifdef FW_ONE_CALL
// One Newton iteration as ONE non-inlined function (one save / restore of the callee-saved registers per iteration instead of
// three).  Returns 0 = stepped (red[0] = t, red[1] = collapsed), 1 = converged before the step, 2 / 3 = Phi / Schur complement
// not positive definite.
#ifdef FW_INLINE_ITER                             // A/B switch: the iteration inlined into the kernel spills 1148 bytes per lane: 8 % slower
#define FW_ITER FW_IN
#else
#define FW_ITER FW_FN
#endif
// test_only: evaluate the exit test of this iteration and return 4 instead of stepping (pphase 4: the steps behind the first
// one are taken by the continuation launch over the compacted list).
template <int N, bool EX>
FW_ITER int fw_iteration(FwKP Pin, int p, double* lds_g, double* red_g, int first, int do_init, int test_only) {
    const FwKP P = fw_uniform(Pin);
    test_only = __builtin_amdgcn_readfirstlane(test_only);
    fw_phase_pre<N>(Pin, p, lds_g, red_g, first, do_init);
    const fw_lds_t red = (fw_lds_t)red_g;
    fw_wave_fence();
    const double rd2 = red[0], rp2 = red[1];
    const bool bad = red[2] != 0.0;
    fw_wave_fence();
    const double rho2 = rd2 + rp2;
    if (P->mode != FW_MODE_EXPORT && sqrt(rho2) <= 1e-6 && sqrt(rp2) <= 1e-8) return 1;       // inf_newton_solver.m:19-22
    if (test_only) return 4;
    if (bad) return 2;
    if (fw_phase_mid<N, EX>(Pin, p, lds_g, first)) return 3;
    fw_phase_post<N>(Pin, p, lds_g, red_g, first, rho2);
    return 0;
}
#endif
// the start point / b / nu of a problem outside the merged functions (cold-start path)
template <int N>
FW_FN void fw_phase_init_fn(FwKP Pin, int p, int write_z) { fw_phase_init<N>(Pin, p, write_z); }

// Panel path: step-length / exit decision of problem p from what the two panel kernels left behind.
// The line search (backtracking_inf_newton.m:2-11) accepts t = 1 iff ||e||^2 <= (1-alpha)^2 rho^2
// and the exit test (inf_newton_solver.m:19-22) needs rho and ||r_p||.  Both are decided here only with a wide
// margin (||e||^2 <= rho_lb^2 / 2 with rho_lb <= rho; ||r_p|| or rho_lb a factor 2 above the exit thresholds);
// returns false for every other problem, which the caller then solves exactly, overwriting the panel result.
// Wave-uniform; fixed summation order.
__device__ __forceinline__ bool fw_panel_decide(FwKP P, int p, bool write) {
    const int lane = threadIdx.x & 63, T = P->M.T;
    const double* e0 = P->epsp + ((size_t)(p >> 4) * T) * 16 + (p & 15);
    double e = 0.0;
    for (int j = lane; j < T; j += 64) e += e0[(size_t)j * 16];
    const double e2 = fw_wave_sum(e);
    const double rp2 = P->gate[2 * p], rho2 = P->gate[2 * p + 1];
    const bool fin = rp2 < 1e300 && rho2 < 1e300 && e2 < 1e300;       // false for NaN
    const bool clear = fin && (rp2 > 4e-16 || rho2 > 4e-12) && e2 <= 0.5 * rho2;
    if (clear && write && lane == 0) {
        if (P->status) P->status[p] = FMPC_OK;
        if (P->iters) P->iters[p] = 1;
        if (P->step) for (int q = 0; q < P->step_ld; ++q) P->step[(size_t)p * P->step_ld + q] = q == 0 ? 1.0 : -1.0;
    }
    if (clear && write && P->u0out && !P->u0_done) {     // z is what the d_z kernel wrote (an earlier launch)
        const int m = P->M.m;
        const double* zp = P->zout + (size_t)p * P->zld;
        for (int idx = lane; idx < m; idx += 64) P->u0out[(size_t)p * m + idx] = zp[idx];
    }
    return clear;
}

// Panel path, budgets > 1: is the exit test of the NEXT iteration (inf_newton_solver.m:19-22: ||r|| <= 1e-6 and
// ||r_p|| <= 1e-8 at the new point) clearly met?  After a full step r_p and the x entries of r_d vanish up to
// rounding (measured: <= 1e-19 against the threshold of 1e-12 on the squared norm); the u entries of r_d were summed by
// fmpc_cold_dz<true> with the same formula as the exact path, so the two agree to rounding as well.  "Clearly met" =
// 0.4 % below the threshold on the squared norm; everything else is evaluated exactly by the continuation.
// Wave-uniform; fixed summation order.
__device__ __forceinline__ bool fw_panel_converged(FwKP P, int p) {
    const int lane = threadIdx.x & 63, T = P->M.T;
    const double* r0 = P->rnp + ((size_t)(p >> 4) * T) * 16 + (p & 15);
    double a = 0.0;
    for (int j = lane; j < T; j += 64) a += r0[(size_t)j * 16];
    const double rn2 = fw_wave_sum(a);
    return rn2 <= 0.996e-12;                         // false for NaN
}

// EX: the launch that computes and exports the handle's shared cold-start factor (batch 1, once per barrier weight) -- an instance
// of its own, so that the production instance carries none of its branches and streams its factor as tile dumps.
template <int N, bool EX>
__global__ void __launch_bounds__(FW_THREADS, 2) fmpc_newton_wave(FwParams Pv) {
    using C = FwCfg<N>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const FwKP P = fw_params();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int mp = P->V.mp;
    const int batch = P->batch;
    const bool panel_mode = P->gate != nullptr;
    const int wave_g = blockIdx.x * FW_WAVES + wv, nwaves = gridDim.x * FW_WAVES;
    // pphase 3 ("flag mode", behind fmpc_first_move): list[p] != 0 marks the problems to solve exactly, from scratch
    const int pphase = (panel_mode || P->pphase == 3) ? P->pphase : 0;
    const int nlist = pphase == 2 ? P->handed[1] : 0;          // written by the pphase-1 launch
    const int rounds = pphase == 2 ? (nlist + nwaves - 1) / nwaves : (batch + nwaves - 1) / nwaves;
    if (panel_mode && pphase != 2) {
        // decisions of all problems of this workgroup; leave if none of them needs the exact path
        int need = 0;
        for (int rnd = 0; rnd < rounds; ++rnd) {
            const int q = wave_g + rnd * nwaves;
            if (q < batch) {
                int entry = -1;
                if (!fw_panel_decide(P, q, true)) { entry = q | FW_LIST_HANDED; if (lane == 0 && P->handed) atomicAdd(P->handed, 1); }
                else if (P->max_iter > 1 && !fw_panel_converged(P, q)) entry = q;   // accepted first step, more iterations follow
                if (entry >= 0) {
                    need = 1;
                    if (pphase == 1 && lane == 0) P->list[atomicAdd(P->handed + 1, 1)] = entry;
                }
            }
        }
        if (pphase == 1) return;
        if (!__syncthreads_or(need)) return;
    }
    if (pphase == 2 && nlist == 0) return;
    if (pphase == 3 && P->nflag) {                   // behind the affine kernel: has its running count of flagged problems moved?
        const int cnt = P->nflag[0], seen = P->nflag[1];
        if (cnt == seen) return;
        // (rare) the workgroup that draws the last ticket has seen every workgroup read both counts: it brings `seen` up to date
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(P->nflag + 2, 1) == (int)gridDim.x - 1) { P->nflag[1] = cnt; P->nflag[2] = 0; }
    }
    if (pphase == 3) {                               // nothing flagged among this workgroup's problems (the usual case): leave
        int any = 0;
        for (int rnd = 0; rnd < rounds; ++rnd) { const int q = wave_g + rnd * nwaves; if (q < batch && P->list[q] != 0) any = 1; }
        if (!__syncthreads_or(any)) return;
    }
    for (int i = threadIdx.x; i < mp * FW_LDB; i += FW_THREADS) lds[i] = P->V.BtP[i];
    {   // the block Y_{i,i+2} of the first stage (= of nearly all stages) for the factorisations' column loads
        double* sY2 = lds + (size_t)mp * FW_LDB + (size_t)FW_WAVES * C::PER_WAVE + FW_WAVES * 4;
        const double* src = P->V.img + (size_t)P->V.i2[0] * C::IMG_STRIDE + C::IMG_D + C::IMG_1;
        for (int i = threadIdx.x; i < N * N; i += FW_THREADS) { const int j = i / N, c = i - j * N; sY2[i] = src[j * 32 + c]; }
    }
    {   // this wave's tiles: finite everywhere (pad rows/columns are read by the layout changes)
        double* t = lds + (size_t)mp * FW_LDB + (size_t)wv * C::PER_WAVE;
        for (int i = lane; i < C::PER_WAVE; i += 64) t[i] = 0.0;
    }
    __syncthreads();                       // the only workgroup barrier: waves are independent below
    // scalars handed back by the phases: a 4-double LDS slot per wave behind the tiles
    double* red = lds + (size_t)mp * FW_LDB + (size_t)FW_WAVES * C::PER_WAVE + wv * 4;

    const int max_iter = P->max_iter;
#ifdef FW_TIMING
    unsigned long long _k0 = __builtin_readcyclecounter(), _k1, _ka[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define FW_KTICK(k) do { _k1 = __builtin_readcyclecounter(); _ka[k] += _k1 - _k0; _k0 = _k1; } while (0)
#else
#define FW_KTICK(k)
#endif
    // cold: first Newton step from the mid-box start with the handle's shared factor and constants.
    // All waves of a workgroup run the same number of rounds (the shared sweeps are collective).
    const bool cold_mode = P->mode == FW_MODE_SHARED && *P->sh_ok != 0;
    for (int rnd = 0; rnd < rounds; ++rnd) {
        // panel mode: `accepted` = the panel kernels' first Newton step stands (z, nu+ are in place)
        int p = wave_g + rnd * nwaves;
        bool accepted, active;
        if (pphase == 2) {                               // the compacted list of the problems that go on
            const int entry = p < nlist ? P->list[p] : -1;
            accepted = entry >= 0 && !(entry & FW_LIST_HANDED);
            active = entry >= 0;
            p = entry >= 0 ? (entry & (FW_LIST_HANDED - 1)) : batch;
        } else if (pphase == 3) {
            accepted = false;
            active = p < batch && P->list[p] != 0;
            if (active && lane == 0 && P->handed) atomicAdd(P->handed, 1);
        } else {
            accepted = p < batch && panel_mode && fw_panel_decide(P, p, false);
            active = p < batch && !(accepted && (max_iter <= 1 || fw_panel_converged(P, p)));
        }
        if (cold_mode) {
            // [cu | hc | wc | ubar] into LDS for the cold step's epilogue.  The region overlaps the per-wave
            // tiles of the general path, so wait until every wave has left the previous round.
            __syncthreads();
            const fw_lds_t sC = fw_cold_consts<N>(lds, mp);
            for (int i = threadIdx.x; i < 4 * mp; i += FW_THREADS)
                sC[i] = i < 3 * mp ? P->cold[i] : (i - 3 * mp < P->M.m ? P->M.umid[i - 3 * mp] : 0.0);
            __syncthreads();
        }
        FW_KTICK(7);
        int st = FMPC_OK, nsteps = 0, it0 = 0;
        bool done = !active;
        if (cold_mode) {
            double rho2 = 0.0;
            int go = 0;
            if (active && !accepted) {
                fw_phase_init_fn<N>(P, p, 0);
                FW_KTICK(0);
                fw_cold_resid<N>(P, p, lds, red);
                fw_wave_fence();
                const double rd2 = red[0], rp2 = red[1];
                fw_wave_fence();
                rho2 = rd2 + rp2;
                FW_KTICK(1);
                if (sqrt(rho2) <= 1e-6 && sqrt(rp2) <= 1e-8) done = true;     // inf_newton_solver.m:19-22
                else { go = 1; fw_cold_rhs<N>(P, p, lds); }
                FW_KTICK(2);
            }
            fw_phase_sweep_shared<N, 0>(P, p, go, lds);
            FW_KTICK(3);
            fw_phase_sweep_shared<N, 1>(P, p, go, lds);
            FW_KTICK(4);
            if (go) {
                const double al = 1e-4;
                double t = 1.0;
                // cheap dots (27-dimensional quadratic forms); accept t = 1 only with a wide margin
                fw_cold_dots<N>(P, p, lds, red);
                fw_wave_fence();
                const double eps2q = red[1];
                fw_wave_fence();
                const bool clear = eps2q >= 0.0 && (-1.0 + 2.0 * al - al * al) * rho2 + eps2q <= -0.5 * rho2;
                if (clear) {
                    fw_cold_step<N>(P, p, lds, red, 2, 1.0);           // one product B'(nu + d_nu), z written once
                } else {
                    fw_cold_step<N>(P, p, lds, red, 0, 0.0);           // element-wise dots, z written for t = 1
                    fw_wave_fence();
                    const double beta_e = red[0], eps2 = red[1];
                    fw_wave_fence();
                    int halv = 0;
                    while (true) {      // closed form of backtracking_inf_newton.m:2-11 (frozen d)
                        const double gq = (t - 2.0 + 2.0 * al - al * al * t) * rho2
                                          - 2.0 * (1.0 - t) * beta_e + t * eps2;
                        if (gq <= 0.0) break;
                        t *= 0.5;
                        if (++halv >= FW_MAX_HALVINGS) { t = 0.0; st = FMPC_W_LINESEARCH; break; }
                    }
                    if (t != 1.0) fw_cold_step<N>(P, p, lds, red, 1, t);
                }
                fw_cold_nu_update<N>(P, p, lds, t);
                if (P->step && lane == 0 && P->step_ld > 0) P->step[(size_t)p * P->step_ld] = t;
                nsteps = 1;
                FW_KTICK(5);
            }
            if (active && accepted) {
                // continue after the panel kernels' step: b, the step record, and nu+ from the panel workspace
                fw_phase_init_fn<N>(P, p, 0);
                const FwView<N> W(P, p);
                const int nbn = W.nb * N;
                const double* src = P->nuws + ((size_t)(p >> 4) * nbn) * 16 + (p & 15);
                for (int idx = lane; idx < nbn; idx += 64) W.nu[idx] = src[(size_t)idx * 16];
                if (P->step && lane == 0 && P->step_ld > 0) P->step[(size_t)p * P->step_ld] = 1.0;
                fw_mem_fence();
                nsteps = 1;
            }
            it0 = 1;
        }
        int do_init = (!cold_mode && active) ? (P->zinit ? 0 : 1) : -1;       // an explicit start point is read where it lies (FwView::zs)
        for (int it = it0; it < max_iter && !done; ++it) {
            const int first = (P->zinit != nullptr && nsteps == 0) ? 1 : 0;     // z_out not written yet: read z_init
#ifdef FW_ONE_CALL
            // pphase 4 (explicit-start batches with a budget > 1): this launch takes the FIRST step of every problem and, for the
            // next one, only the exit test (inf_newton_solver.m:19-22); a problem that goes on is appended to the list the
            // continuation launch works through -- a few per cent of the batch would otherwise hold the whole launch for a
            // second iteration (with one problem per wavefront slot the launch lasts as long as its slowest wavefront)
            const int code = fw_iteration<N, EX>(P, p, lds, red, first, do_init, (P->pphase == 4 && nsteps >= 1) ? 1 : 0);
            do_init = -1;
            if (code == 1) break;
            if (code == 4) { if (lane == 0) P->list[atomicAdd(P->handed + 1, 1)] = p | FW_LIST_GENERAL; break; }
            if (code == 2) { st = FMPC_E_NOT_PD_PHI; break; }
            if (code == 3) { st = FMPC_E_NOT_PD_SCHUR; break; }
#else
            fw_phase_pre<N>(P, p, lds, red, first, do_init);               // [start,] r_d, r_p, right-hand side
            do_init = -1;
            fw_wave_fence();
            const double rd2 = red[0], rp2 = red[1];
            const bool bad = red[2] != 0.0;
            fw_wave_fence();
            FW_KTICK(1);
            const double rho2 = rd2 + rp2;
            if (P->mode != FW_MODE_EXPORT && sqrt(rho2) <= 1e-6 && sqrt(rp2) <= 1e-8) break;   // inf_newton_solver.m:19-22
            if (P->pphase == 4 && nsteps >= 1) { if (lane == 0) P->list[atomicAdd(P->handed + 1, 1)] = p | FW_LIST_GENERAL; break; }
            if (bad) { st = FMPC_E_NOT_PD_PHI; break; }
            FW_KTICK(2);
            const int npd = fw_phase_mid<N, EX>(P, p, lds, first);
            if (npd) { st = FMPC_E_NOT_PD_SCHUR; break; }
            FW_KTICK(4);
            fw_phase_post<N>(P, p, lds, red, first, rho2);  // d_z and z + d_z, line search, nu += t d_nu
#endif
            fw_wave_fence();
            const double t = red[0];
            if (red[1] != 0.0) st = FMPC_W_LINESEARCH;
            fw_wave_fence();
            if (P->step && lane == 0 && it < P->step_ld) P->step[(size_t)p * P->step_ld + it] = t;
            ++nsteps;
            FW_KTICK(5);
        }
        if (!active) continue;
        if (cold_mode && nsteps == 0) fw_phase_init_fn<N>(P, p, 2);       // left before stepping: z is the start point
        if (!cold_mode && nsteps == 0 && P->zinit) fw_phase_zcopy<N>(P, p);
        FW_KTICK(6);
        if (P->nuout) {
            const FwView<N> W(P, p);
            const int nbn = W.nb * N;
            for (int idx = lane; idx < nbn; idx += 64) P->nuout[(size_t)p * nbn + idx] = W.nu[idx];
        }
        if (lane == 0) {
            if (P->status) P->status[p] = st;
            if (P->iters) P->iters[p] = nsteps;
        }
        if (P->u0out) {                              // z of this problem was written by this wave's own lanes
            fw_mem_fence();
            const int m = P->M.m;
            const double* zp = P->zout + (size_t)p * P->zld;
            for (int idx = lane; idx < m; idx += 64) P->u0out[(size_t)p * m + idx] = zp[idx];
        }
    }
#ifdef FW_TIMING
    if (lane == 0) for (int q = 0; q < 8; ++q) atomicAdd(&fw_timing[4 + q], _ka[q]);
#endif
}

// ---------------------------------------------------------------- host side of the wave kernel
size_t fmpc_wave_lds_bytes(int n, int mp) {
    if (n != 27) return 0;
    return ((size_t)mp * FW_LDB + (size_t)FW_WAVES * FwCfg<27>::PER_WAVE + FW_WAVES * 4 + FwCfg<27>::Y2LDS) * sizeof(double);
}
bool fmpc_wave_supports(int n) { return n == 27; }
int fmpc_wave_mp(int m) { const int q = 4 * FW_KCH; return (m + q - 1) / q * q; }
int fmpc_wave_img_stride(int n) { return n == 27 ? FwCfg<27>::IMG_STRIDE : 0; }
int fmpc_wave_waves_per_wg() { return FW_WAVES; }
size_t fmpc_wave_ws_doubles(int n, int m, int mp, int T, int nb) {
    return fw_ws_layout(n, m, mp, T, nb, FwCfg<27>::LDG, FwCfg<27>::FST).total;
}
size_t fmpc_wave_shared_fac_doubles(int n, int nb) { return (size_t)nb * 6 * n * FwCfg<27>::LDG; }
void fmpc_wave_cold_layout(int n, int mp, int* off9) {      // 14 entries
    const FwCold c = fw_cold_layout(n, mp);
    off9[0] = c.cu; off9[1] = c.hc; off9[2] = c.wc; off9[3] = c.G; off9[4] = c.cbu; off9[5] = c.cp0; off9[6] = c.cp1;
    off9[7] = c.cp2; off9[8] = c.total; off9[9] = c.Ma; off9[10] = c.Ma2; off9[11] = c.va; off9[12] = c.va2; off9[13] = c.sa;
}

// Fill the three images of one n x n row-major block (see FwCfg): host helper.
void fmpc_wave_make_images(int n, const double* blk, double* out) {
    using C = FwCfg<27>;
    for (int i = 0; i < C::IMG_STRIDE; ++i) out[i] = 0.0;
    auto at = [&](int r, int c) { return (r < n && c < n) ? blk[r * n + c] : 0.0; };
    const int subs[3][2] = {{0, 0}, {0, 1}, {1, 1}};
    for (int sidx = 0; sidx < 3; ++sidx)
        for (int r = 0; r < 4; ++r)
            for (int l = 0; l < 64; ++l)
                out[(sidx * 4 + r) * 64 + l] = at(16 * subs[sidx][0] + 4 * r + (l >> 4), 16 * subs[sidx][1] + (l & 15));
    double* o1 = out + C::IMG_D;
    for (int I = 0; I < 2; ++I)
        for (int J = 0; J < 2; ++J)
            for (int r = 0; r < 4; ++r)
                for (int l = 0; l < 64; ++l)
                    o1[((I * 2 + J) * 4 + r) * 64 + l] = at(16 * I + 4 * r + (l >> 4), 16 * J + (l & 15));
    double* o2 = o1 + C::IMG_1;
    for (int j = 0; j < n; ++j)
        for (int c = 0; c < 32; ++c) o2[j * 32 + c] = at(j, c);
    // paired copies of the subtile images: [subtile][r / 2][lane][r % 2]
    double* op = out + C::IMG_P;
    for (int sub = 0; sub < 3 + 4; ++sub)
        for (int r = 0; r < 4; ++r)
            for (int l = 0; l < 64; ++l)
                op[((sub * 2 + r / 2) * 64 + l) * 2 + (r & 1)] = out[(sub * 4 + r) * 64 + l];     // (IMG_D and IMG_1 are contiguous: 7 subtiles)
}

hipError_t fmpc_wave_prepare(int n, size_t lds_bytes) {
    if (n != 27) return hipErrorInvalidValue;
    const hipError_t e = hipFuncSetAttribute((const void*)fmpc_newton_wave<27, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)fmpc_newton_wave<27, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

hipError_t fmpc_launch_wave(const FmpcDevModel& M, const FwModel& V, int batch, int grid,
                            const double* x0, const double* x0p, const double* w, const double* zinit,
                            const double* nu0, int max_iter, double kbar, double* zout, double* nuout,
                            int* status, int* iters, double* step, int step_ld, double* ws,
                            size_t ws_stride, size_t lds_bytes, hipStream_t stream,
                            int mode, double* sh_fac, double* sh_rs, int* sh_ok, const double* cold,
                            const double* gate, const double* epsp, int* handed, const double* nuws, double* u0out,
                            int pphase, const double* rnp, int* list, int u0_done, int* nflag, int zld) {
    if (M.n != 27) return hipErrorInvalidValue;
    FwParams P;
    P.zld = zld > 0 ? zld : M.T * (M.n + M.m);
    P.gate = gate; P.epsp = epsp; P.handed = handed; P.nuws = nuws; P.u0out = u0out;
    P.pphase = pphase; P.rnp = rnp; P.list = list; P.u0_done = u0_done; P.nflag = nflag;
    static const int env_flags = [] { const char* e = getenv("FMPC_WAVE_FLAGS"); return e && e[0] ? atoi(e) : 0; }();
    P.flags = env_flags;
    if (pphase == 1) lds_bytes = 0;                  // the decide-only launch touches no LDS: cheap to place
    P.M = M; P.V = V; P.batch = batch; P.max_iter = max_iter; P.step_ld = step_ld; P.mode = mode; P.sh_fac = sh_fac; P.sh_rs = sh_rs; P.sh_ok = sh_ok; P.cold = cold;
    P.kbar = kbar; P.x0 = x0; P.x0p = x0p; P.w = w; P.zinit = zinit; P.nu0 = nu0; P.zout = zout;
    P.nuout = nuout; P.status = status; P.iters = iters; P.step = step; P.ws = ws; P.ws_stride = ws_stride;
    if (mode == FW_MODE_EXPORT) hipLaunchKernelGGL((fmpc_newton_wave<27, true>), dim3(grid), dim3(FW_THREADS), lds_bytes, stream, P);
    else hipLaunchKernelGGL((fmpc_newton_wave<27, false>), dim3(grid), dim3(FW_THREADS), lds_bytes, stream, P);
    return hipGetLastError();
}
