// CDNA4 fastMPC, cold-start step WITHOUT w as one matrix product: the affine form of the whole step (n = 27).
//
// From the cold start (fast_mpc_init.m: z0 = box centres, README.md:538-540) and with a Newton budget of 1 -- the
// reference's own call, Fast_MPC2(..., x_init = []).mpc_fixed_log_newton(1, k), README.md:548-556 -- the factor depends on
// k only and the step is an AFFINE map of the data d = [x0 ; x0_pre] (w = [] in the reference's replay):
//     nu+ = nuc + J d                      (fmpc_kernel_inv.hip: the dense form of the dual solve)
//     z+  = z0 + d_z = zc + Kz d           (d_z = -Phi^-1 (r_d + C' nu+), inf_newton_solver.m:34-35, is linear in nu+)
// so a batch of B problems is ONE product  Z (T(n+m) x B) = [Kz | zc] (T(n+m) x 56) * [D ; 1] (56 x B)  on the fp64 matrix
// cores, written straight into the caller's z: 14 k-steps per 16 x 16 tile of z, i.e. 7 matrix instructions per KB of
// output -- at 2000 problems 0.56 M instructions, 17 us of the chip's matrix pipes, for 82 MB of z.  The three kernels this
// replaces (dual solve 14 us, d_z 25 us, decision 5 us) spent most of their time on launch boundaries and on staging nu+.
// Kz is built once per (handle, k) on the host in long double from the same shared factor as J (fmpc_host_build_affine).
//
// The step-length / exit decision (backtracking_inf_newton.m:2-11, inf_newton_solver.m:19-22) needs
// ||e||^2 and ||r_p||^2: quadratic forms of d (fmpc_kernel_first.hip has the same forms for the closed loop), evaluated
// here on the matrix cores as well, 16 problems per wavefront, with the rounding guard of the first-move kernel: t = 1 is
// accepted only with a wide margin, anything else is flagged in `need` and redone EXACTLY by the launch that follows
// (fmpc_newton_wave in flag mode, which returns at once when no flag is set and overwrites z of the flagged problems).
//
// Work split: a task = (group of 4 column tiles = 64 problems, one 16-row tile of z) = 56 matrix instructions; the tasks are
// dealt in contiguous ranges to 8 wavefronts per CU (two per SIMD), which keep their 64 problems' data (the B operand, 112
// registers) across the tasks of a group; the A operand (a 512-byte image per k-step, L2-resident: 2.3 MB in all) is
// requested one task ahead.  First moves only (z_out = NULL): the first m rows alone.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include "fmpc_device.h"
#include "fmpc_affine.h"
#include "../../include/fastmpc.h"

#ifdef FW_TIMING
// diagnostic build: per workgroup (wavefront 0) time stamps of the constant 100 MHz clock: start, data staged, first tile done, end
__device__ unsigned long long fa_trace[1024 * 8];
extern "C" int fmpc_debug_affine_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fa_trace), sizeof(unsigned long long) * 1024 * 8) == hipSuccess ? 0 : -1;
}
#define FA_TICK(k) do { if (tid == 0 && blockIdx.x < 1024) fa_trace[blockIdx.x * 8 + (k)] = (unsigned long long)wall_clock64(); } while (0)
#else
#define FA_TICK(k)
#endif
typedef double d4a __attribute__((ext_vector_type(4)));
typedef double d2a __attribute__((ext_vector_type(2)));
#define FA_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define FA_THREADS 256
#define FA_CT 4                          // column tiles (16 problems each) per task

// The data operand of a tile of 16 problems: lane (g = lane / 16, c = lane % 16) holds d'[p0 + c][4 q + g], d' = [x0 ; x0_pre ; 1 ; 0]
__device__ __forceinline__ void fa_load_d(double (&D)[FA_KS], const FaParams& P, int p0, int g, int c) {
    const int n = P.n;
    const int p = p0 + c < P.batch ? p0 + c : P.batch - 1;
    const double* x0 = P.x0 + (size_t)p * n;
    const double* xp = P.x0p ? P.x0p + (size_t)p * n : x0;
    // one load per entry from an address chosen BEFORE the load, all requested up front; the selects follow (a select right
    // behind its load makes the compiler wait for each load in turn: one memory round trip per entry)
    double t[FA_KS];
#pragma unroll
    for (int q = 0; q < FA_KS; ++q) {
        const int k = 4 * q + g;
        const double* src = k < n ? x0 + k : xp + (k < 2 * n ? k - n : 0);
        t[q] = *src;
    }
#pragma unroll
    for (int q = 0; q < FA_KS; ++q) {
        const int k = 4 * q + g;
        const bool on = k < n || (k < 2 * n && P.x0p != nullptr);
        D[q] = on ? t[q] : (k == 2 * n ? 1.0 : 0.0);
    }
}

__device__ __forceinline__ void fa_load_a(double (&A)[FA_KS], const double* img, int tile, int lane) {
    const double* ip = img + (size_t)tile * FA_KS * 64 + lane;
#pragma unroll
    for (int q = 0; q < FA_KS; ++q) A[q] = ip[q * 64];
}

// The product's operand prefetch, by hand.  gfx950 counts loads and stores in ONE in-order counter (vmcnt), and the compiler's
// wait insertion assumes at a loop head that nothing was issued behind a load of the previous iteration: it waits until the
// 16 stores of the previous tile have been written, every tile (measured: 40 us per 2000 problems, the matrix pipes idle).
// Loads the compiler does not see + waits with the count we know (exactly the 16 stores of a tile follow the request):
__device__ __forceinline__ void fa_request_a(double (&A)[FA_KS], const double* img, int tile, int lane) {
    const double* ip = img + (size_t)tile * FA_KS * 64 + lane;
    const double* ip2 = ip + 8 * 64;
#define FA_LD(q, base, off) asm volatile("global_load_dwordx2 %0, %1, off offset:" #off : "=v"(A[q]) : "v"(base))
    FA_LD(0, ip, 0); FA_LD(1, ip, 512); FA_LD(2, ip, 1024); FA_LD(3, ip, 1536); FA_LD(4, ip, 2048); FA_LD(5, ip, 2560); FA_LD(6, ip, 3072); FA_LD(7, ip, 3584);
    FA_LD(8, ip2, 0); FA_LD(9, ip2, 512); FA_LD(10, ip2, 1024); FA_LD(11, ip2, 1536); FA_LD(12, ip2, 2048); FA_LD(13, ip2, 2560);
#undef FA_LD
}
// all FA_KS values requested by fa_request_a have arrived once at most `BEHIND` later memory operations are outstanding
template <int BEHIND>
__device__ __forceinline__ void fa_await_a(double (&A)[FA_KS]) {
    asm volatile("s_waitcnt vmcnt(%14)"
                 : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]), "+v"(A[8]), "+v"(A[9]),
                   "+v"(A[10]), "+v"(A[11]), "+v"(A[12]), "+v"(A[13])
                 : "n"(BEHIND));
}

__device__ __forceinline__ bool fa_decide(const FaParams& P, double qe, double qp, double rdl, double dn2) {
    const double dn = sqrt(dn2);
    double e2 = qe + P.e0, rp2 = qp + P.ep0;
    // rounding of the forms: |error| <= c eps (|d|^2 |M|_F + 2 |v| |d| + |const|), c generous (fmpc_kernel_first.hip)
    const double ce = 4096.0 * 2.220446049250313e-16;
    const double de = ce * (dn2 * P.normE + 2.0 * P.norme * dn + fabs(P.e0));
    const double dp = ce * (dn2 * P.normEp + 2.0 * P.normep * dn + fabs(P.ep0));
    e2 += de;                                                     // upper bound of ||e||^2
    rp2 = rp2 - dp > 0.0 ? rp2 - dp : 0.0;                        // lower bound of ||r_p||^2
    const double rho2 = rp2 + rdl;                                // lower bound of rho^2
    const bool fin = rp2 < 1e300 && rho2 < 1e300 && e2 < 1e300 && e2 >= 0.0;
    return fin && (rp2 > 4e-16 || rho2 > 4e-12) && e2 <= 0.5 * rho2;
}

// Operand roles: the PROBLEMS are the rows of the matrix instruction (A operand: lane (g, i) holds d'[p0 + i][4 q + g]), the rows
// of z its columns (B operand: lane (g, j) holds Kz[16 t + j][4 q + g], a 512-byte image per k-step).  Result register r of
// lane (g, j) is then z[p0 + 4 r + g][16 t + j]: the 16 lanes of a row group hold 16 CONSECUTIVE entries of one problem's z --
// a store instruction writes four full 128-byte runs.
//
// A workgroup = one group of 64 problems x a share of the tiles of z, dealt round-robin to its four wavefronts.  The group's
// data come into LDS in one coalesced pass (64 x 27 consecutive doubles of x0, of x0_pre) and go from there into the
// wavefronts' operand registers (gathered straight from memory they were four dependent batches of scattered loads).  The
// last four workgroups of a group first evaluate the decision forms of its four column tiles (one each; wavefront t the rows
// 16 t .. 16 t + 15 of E and Ep, 28 matrix instructions), from the same LDS copy of the data.
// Measured (timing build, scripts/affine_trace.py): a tile of 56 matrix instructions takes 2.05 us of a SIMD's matrix pipe
// (64 cycles each at 1.75 GHz), two wavefronts per SIMD keep it busy; the kernel is bound by that pipe.
// NT: the z stores of the full rounds bypass the L2 (non-temporal).  Only with the z rows of consecutive problems a multiple of
// 128 bytes apart from a 128-byte aligned base (fmpc_set_z_ld): every 128-byte run of a tile is then exactly one cache line
// and nothing is left for the L2 to merge or to write back when the kernel ends (34.6 against 39.1 us per 2000-problem step,
// same box; with rows that straddle lines the same stores take 60 us).
template <bool ZOUT, bool NT = false>
__global__ void __launch_bounds__(FA_THREADS, 2) fmpc_cold_affine(FaParams P) {
    // the group's data in OPERAND order: entry (problem 16 ct + c, k = 4 q + g) at ((q FA_CT + ct) 4 + g) 16 + c, so that a
    // wavefront's read of an operand register is 64 consecutive doubles (row-major [problem][k] with an odd stride had 2-4
    // lanes per bank: eight wavefronts x 56 reads took 4.4 us per workgroup, measured)
    __shared__ double sD[FA_KS * FA_CT * 64];
#define FA_SD(ct, cc, k) sD[((((k) >> 2) * FA_CT + (ct)) * 4 + ((k) & 3)) * 16 + (cc)]
    __shared__ double sF[4][3][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, c = lane & 15;
    const int n = P.n;
    const int wg = (int)blockIdx.x;
    const int gi = wg / P.wgs_per_group, slot = wg - gi * P.wgs_per_group;
    const int tiles = P.tiles_used, rows = P.rows, m = P.m;
    const int tstep = 4 * P.wgs_per_group;
    int tile = slot * 4 + wv;
    FA_TICK(0);
    if (wg == 0 && tid == 0 && P.handed) *P.handed = 0;
    double A[FA_KS], An[FA_KS];
    fa_request_a(A, P.img, tile < tiles ? tile : 0, lane);           // in flight while the data are staged
    const int p0 = gi * FA_CT * 16;
    {
        const int np = P.batch - p0 < FA_CT * 16 ? P.batch - p0 : FA_CT * 16;       // problems of this group
        const double* s0 = P.x0 + (size_t)p0 * n;
        const double* s1 = P.x0p ? P.x0p + (size_t)p0 * n : s0;
        double v0[7], v1[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int idx = tid + FA_THREADS * j, ic = idx < np * n ? idx : 0;
            v0[j] = s0[ic]; v1[j] = s1[ic];
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int idx = tid + FA_THREADS * j;
            if (idx < FA_CT * 16 * n) {
                const int pr = idx / n, k = idx - pr * n;
                const bool on = idx < np * n;
                FA_SD(pr >> 4, pr & 15, k) = on ? v0[j] : 0.0;
                FA_SD(pr >> 4, pr & 15, n + k) = (on && P.x0p) ? v1[j] : 0.0;
            }
        }
        if (tid < FA_CT * 16) { FA_SD(tid >> 4, tid & 15, 2 * n) = 1.0; FA_SD(tid >> 4, tid & 15, 2 * n + 1) = 0.0; }
    }
    __syncthreads();
    FA_TICK(1);
    // ================================================================ decision forms of the group's column tiles
    for (int fct = P.wgs_per_group - 1 - slot; fct < FA_CT; fct += P.wgs_per_group) {
        if (fct < 0 || p0 + fct * 16 >= P.batch) continue;                         // (uniform)
        const int t = wv;                                                          // rows 16 t .. 16 t + 15 of E, Ep
        double E1[FA_KS], E2[FA_KS], Df[FA_KS];
        fa_load_a(E1, P.imgE, t, lane);
        fa_load_a(E2, P.imgEp, t, lane);
        const int k = 16 * t + c;
        const double le = P.elin[k], lp = P.eplin[k];                              // 2 e and -2 ep, zero beyond 2 n (64 entries)
        // lower bound of ||r_d(nu0)||^2: its x entries of the last stage (as the gate of the panel path), lane c: entries c, c + 16
        double rdl[4] = {P.rd2_0, P.rd2_0, P.rd2_0, P.rd2_0};
        if (P.nu0 && t == 0) {
            double xa[4][2];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pr = p0 + fct * 16 + 4 * r + g < P.batch ? p0 + fct * 16 + 4 * r + g : P.batch - 1;
                const double* nu = P.nu0 + (size_t)pr * P.nb * n;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int e = c + 16 * j, ec = e < n ? e : 0;
                    xa[r][j] = P.dx0T[ec] + nu[(P.T - 1) * n + ec] + (P.has_xf ? nu[P.T * n + ec] : 0.0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double acc = xa[r][0] * xa[r][0] + (c + 16 < n ? xa[r][1] * xa[r][1] : 0.0);
                acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64); acc += __shfl_xor(acc, 8, 64);
                rdl[r] = acc;
            }
        }
#pragma unroll
        for (int q = 0; q < FA_KS; ++q) Df[q] = sD[(q * FA_CT + fct) * 64 + lane];        // (the images' column 2 n is zero: the constant 1 drops out)
        d4a ce = {0, 0, 0, 0}, cp = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < FA_KS; ++q) { ce = FA_MFMA(Df[q], E1[q], ce); cp = FA_MFMA(Df[q], E2[q], cp); }
        // register r <-> problem 4 r + g of the tile, entry k = 16 t + c of d
        double qe[4], qp[4], dn2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double dv = k < 2 * n ? FA_SD(fct, 4 * r + g, k < 2 * n ? k : 0) : 0.0;
            qe[r] = dv * (ce[r] + le); qp[r] = dv * (cp[r] + lp); dn2[r] = dv * dv;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { qe[r] += __shfl_xor(qe[r], o, 64); qp[r] += __shfl_xor(qp[r], o, 64); dn2[r] += __shfl_xor(dn2[r], o, 64); }
            if (c == 0) { sF[t][0][4 * r + g] = qe[r]; sF[t][1][4 * r + g] = qp[r]; sF[t][2][4 * r + g] = dn2[r]; }
        }
        __syncthreads();
        if (t == 0 && c == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * r + g, pp = p0 + fct * 16 + i;
                if (pp < P.batch) {
                    const double se = (sF[0][0][i] + sF[1][0][i]) + (sF[2][0][i] + sF[3][0][i]);
                    const double sp = (sF[0][1][i] + sF[1][1][i]) + (sF[2][1][i] + sF[3][1][i]);
                    const double sn = (sF[0][2][i] + sF[1][2][i]) + (sF[2][2][i] + sF[3][2][i]);
                    const bool clear = fa_decide(P, se, sp, rdl[r], sn);
                    P.need[pp] = clear ? 0 : 1;
                    if (!clear && P.nflag) atomicAdd(P.nflag, 1);
                    if (clear) {
                        if (P.status) P.status[pp] = FMPC_OK;
                        if (P.iters) P.iters[pp] = 1;
                        if (P.step) for (int q = 0; q < P.step_ld; ++q) P.step[(size_t)pp * P.step_ld + q] = q == 0 ? 1.0 : -1.0;
                    }
                }
            }
        }
        __syncthreads();
    }
    // ================================================================ z = [D ; 1]' [Kz | zc]'
    // Full rounds of tiles go round-robin to the group's wavefronts.  The workgroups that evaluated the decision forms (the
    // last four of the group: 6 us, measured as the kernel's tail) leave out their tiles of the last round; those 16 tiles and
    // the tiles beyond the full rounds (z has 320 full tiles + one of 10 rows at (27, 144, 30): one) are dealt by COLUMN TILE,
    // 14 matrix instructions apiece, to all wavefronts of the group -- a whole extra tile on one wavefront holds up its SIMD for
    // a tile's time, 10 % of the kernel.
    const int rounds = tiles / tstep;
    const bool forms_wg = P.wgs_per_group >= 8 && rounds >= 2 && slot >= P.wgs_per_group - FA_CT;
    const int nskip = (P.wgs_per_group >= 8 && rounds >= 2) ? 4 * FA_CT : 0;  // tiles of the last round the forms workgroups leave out
    const int skip0 = (rounds - 1) * tstep + (P.wgs_per_group - FA_CT) * 4;    // ... a contiguous range
    const int tfull = (rounds - (forms_wg ? 1 : 0)) * tstep;                 // this wavefront's rounds end here
    double* dump = P.dump + ((blockIdx.x & 15) * FA_THREADS + tid);          // 16 x 256 doubles: nobody reads them
    {
        const int wg_w = slot * 4 + wv;                                          // wavefront of the group
        const int nleft = nskip + (tiles - rounds * tstep);
        for (int task = wg_w; task < nleft * FA_CT; task += tstep) {
            const int j = task / FA_CT, ct = task % FA_CT;
            const int lt = j < nskip ? skip0 + j : rounds * tstep + (j - nskip);
            double Al[FA_KS], Dl[FA_KS];
            fa_load_a(Al, P.img, lt, lane);
#pragma unroll
            for (int q = 0; q < FA_KS; ++q) Dl[q] = sD[(q * FA_CT + ct) * 64 + lane];
            d4a acc = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < FA_KS; ++q) acc = FA_MFMA(Dl[q], Al[q], acc);
            const int row = 16 * lt + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pp = (gi * FA_CT + ct) * 16 + 4 * r + g;
                if (ZOUT && lt < P.tiles && row < rows && pp < P.batch) P.zout[(size_t)pp * P.ldz + row] = acc[r];
                if (ZOUT && lt >= P.tiles && row - 16 * P.tiles < P.nu_rows && pp < P.batch) P.nuout[(size_t)pp * P.nu_rows + (row - 16 * P.tiles)] = acc[r];
                if (P.u0out != nullptr && row < m && pp < P.batch) P.u0out[(size_t)pp * m + row] = acc[r];
            }
        }
    }
    if (tile >= tfull) return;
    double D[FA_CT][FA_KS];
#pragma unroll
    for (int ct = 0; ct < FA_CT; ++ct)
#pragma unroll
        for (int q = 0; q < FA_KS; ++q) D[ct][q] = sD[(q * FA_CT + ct) * 64 + lane];
    FA_TICK(4);
    fa_await_a<0>(A);
    FA_TICK(5);
    for (; tile < tfull; tile += tstep) {
        const int nxt = tile + tstep < tfull ? tile + tstep : tile;
        fa_request_a(An, P.img, nxt, lane);         // the next tile's operand is requested BEFORE this tile's 56 matrix instructions
        d4a acc[FA_CT];
#pragma unroll
        for (int ct = 0; ct < FA_CT; ++ct) acc[ct] = (d4a){0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < FA_KS; ++q)
#pragma unroll
            for (int ct = 0; ct < FA_CT; ++ct) acc[ct] = FA_MFMA(D[ct][q], A[q], acc[ct]);
        // Stores without branches: a lane that has nothing to write (a problem beyond the batch, a row beyond z) writes to a
        // dump line instead.  (A store under a condition is a branch, and behind a branch the compiler no longer knows how many
        // stores follow the request for the next operand.)  Addresses: a uniform base per (column tile, register) + one per-lane offset.
        const int row = 16 * tile + c;
        if (ZOUT) {
            // (tiles beyond those of z are rows of nu+: another base and row count, the same 16 stores)
            const bool isnu = tile >= P.tiles;
            const int rloc = isnu ? row - 16 * P.tiles : row, rcnt = isnu ? P.nu_rows : rows, ld = isnu ? P.nu_rows : P.ldz;
            double* obase = isnu ? P.nuout : P.zout;
            const unsigned voz = (unsigned)(g * ld + rloc);
            const bool rok = rloc < rcnt;
#pragma unroll
            for (int ct = 0; ct < FA_CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pb = (gi * FA_CT + ct) * 16 + 4 * r;       // uniform
                    double* zb = obase + (size_t)pb * ld + voz;
                    double* dst = (rok && pb + g < P.batch) ? zb : dump;
                    if (NT) __builtin_nontemporal_store(acc[ct][r], dst);
                    else *dst = acc[ct][r];
                }
        }
        // Without z (ZOUT = false) the first moves are the ONLY stores behind the request above: they must not sit under a
        // condition, or a wavefront that skipped them would pass fa_await_a<16> with its 14 loads still in flight (found by
        // tests/test_isa_affine_hazard.py; the launcher only deals the tiles 16 t < m then, and rows >= m go to the dump line).
        if (!ZOUT || (P.u0out != nullptr && 16 * tile < m)) {    // uniform: the first m rows again, as the first moves
            const unsigned vou = (unsigned)(g * m + row);
            const bool rok = row < m;
#pragma unroll
            for (int ct = 0; ct < FA_CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pb = (gi * FA_CT + ct) * 16 + 4 * r;
                    double* ub = P.u0out + (size_t)pb * m + vou;
                    double* dst = (rok && pb + g < P.batch) ? ub : dump;
                    *dst = acc[ct][r];
                }
        }
#ifdef FW_TIMING
        if (tile == slot * 4 + wv) { asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0])); FA_TICK(6); }
#endif
        // exactly 16 stores (z, or the first moves when z is not wanted) + possibly 16 more follow the request above
        fa_await_a<16>(An);
#pragma unroll
        for (int q = 0; q < FA_KS; ++q) A[q] = An[q];
#ifdef FW_TIMING
        if (tile == slot * 4 + wv) FA_TICK(2);
#endif
    }
    FA_TICK(3);
}

hipError_t fmpc_launch_affine(FaParams P, int num_cu, hipStream_t stream) {
    if (P.n != 27 || 2 * P.n + 2 > FA_KC || 7 * FA_THREADS < FA_CT * 16 * P.n) return hipErrorInvalidValue;
    if (!P.zout && !P.u0out) return hipErrorInvalidValue;         // (fmpc_cold_affine<false> stores the first moves unconditionally)
    const int ncol = (P.batch + 15) / 16, ngroups = (ncol + FA_CT - 1) / FA_CT;
    P.tiles_used = P.zout ? P.tiles + (P.nuout ? P.nu_tiles : 0) : (P.m + 15) / 16;
    if (P.ldz < P.rows) P.ldz = P.rows;
    static const bool no_nt = [] { const char* e = getenv("FMPC_AFFINE_NO_NT"); return e && e[0] == '1'; }();       // A/B switch
    const bool nt = P.zout && !no_nt && P.ldz % 16 == 0 && ((size_t)P.zout & 127) == 0;
    // two workgroups of four wavefronts per CU are resident: that many workgroups share the groups of 64 problems (a workgroup
    // beyond the resident set would start when another ends)
    int wpg = (2 * num_cu) / ngroups;                            // workgroups per group
    const int wpg_max = (P.tiles_used + 3) / 4;                  // one tile per wavefront at least
    if (wpg < 1) wpg = 1;
    if (wpg > wpg_max) wpg = wpg_max;
    P.wgs_per_group = wpg;
    const int grid = ngroups * wpg;
    if (P.zout && nt) hipLaunchKernelGGL((fmpc_cold_affine<true, true>), dim3(grid), dim3(FA_THREADS), 0, stream, P);
    else if (P.zout) hipLaunchKernelGGL((fmpc_cold_affine<true, false>), dim3(grid), dim3(FA_THREADS), 0, stream, P);
    else hipLaunchKernelGGL((fmpc_cold_affine<false, false>), dim3(grid), dim3(FA_THREADS), 0, stream, P);
    return hipGetLastError();
}
