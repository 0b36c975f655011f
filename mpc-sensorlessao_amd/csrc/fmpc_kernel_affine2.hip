// CDNA4 fastMPC, cold-start step WITHOUT w (the reference's replay call, README.md:548-556) in its TWO-STAGE affine form.
//
// fmpc_kernel_affine.hip evaluates z+ = zc + Kz d as ONE product: 14 k-steps for every 16 x 16 tile of z, 321 row tiles per 16
// problems.  But the u rows of a stage factor through that stage's 27 multipliers (inf_newton_solver.m:34-35 on the u entries,
// fast_mpc_eq_const.m:38-47 for the structure of C):
//     nu+_s = nuc_s + J_s d                       27 rows  x 14 k-steps            (the dense form of the dual solve)
//     u_s   = (umid - wc o cu) + diag(wc) B' nu+_s   144 rows x  7 k-steps         (B' nu+_s: K = 27, + the constant: K = 28)
//     x_s+1 = the x rows of Kz d                  27 rows  x 14 k-steps            (they couple three stages: taken as they are)
// i.e. 2 x 14 + 9 x 7 + 2 x 14 = 119 matrix instructions per stage and 16 problems instead of 154 (at n = 27, m = 144), and the
// result tile of the first product IS the operand of the second (v_mfma_f64_16x16x4_f64 fragment maps: a result tile with its
// row index as the contraction index is directly an A operand, scripts/mfma_f64_probe.hip) -- nothing is staged in between.
//
// Work split: ONE task per wavefront, no loop over tiles whose operands have to be prefetched around stores.  A workgroup of
// EIGHT wavefronts (one per CU) = 64 problems x FOUR consecutive stages; wavefront (slot, h) takes stage slot of the four and
// the 16-row half h of nu+_s and of the x rows for all four column tiles (56 + 56 matrix instructions, ONE 7 KB operand image
// each: every image register feeds four instructions and no two wavefronts load the same image), hands its half of nu+_s to
// its partner through LDS, and takes five (h = 0) or four (h = 1) of the nine u row tiles (28 instructions each); wavefronts
// w and w + 4 -- one SIMD -- are the two halves of one stage: 252 + 224 instructions per SIMD.  Every global load of a
// wavefront is requested at its very start (first what the workgroup stages in LDS -- data and Bw --, then its images) and
// none follows a store: vmcnt counts loads and stores in one in-order counter, a load behind a store waits for the store to be
// written.  Why one workgroup per CU: what limits the prologue is the bytes a CU pulls through its L1 (measured: about 20 B
// per cycle): with two workgroups of four wavefronts (column tiles split instead of rows) every image was loaded twice and Bw
// staged twice per CU, 342 KB per CU, and the workgroup served second began its products 5 us after the first (timing
// build, scripts/affine2_trace.py); now 171 KB.  At (27, 144, 30) and 2000 problems: 32 groups x 8 quads = 256 workgroups; the
// last quad of a group holds two stages, its four free wavefronts evaluate the step-length / exit decision forms of the
// group's four column tiles (as fmpc_kernel_affine.hip does: same forms, same rounding guard), one column tile each.
//
// Rounding: u_s is computed from the fp64 nu+_s here, from Kz (built in long double) there: both are within 1e-13 of the
// oracle; tests/test_gpu_affine.py holds both to 1e-9 against it and to 1e-11 against each other.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include "fmpc_device.h"
#include "fmpc_host.h"
#include "fmpc_affine.h"
#include "../../include/fastmpc.h"

#ifdef FW_TIMING
// diagnostic build: per wavefront time stamps of the constant 100 MHz clock (scripts/affine2_trace.py)
__device__ unsigned long long fb_trace[2048 * 8];   // (256 workgroups x 8 wavefronts)
extern "C" int fmpc_debug_affine2_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fb_trace), sizeof(unsigned long long) * 2048 * 8) == hipSuccess ? 0 : -1;
}
#define FB_TICK(k) do { if (lane == 0 && blockIdx.x < 256) fb_trace[(blockIdx.x * 8 + wv) * 8 + (k)] = (unsigned long long)wall_clock64(); } while (0)
#else
#define FB_TICK(k)
#endif
typedef double d4b __attribute__((ext_vector_type(4)));
typedef double d2b __attribute__((ext_vector_type(2)));
#define FB_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define FB_THREADS 512
// experiment switch for the z stores: plain / non-temporal
#ifdef FB_NT_STORES
#define FB_ST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define FB_ST(p, v) (*(p) = (v))
#endif
#define FB_CT 4                          // column tiles (16 problems each) per workgroup
#define FB_MT 9                          // u row tiles at most (m <= 144)

__device__ __forceinline__ void fb_lds_barrier() {         // orders LDS traffic only; global loads and stores stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ bool fb_decide(const FaParams& P, double qe, double qp, double rdl, double dn2) {
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

// grid: groups x quads workgroups (+ groups more when the last quad has no four free wavefronts for the decision forms);
// P.tiles_used = quads, P.wgs_per_group = groups (launcher)
__global__ void __launch_bounds__(FB_THREADS, 2) fmpc_cold_affine2(FaParams P) {
    // LDS: the group's data in OPERAND order (as fmpc_kernel_affine.hip): entry (problem 16 ct + c, k = 4 q + g) at
    // ((q FB_CT + ct) 4 + g) 16 + c -- a wavefront's read of an operand register is 64 consecutive doubles; the images of Bw;
    // the exchange of the nu+ halves [slot][half][ct][register][lane]
    extern __shared__ __attribute__((aligned(16))) double fb_lds[];
    double* const sD = fb_lds;
    double* const sBw = sD + FA_KS * FB_CT * 64;
    double* const sX = sBw + FB_MT * FA2_KB * 64;
#define FB_SD(ct, cc, k) sD[((((k) >> 2) * FB_CT + (ct)) * 4 + ((k) & 3)) * 16 + (cc)]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, c = lane & 15;
    const int n = P.n, m = P.m, T = P.T, s = n + m;
    const int quads = P.tiles_used, groups = P.wgs_per_group;
    const int wg = (int)blockIdx.x;
    const bool extra = wg >= groups * quads;                       // a workgroup that only evaluates the decision forms
    const int gi = extra ? wg - groups * quads : wg / quads;
    const int qd = extra ? quads : wg - gi * quads;
    const int slot = wv & 3, half = wv >> 2;                       // wavefronts w and w + 4 share a SIMD: the two halves of one stage
    const int st = 4 * qd + slot;
    const bool live = !extra && st < T;
    // the decision forms go to four wavefronts without a stage: those of the last quad if it has two free slots, else of an extra workgroup
    const int nfree = 4 * quads - T;
    const bool forms = extra ? wv < 4 : (nfree >= 2 && qd == quads - 1 && slot >= 2);
    const int fct = extra ? wv : (slot - 2) + 2 * half;            // ... one column tile each
    FB_TICK(0);
    if (wg == 0 && tid == 0 && P.handed) *P.handed = 0;
    // ---- every global load of this wavefront, up front: first what the workgroup stages in LDS (data, Bw), then this wavefront's
    //      operand images (registers) -- vmcnt is in order, so the LDS writes below wait for the first group only and the images
    //      stay in flight across the barrier, which orders LDS traffic alone.  16-byte loads throughout: a CU's load path takes a
    //      wave instruction every 12-17 cycles whatever its width.
    const int p0 = gi * FB_CT * 16;
    const int mt = (m + 15) >> 4;
    const int np = P.batch - p0 < FB_CT * 16 ? P.batch - p0 : FB_CT * 16;           // problems of this group
    constexpr int NV = (FB_CT * 16 * 27 / 2 + FB_THREADS - 1) / FB_THREADS;         // pairs of the group's x0 (64 x 27 doubles) per thread
    constexpr int NBW = (FB_MT * FA2_KB * 64 / 2 + FB_THREADS - 1) / FB_THREADS;    // pairs of Bw per thread
    d2b v0[NV], v1[NV], bw[NBW];
    const int ndg = np * n, nfull = ndg >> 1;                                       // doubles / whole pairs of the group in x0
    double t0 = 0.0, t1 = 0.0;                                                      // the odd element out (never a pair across the end of x0)
    {
        const double* d0 = P.x0 + (size_t)p0 * n;
        const double* d1 = P.x0p ? P.x0p + (size_t)p0 * n : d0;
        const d2b* s0 = (const d2b*)d0;
        const d2b* s1 = (const d2b*)d1;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int pi = tid + FB_THREADS * j, pc = pi < nfull ? pi : 0;
            v0[j] = s0[pc]; v1[j] = s1[pc];
        }
        if (ndg & 1) { t0 = d0[ndg - 1]; t1 = d1[ndg - 1]; }                        // (uniform)
    }
    const int nbw2 = mt * FA2_KB * 32;                                              // pairs of Bw
    if (!extra) {
        const d2b* bsrc = (const d2b*)P.imgBw;
#pragma unroll
        for (int j = 0; j < NBW; ++j) { const int pi = tid + FB_THREADS * j; bw[j] = bsrc[pi < nbw2 ? pi : 0]; }
    }
    double Ji[FA_KS], Xi[FA_KS];
    if (live) {
        const d2b* ij = (const d2b*)P.imgJ + ((size_t)st * 2 + half) * (FA_KS / 2) * 64 + lane;
        const d2b* ix = (const d2b*)P.imgX + ((size_t)st * 2 + half) * (FA_KS / 2) * 64 + lane;
#pragma unroll
        for (int q = 0; q < FA_KS / 2; ++q) { const d2b v = ij[q * 64]; Ji[2 * q] = v.x; Ji[2 * q + 1] = v.y; }
#pragma unroll
        for (int q = 0; q < FA_KS / 2; ++q) { const d2b v = ix[q * 64]; Xi[2 * q] = v.x; Xi[2 * q + 1] = v.y; }
    }
    {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int pi = tid + FB_THREADS * j;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int idx = 2 * pi + e;
                if (idx < FB_CT * 16 * n) {
                    const int pr = idx / n, k = idx - pr * n;
                    const bool on = idx < ndg;
                    const double a0 = pi < nfull ? (e ? v0[j].y : v0[j].x) : t0, a1 = pi < nfull ? (e ? v1[j].y : v1[j].x) : t1;
                    FB_SD(pr >> 4, pr & 15, k) = on ? a0 : 0.0;
                    FB_SD(pr >> 4, pr & 15, n + k) = (on && P.x0p) ? a1 : 0.0;
                }
            }
        }
        if (tid < FB_CT * 16) { FB_SD(tid >> 4, tid & 15, 2 * n) = 1.0; FB_SD(tid >> 4, tid & 15, 2 * n + 1) = 0.0; }
        if (!extra) {
            d2b* bdst = (d2b*)sBw;
#pragma unroll
            for (int j = 0; j < NBW; ++j) { const int pi = tid + FB_THREADS * j; if (pi < nbw2) bdst[pi] = bw[j]; }
        }
    }
    fb_lds_barrier();
    FB_TICK(1);
    if (!live) {
        // (every wavefront of a task workgroup takes part in the second barrier: the exchange of the nu+ halves)
        if (!extra) fb_lds_barrier();
        if (!forms || p0 + fct * 16 >= P.batch) return;
        // ============================================================ decision forms: ONE column tile per wavefront (the four row
        // blocks t of E and Ep in turn, 4 x 28 matrix instructions), no exchange between the wavefronts
        // lower bound of ||r_d(nu0)||^2: its x entries of the last stage (as the gate of the panel path), lane c: entries c, c + 16
        double rdl[4] = {P.rd2_0, P.rd2_0, P.rd2_0, P.rd2_0};
        double xa[4][2];
        if (P.nu0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pr = p0 + fct * 16 + 4 * r + g < P.batch ? p0 + fct * 16 + 4 * r + g : P.batch - 1;
                const double* nu = P.nu0 + (size_t)pr * P.nb * n;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int e = c + 16 * j, ec = e < n ? e : 0;
                    xa[r][j] = P.dx0T[ec] + nu[(T - 1) * n + ec] + (P.has_xf ? nu[T * n + ec] : 0.0);
                }
            }
        }
        double Df[FA_KS];
#pragma unroll
        for (int q = 0; q < FA_KS; ++q) Df[q] = sD[(q * FB_CT + fct) * 64 + lane];     // (the images' column 2 n is zero: the constant 1 drops out)
        double pe[4] = {0, 0, 0, 0}, pq[4] = {0, 0, 0, 0}, pn[4] = {0, 0, 0, 0}, te[4] = {0, 0, 0, 0}, tp[4] = {0, 0, 0, 0}, tn[4] = {0, 0, 0, 0};
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {
            const double* ie = P.imgE + (size_t)t * FA_KS * 64 + lane;
            const double* ip = P.imgEp + (size_t)t * FA_KS * 64 + lane;
            double E1[FA_KS], E2[FA_KS];
#pragma unroll
            for (int q = 0; q < FA_KS; ++q) { E1[q] = ie[q * 64]; E2[q] = ip[q * 64]; }
            const int k = 16 * t + c;
            const double le = P.elin[k], lp = P.eplin[k];                          // 2 e and -2 ep, zero beyond 2 n (64 entries)
            d4b ce = {0, 0, 0, 0}, cp = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < FA_KS; ++q) { ce = FB_MFMA(Df[q], E1[q], ce); cp = FB_MFMA(Df[q], E2[q], cp); }
            // register r <-> problem 4 r + g of the tile, entry k = 16 t + c of d; summed over t as fmpc_kernel_affine.hip does:
            // (t0 + t1) + (t2 + t3)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double dv = k < 2 * n ? FB_SD(fct, 4 * r + g, k < 2 * n ? k : 0) : 0.0;
                double qe = dv * (ce[r] + le), qp = dv * (cp[r] + lp), dn2 = dv * dv;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { qe += __shfl_xor(qe, o, 64); qp += __shfl_xor(qp, o, 64); dn2 += __shfl_xor(dn2, o, 64); }
                if ((t & 1) == 0) { pe[r] = qe; pq[r] = qp; pn[r] = dn2; }
                else {
                    pe[r] += qe; pq[r] += qp; pn[r] += dn2;
                    te[r] = t == 1 ? pe[r] : te[r] + pe[r]; tp[r] = t == 1 ? pq[r] : tp[r] + pq[r]; tn[r] = t == 1 ? pn[r] : tn[r] + pn[r];
                }
            }
        }
        if (P.nu0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double acc = xa[r][0] * xa[r][0] + (c + 16 < n ? xa[r][1] * xa[r][1] : 0.0);
                acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64); acc += __shfl_xor(acc, 8, 64);
                rdl[r] = acc;
            }
        }
        if (c == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * r + g, pp = p0 + fct * 16 + i;
                if (pp < P.batch) {
                    const bool clear = fb_decide(P, te[r], tp[r], rdl[r], tn[r]);
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
        FB_TICK(5);
        return;
    }
    const int rows = P.rows;
    double* const zrow = P.zout + (size_t)st * s;
    const bool uo = P.u0out != nullptr && st == 0;
    // stores: register r of lane (g, j) is (problem 16 ct + 4 r + g, row 16 tile + j): 16 lanes write 128 consecutive bytes
    // ============================================================ this half of nu+_s = [J_s | nuc_s] d'  (rows = multipliers, columns =
    // problems) and of the x rows (problems as the rows of the product); both from registers and LDS alone
    // Order: nu+ first (everything else of the stage waits for it), then the u rows -- the bulk of the stores -- and the x rows
    // last: once the stores flow the kernel is bound by the HBM write path (82 MB at 2000 problems: 15-17 us), so what counts
    // is how soon the FIRST store leaves and that the stream of stores never pauses.
    d4b nu[FB_CT];
#pragma unroll
    for (int ct = 0; ct < FB_CT; ++ct) nu[ct] = (d4b){0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < FA_KS; ++q) {
#pragma unroll
        for (int ct = 0; ct < FB_CT; ++ct) nu[ct] = FB_MFMA(Ji[q], sD[(q * FB_CT + ct) * 64 + lane], nu[ct]);
    }
    // the half goes to the partner through LDS
    {
        double* xo = sX + (size_t)((slot * 2 + half) * FB_CT) * 256 + lane;
#pragma unroll
        for (int ct = 0; ct < FB_CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) xo[(ct * 4 + r) * 64] = nu[ct][r];
    }
    FB_TICK(2);
    fb_lds_barrier();
    d4b nup[FB_CT];
    {
        const double* xi = sX + (size_t)((slot * 2 + (1 - half)) * FB_CT) * 256 + lane;
#pragma unroll
        for (int ct = 0; ct < FB_CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) nup[ct][r] = xi[(ct * 4 + r) * 64];
    }
    FB_TICK(3);
    // ============================================================ u_s = Bw [nu+_s ; 1]: the result tiles above are the A operand
    // (k-step q: rows 4 q .. 4 q + 3 of nu+_s: register q % 4 of half q / 4)
    d4b nlo[FB_CT], nhi[FB_CT];                                    // rows 0 .. 15 and 16 .. 31 of nu+_s (one select per register, once)
#pragma unroll
    for (int ct = 0; ct < FB_CT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) { nlo[ct][r] = half == 0 ? nu[ct][r] : nup[ct][r]; nhi[ct][r] = half == 0 ? nup[ct][r] : nu[ct][r]; }
    const int ut0 = half == 0 ? 0 : (mt + 1) / 2, ut1 = half == 0 ? (mt + 1) / 2 : mt;
    for (int ut = ut0; ut < ut1; ++ut) {
        const double* bwp = sBw + (size_t)ut * FA2_KB * 64 + lane;
        d4b acc[FB_CT];
#pragma unroll
        for (int ct = 0; ct < FB_CT; ++ct) acc[ct] = (d4b){0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < FA2_KB; ++q) {
            const double b = bwp[q * 64];
#pragma unroll
            for (int ct = 0; ct < FB_CT; ++ct) {
                acc[ct] = FB_MFMA((q >> 2) == 0 ? nlo[ct][q & 3] : nhi[ct][q & 3], b, acc[ct]);
            }
        }
        const int ur = 16 * ut + c;
        if (ur < m) {
#pragma unroll
            for (int ct = 0; ct < FB_CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pp = p0 + ct * 16 + 4 * r + g;
                    if (pp < P.batch) { FB_ST(&zrow[(size_t)pp * rows + ur], acc[ct][r]); if (uo) P.u0out[(size_t)pp * m + ur] = acc[ct][r]; }
                }
        }
    }
    // ============================================================ this half of the x rows (problems as the rows of the product)
    {
        d4b xr_[FB_CT];
#pragma unroll
        for (int ct = 0; ct < FB_CT; ++ct) xr_[ct] = (d4b){0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < FA_KS; ++q) {
#pragma unroll
            for (int ct = 0; ct < FB_CT; ++ct) xr_[ct] = FB_MFMA(sD[(q * FB_CT + ct) * 64 + lane], Xi[q], xr_[ct]);
        }
        const int xr = 16 * half + c;
        if (xr < n) {
#pragma unroll
            for (int ct = 0; ct < FB_CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pp = p0 + ct * 16 + 4 * r + g;
                    if (pp < P.batch) FB_ST(&zrow[(size_t)pp * rows + m + xr], xr_[ct][r]);
                }
        }
    }
    FB_TICK(4);
}

static size_t fb_lds_bytes() { return (size_t)(FA_KS * FB_CT * 64 + FB_MT * FA2_KB * 64 + 4 * 2 * FB_CT * 256) * sizeof(double); }

bool fmpc_affine2_applies(const FaParams& P) {
    // OPT-IN (FMPC_AFFINE2=1): measured 3 us SLOWER per 2000-problem step than fmpc_cold_affine although it issues 22 % fewer matrix
    // instructions and its wavefronts end at the same time (29 us): the step is bound by the HBM write path (82 MB; the L2 has to
    // merge the 128-byte runs of neighbouring tiles, non-temporal stores take twice as long) and by the write-back at the kernel's
    // end, not by the matrix pipes -- DESIGN.md, round 4.  Kept under test as the second, independent evaluation of the affine map.
    const char* e = getenv("FMPC_AFFINE2");                       // (read per call: tests switch it)
    return e && e[0] == '1' && P.imgJ != nullptr && P.n == 27 && P.m <= 16 * FB_MT && P.zout != nullptr && P.nuout == nullptr &&
           2 * P.n + 2 <= FA_KC && P.n + 1 <= 4 * FA2_KB;
}

hipError_t fmpc_launch_affine2(FaParams P, hipStream_t stream) {
    if (!fmpc_affine2_applies(P)) return hipErrorInvalidValue;
    static const hipError_t prep = hipFuncSetAttribute((const void*)fmpc_cold_affine2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fb_lds_bytes());
    if (prep != hipSuccess) return prep;
    const int ncol = (P.batch + 15) / 16, groups = (ncol + FB_CT - 1) / FB_CT, quads = (P.T + 3) / 4;
    P.tiles_used = quads; P.wgs_per_group = groups;
    const bool embedded = 4 * quads - P.T >= 2;                    // the last quad has four wavefronts without a stage
    hipLaunchKernelGGL(fmpc_cold_affine2, dim3(groups * quads + (embedded ? 0 : groups)), dim3(FB_THREADS), fb_lds_bytes(), stream, P);
    return hipGetLastError();
}
