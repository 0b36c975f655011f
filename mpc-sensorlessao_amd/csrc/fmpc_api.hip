// C-ABI of the MI355X fastMPC solver (include/fastmpc.h): handle, host-side assembly of the
// iteration-invariant blocks, launches.  No CPU solve path exists in this library.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/fastmpc.h"
#include "fmpc_device.h"
#include "fmpc_panel.h"
#include "fmpc_host.h"
#include "fmpc_first.h"
#include "fmpc_affine.h"
#include "fmpc_loopu0.h"
#include "fmpc_tiled.h"
#include "fmpc_rampcold.h"
#define FMPC_PRODUCT_MIN_BATCH_DEFAULT 65   // closed-loop steps with first moves only: the product form from this many realisations on
#include "fmpc_alloc.h"                    // counted hipMalloc / hipFree (fmpc_alloc_generation), shared with fmpc_est_api.hip

std::atomic<unsigned long long> fmpc_alloc_gen{0};
extern "C" unsigned long long fmpc_alloc_generation(void) { return fmpc_alloc_gen.load(); }


// kernels / launchers (fmpc_kernel_generic.hip)
size_t fmpc_generic_lds_bytes(int n, int m);
size_t fmpc_generic_big_lds_bytes(int n, int m);
hipError_t fmpc_generic_prepare(size_t lds_bytes, int big);
// fmpc_kernel_ramp.hip
size_t fmpc_ramp_lds_bytes(int n, int m, int nbn);
size_t fmpc_ramp_ws_doubles(int n, int m, int T, int nb);
hipError_t fmpc_ramp_prepare(size_t lds_bytes);
hipError_t fmpc_launch_ramp(const FmpcDevModel& M, const double* dumin, const double* dumax, int batch, int grid,
                            const double* x0, const double* x0p, const double* w, const double* uprev,
                            const double* zinit, const double* nu0, int max_iter, double kbar, double* zout,
                            double* nuout, int* status, int* iters, double* step, int step_ld, double* ws,
                            size_t ws_stride, int threads, hipStream_t stream, int it0 = 0);
hipError_t fmpc_launch_generic(const FmpcDevModel& M, int batch, int grid, const double* x0,
                               const double* x0p, const double* w, const double* zinit,
                               const double* nu0, int max_iter, double kbar, double* zout,
                               double* nuout, int* status, int* iters, double* step, int step_ld,
                               double* ws, size_t ws_stride, hipStream_t stream, int big);
hipError_t fmpc_launch_unpack(int n, int m, int T, int batch, const double* z, double* U,
                              double* X, double* u0, hipStream_t stream);
hipError_t fmpc_launch_loop_inputs(int n, int m, int T, int batch, const double* Bt, const double* M1, const double* M2,
                                   const double* a, const double* x0_last, const double* u1, const double* u2,
                                   double* x0, double* x0_pre, double* w, hipStream_t stream, double* lv = nullptr);

// one-wave-per-problem MFMA kernel (fmpc_kernel_wave.hip)
size_t fmpc_wave_lds_bytes(int n, int mp);
bool fmpc_wave_supports(int n);
int fmpc_wave_img_stride(int n);
int fmpc_wave_mp(int m);
int fmpc_wave_waves_per_wg();
size_t fmpc_wave_ws_doubles(int n, int m, int mp, int T, int nb);
void fmpc_wave_make_images(int n, const double* blk, double* out);
hipError_t fmpc_wave_prepare(int n, size_t lds_bytes);
hipError_t fmpc_launch_wave(const FmpcDevModel& M, const FwModel& V, int batch, int grid,
                            const double* x0, const double* x0p, const double* w, const double* zinit,
                            const double* nu0, int max_iter, double kbar, double* zout, double* nuout,
                            int* status, int* iters, double* step, int step_ld, double* ws,
                            size_t ws_stride, size_t lds_bytes, hipStream_t stream,
                            int mode, double* sh_fac, double* sh_rs, int* sh_ok, const double* cold,
                            const double* gate = nullptr, const double* epsp = nullptr, int* handed = nullptr,
                            const double* nuws = nullptr, double* u0out = nullptr,
                            int pphase = 0, const double* rnp = nullptr, int* list = nullptr, int u0_done = 0,
                            int* nflag = nullptr, int zld = 0);
size_t fmpc_wave_shared_fac_doubles(int n, int nb);
void fmpc_wave_cold_layout(int n, int mp, int* off9);

hipError_t fmpc_launch_var_identify(int n, int num_train, int num_samples, int batch, const double* series, double* A1,
                                    double* A2, int* status, hipStream_t stream);

#define FMPC_LDS_LIMIT (160 * 1024)

struct fmpc_handle_s {
    int n, m, T, nb, var_order, has_xf, device;
    int num_cu;
    FmpcDevModel dev;            // device pointers into `pool`
    const double* loop_M1; const double* loop_M2;   // closed-loop prediction matrices (T n) x n, row-major
    double* pool_d;              // one allocation for all shared doubles
    int* pool_i;                 // and one for the index arrays
    size_t lds_bytes;
    int wg_per_cu;
    // wave kernel (n = 27): used when available unless FMPC_FORCE_GENERIC=1
    int use_wave;
    double* fm_forms;            // diagnostic: where the one-step first-move kernel leaves the bounds its decision used (fmpc_debug_first_move_forms)
    double* fm_compact;          // compact batch of the stopped realisations of a walk (fmpc_loop_run_walk)
    int* fm_walk_i; size_t fm_walk_cap;   // start / stop step per realisation of a walk
    // affine form of the cold-start step without w (fmpc_kernel_affine.hip), built with the first-move form
    FaParams fa_P; int fa_valid, fa_disabled; int* fa_need; size_t fa_need_cap;
    double* ao_scr; size_t ao_cap;       // fmpc_ao_step_device: a zero a[k] and scratch x0 / x0_pre for the loop-input kernel (3 batch n)
    FlParams fs_P; FlStepIn fs_I; int fs_valid, fs_disabled, fs_min_batch;   // the fused step (fmpc_loop_step27): images in its column order, B's images
    FlParams fl_P; int fl_valid, fl_disabled;       // first-move form as a product: closed-loop steps of > 64 realisations (fmpc_kernel_loopu0.hip)
    FwModel wave;
    double* wave_pool_d;
    int* wave_pool_i;
    size_t wave_lds;
    // shared cold-start factor (SURVEY §7.2a regime (ii)): Phi, Y and its Cholesky factor are the
    // same for every problem in the first Newton step from the mid-box start; it depends only on the
    // model and on k, so it is computed once per (handle, k) and kept in HBM (L2-resident, 0.5 MB).
    double* sh_fac; double* sh_rs; int* sh_ok; double* sh_scratch;
    double sh_k; int sh_valid; int sh_enabled;
    double* cold_d;                      // cold-start constants on the device (FwCold layout)
    // panel kernel (fmpc_kernel_panel.hip): the cold-start step on 16-problem panels, n_newton = 1
    int pn_enabled, pn_valid, pn_mp;
    int last_path;
    size_t pn_lds;
    double* pn_pool;                     // [simg | btimg | aimg | vec | ucon]
    size_t pn_o_simg, pn_o_limg, pn_o_bt, pn_o_aimg, pn_o_vec, pn_o_ucon, pn_o_dump, pn_doubles;
    int* pn_cnt;                         // problems the exact path had to solve in the last call (diagnostic)
    int* gn_list; double* gn_nu; int* gn_cnt; int* gn_cnt_host; size_t gn_cap; int gn_split;   // explicit-start batches with a budget > 1: first step / continuation split (FMPC_NO_GENERAL_SPLIT=1: one launch)
    int* fa_nflag;                       // affine form: [running count of flagged problems | the count the last exact-path launch dealt with | its
                                         // ticket]: the exact-path launch leaves at once while the two counts agree (FwParams::nflag)
    size_t pn_cap;                       // per-batch buffers of the panel path, grown together
    double* pn_gate; double* pn_epsp; double* pn_nuws;
    double* pn_rnp; int* pn_list;        // budgets > 1: next-exit-test partials, compacted list of the problems that go on
    int* pn_cnt_host;                    // pinned copy of pn_cnt of an EARLIER call (never waited for): sizes the continuation grid
    size_t pn_o_dz; int pn_dz_len;
    size_t pn_dz_lds;
    double pn_rd2_0, pn_rp2c;
    // dense form of the cold-start dual solve (fmpc_kernel_inv.hip): nu+ = nuc + J d, built per (handle, k) on first use
    int inv_enabled, inv_valid, inv_jks, inv_max_batch, inv_last; double inv_k;
    int inv_failed; double inv_failed_k;                         // the build for this k gave a non-finite J: not retried
    double* inv_jimg; double* inv_nuc; double* inv_eimg;
    double* inv_jst; double* inv_nucst; int inv_fuse;            // per-stage rows of J_x and nuc: the dual solve fused into d_z (w = NULL, no xf, budget 1)
    double* inv_jimg2; int inv_jks2;     // J' = [J_x | -J_w M1 | -J_w M2]: closed-loop steps, w = -M1 B u1 - M2 B u2 (fmpc_loop_step_device)
    double* lp_v; size_t lp_cap; int lp_hint;                    // [B u1 ; B u2] per problem of the running loop step
    // first-move form of the closed-loop step (fmpc_kernel_first.hip): host copies of J (columns [x0 | x0_pre | B u1 | B u2]) and
    // nuc kept by fmpc_build_inverse, the derived matrices on the device per (handle, k), a flag per realisation
    std::vector<double> hm_J4, hm_nuc; double hm_J4_k; int hm_J4_valid;
    double* fm_pool; int fm_valid, fm_disabled; double fm_k; FmParams fm_P; int* fm_need; size_t fm_need_cap;
    std::vector<double> hm_m1, hm_m2;
    std::vector<double> hm_Q2, hm_Qf2, hm_ql, hm_qfl, hm_xf, hm_blocks;
    std::vector<int> hm_idxD, hm_idx1, hm_idx2;
    int* pn_sched; int pn_nsf, pn_nsb, pn_limg_cap;
    std::vector<double> hm_R2, hm_rl, hm_umin, hm_umax, hm_umid, hm_xmid, hm_bt, hm_a1, hm_a2;   // host copies
    // tiled kernel (fmpc_kernel_tiled.hip): workgroup per problem, any n <= 79; images per arithmetic type, built on first use
    int generic_ok;                      // the generic kernel's LDS tiles fit (n <= 64), or generic_big
    int prefer_tiled;                    // the fp64 tiled kernel exists for this size: it is 1.3 (n = 4) to 9 (n = 45) times faster than the
                                         // generic kernel at every batch size (round 5 measurement): the default; FMPC_FORCE_GENERIC=1: generic
    int generic_big;                     // sizes no other kernel takes (n > 79 ...), diagonal weights: the generic kernel with its tiles in the workspace
    int prec;                            // FMPC_PREC_F64 / FMPC_PREC_F32_MIXED of the per-problem-factor path
    int force_tiled;                     // FMPC_TILED=1: route every solve through the tiled kernel (tests, profiles)
    struct Tiled { int ready, NB, NW; size_t lds; void* pool; int* ipool; double* bm; FtModel V; } tl[2];   // bm: padded fp64 images   // [0] fp64, [1] fp32
    double* tl_ws; size_t tl_ws_doubles; int tl_prepared;         // (bit NW: that wavefront count of the fp64 instance is prepared)
    int tl_last_nw;                       // wavefronts per problem of the last tiled launch (diagnostic)
    int z_ld;                             // fmpc_set_z_ld: doubles between the z rows of consecutive problems (0: T (n + m))
    int small_tiled;                      // per-problem-factor solves of few problems go to the tiled kernel (FMPC_NO_SMALL_TILED=1: off)
    int small_nw;                         // ... with this many wavefronts per problem: 4 (default since round 5) or 2 (FMPC_SMALL_TILED_NW=2 /
                                          // fmpc_set_small_batch_kernel(h, 4): opt-in, see the note at FT_DISPATCH in fmpc_kernel_tiled.hip)
    std::vector<double> hm_b;            // B row-major n x m
    std::vector<double> hm_a1f, hm_a2f;  // A1, A2 row-major (always kept: the tiled kernel's images)
    int denseQ;                          // Q or Qf not diagonal: tiled kernel only
    int denseR;                          // R not diagonal: tiled kernel, fp64
    std::vector<double> hm_r2full;       // 2R, m x m row-major
    std::vector<double> hm_q2m, hm_qf2m, hm_xm, hm_xfm;   // 2Q, 2Qf, (2Q)^-1, (2Qf)^-1 row-major
    // ramp-rate rows (VAR_1): bounds on the device, own workspace (dense Y per workgroup)
    double* ramp_du;             // [du_min | du_max], 2 m doubles; nullptr until fmpc_set_ramp
    double* ramp_ws; size_t ramp_ws_doubles;
    // cold-start step with the ramp rows in its Woodbury form (fmpc_ramp_cold): constants per (handle, k, ramp bounds)
    std::vector<double> hm_dumin, hm_dumax;
    int rc_disabled, rc_valid, rc_failed, rc_last; double rc_k, rc_failed_k;
    double* rc_pool; FrColdParams rc_P;
    double* rc_ws; size_t rc_ws_doubles;
    double* rc_nu; int* rc_si; size_t rc_cap;     // nu / status / iters of the first step when the caller passes none (budgets > 1)
    // A handle's device workspaces serve ONE solve at a time.  Solves enqueued on different streams are ordered on
    // the device: every solve records `ev`, and a solve on another stream than the last one waits for it first.
    hipEvent_t ev; int ev_valid; hipStream_t last_stream;
    // workspace, grown on demand; `mu` serialises the host-side enqueue (a handle may be shared between threads)
    std::mutex mu;
    std::mutex host_mu;          // serialises the host-pointer entry points (shared staging)
    double* ws;
    size_t ws_doubles;
    // z_out == NULL (first moves only): the working iterate of the problems a kernel has to iterate on lives here
    double* zs; size_t zs_doubles;
    // staging for the host-pointer entry points
    void* stage;
    size_t stage_bytes;
    void* pin; void* pin_dev; size_t pin_bytes;         // pinned host twin of the staging block for small host-pointer solves (fmpc_solve_host)
};

extern "C" int fmpc_version(void) { return FMPC_VERSION; }

extern "C" const char* fmpc_strerror(int code) {
    switch (code) {
        case FMPC_OK: return "ok";
        case FMPC_W_LINESEARCH: return "line search collapsed to t = 0";
        case FMPC_E_NULL: return "Define the state dynamics/equality constrained matrix";
        case FMPC_E_DIM: return "size mismatch";
        case FMPC_E_UNSUPPORTED: return "not supported by the device path yet";
        case FMPC_E_NOT_PD_PHI: return "Matrix must be positive definite (KKT_H)";
        case FMPC_E_NOT_PD_SCHUR: return "Matrix must be positive definite (Schur)";
        case FMPC_E_HIP: return "HIP runtime error";
        case FMPC_E_ALLOC: return "allocation failed";
        case FMPC_E_NO_DEVICE: return "no HIP device (this library has no CPU path)";
        default: return "unknown fastmpc status";
    }
}

extern "C" int fmpc_step_ld(int n_newton) { return n_newton > 0 ? n_newton : 1000; }

namespace {

// column-major (MATLAB) element
inline double cm(const double* M, int rows, int r, int c) { return M[(size_t)r + (size_t)c * rows]; }

bool is_diag(const double* M, int n) {
    for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r)
            if (r != c && cm(M, n, r, c) != 0.0) return false;
    return true;
}

// out (row-major n x n) += A diag(x) B'   with A, B row-major n x n
void add_AxBt(std::vector<double>& out, const std::vector<double>& A, const std::vector<double>& x,
              const std::vector<double>& B, int n, double sign) {
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            double t = 0.0;
            for (int c = 0; c < n; ++c) t += A[a * n + c] * x[c] * B[b * n + c];
            out[a * n + b] += sign * t;
        }
}

// inverse of a symmetric positive definite matrix (row-major n x n) by Cholesky in extended precision; false: not PD
bool spd_inverse(const std::vector<double>& A, int n, std::vector<double>& inv) {
    typedef long double ld;
    std::vector<ld> L((size_t)n * n, 0.0L), Li((size_t)n * n, 0.0L);
    for (int c = 0; c < n; ++c) {
        ld d = A[c * n + c];
        for (int k = 0; k < c; ++k) d -= L[c * n + k] * L[c * n + k];
        if (!(d > 0.0L)) return false;
        const ld l = sqrtl(d);
        L[c * n + c] = l;
        for (int r = c + 1; r < n; ++r) {
            ld t = A[r * n + c];
            for (int k = 0; k < c; ++k) t -= L[r * n + k] * L[c * n + k];
            L[r * n + c] = t / l;
        }
    }
    for (int c = 0; c < n; ++c) {                    // Li = L^-1
        Li[c * n + c] = 1.0L / L[c * n + c];
        for (int r = c + 1; r < n; ++r) {
            ld t = 0.0L;
            for (int k = c; k < r; ++k) t += L[r * n + k] * Li[k * n + c];
            Li[r * n + c] = -t / L[r * n + r];
        }
    }
    inv.assign((size_t)n * n, 0.0);
    for (int a = 0; a < n; ++a)
        for (int b = 0; b <= a; ++b) {
            ld t = 0.0L;
            for (int k = a; k < n; ++k) t += Li[k * n + a] * Li[k * n + b];
            inv[a * n + b] = inv[b * n + a] = (double)t;
        }
    return true;
}

}  // namespace

extern "C" int fmpc_create(fmpc_handle* out, int n, int m, int T, int var_order,
                           const double* A1, const double* A2, const double* B,
                           const double* Q, const double* R, const double* Qf,
                           const double* q, const double* r, const double* qf,
                           const double* x_min, const double* x_max,
                           const double* u_min, const double* u_max,
                           const double* xf, int device) {
    if (!out) return FMPC_E_NULL;
    *out = nullptr;
    if (n <= 0 || m <= 0 || T <= 0 || (var_order != 1 && var_order != 2)) return FMPC_E_DIM;
    if (!A1 || !B || (var_order == 2 && !A2)) return FMPC_E_NULL;   // fast_mpc_eq_const.m:19-25
    if (!Q || !R || !Qf || !x_min || !x_max || !u_min || !u_max) return FMPC_E_NULL;
    if (device < 0) return FMPC_E_NO_DEVICE;
    // dense (symmetric positive definite) R: fast_mpc_objective.m:51-54 takes any square R.  Handled by the tiled kernel in
    // fp64 (a per-stage m x m factorisation in LDS): n <= 47 and m (m + 1) / 2 + m (n + 2) doubles of LDS.
    const bool denseR = !is_diag(R, m);
    // dense (symmetric positive definite) Q, Qf: fast_mpc_objective.m:52-55 takes any square Q, Qf.  Handled by the
    // tiled kernel only (the other kernels keep the state weights as diagonals).
    const bool denseQ = !is_diag(Q, n) || !is_diag(Qf, n);
    for (int i = 0; i < n; ++i)
        if (!(cm(Q, n, i, i) > 0.0) || !(cm(Qf, n, i, i) > 0.0)) return FMPC_E_NOT_PD_PHI;
    for (int i = 0; i < m; ++i)
        if (!(cm(R, m, i, i) > 0.0)) return FMPC_E_NOT_PD_PHI;
    if (denseQ)
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < a; ++b)
                if (cm(Q, n, a, b) != cm(Q, n, b, a) || cm(Qf, n, a, b) != cm(Qf, n, b, a)) return FMPC_E_NOT_PD_PHI;
    std::vector<double> R2full;
    if (denseR) {
        R2full.assign((size_t)m * m, 0.0);
        for (int a = 0; a < m; ++a)
            for (int b = 0; b < m; ++b) {
                if (cm(R, m, a, b) != cm(R, m, b, a)) return FMPC_E_NOT_PD_PHI;
                R2full[(size_t)a * m + b] = 2.0 * cm(R, m, a, b);
            }
        std::vector<double> tmp;
        if (!spd_inverse(R2full, m, tmp)) return FMPC_E_NOT_PD_PHI;      // (the inverse itself is not used: a positive definiteness test)
    }
    // per-problem-factor paths: the generic kernel (n <= 64 and its tiles fit the LDS), else the tiled kernel in fp64
    // (n <= 47), else the tiled kernel with an fp32 factor (n <= 79: "fp32 mixed precision", BASELINE configs[4])
    size_t lds = fmpc_generic_lds_bytes(n, m);
    bool generic_ok = !denseQ && !denseR && n <= 64 && lds <= FMPC_LDS_LIMIT;
    const int nb_ = T + (xf ? 1 : 0);
    // (fp64 tiled instances of 4 and 5 blocks of 16, 47 < n <= 79: round 5)
    int NB64 = 0;
    const bool tiled64_any = fmpc_tiled_supports(n, m, nb_, 0, &NB64, nullptr, denseR);
    // (likewise the fp32-factor instances of 6 and 7 blocks, 79 < n <= 111: on request; the default beyond n = 79 is the exact fp64 fallback)
    int NB32 = 0;
    const bool tiled32_any = fmpc_tiled_supports(n, m, nb_, 1, &NB32, nullptr, denseR);
    const bool tiled64 = tiled64_any && NB64 <= 3, tiled32 = tiled32_any && NB32 <= 5;
    // Any other size with diagonal weights (the reference checks shapes only, fast_mpc_objective.m:17-47): the generic kernel
    // with its tiles in the HBM workspace ("big": a size fallback, fp64, no speed claim).  FMPC_GENERIC_BIG=1 forces it (tests).
    bool generic_big = false;
    { const char* fb = getenv("FMPC_GENERIC_BIG");
      if ((!generic_ok && !tiled64 && !tiled32) || (fb && fb[0] == '1'))
          generic_big = fmpc_generic_big_lds_bytes(n, m) <= FMPC_LDS_LIMIT; }   // (dense Q, Qf, R: taken)
    if (generic_big) { generic_ok = true; lds = fmpc_generic_big_lds_bytes(n, m); }
    if (!generic_ok && !tiled64 && !tiled32) return FMPC_E_UNSUPPORTED;
    if (!generic_ok) lds = 0;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FMPC_E_NO_DEVICE;
    if (device >= ndev) return FMPC_E_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return FMPC_E_HIP;

    fmpc_handle h = new (std::nothrow) fmpc_handle_s();
    if (!h) return FMPC_E_ALLOC;
    h->n = n; h->m = m; h->T = T; h->var_order = var_order; h->has_xf = xf ? 1 : 0;
    h->nb = T + h->has_xf; h->device = device;
    h->pool_d = nullptr; h->pool_i = nullptr; h->ws = nullptr; h->ws_doubles = 0; h->zs = nullptr; h->zs_doubles = 0;
    h->stage = nullptr; h->stage_bytes = 0; h->lds_bytes = lds;
    h->pin = nullptr; h->pin_bytes = 0; h->pin_dev = nullptr;
    h->ramp_du = nullptr; h->ramp_ws = nullptr; h->ramp_ws_doubles = 0;
    { const char* e = getenv("FMPC_NO_RAMP_COLD"); h->rc_disabled = (e && e[0] == '1') ? 1 : 0; }
    h->rc_valid = 0; h->rc_failed = 0; h->rc_last = 0; h->rc_k = 0.0; h->rc_failed_k = 0.0; h->rc_pool = nullptr;
    h->rc_ws = nullptr; h->rc_ws_doubles = 0; h->rc_nu = nullptr; h->rc_si = nullptr; h->rc_cap = 0;
    h->ev = nullptr; h->ev_valid = 0; h->last_stream = nullptr;
    h->generic_ok = generic_ok ? 1 : 0; h->generic_big = generic_big ? 1 : 0;
    { const char* fg = getenv("FMPC_FORCE_GENERIC"); h->prefer_tiled = (tiled64_any && !generic_big && !(fg && fg[0] == '1')) ? 1 : 0; }
    // fp64 wherever an fp64 kernel on the matrix cores (or the generic kernel) exists -- the reference is fp64 throughout and the literal
    // drop-in call (fmpc_solve_once) has no precision argument; the fp32 factor (BASELINE configs[4]) is a request since round 5
    h->prec = (generic_ok || tiled64_any) ? FMPC_PREC_F64 : FMPC_PREC_F32_MIXED;
    { const char* ft = getenv("FMPC_TILED"); h->force_tiled = (ft && ft[0] == '1') ? 1 : 0; }
    memset(h->tl, 0, sizeof(h->tl)); h->tl_ws = nullptr; h->tl_ws_doubles = 0; h->tl_prepared = 0;
    { const char* ns = getenv("FMPC_NO_SMALL_TILED"); h->small_tiled = (ns && ns[0] == '1') ? 0 : 1; }
    h->z_ld = 0;
    { const char* nw = getenv("FMPC_SMALL_TILED_NW"); h->small_nw = (nw && nw[0] == '2') ? 2 : 4; }      // (four: the default again, round 5)
    h->use_wave = 0; h->wave_pool_d = nullptr; h->wave_pool_i = nullptr; h->wave_lds = 0;
    h->sh_fac = nullptr; h->sh_rs = nullptr; h->sh_ok = nullptr; h->sh_scratch = nullptr; h->sh_k = 0.0; h->sh_valid = 0; h->sh_enabled = 0; h->cold_d = nullptr;
    h->last_path = 0; h->pn_sched = nullptr; h->pn_nsf = 0; h->pn_nsb = 0; h->pn_limg_cap = 0; h->pn_enabled = 0; h->pn_valid = 0; h->pn_mp = 0; h->pn_lds = 0; h->pn_pool = nullptr;
    h->hm_J4_valid = 0; h->hm_J4_k = 0.0; h->fm_pool = nullptr; h->fm_valid = 0; h->fm_disabled = 0; h->fm_k = 0.0; h->fm_need = nullptr; h->fm_need_cap = 0; h->fm_compact = nullptr; h->fm_forms = nullptr; h->fm_walk_i = nullptr; h->fm_walk_cap = 0;
    h->tl_last_nw = 0; h->ao_scr = nullptr; h->ao_cap = 0;
    h->fa_valid = 0; h->fa_need = nullptr; h->fa_need_cap = 0; h->fl_valid = 0; h->fs_valid = 0;
    { const char* na = getenv("FMPC_NO_AFFINE"); h->fa_disabled = (na && na[0] == '1') ? 1 : 0; }
    { const char* na = getenv("FMPC_NO_LOOP_U0"); h->fl_disabled = (na && na[0] == '1') ? 1 : 0; }
    { const char* na = getenv("FMPC_NO_LOOP_FUSE"); h->fs_disabled = (na && na[0] == '1') ? 1 : 0; }
    { const char* na = getenv("FMPC_PRODUCT_MIN_BATCH"); h->fs_min_batch = (na && atoi(na) >= 1) ? atoi(na) : FMPC_PRODUCT_MIN_BATCH_DEFAULT; }
    { const char* nf = getenv("FMPC_NO_FIRST_MOVE"); h->fm_disabled = (nf && nf[0] == '1') ? 1 : 0; }
    h->inv_failed = 0; h->inv_failed_k = 0.0; h->inv_enabled = 0; h->inv_last = 0; h->inv_valid = 0; h->inv_jks = 0; h->inv_max_batch = 768; h->inv_k = 0.0; h->inv_jimg = nullptr; h->inv_nuc = nullptr; h->inv_eimg = nullptr; h->inv_jst = nullptr; h->inv_nucst = nullptr; h->inv_fuse = 0; h->inv_jimg2 = nullptr; h->inv_jks2 = 0; h->lp_v = nullptr; h->lp_cap = 0; h->lp_hint = 0;
    h->fa_nflag = nullptr;
    h->gn_list = nullptr; h->gn_nu = nullptr; h->gn_cnt = nullptr; h->gn_cnt_host = nullptr; h->gn_cap = 0;
    { const char* gs = getenv("FMPC_NO_GENERAL_SPLIT"); h->gn_split = (gs && gs[0] == '1') ? 0 : 1; }
    h->pn_cnt = nullptr; h->pn_cap = 0; h->pn_gate = nullptr; h->pn_epsp = nullptr; h->pn_nuws = nullptr; h->pn_rnp = nullptr; h->pn_list = nullptr; h->pn_cnt_host = nullptr; h->pn_dz_lds = 0; h->pn_rd2_0 = 0.0; h->pn_rp2c = 0.0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete h; return FMPC_E_HIP; }
    h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->wg_per_cu = lds ? (int)(FMPC_LDS_LIMIT / lds) : 1;
    if (h->wg_per_cu < 1) h->wg_per_cu = 1;
    if (h->wg_per_cu > 8) h->wg_per_cu = 8;
    if (generic_big && h->wg_per_cu > 2) h->wg_per_cu = 2;         // (a workspace slot holds the whole factor: nb 3 n (n + 1) doubles)

    const bool var2 = var_order == 2;
    const int nn = n * n;
    // ---- row-major copies
    std::vector<double> a1(nn), a2(nn, 0.0), a1t(nn), a2t(nn, 0.0), bt((size_t)m * n);
    for (int rr = 0; rr < n; ++rr)
        for (int c = 0; c < n; ++c) {
            a1[rr * n + c] = cm(A1, n, rr, c);
            a1t[c * n + rr] = a1[rr * n + c];
            if (var2) {
                a2[rr * n + c] = cm(A2, n, rr, c);
                a2t[c * n + rr] = a2[rr * n + c];
            }
        }
    for (int rr = 0; rr < n; ++rr)
        for (int c = 0; c < m; ++c) bt[(size_t)c * n + rr] = cm(B, n, rr, c);
    std::vector<double> R2(m), Q2(n), Qf2(n);
    for (int i = 0; i < m; ++i) R2[i] = 2.0 * cm(R, m, i, i);
    for (int i = 0; i < n; ++i) { Q2[i] = 2.0 * cm(Q, n, i, i); Qf2[i] = 2.0 * cm(Qf, n, i, i); }
    // X = (2Q)^-1, Xf = (2Qf)^-1 as dense row-major matrices (diagonal weights: diagonal matrices)
    std::vector<double> Q2m(nn, 0.0), Qf2m(nn, 0.0), X(nn, 0.0), Xf(nn, 0.0);
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) { Q2m[a * n + b] = 2.0 * cm(Q, n, a, b); Qf2m[a * n + b] = 2.0 * cm(Qf, n, a, b); }
    if (denseQ) {
        if (!spd_inverse(Q2m, n, X) || !spd_inverse(Qf2m, n, Xf)) { delete h; return FMPC_E_NOT_PD_PHI; }
    } else {
        for (int i = 0; i < n; ++i) { X[i * n + i] = 1.0 / Q2[i]; Xf[i * n + i] = 1.0 / Qf2[i]; }
    }
    // ---- iteration-invariant Y blocks, deduplicated
    //   Yd_i = X_{i+1} + [i>=1] A1 X_i A1' + [i>=2] A2 X_{i-1} A2'
    //   Y1_i = -X_{i+1} A1' + [i>=1] A1 X_i A2'      (rows i, i+1 < T)
    //   Y2_i = -X_{i+1} A2'                          (rows i, i+2 < T)
    //   xf:   Yd_T = Xf, Y1_{T-1} = Xf
    std::vector<std::vector<double>> blocks;
    std::vector<int> idxD, idx1, idx2;
    fmpc_host_y_blocks(n, T, var2, xf != nullptr, a1, a2, X, Xf, blocks, idxD, idx1, idx2);      // (fmpc_host.cpp)
    // ---- pack the pool
    std::vector<double> pool;
    auto push = [&](const double* p, size_t cnt) {
        size_t off = pool.size();
        pool.insert(pool.end(), p, p + cnt);
        while (pool.size() % 2) pool.push_back(0.0);   // keep 16-byte alignment
        return off;
    };
    auto push_opt = [&](const double* p, size_t cnt) {
        std::vector<double> z(cnt, 0.0);
        return push(p ? p : z.data(), cnt);
    };
    const size_t oA1 = push(a1.data(), nn), oA2 = push(a2.data(), nn);
    const size_t oA1t = push(a1t.data(), nn), oA2t = push(a2t.data(), nn);
    const size_t oBt = push(bt.data(), bt.size());
    const size_t oR2 = push(R2.data(), m), oQ2 = push(Q2.data(), n), oQf2 = push(Qf2.data(), n);
    const size_t orl = push_opt(r, m), oql = push_opt(q, n), oqfl = push_opt(qf, n);
    const size_t oumin = push(u_min, m), oumax = push(u_max, m);
    std::vector<double> umid(m), xmid(n);
    for (int i = 0; i < m; ++i) umid[i] = (u_min[i] + u_max[i]) / 2;    // fast_mpc_init.m:19-20
    for (int i = 0; i < n; ++i) xmid[i] = (x_min[i] + x_max[i]) / 2;
    const size_t oumid = push(umid.data(), m), oxmid = push(xmid.data(), n);
    const size_t oxf = push_opt(xf, n);
    // dense Q, Qf for the generic kernel's workspace instance: 2Q, 2Qf and their inverses
    const size_t oQ2m = push(Q2m.data(), nn), oQf2m = push(Qf2m.data(), nn), oXm = push(X.data(), nn), oXfm = push(Xf.data(), nn);
    const size_t oR2m = denseR ? push(R2full.data(), R2full.size()) : 0;
    // closed-loop prediction matrices (main.mlx, MPC_DesignMatrices): M1_0 = A1, M2_0 = A2, M1_1 = A1^2 + A2,
    // M2_1 = A1 A2, M1_i = A1 M1_{i-1} + A2 M1_{i-2}, M2_i = M1_{i-1} A2
    std::vector<double> lm1((size_t)T * nn, 0.0), lm2((size_t)T * nn, 0.0);
    {
        auto mul = [&](const double* X, const double* Y, double* Z, double beta) {     // Z = beta Z + X Y (n x n row-major)
            for (int rr = 0; rr < n; ++rr)
                for (int c = 0; c < n; ++c) {
                    double t = 0.0;
                    for (int qq = 0; qq < n; ++qq) t += X[rr * n + qq] * Y[qq * n + c];
                    Z[rr * n + c] = beta * Z[rr * n + c] + t;
                }
        };
        for (int i = 0; i < T; ++i) {
            double* m1 = lm1.data() + (size_t)i * nn; double* m2 = lm2.data() + (size_t)i * nn;
            if (i == 0) { for (int e = 0; e < nn; ++e) { m1[e] = a1[e]; m2[e] = a2[e]; } }
            else if (i == 1) { for (int e = 0; e < nn; ++e) m1[e] = a2[e]; mul(a1.data(), a1.data(), m1, 1.0); mul(a1.data(), a2.data(), m2, 0.0); }
            else {
                mul(a1.data(), m1 - nn, m1, 0.0); mul(a2.data(), m1 - 2 * nn, m1, 1.0);
                mul(m1 - nn, a2.data(), m2, 0.0);
            }
        }
    }
    const size_t oM1 = push(lm1.data(), lm1.size()), oM2 = push(lm2.data(), lm2.size());
    h->hm_m1 = lm1; h->hm_m2 = lm2;
    // blocks are indexed as Yblk + idx*n*n by the kernels: pack them without padding
    std::vector<double> yall;
    for (auto& bk : blocks) yall.insert(yall.end(), bk.begin(), bk.end());
    const size_t oY = push(yall.data(), yall.size());

    std::vector<int> ipool;
    ipool.insert(ipool.end(), idxD.begin(), idxD.end());
    ipool.insert(ipool.end(), idx1.begin(), idx1.end());
    ipool.insert(ipool.end(), idx2.begin(), idx2.end());

    if (hipMalloc((void**)&h->pool_d, pool.size() * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&h->pool_i, ipool.size() * sizeof(int)) != hipSuccess) {
        fmpc_destroy(h);
        return FMPC_E_ALLOC;
    }
    if (hipMemcpy(h->pool_d, pool.data(), pool.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->pool_i, ipool.data(), ipool.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        fmpc_destroy(h);
        return FMPC_E_HIP;
    }
    FmpcDevModel& D = h->dev;
    D.n = n; D.m = m; D.T = T; D.nb = h->nb; D.has_xf = h->has_xf; D.var2 = var2 ? 1 : 0;
    D.A1 = h->pool_d + oA1; D.A2 = h->pool_d + oA2; D.A1t = h->pool_d + oA1t; D.A2t = h->pool_d + oA2t;
    D.Bt = h->pool_d + oBt; D.R2 = h->pool_d + oR2; D.Q2 = h->pool_d + oQ2; D.Qf2 = h->pool_d + oQf2;
    D.rl = h->pool_d + orl; D.ql = h->pool_d + oql; D.qfl = h->pool_d + oqfl;
    D.umin = h->pool_d + oumin; D.umax = h->pool_d + oumax; D.umid = h->pool_d + oumid;
    D.xmid = h->pool_d + oxmid; D.xf = h->pool_d + oxf; D.Yblk = h->pool_d + oY;
    D.idxD = h->pool_i; D.idx1 = h->pool_i + h->nb; D.idx2 = h->pool_i + 2 * h->nb;
    D.denseR = denseR ? 1 : 0; D.R2m = denseR ? h->pool_d + oR2m : nullptr;
    D.denseQ = denseQ ? 1 : 0; D.Q2m = h->pool_d + oQ2m; D.Qf2m = h->pool_d + oQf2m; D.Xm = h->pool_d + oXm; D.Xfm = h->pool_d + oXfm;
    h->loop_M1 = h->pool_d + oM1; h->loop_M2 = h->pool_d + oM2;

    if (generic_ok && fmpc_generic_prepare(lds, generic_big ? 1 : 0) != hipSuccess) { fmpc_destroy(h); return FMPC_E_HIP; }
    // host copies the tiled kernel's images are built from (on first use of an arithmetic type)
    h->hm_blocks = yall; h->hm_idxD = idxD; h->hm_idx1 = idx1; h->hm_idx2 = idx2; h->hm_bt = bt;
    h->hm_a1f = a1; h->hm_a2f = a2;
    // (always kept: the cold-start form of the ramp path is built from them for any n, fmpc_ensure_ramp_cold)
    h->hm_R2 = R2; h->hm_rl.assign(m, 0.0); if (r) h->hm_rl.assign(r, r + m);
    h->hm_umin.assign(u_min, u_min + m); h->hm_umax.assign(u_max, u_max + m); h->hm_umid = umid; h->hm_xmid = xmid;
    h->hm_Q2 = Q2; h->hm_Qf2 = Qf2;
    h->hm_ql.assign(n, 0.0); if (q) h->hm_ql.assign(q, q + n);
    h->hm_qfl.assign(n, 0.0); if (qf) h->hm_qfl.assign(qf, qf + n);
    h->hm_xf.clear(); if (xf) h->hm_xf.assign(xf, xf + n);
    h->denseR = denseR ? 1 : 0; h->hm_r2full = R2full;
    h->denseQ = denseQ ? 1 : 0; h->hm_q2m = Q2m; h->hm_qf2m = Qf2m; h->hm_xm = X; h->hm_xfm = Xf;
    h->hm_b.resize((size_t)n * m);
    for (int rr = 0; rr < n; ++rr)
        for (int c = 0; c < m; ++c) h->hm_b[(size_t)rr * m + c] = cm(B, n, rr, c);

    // ---- wave kernel: MFMA-layout images of the constant blocks (+ a zero block), padded B'
    const char* force = getenv("FMPC_FORCE_GENERIC");
    const int mp = fmpc_wave_mp(m);
    if (!denseQ && !denseR && fmpc_wave_supports(n) && !(force && force[0] == '1') &&
        fmpc_wave_lds_bytes(n, mp) <= FMPC_LDS_LIMIT) {
        const int stride = fmpc_wave_img_stride(n);
        const int nblk = (int)blocks.size();
        std::vector<double> wpool((size_t)(nblk + 1) * stride, 0.0);      // last = zero block
        for (int k = 0; k < nblk; ++k) fmpc_wave_make_images(n, blocks[k].data(), wpool.data() + (size_t)k * stride);
        const size_t oBtP = wpool.size();
        wpool.resize(oBtP + (size_t)mp * 33, 0.0);
        for (int c = 0; c < m; ++c)
            for (int rr = 0; rr < n; ++rr) wpool[oBtP + (size_t)c * 33 + rr] = bt[(size_t)c * n + rr];
        std::vector<int> wi;
        for (int i = 0; i < h->nb; ++i) wi.push_back(idxD[i]);
        for (int i = 0; i < h->nb; ++i) wi.push_back(idx1[i] >= 0 ? idx1[i] : nblk);
        for (int i = 0; i < h->nb; ++i) wi.push_back(idx2[i] >= 0 ? idx2[i] : nblk);
        if (hipMalloc((void**)&h->wave_pool_d, wpool.size() * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&h->wave_pool_i, wi.size() * sizeof(int)) != hipSuccess) {
            fmpc_destroy(h);
            return FMPC_E_ALLOC;
        }
        if (hipMemcpy(h->wave_pool_d, wpool.data(), wpool.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->wave_pool_i, wi.data(), wi.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
            fmpc_destroy(h);
            return FMPC_E_HIP;
        }
        h->wave.mp = mp;
        h->wave.img = h->wave_pool_d;
        h->wave.BtP = h->wave_pool_d + oBtP;
        h->wave.iD = h->wave_pool_i;
        h->wave.i1 = h->wave_pool_i + h->nb;
        h->wave.i2 = h->wave_pool_i + 2 * h->nb;
        h->wave_lds = fmpc_wave_lds_bytes(n, mp);
        if (fmpc_wave_prepare(n, h->wave_lds) != hipSuccess) { fmpc_destroy(h); return FMPC_E_HIP; }
        h->use_wave = 1;
        const char* nosh = getenv("FMPC_NO_SHARED");
        if (!(nosh && nosh[0] == '1') && h->nb * n <= 840) {     // the cold path keeps rhs/y/d_nu in an 840-double LDS vector
            const size_t nf = fmpc_wave_shared_fac_doubles(n, h->nb);
            const size_t nscr = (size_t)T * (n + m) + 64 + (size_t)n;      // zero state + z of the export solve
            if (hipMalloc((void**)&h->sh_fac, nf * sizeof(double)) != hipSuccess ||
                hipMalloc((void**)&h->sh_rs, (size_t)h->nb * 32 * sizeof(double)) != hipSuccess ||
                hipMalloc((void**)&h->sh_ok, sizeof(int)) != hipSuccess ||
                hipMalloc((void**)&h->sh_scratch, nscr * sizeof(double)) != hipSuccess) {
                fmpc_destroy(h);
                return FMPC_E_ALLOC;
            }
            (void)hipMemset(h->sh_ok, 0, sizeof(int));
            (void)hipMemset(h->sh_scratch, 0, nscr * sizeof(double));
            int off9[14];
            fmpc_wave_cold_layout(n, mp, off9);
            if (hipMalloc((void**)&h->cold_d, (size_t)off9[8] * sizeof(double)) != hipSuccess) { fmpc_destroy(h); return FMPC_E_ALLOC; }
            h->hm_R2 = R2; h->hm_rl.assign(m, 0.0); if (r) h->hm_rl.assign(r, r + m);
            h->hm_umin.assign(u_min, u_min + m); h->hm_umax.assign(u_max, u_max + m);
            h->hm_umid = umid; h->hm_xmid = xmid; h->hm_bt = bt; h->hm_a1 = a1; h->hm_a2 = a2;
            h->sh_enabled = 1;
            // panel kernel: needs the panel + images in LDS
            const char* nopn = getenv("FMPC_NO_PANEL");
            const int pmp = (m + 15) & ~15;
            const size_t plds = fmpc_panel_lds_bytes(h->nb, pmp);
            if (n == FP_N && !(nopn && nopn[0] == '1') && plds <= FMPC_LDS_LIMIT) {
                FmpcPanelIn PL;                                     // pool layout: fmpc_host_panel_layout (fmpc_host.cpp)
                size_t o = 0;
                fmpc_host_panel_layout(n, m, T, h->nb, pmp, PL, &h->pn_o_dump, &o, &h->pn_dz_len);
                h->pn_o_simg = PL.o_simg; h->pn_limg_cap = PL.limg_cap; h->pn_o_limg = PL.o_limg; h->pn_o_bt = PL.o_bt;
                h->pn_o_aimg = PL.o_aimg; h->pn_o_vec = PL.o_vec; h->pn_o_ucon = PL.o_ucon; h->pn_o_dz = PL.o_dz;
                h->pn_doubles = PL.pool_doubles;
                if (hipMalloc((void**)&h->pn_pool, o * sizeof(double)) != hipSuccess ||
                    hipMalloc((void**)&h->pn_cnt, 2 * sizeof(int)) != hipSuccess || hipMalloc((void**)&h->fa_nflag, 4 * sizeof(int)) != hipSuccess ||
                    hipMalloc((void**)&h->pn_sched, (size_t)2 * FP_MAX_STEPS(h->nb) * FP_STEP_INTS * sizeof(int)) != hipSuccess) { fmpc_destroy(h); return FMPC_E_ALLOC; }
                (void)hipMemset(h->pn_cnt, 0, 2 * sizeof(int));     // [handed over, length of the continuation list]
                (void)hipMemset(h->fa_nflag, 0, 4 * sizeof(int));
                if (hipHostMalloc((void**)&h->pn_cnt_host, 2 * sizeof(int), hipHostMallocDefault) != hipSuccess) { fmpc_destroy(h); return FMPC_E_ALLOC; }
                h->pn_cnt_host[0] = -1; h->pn_cnt_host[1] = -1;   // nothing known yet
                h->pn_dz_lds = fmpc_dz_lds_bytes(pmp, 1);
                if (fmpc_panel_prepare(plds) != hipSuccess || fmpc_dz_prepare(pmp) != hipSuccess) { fmpc_destroy(h); return FMPC_E_HIP; }
                h->hm_Q2 = Q2; h->hm_Qf2 = Qf2; h->hm_blocks = yall; h->hm_idxD = idxD; h->hm_idx1 = idx1; h->hm_idx2 = idx2;
                h->hm_ql.assign(n, 0.0); if (q) h->hm_ql.assign(q, q + n);
                h->hm_qfl.assign(n, 0.0); if (qf) h->hm_qfl.assign(qf, qf + n);
                if (xf) h->hm_xf.assign(xf, xf + n);
                h->pn_mp = pmp; h->pn_lds = plds; h->pn_enabled = 1;
                {   // the dense form: FMPC_NO_INV=1 switches it off, FMPC_INV_MAX_BATCH bounds its use when w is given
                    const char* noinv = getenv("FMPC_NO_INV");
                    const char* mb = getenv("FMPC_INV_MAX_BATCH");
                    h->inv_enabled = !(noinv && noinv[0] == '1');
                    { const char* nf = getenv("FMPC_FUSE_DZ"); h->inv_fuse = nf && nf[0] == '1'; }    // (measured slower: DESIGN.md §7; off unless asked for)
                    if (mb && mb[0]) h->inv_max_batch = atoi(mb);
                    h->inv_jks = FP_XKS + (T * n + 3) / 4;
                    h->inv_jks2 = 2 * FP_XKS;                       // [x0 ; x0_pre ; 0 0 ; B u1 ; B u2 ; 0 0]
                    const size_t nrt = ((size_t)h->nb * n + 15) / 16;
                    if (h->inv_enabled && (hipMalloc((void**)&h->inv_jimg, nrt * (size_t)(16 * ((h->inv_jks + 15) / 16 + 3)) * 64 * sizeof(double)) != hipSuccess ||
                                           hipMalloc((void**)&h->inv_nuc, nrt * 16 * sizeof(double)) != hipSuccess ||
                                           hipMalloc((void**)&h->inv_eimg, 4 * FP_XKS * 64 * sizeof(double)) != hipSuccess ||
                                           hipMalloc((void**)&h->inv_jst, (size_t)h->nb * 2 * FP_XKS * 64 * sizeof(double)) != hipSuccess ||
                                           hipMalloc((void**)&h->inv_nucst, (size_t)h->nb * 32 * sizeof(double)) != hipSuccess ||
                                           hipMalloc((void**)&h->inv_jimg2, nrt * (size_t)(16 * ((2 * FP_XKS + 15) / 16 + 3)) * 64 * sizeof(double)) != hipSuccess)) { fmpc_destroy(h); return FMPC_E_ALLOC; }
                }
            }
        }
    }
    *out = h;
    return FMPC_OK;
}

extern "C" int fmpc_destroy(fmpc_handle h) {
    if (!h) return FMPC_E_NULL;
    (void)hipSetDevice(h->device);
    if (h->pool_d) (void)hipFree(h->pool_d);
    if (h->pool_i) (void)hipFree(h->pool_i);
    if (h->wave_pool_d) (void)hipFree(h->wave_pool_d);
    if (h->wave_pool_i) (void)hipFree(h->wave_pool_i);
    if (h->sh_fac) (void)hipFree(h->sh_fac);
    if (h->sh_rs) (void)hipFree(h->sh_rs);
    if (h->sh_ok) (void)hipFree(h->sh_ok);
    if (h->sh_scratch) (void)hipFree(h->sh_scratch);
    if (h->cold_d) (void)hipFree(h->cold_d);
    if (h->inv_jimg) (void)hipFree(h->inv_jimg);
    if (h->inv_nuc) (void)hipFree(h->inv_nuc);
    if (h->inv_eimg) (void)hipFree(h->inv_eimg);
    if (h->inv_jst) (void)hipFree(h->inv_jst);
    if (h->inv_nucst) (void)hipFree(h->inv_nucst);
    if (h->inv_jimg2) (void)hipFree(h->inv_jimg2);
    if (h->fm_pool) (void)hipFree(h->fm_pool);
    if (h->fm_need) (void)hipFree(h->fm_need);
    if (h->fa_need) (void)hipFree(h->fa_need);
    if (h->ao_scr) (void)hipFree(h->ao_scr);
    if (h->fm_compact) (void)hipFree(h->fm_compact);
    if (h->fm_walk_i) (void)hipFree(h->fm_walk_i);
    if (h->lp_v) (void)hipFree(h->lp_v);
    if (h->pn_pool) (void)hipFree(h->pn_pool);
    if (h->pn_cnt) (void)hipFree(h->pn_cnt);
    if (h->fa_nflag) (void)hipFree(h->fa_nflag);
    if (h->gn_list) (void)hipFree(h->gn_list);
    if (h->gn_nu) (void)hipFree(h->gn_nu);
    if (h->gn_cnt) (void)hipFree(h->gn_cnt);
    if (h->gn_cnt_host) (void)hipHostFree(h->gn_cnt_host);
    if (h->pn_sched) (void)hipFree(h->pn_sched);
    if (h->pn_gate) (void)hipFree(h->pn_gate);
    if (h->pn_epsp) (void)hipFree(h->pn_epsp);
    if (h->pn_nuws) (void)hipFree(h->pn_nuws);
    if (h->pn_rnp) (void)hipFree(h->pn_rnp);
    if (h->pn_list) (void)hipFree(h->pn_list);
    if (h->pn_cnt_host) (void)hipHostFree(h->pn_cnt_host);
    for (int t = 0; t < 2; ++t) {
        if (h->tl[t].pool) (void)hipFree(h->tl[t].pool);
        if (h->tl[t].ipool) (void)hipFree(h->tl[t].ipool);
        if (h->tl[t].bm) (void)hipFree(h->tl[t].bm);
    }
    if (h->tl_ws) (void)hipFree(h->tl_ws);
    if (h->ev) (void)hipEventDestroy(h->ev);
    if (h->ramp_du) (void)hipFree(h->ramp_du);
    if (h->ramp_ws) (void)hipFree(h->ramp_ws);
    if (h->rc_pool) (void)hipFree(h->rc_pool);
    if (h->rc_ws) (void)hipFree(h->rc_ws);
    if (h->rc_nu) (void)hipFree(h->rc_nu);
    if (h->rc_si) (void)hipFree(h->rc_si);
    if (h->ws) (void)hipFree(h->ws);
    if (h->zs) (void)hipFree(h->zs);
    if (h->stage) (void)hipFree(h->stage);
    if (h->pin) (void)hipHostFree(h->pin);
    delete h;
    return FMPC_OK;
}

extern "C" int fmpc_dims(fmpc_handle h, int* n, int* m, int* T, int* nz, int* nu_len) {
    if (!h) return FMPC_E_NULL;
    if (n) *n = h->n;
    if (m) *m = h->m;
    if (T) *T = h->T;
    if (nz) *nz = h->T * (h->n + h->m);
    if (nu_len) *nu_len = h->nb * h->n;
    return FMPC_OK;
}

// Cold-start constants for barrier weight k (see FwCold in fmpc_kernel_wave.hip): the first Newton
// step from u = ubar, x = xbar only needs these k-dependent vectors and the 27 x 27 matrix G.
static int fmpc_upload_cold(fmpc_handle h, double k, hipStream_t stream) {
    const int n = h->n, m = h->m, mp = h->wave.mp;
    int o[14];
    fmpc_wave_cold_layout(n, mp, o);
    std::vector<double> c((size_t)o[8], 0.0);
    for (int j = 0; j < m; ++j) {
        const double sp = h->hm_umax[j] - h->hm_umid[j], sm = h->hm_umid[j] - h->hm_umin[j];
        const double dp = 1.0 / sp, dm = 1.0 / sm;
        const double hc = k * (dp * dp + dm * dm);
        c[o[0] + j] = h->hm_R2[j] * h->hm_umid[j] + h->hm_rl[j] + k * (dp - dm);
        c[o[1] + j] = hc;
        c[o[2] + j] = 1.0 / (h->hm_R2[j] + hc);
    }
    const double* bt = h->hm_bt.data();                         // bt[c*n + r] = B[r][c]
    for (int a = 0; a < n; ++a) {
        double cbu = 0.0, bu = 0.0;
        for (int j = 0; j < m; ++j) {
            cbu += bt[(size_t)j * n + a] * c[o[2] + j] * c[o[0] + j];
            bu += bt[(size_t)j * n + a] * h->hm_umid[j];
        }
        c[o[4] + a] = cbu;
        double a1x = 0.0, a2x = 0.0;
        for (int q = 0; q < n; ++q) { a1x += h->hm_a1[a * n + q] * h->hm_xmid[q]; a2x += h->hm_a2[a * n + q] * h->hm_xmid[q]; }
        c[o[5] + a] = h->hm_xmid[a] - bu;
        c[o[6] + a] = h->hm_xmid[a] - bu - a1x;
        c[o[7] + a] = h->hm_xmid[a] - bu - a1x - a2x;
        double va = 0.0, va2 = 0.0;
        for (int j = 0; j < m; ++j) {
            const double aj = c[o[1] + j] * c[o[2] + j];
            va += bt[(size_t)j * n + a] * aj * c[o[0] + j];
            va2 += bt[(size_t)j * n + a] * aj * aj * c[o[0] + j];
        }
        c[o[11] + a] = va; c[o[12] + a] = va2;
        for (int b = 0; b < n; ++b) {
            double g = 0.0, ma = 0.0, ma2 = 0.0;
            for (int j = 0; j < m; ++j) {
                const double bb = bt[(size_t)j * n + a] * bt[(size_t)j * n + b];
                const double aj = c[o[1] + j] * c[o[2] + j];
                g += bb * c[o[2] + j]; ma += bb * aj; ma2 += bb * aj * aj;
            }
            c[o[3] + a * n + b] = g; c[o[9] + a * n + b] = ma; c[o[10] + a * n + b] = ma2;
        }
    }
    for (int j = 0; j < m; ++j) {
        const double aj = c[o[1] + j] * c[o[2] + j], cu = c[o[0] + j];
        c[o[13]] += aj * cu * cu; c[o[13] + 1] += aj * aj * cu * cu;
    }
    return hipMemcpyAsync(h->cold_d, c.data(), c.size() * sizeof(double), hipMemcpyHostToDevice, stream) == hipSuccess
               ? FMPC_OK : FMPC_E_HIP;
}

// Constants of the panel kernel for barrier weight k: built on the host (fmpc_host_build_panel in fmpc_host.cpp: twisted block
// factorisation of Y in long double, operator images, sweep schedules) and uploaded.  Called once per (handle, k).
static int fmpc_upload_panel(fmpc_handle h, double k, hipStream_t stream) {
    (void)stream;
    h->pn_valid = 0;
    FmpcPanelIn In;
    In.n = h->n; In.m = h->m; In.T = h->T; In.nb = h->nb; In.mp = h->pn_mp; In.var_order = h->var_order;
    In.pool_doubles = h->pn_doubles; In.o_simg = h->pn_o_simg; In.o_limg = h->pn_o_limg; In.o_bt = h->pn_o_bt;
    In.o_aimg = h->pn_o_aimg; In.o_vec = h->pn_o_vec; In.o_ucon = h->pn_o_ucon; In.o_dz = h->pn_o_dz; In.limg_cap = h->pn_limg_cap;
    In.umax = h->hm_umax.data(); In.umin = h->hm_umin.data(); In.umid = h->hm_umid.data(); In.xmid = h->hm_xmid.data();
    In.R2 = h->hm_R2.data(); In.rl = h->hm_rl.data(); In.Q2 = h->hm_Q2.data(); In.Qf2 = h->hm_Qf2.data();
    In.ql = h->hm_ql.data(); In.qfl = h->hm_qfl.data(); In.xf = h->hm_xf.data();
    In.bt = h->hm_bt.data(); In.a1 = h->hm_a1.data(); In.a2 = h->hm_a2.data(); In.blocks = h->hm_blocks.data();
    In.idxD = h->hm_idxD.data(); In.idx1 = h->hm_idx1.data(); In.idx2 = h->hm_idx2.data();
    FmpcPanelOut Out;
    fmpc_host_build_panel(In, k, Out);
    if (!Out.valid) return FMPC_OK;                                // not PD at the start point: the exact path reports it
    h->pn_nsf = Out.nsf; h->pn_nsb = Out.nsb; h->pn_rp2c = Out.rp2c; h->pn_rd2_0 = Out.rd2_0;
    if (hipMemcpy(h->pn_sched, Out.sched.data(), Out.sched.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return FMPC_E_HIP;
    if (hipMemcpy(h->pn_pool, Out.pool.data(), Out.pool.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return FMPC_E_HIP;
    h->pn_valid = 1;
    return FMPC_OK;
}

// Images of the tiled kernel for one arithmetic type (0: fp64, 1: fp32): the unique constant Y blocks and B' as
// zero-padded 16 x 16 tiles, block ids with the zero block substituted for "none".  Built on first use.
template <typename R>
static int fmpc_tiled_build(fmpc_handle h, int t) {
    fmpc_handle_s::Tiled& X = h->tl[t];
    const int n = h->n, m = h->m, nb = h->nb, nn = n * n;
    int NB = 0, NW = 0;
    if (!fmpc_tiled_supports(n, m, nb, t, &NB, &NW, h->denseR)) return FMPC_E_UNSUPPORTED;
    const int mb = (m + 15) / 16, NQ = NB * NB;
    const int nblk = (int)(h->hm_blocks.size() / nn);
    std::vector<R> img((size_t)(nblk + 1) * NQ * FT_TILE + (size_t)mb * NB * FT_TILE, (R)0);
    for (int k = 0; k < nblk; ++k)
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b)
                img[((size_t)k * NQ + (a / 16) * NB + b / 16) * FT_TILE + (a % 16) * 16 + b % 16] = (R)h->hm_blocks[(size_t)k * nn + a * n + b];
    const size_t obt = (size_t)(nblk + 1) * NQ * FT_TILE;
    for (int c = 0; c < m; ++c)
        for (int r = 0; r < n; ++r)
            img[obt + ((size_t)(c / 16) * NB + r / 16) * FT_TILE + (c % 16) * 16 + r % 16] = (R)h->hm_bt[(size_t)c * n + r];
    std::vector<int> ids;
    for (int i = 0; i < nb; ++i) ids.push_back(h->hm_idxD[i]);
    for (int i = 0; i < nb; ++i) ids.push_back(h->hm_idx1[i] >= 0 ? h->hm_idx1[i] : nblk);
    for (int i = 0; i < nb; ++i) ids.push_back(h->hm_idx2[i] >= 0 ? h->hm_idx2[i] : nblk);
    // zero-padded fp64 images of B', B, A1, A2, A1', A2' for the residual GEMMs
    const int NP = 16 * NB, MP = 16 * mb;
    const size_t oBt = 0, oBm = oBt + (size_t)MP * NP, oA1 = oBm + (size_t)NP * MP, oA2 = oA1 + (size_t)NP * NP,
                 oA1t = oA2 + (size_t)NP * NP, oA2t = oA1t + (size_t)NP * NP, oQ2 = oA2t + (size_t)NP * NP,
                 oQf2 = oQ2 + (size_t)NP * NP, oX = oQf2 + (size_t)NP * NP, oXf = oX + (size_t)NP * NP, oR2 = oXf + (size_t)NP * NP,
                 ptot = oR2 + (h->denseR ? (size_t)MP * MP : 0);
    std::vector<double> pad(ptot, 0.0);
    for (int c = 0; c < m; ++c)
        for (int r = 0; r < n; ++r) {
            pad[oBt + (size_t)c * NP + r] = h->hm_bt[(size_t)c * n + r];
            pad[oBm + (size_t)r * MP + c] = h->hm_bt[(size_t)c * n + r];
        }
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            pad[oA1 + (size_t)a * NP + b] = h->hm_a1f[(size_t)a * n + b]; pad[oA1t + (size_t)b * NP + a] = h->hm_a1f[(size_t)a * n + b];
            pad[oA2 + (size_t)a * NP + b] = h->hm_a2f[(size_t)a * n + b]; pad[oA2t + (size_t)b * NP + a] = h->hm_a2f[(size_t)a * n + b];
            pad[oQ2 + (size_t)a * NP + b] = h->hm_q2m[(size_t)a * n + b]; pad[oQf2 + (size_t)a * NP + b] = h->hm_qf2m[(size_t)a * n + b];
            pad[oX + (size_t)a * NP + b] = h->hm_xm[(size_t)a * n + b]; pad[oXf + (size_t)a * NP + b] = h->hm_xfm[(size_t)a * n + b];
        }
    if (h->denseR)
        for (int a = 0; a < m; ++a)
            for (int b = 0; b < m; ++b) pad[oR2 + (size_t)a * MP + b] = h->hm_r2full[(size_t)a * m + b];
    if (hipMalloc(&X.pool, img.size() * sizeof(R)) != hipSuccess ||
        hipMalloc((void**)&X.ipool, ids.size() * sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&X.bm, pad.size() * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
    if (hipMemcpy(X.pool, img.data(), img.size() * sizeof(R), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(X.ipool, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(X.bm, pad.data(), pad.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return FMPC_E_HIP;
    X.NB = NB; X.NW = NW;
    X.lds = fmpc_tiled_lds_bytes(NB, mb, NW, t, nb, h->denseR ? ft_pr_doubles(n, m) : 0);
    if (fmpc_tiled_prepare(n, NB, NW, t, X.lds, h->denseR) != hipSuccess) return FMPC_E_HIP;
    X.V.NB = NB; X.V.mb = mb; X.V.cn = n / 16; X.V.nl = n % 16; X.V.nblk = nblk;
    X.V.yimg = X.pool; X.V.btimg = (const R*)X.pool + obt;
    X.V.iD = X.ipool; X.V.i1 = X.ipool + nb; X.V.i2 = X.ipool + 2 * nb;
    X.V.BtP = X.bm + oBt; X.V.BmP = X.bm + oBm; X.V.A1P = X.bm + oA1; X.V.A2P = X.bm + oA2; X.V.A1tP = X.bm + oA1t; X.V.A2tP = X.bm + oA2t;
    X.V.denseR = h->denseR; X.V.R2P = h->denseR ? X.bm + oR2 : nullptr;
    X.V.denseQ = h->denseQ; X.V.Q2P = X.bm + oQ2; X.V.Qf2P = X.bm + oQf2; X.V.XP = X.bm + oX; X.V.XfP = X.bm + oXf;
    X.ready = 1;
    return FMPC_OK;
}

// launches the tiled kernel; caller holds h->mu
// nw_override: wavefronts per problem other than the handle's default (2 or 4, fp64 without dense R): few problems per CU
// are solved faster by more wavefronts each (the latency of a problem is what counts then)
static bool fmpc_capturing(hipStream_t stream);
__global__ void fmpc_zero_ints(int* p, int n) { if ((int)threadIdx.x < n) p[threadIdx.x] = 0; }
struct FmpcTiledPlan { int NWu; size_t ldsu; int grid; size_t slot; };
// Everything a tiled launch needs BEFORE anything is enqueued: the instance is built and prepared, the launch is sized and its
// workspace exists.  A caller that enqueues another launch first (the first-step / continuation split) plans first and falls
// back when this fails, so a failure never leaves a half-finished solve.  While `stream` is being captured into a
// graph nothing is allocated or released (that would end the capture): grid_hint is ignored and the launch is clamped to the
// slots the workspace already has -- the kernel walks its problems with a workgroup stride, any grid is correct.
static int fmpc_tiled_plan(fmpc_handle h, int t, int batch, hipStream_t stream, int nw_override, int grid_hint, FmpcTiledPlan* out) {
    fmpc_handle_s::Tiled& X = h->tl[t];
    if (!X.ready) {
        const int rc = t ? fmpc_tiled_build<float>(h, 1) : fmpc_tiled_build<double>(h, 0);
        if (rc != FMPC_OK) return rc;
    }
    int NWu = X.NW; size_t ldsu = X.lds;
    if (nw_override && nw_override != X.NW && !t && !h->denseR) {
        NWu = nw_override;
        ldsu = fmpc_tiled_lds_bytes(X.NB, X.V.mb, NWu, t, h->nb, 0);
        if (!(h->tl_prepared & (1 << NWu))) {
            if (ldsu > FMPC_LDS_LIMIT || fmpc_tiled_prepare(h->n, X.NB, NWu, t, ldsu, 0) != hipSuccess) { NWu = X.NW; ldsu = X.lds; }
            else h->tl_prepared |= 1 << NWu;
        }
    }
    int wgs = (int)(FMPC_LDS_LIMIT / ldsu);
    if (wgs < 1) wgs = 1;
    if (wgs * NWu > 8) wgs = 8 / NWu > 0 ? 8 / NWu : 1;                // two waves per SIMD (the kernel's launch bound)
    const int cap = h->num_cu * wgs;
    const bool capturing = fmpc_capturing(stream);
    int grid = batch < cap ? batch : cap;
    if (!capturing && grid_hint > 0 && grid_hint < grid) grid = grid_hint;   // (a list: as many workgroups as it is expected to be long)
    const FtWs L = ft_ws_layout(h->n, h->m, h->T, h->nb, X.NB, t ? 4 : 8, h->denseR);
    // one workspace slot per LAUNCHED workgroup (not per workgroup the chip could hold: a warm start of one problem or a
    // continuation list of 200 would otherwise allocate 0.5-1.4 GB per handle); grown geometrically up to the full grid,
    // so that a growing sequence of batch sizes reallocates (and synchronises) a logarithmic number of times
    const size_t need = L.total * (size_t)grid, full = L.total * (size_t)cap;
    if (need > h->tl_ws_doubles) {
        if (capturing) {
            const size_t have = h->tl_ws_doubles / L.total;
            if (have < 1) return FMPC_E_ALLOC;
            grid = (int)have;
        } else {
            size_t want = 2 * h->tl_ws_doubles;
            if (want < need) want = need;
            if (want > full) want = full;
            if (h->tl_ws) { (void)hipDeviceSynchronize(); (void)hipFree(h->tl_ws); h->tl_ws = nullptr; h->tl_ws_doubles = 0; }
            if (hipMalloc((void**)&h->tl_ws, want * sizeof(double)) != hipSuccess) {
                if (want == need || hipMalloc((void**)&h->tl_ws, need * sizeof(double)) != hipSuccess) { h->tl_ws = nullptr; return FMPC_E_ALLOC; }
                want = need;
            }
            h->tl_ws_doubles = want;
        }
    }
    out->NWu = NWu; out->ldsu = ldsu; out->grid = grid; out->slot = L.total;
    return FMPC_OK;
}

static int fmpc_solve_tiled(fmpc_handle h, int t, int batch, const double* x0, const double* x0_pre, const double* w,
                            const double* z_init, const double* nu0, int n_newton, double k, double* z_out,
                            double* nu_out, int* status, int* iters, double* step, double* u0_out, hipStream_t stream,
                            int nw_override = 0, const int* list = nullptr, const int* nlist = nullptr, const double* nuws = nullptr,
                            int grid_hint = 0, const FmpcTiledPlan* planned = nullptr) {
    FmpcTiledPlan plan;
    if (planned) plan = *planned;
    else { const int rc = fmpc_tiled_plan(h, t, batch, stream, nw_override, grid_hint, &plan); if (rc != FMPC_OK) return rc; }
    fmpc_handle_s::Tiled& X = h->tl[t];
    FtParams P;
    P.M = h->dev; P.V = X.V; P.batch = batch;
    P.x0 = x0; P.x0p = x0_pre; P.w = w; P.zinit = z_init; P.nu0 = nu0;
    P.max_iter = n_newton > 0 ? n_newton : 1000; P.kbar = k;
    P.zout = z_out; P.nuout = nu_out; P.status = status; P.iters = iters; P.step = step; P.step_ld = fmpc_step_ld(n_newton);
    P.ws = h->tl_ws; P.ws_stride = plan.slot; P.u0out = u0_out;
    P.list = list; P.nlist = nlist; P.nuws = nuws;
    h->tl_last_nw = plan.NWu;
    if (!list) h->last_path = t ? FMPC_PATH_TILED_F32 : FMPC_PATH_TILED;
    return fmpc_launch_tiled(P, X.NB, plan.NWu, t, plan.grid, plan.ldsu, stream) == hipSuccess ? FMPC_OK : FMPC_E_HIP;
}

extern "C" int fmpc_set_precision(fmpc_handle h, int mode) {
    if (!h) return FMPC_E_NULL;
    if (mode != FMPC_PREC_F64 && mode != FMPC_PREC_F32_MIXED) return FMPC_E_DIM;
    if (mode == FMPC_PREC_F32_MIXED && !fmpc_tiled_supports(h->n, h->m, h->nb, 1, nullptr, nullptr, h->denseR)) return FMPC_E_UNSUPPORTED;
    std::lock_guard<std::mutex> lk(h->mu);
    if (mode == FMPC_PREC_F64 && !h->generic_ok && !fmpc_tiled_supports(h->n, h->m, h->nb, 0, nullptr, nullptr, h->denseR)) {
        // no fp64 tiled instance (a dense R beyond n = 47, B too large for the fp64 tiles): fp64 on request through the generic
        // kernel with its tiles in the workspace (slow, exact)
        const size_t lds = fmpc_generic_big_lds_bytes(h->n, h->m);
        if (lds > FMPC_LDS_LIMIT) return FMPC_E_UNSUPPORTED;
        if (fmpc_generic_prepare(lds, 1) != hipSuccess) return FMPC_E_HIP;
        h->generic_ok = 1; h->generic_big = 1; h->lds_bytes = lds; h->wg_per_cu = 2;
    }
    h->prec = mode;
    return FMPC_OK;
}

static int fmpc_grid_for(fmpc_handle h, int batch) {
    int cap = h->num_cu * h->wg_per_cu;
    return batch < cap ? batch : cap;
}

// grows the per-workgroup workspace; caller holds h->mu
static int fmpc_ensure_ws(fmpc_handle h, int grid, size_t* stride) {
    const FmpcWsLayout L = fmpc_ws_layout(h->n, h->m, h->T, h->nb, h->generic_big != 0, h->denseR != 0);
    *stride = L.total;
    const size_t need = L.total * (size_t)grid;
    if (need > h->ws_doubles) {
        if (h->ws) { (void)hipDeviceSynchronize(); (void)hipFree(h->ws); h->ws = nullptr; h->ws_doubles = 0; }
        if (hipMalloc((void**)&h->ws, need * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
        h->ws_doubles = need;
    }
    return FMPC_OK;
}

// Orders the solves of a handle that arrive on different streams (they share the handle's workspaces): called with
// h->mu held, before the first launch of a solve and after its last one.
// (Under stream capture -- the solves of a recorded stretch going into a HIP graph, RecordedSolves in handle.py -- the handle's
// event is left alone: an event recorded inside a capture cannot be waited for by a stream outside it, and the order of a
// replay against other work is the order of the stream it is replayed on.)
static bool fmpc_capturing(hipStream_t stream) {            // (declared above fmpc_tiled_plan)
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(stream, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
}
static int fmpc_guard_begin(fmpc_handle h, hipStream_t stream) {
    if (fmpc_capturing(stream)) return FMPC_OK;
    if (!h->ev && hipEventCreateWithFlags(&h->ev, hipEventDisableTiming) != hipSuccess) return FMPC_E_HIP;
    if (h->ev_valid && stream != h->last_stream && hipStreamWaitEvent(stream, h->ev, 0) != hipSuccess) return FMPC_E_HIP;
    return FMPC_OK;
}
static void fmpc_guard_end(fmpc_handle h, hipStream_t stream) {
    if (fmpc_capturing(stream)) return;
    if (h->ev && hipEventRecord(h->ev, stream) == hipSuccess) { h->ev_valid = 1; h->last_stream = stream; }
}

// The parameter block of the panel-path kernels (constants of the handle; the per-call fields are the caller's).
static void fmpc_panel_params(fmpc_handle h, double k, FpParams& Q) {
    memset(&Q, 0, sizeof(Q));
    Q.m = h->m; Q.mp = h->pn_mp; Q.T = h->T; Q.nb = h->nb; Q.has_xf = h->has_xf; Q.var2 = h->var_order == 2 ? 1 : 0;
    Q.simg = h->pn_pool + h->pn_o_simg; Q.limg = h->pn_pool + h->pn_o_limg;
    Q.sched_f = h->pn_sched; Q.sched_b = h->pn_sched + (size_t)FP_MAX_STEPS(h->nb) * FP_STEP_INTS; Q.nsf = h->pn_nsf; Q.nsb = h->pn_nsb;
    Q.btimg = h->pn_pool + h->pn_o_bt; Q.aimg = h->pn_pool + h->pn_o_aimg;
    Q.vec = h->pn_pool + h->pn_o_vec; Q.ucon = h->pn_pool + h->pn_o_ucon;
    Q.rd2_0 = h->pn_rd2_0; Q.rp2c = h->pn_rp2c; Q.dump = h->pn_pool + h->pn_o_dump;
    Q.dzimg = h->pn_pool + h->pn_o_dz; Q.dzimg_len = h->pn_dz_len;
    Q.kbar = k;
    Q.jimg = h->inv_jimg; Q.nuc = h->inv_nuc; Q.jks = h->inv_jks; Q.jksp = 16 * ((h->inv_jks + 15) / 16 + 3); Q.eimg = h->inv_eimg;
    Q.jst = h->inv_jst; Q.nucst = h->inv_nucst; Q.gate_only = 0;
    Q.gw = nullptr; Q.gwn = h->T * h->n;                          // (set per call)
}

// J and nuc of the dense form (fmpc_kernel_inv.hip) for barrier weight k: the panel kernel itself solves for the unit
// vectors of d = [x0 ; x0_pre ; w] (scaled by 2^20: the constant part of the right-hand side then costs no digits), so
// both forms of the dual solve rest on one factorisation.  Once per (handle, k), ~5 ms; synchronises the stream.
static int fmpc_build_inverse(fmpc_handle h, double k, hipStream_t stream) {
    const int n = h->n, T = h->T, nb = h->nb, TN = T * n, nrow = nb * n;
    const int ncol = 2 * n + TN, np = 1 + ncol + 2 * n, npanels = (np + FP_NP - 1) / FP_NP;      // + w = -M1 e_j, -M2 e_j
    const double scale = 1048576.0;
    h->inv_valid = 0;
    std::vector<double> hx0((size_t)np * n, 0.0), hx0p((size_t)np * n, 0.0), hw((size_t)np * TN, 0.0);
    for (int j = 0; j < n; ++j) { hx0[(size_t)(1 + j) * n + j] = scale; hx0p[(size_t)(1 + n + j) * n + j] = scale; }
    for (int j = 0; j < TN; ++j) hw[(size_t)(1 + 2 * n + j) * TN + j] = scale;
    for (int j = 0; j < n; ++j)                                       // w = -M1 (B u1) - M2 (B u2): the columns of [B u1 ; B u2]
        for (int e = 0; e < TN; ++e) {
            hw[(size_t)(1 + ncol + j) * TN + e] = -scale * h->hm_m1[(size_t)e * n + j];
            hw[(size_t)(1 + ncol + n + j) * TN + e] = -scale * h->hm_m2[(size_t)e * n + j];
        }
    double* d = nullptr;
    const size_t o_x0 = 0, o_x0p = o_x0 + hx0.size(), o_w = o_x0p + hx0p.size(), o_gate = o_w + hw.size(),
                 o_nu = o_gate + 2 * (size_t)np, total = o_nu + (size_t)npanels * nrow * FP_NP;
    if (hipMalloc((void**)&d, total * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
    int rc = FMPC_OK;
    std::vector<double> nu((size_t)npanels * nrow * FP_NP);
    if (hipMemcpyAsync(d + o_x0, hx0.data(), hx0.size() * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipMemcpyAsync(d + o_x0p, hx0p.data(), hx0p.size() * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipMemcpyAsync(d + o_w, hw.data(), hw.size() * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess) rc = FMPC_E_HIP;
    if (rc == FMPC_OK) {
        FpParams Q;
        fmpc_panel_params(h, k, Q);
        Q.batch = np; Q.npanels = npanels; Q.step_ld = 1;
        Q.x0 = d + o_x0; Q.x0p = d + o_x0p; Q.w = d + o_w; Q.nu0 = nullptr;
        Q.nuws = d + o_nu; Q.gate = d + o_gate; Q.handed = nullptr;
        const int pgrid = npanels < h->num_cu ? npanels : h->num_cu;
        if (fmpc_launch_panel(Q, pgrid, fmpc_panel_lds_used(h->nb, h->pn_mp, h->pn_nsf + h->pn_nsb), stream) != hipSuccess ||
            hipMemcpyAsync(nu.data(), d + o_nu, nu.size() * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) rc = FMPC_E_HIP;
    }
    (void)hipFree(d);
    if (rc != FMPC_OK) return rc;
    auto at = [&](int p, int row) { return nu[((size_t)(p / FP_NP) * nrow + row) * FP_NP + (p % FP_NP)]; };
    const int nrt = (nrow + 15) / 16, jks = h->inv_jks, jksp = 16 * ((jks + 15) / 16 + 3);
    std::vector<double> img((size_t)nrt * jksp * 64, 0.0), nuc((size_t)nrt * 16, 0.0);
    for (int r = 0; r < nrow; ++r) {
        nuc[r] = at(0, r);
        if (!std::isfinite(nuc[r])) return FMPC_OK;                  // the panel path reports what is wrong; this form stays off
    }
    const bool var2 = h->var_order == 2;
    for (int rt = 0; rt < nrt; ++rt)
        for (int ks = 0; ks < jks; ++ks)
            for (int l = 0; l < 64; ++l) {
                const int row = 16 * rt + (l & 15), kk = 4 * ks + (l >> 4);
                int col = -1;                                        // column of d -> index of the unit-vector problem
                if (kk < n) col = kk;
                else if (kk < 2 * n) col = var2 ? kk : -1;           // VAR(1): x0_pre does not enter
                else if (kk >= 4 * FP_XKS && kk - 4 * FP_XKS < TN) col = 2 * n + (kk - 4 * FP_XKS);
                if (row < nrow && col >= 0) {
                    const double v = (at(1 + col, row) - nuc[row]) / scale;
                    if (!std::isfinite(v)) return FMPC_OK;
                    img[((size_t)rt * jksp + ks) * 64 + l] = v;
                }
            }
    // per stage: the rows of J_x as two 16-row A tiles (rows beyond n: zero) and nuc padded to 32 (d_z with the dual solve fused in)
    {
        std::vector<double> jst((size_t)nb * 2 * FP_XKS * 64, 0.0), ncs((size_t)nb * 32, 0.0);
        for (int st = 0; st < nb; ++st) {
            for (int r = 0; r < n; ++r) ncs[(size_t)st * 32 + r] = nuc[(size_t)st * n + r];
            for (int I = 0; I < 2; ++I)
                for (int ks = 0; ks < FP_XKS; ++ks)
                    for (int l = 0; l < 64; ++l) {
                        const int rl = 16 * I + (l & 15), kk = 4 * ks + (l >> 4);
                        if (rl >= n) continue;
                        const int row = st * n + rl;
                        jst[(((size_t)st * 2 + I) * FP_XKS + ks) * 64 + l] = img[((size_t)(row >> 4) * jksp + ks) * 64 + ((kk & 3) << 4) + (row & 15)];
                    }
        }
        if (hipMemcpy(h->inv_jst, jst.data(), jst.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->inv_nucst, ncs.data(), ncs.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return FMPC_E_HIP;
    }
    // J' = [J_x | d nu+ / d (B u1) | d nu+ / d (B u2)]
    const int jks2 = h->inv_jks2, jksp2 = 16 * ((jks2 + 15) / 16 + 3);
    std::vector<double> img2((size_t)nrt * jksp2 * 64, 0.0);
    for (int rt = 0; rt < nrt; ++rt)
        for (int ks = 0; ks < jks2; ++ks)
            for (int l = 0; l < 64; ++l) {
                const int row = 16 * rt + (l & 15), kk = 4 * ks + (l >> 4);
                if (ks < FP_XKS) { img2[((size_t)rt * jksp2 + ks) * 64 + l] = img[((size_t)rt * jksp + ks) * 64 + l]; continue; }
                const int c = kk - 4 * FP_XKS;
                if (row < nrow && c < 2 * n) {
                    const double v = (at(1 + ncol + c, row) - nuc[row]) / scale;
                    if (!std::isfinite(v)) return FMPC_OK;
                    img2[((size_t)rt * jksp2 + ks) * 64 + l] = v;
                }
            }
    if (hipMemcpy(h->inv_jimg2, img2.data(), img2.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return FMPC_E_HIP;
    // E = [A1 A2 ; A2 0] (fast_mpc_eq_const.m:39-44), rows 0..53, columns [x0 ; x0_pre]
    std::vector<double> eimg((size_t)4 * FP_XKS * 64, 0.0);
    for (int I = 0; I < 4; ++I)
        for (int ks = 0; ks < FP_XKS; ++ks)
            for (int l = 0; l < 64; ++l) {
                const int row = 16 * I + (l & 15), kk = 4 * ks + (l >> 4);
                double v = 0.0;
                if (row < n && kk < n) v = h->hm_a1[row * n + kk];
                else if (row < n && kk < 2 * n) v = var2 ? h->hm_a2[row * n + kk - n] : 0.0;
                else if (row < 2 * n && kk < n) v = var2 ? h->hm_a2[(row - n) * n + kk] : 0.0;
                eimg[((size_t)I * FP_XKS + ks) * 64 + l] = v;
            }
    if (hipMemcpy(h->inv_eimg, eimg.data(), eimg.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->inv_jimg, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->inv_nuc, nuc.data(), nuc.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return FMPC_E_HIP;
    // host copy for the first-move form: J with the columns [x0 | x0_pre | B u1 | B u2], nuc
    {
        const int nc = 4 * n;
        h->hm_J4.assign((size_t)nrow * nc, 0.0); h->hm_nuc.assign(nuc.begin(), nuc.begin() + nrow);
        for (int r = 0; r < nrow; ++r)
            for (int c = 0; c < nc; ++c) {
                int pidx = -1;
                if (c < n) pidx = 1 + c;
                else if (c < 2 * n) pidx = var2 ? 1 + c : -1;
                else pidx = 1 + ncol + (c - 2 * n);
                if (pidx >= 0) h->hm_J4[(size_t)r * nc + c] = (at(pidx, r) - nuc[r]) / scale;
            }
        h->hm_J4_valid = 1; h->hm_J4_k = k; h->fm_valid = 0;
    }
    h->inv_valid = 1; h->inv_k = k;
    return FMPC_OK;
}

// Matrices of the first-move form for barrier weight k (fmpc_host_build_first_move), uploaded as one pool.
static int fmpc_build_first_move(fmpc_handle h, double k) {
    h->fm_valid = 0;
    if (!h->hm_J4_valid || h->hm_J4_k != k) return FMPC_E_UNSUPPORTED;
    const int n = h->n, m = h->m, T = h->T, TN = T * n;
    FmpcFirstIn In;
    In.n = n; In.m = m; In.T = T; In.nb = h->nb; In.var2 = h->var_order == 2 ? 1 : 0; In.has_xf = h->has_xf;
    In.bt = h->hm_bt.data(); In.umax = h->hm_umax.data(); In.umin = h->hm_umin.data(); In.umid = h->hm_umid.data();
    In.xmid = h->hm_xmid.data(); In.R2 = h->hm_R2.data(); In.rl = h->hm_rl.data(); In.a1 = h->hm_a1.data(); In.a2 = h->hm_a2.data();
    In.m1 = h->hm_m1.data(); In.m2 = h->hm_m2.data(); In.xf = h->hm_xf.empty() ? h->hm_xmid.data() : h->hm_xf.data();
    In.J = h->hm_J4.data(); In.nuc = h->hm_nuc.data(); In.k = k;
    FmpcFirstOut O;
    fmpc_host_build_first_move(In, O);
    for (double v : O.K0t) if (!std::isfinite(v)) return FMPC_E_UNSUPPORTED;
    for (double v : O.E) if (!std::isfinite(v)) return FMPC_E_UNSUPPORTED;
    std::vector<double> pool;
    auto push = [&](const std::vector<double>& v) { const size_t o = pool.size(); pool.insert(pool.end(), v.begin(), v.end()); while (pool.size() % 2) pool.push_back(0.0); return o; };
    std::vector<double> m12t((size_t)2 * n * TN), dx0T(n);
    for (int e = 0; e < TN; ++e)
        for (int q = 0; q < n; ++q) { m12t[(size_t)q * TN + e] = h->hm_m1[(size_t)e * n + q]; m12t[(size_t)(n + q) * TN + e] = h->hm_m2[(size_t)e * n + q]; }
    for (int r = 0; r < n; ++r) dx0T[r] = h->hm_Qf2[r] * h->hm_xmid[r] + h->hm_qfl[r];
    const size_t oK = push(O.K0t), ou = push(O.u0c), oE = push(O.Ec), oe = push(O.e), oEp = push(O.Epc), oep = push(O.ep), om = push(m12t), od = push(dx0T);
    // affine form of the whole step without w: [Kz | zc] as matrix-core images; the (x0, x0_pre) blocks of E, Ep likewise
    h->fa_valid = 0;
    FmpcAffineOut AO;
    size_t oA = 0, oAE = 0, oAEp = 0, oAl = 0, oAlp = 0, oAd = 0;
    bool fa_ok = !h->fa_disabled && 2 * n + 2 <= FA_KC;
    if (fa_ok) {
        FmpcAffineIn AI;
        AI.n = n; AI.m = m; AI.T = T; AI.nb = h->nb; AI.has_xf = h->has_xf; AI.ncJ = 4 * n;
        AI.bt = In.bt; AI.umax = In.umax; AI.umin = In.umin; AI.umid = In.umid; AI.xmid = In.xmid; AI.R2 = In.R2; AI.rl = In.rl;
        AI.Q2 = h->hm_Q2.data(); AI.Qf2 = h->hm_Qf2.data(); AI.ql = h->hm_ql.data(); AI.qfl = h->hm_qfl.data(); AI.a1 = In.a1; AI.a2 = In.a2;
        AI.J = In.J; AI.nuc = In.nuc; AI.k = k;
        fmpc_host_build_affine(AI, AO);
        for (double v : AO.img) if (!std::isfinite(v)) { fa_ok = false; break; }
    }
    if (fa_ok) {
        const int nc = 4 * n, nd = 2 * n;
        std::vector<double> E64((size_t)64 * FA_KC, 0.0), Ep64((size_t)64 * FA_KC, 0.0), el(64, 0.0), epl(64, 0.0), imgE, imgEp;
        for (int r = 0; r < nd; ++r) {
            for (int c = 0; c < nd; ++c) { E64[(size_t)r * FA_KC + c] = O.E[(size_t)r * nc + c]; Ep64[(size_t)r * FA_KC + c] = O.Ep[(size_t)r * nc + c]; }
            el[r] = 2.0 * O.e[r]; epl[r] = -2.0 * O.ep[r];
        }
        fmpc_host_mfma_a_images(E64.data(), 64, imgE);
        fmpc_host_mfma_a_images(Ep64.data(), 64, imgEp);
        oA = push(AO.img); oAE = push(imgE); oAEp = push(imgEp); oAl = push(el); oAlp = push(epl);
        oAd = push(std::vector<double>(4096, 0.0));
    }
    // the first-move form as a product over many realisations (fmpc_kernel_loopu0.hip): [K0 | u0c], E, Ep as matrix-core images
    // over d -- in the order [x0; x0_pre; B u1; B u2; 1] behind the loop-input kernel, in blocks of 28 with the constant in
    // column 111 for the one-launch step (which also takes B as operand images); built by fmpc_host_build_loop_images
    h->fl_valid = 0; h->fs_valid = 0;
    size_t oLU = 0, oLE = 0, oLEp = 0, oSU = 0, oSE = 0, oSEp = 0, oSB = 0;
    const bool fl_ok = !h->fl_disabled && 4 * n + 1 <= 4 * FL_KS;
    const bool fs_ok = fl_ok && !h->fs_disabled && n == 27 && m <= 144;
    if (fl_ok) {
        FmpcLoopImages LI;
        fmpc_host_build_loop_images(O, n, m, FL_KS, false, nullptr, LI);
        oLU = push(LI.imgU); oLE = push(LI.imgE); oLEp = push(LI.imgEp);
    }
    if (fs_ok) {
        FmpcLoopImages LI;
        fmpc_host_build_loop_images(O, n, m, FL_KS, true, In.bt, LI);
        oSU = push(LI.imgU); oSE = push(LI.imgE); oSEp = push(LI.imgEp); oSB = push(LI.imgB);
    }
    if (h->fm_pool) { (void)hipDeviceSynchronize(); (void)hipFree(h->fm_pool); h->fm_pool = nullptr; }
    if (hipMalloc((void**)&h->fm_pool, pool.size() * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
    if (hipMemcpy(h->fm_pool, pool.data(), pool.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return FMPC_E_HIP;
    FmParams& P = h->fm_P;
    memset(&P, 0, sizeof(P));
    P.n = n; P.m = m; P.T = T; P.nb = h->nb; P.var2 = In.var2; P.has_xf = h->has_xf;
    P.bt = h->dev.Bt; P.K0t = h->fm_pool + oK; P.u0c = h->fm_pool + ou; P.E = h->fm_pool + oE; P.e = h->fm_pool + oe;
    P.Ep = h->fm_pool + oEp; P.ep = h->fm_pool + oep; P.m12t = h->fm_pool + om; P.dx0T = h->fm_pool + od;
    P.e0 = O.e0; P.ep0 = O.ep0; P.normE = O.normE; P.norme = O.norme; P.normEp = O.normEp; P.normep = O.normep; P.rd2_0 = h->pn_rd2_0;
    h->fm_valid = 1; h->fm_k = k;
    if (fa_ok) {
        FaParams& A = h->fa_P;
        memset(&A, 0, sizeof(A));
        A.n = n; A.m = m; A.T = T; A.nb = h->nb; A.has_xf = h->has_xf; A.rows = AO.rows; A.tiles = AO.tiles; A.nu_rows = AO.nu_rows; A.nu_tiles = AO.nu_tiles;
        A.img = h->fm_pool + oA; A.imgE = h->fm_pool + oAE; A.imgEp = h->fm_pool + oAEp; A.elin = h->fm_pool + oAl; A.eplin = h->fm_pool + oAlp; A.dump = h->fm_pool + oAd;
        A.dx0T = P.dx0T; A.e0 = O.e0; A.ep0 = O.ep0; A.normE = O.normE; A.norme = O.norme; A.normEp = O.normEp; A.normep = O.normep; A.rd2_0 = h->pn_rd2_0;
        h->fa_valid = 1;
    }
    if (fl_ok) {
        FlParams& L = h->fl_P;
        memset(&L, 0, sizeof(L));
        L.n = n; L.m = m; L.T = T; L.nb = h->nb; L.has_xf = h->has_xf; L.var2 = In.var2;
        L.imgU = h->fm_pool + oLU; L.imgE = h->fm_pool + oLE; L.imgEp = h->fm_pool + oLEp;
        L.dx0T = P.dx0T; L.e0 = O.e0; L.ep0 = O.ep0; L.normE = O.normE; L.norme = O.norme; L.normEp = O.normEp; L.normep = O.normep; L.rd2_0 = h->pn_rd2_0;
        h->fl_valid = 1;
        if (fs_ok) {
            h->fs_P = L;
            h->fs_P.imgU = h->fm_pool + oSU; h->fs_P.imgE = h->fm_pool + oSE; h->fs_P.imgEp = h->fm_pool + oSEp;
            memset(&h->fs_I, 0, sizeof(h->fs_I));
            const int Tn = T * n;
            h->fs_I.rs = (Tn + 63) / 64; h->fs_I.rows = (Tn + h->fs_I.rs - 1) / h->fs_I.rs;
            h->fs_I.imgB = h->fm_pool + oSB; h->fs_I.M1 = h->loop_M1; h->fs_I.M2 = h->loop_M2;
            h->fs_valid = 1;
        }
    }
    return FMPC_OK;
}

// The wave kernel's workspace: one slot per resident wavefront of the whole chip.  Caller holds h->mu.
static int fmpc_ensure_wave_ws(fmpc_handle h, size_t* stride_out) {
    const int wpw = fmpc_wave_waves_per_wg();
    const size_t stride = fmpc_wave_ws_doubles(h->n, h->m, h->wave.mp, h->T, h->nb);
    const size_t need = stride * (size_t)h->num_cu * wpw;
    if (need > h->ws_doubles) {
        if (h->ws) { (void)hipDeviceSynchronize(); (void)hipFree(h->ws); h->ws = nullptr; h->ws_doubles = 0; }
        if (hipMalloc((void**)&h->ws, need * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
        h->ws_doubles = need;
    }
    *stride_out = stride;
    return FMPC_OK;
}

// Cold-start constants of barrier weight k: the shared factor (an export launch of the wave kernel on one problem), the
// k-dependent constants of the exact path and of the panel path.  Once per (handle, k).  Caller holds h->mu.
static int fmpc_ensure_cold(fmpc_handle h, double k, size_t stride, hipStream_t stream) {
    if (h->sh_valid && h->sh_k == k) return FMPC_OK;
    // earlier solves (ordered before this point on `stream` by the guard) may still read the constants of
    // the previous k, which the blocking uploads below overwrite
    if (h->sh_valid && hipStreamSynchronize(stream) != hipSuccess) return FMPC_E_HIP;
    double* scr = h->sh_scratch;                 // a zero state: x0 = x0_pre = 0, w = nu0 = 0
    const hipError_t e = fmpc_launch_wave(h->dev, h->wave, 1, 1, scr, scr, nullptr, nullptr, nullptr, 1, k,
                                          scr + ((h->n + 15) & ~15), nullptr, nullptr, nullptr, nullptr, 1, h->ws, stride,
                                          h->wave_lds, stream, 2, h->sh_fac, h->sh_rs, h->sh_ok, h->cold_d);
    if (e != hipSuccess) return FMPC_E_HIP;
    if (fmpc_upload_cold(h, k, stream) != FMPC_OK) return FMPC_E_HIP;
    if (h->pn_enabled && fmpc_upload_panel(h, k, stream) != FMPC_OK) return FMPC_E_HIP;
    h->sh_valid = 1; h->sh_k = k;
    return FMPC_OK;
}

// set by the host-pointer entry around its device solve: its staging block has contiguous rows whatever fmpc_set_z_ld says
static thread_local int fmpc_tl_contiguous_z = 0;
// set by fmpc_solve_u0_device_ld around its solve: the row distance of THIS call's z_out (>= 0), whatever the handle's persistent
// value says (-1: none given, the handle's fmpc_set_z_ld value applies).  Thread-local, so a concurrent solve on the same handle
// from another thread never inherits it (the set / solve / restore of fmpc_set_z_ld was not atomic).
static thread_local int fmpc_tl_zld_explicit = -1;
static inline int fmpc_effective_zld(fmpc_handle h) { return fmpc_tl_zld_explicit >= 0 ? fmpc_tl_zld_explicit : h->z_ld; }

// fmpc_solve_device (u0_out == NULL) / fmpc_solve_u0_device: the first moves are written by the solve's last kernel
// where that kernel visits every problem anyway (the wave kernel), by the unpack kernel otherwise
static int fmpc_solve_device_inner(fmpc_handle h, int batch,
                                  const double* x0, const double* x0_pre, const double* w,
                                  const double* z_init, const double* nu0,
                                  int n_newton, double k,
                                  double* z_out, double* nu_out, int* status, int* iters, double* step,
                                  double* u0_out, void* stream) {
    if (!h || !x0 || (!z_out && !u0_out)) return FMPC_E_NULL;
    if (batch < 0) return FMPC_E_DIM;
    if (batch == 0) return FMPC_OK;
    const int max_iter = n_newton > 0 ? n_newton : 1000;
    hipError_t e;
    // Output options (README.md:558-570,589: the loop uses U(1:nu) only): z_out == NULL with u0_out given returns the first
    // moves alone.  The kernels that iterate on z then work in a scratch array of the handle; the cold-start panel path
    // with a budget of 1 does not write z at all (fmpc_cold_dz<.., .., true>).
    const bool z_null = z_out == nullptr;
    if (z_null) {
        const size_t need = (size_t)batch * h->T * (h->n + h->m);
        if (need > h->zs_doubles) {
            if (h->zs) { (void)hipDeviceSynchronize(); (void)hipFree(h->zs); h->zs = nullptr; h->zs_doubles = 0; }
            if (hipMalloc((void**)&h->zs, need * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
            h->zs_doubles = need;
        }
        z_out = h->zs;
    }
    // Padded z rows (fmpc_set_z_ld): the affine form of the cold-start step and its exact path honour them, nothing else does
    const int zld = (!z_null && !fmpc_tl_contiguous_z && fmpc_effective_zld(h) > h->T * (h->n + h->m)) ? fmpc_effective_zld(h) : 0;
    if (zld && (w != nullptr || z_init != nullptr || max_iter != 1 || h->prec == FMPC_PREC_F32_MIXED || h->force_tiled || h->denseQ ||
                h->denseR || !h->use_wave || !h->sh_enabled || !h->pn_enabled)) return FMPC_E_UNSUPPORTED;
    if (h->prec == FMPC_PREC_F32_MIXED || h->force_tiled || ((h->denseQ || h->denseR) && !h->generic_big) ||
        (!h->use_wave && (!h->generic_ok || h->prefer_tiled)))
    {
        // fp64, one or two blocks of 16 (n <= 31), few problems per CU: four wavefronts per problem instead of two -- 7 to 24 % less
        // time up to 256 problems, 20 to 60 % MORE at 2048 (round 5 sweep, n = 8 .. 31).  (n = 27 has its own dispatch below; a size
        // that reaches this point with the wave kernel switched off keeps the handle's setting.)
        const int t_ = h->prec == FMPC_PREC_F32_MIXED ? 1 : 0;
        const int few = (!t_ && !h->denseR && h->n <= 31 && h->n != FP_N && batch <= 512 && !h->force_tiled) ? 4 : 0;
        return fmpc_solve_tiled(h, t_, batch, x0, x0_pre, w, z_init, nu0, n_newton, k,
                                z_out, nu_out, status, iters, step, u0_out, (hipStream_t)stream, few);
    }
    if (h->use_wave) {
        // one wavefront per problem, 8 per workgroup, one workgroup per CU
        const int wpw = fmpc_wave_waves_per_wg();
        int grid = (batch + wpw - 1) / wpw;
        if (grid > h->num_cu) grid = h->num_cu;
        size_t stride = 0;
        { const int rcw = fmpc_ensure_wave_ws(h, &stride); if (rcw != FMPC_OK) return rcw; }
        int mode = 0;
        if (h->sh_enabled && z_init == nullptr) {
            // cold start: the first Newton step of every problem shares one factor (depends on k only)
            const int rce = fmpc_ensure_cold(h, k, stride, (hipStream_t)stream);
            if (rce != FMPC_OK) return rce;
            mode = 1;
        }
        if (mode == 1 && h->pn_enabled && h->pn_valid) {
            // first Newton step from the cold start (the reference's own call has just this one): 16-problem panels
            // on the matrix cores; the exact-path launch below decides every problem's step length, redoes the
            // problems whose decision is not clear-cut and runs the remaining iterations of the budget
            const int npanels = (batch + FP_NP - 1) / FP_NP;
            if ((size_t)batch > h->pn_cap) {
                (void)hipDeviceSynchronize();
                if (h->pn_gate) (void)hipFree(h->pn_gate);
                if (h->pn_epsp) (void)hipFree(h->pn_epsp);
                if (h->pn_nuws) (void)hipFree(h->pn_nuws);
                if (h->pn_rnp) (void)hipFree(h->pn_rnp);
                if (h->pn_list) (void)hipFree(h->pn_list);
                h->pn_gate = nullptr; h->pn_epsp = nullptr; h->pn_nuws = nullptr; h->pn_rnp = nullptr; h->pn_list = nullptr; h->pn_cap = 0;
                if (hipMalloc((void**)&h->pn_gate, (size_t)batch * 2 * sizeof(double)) != hipSuccess ||
                    hipMalloc((void**)&h->pn_epsp, (size_t)npanels * h->T * FP_NP * sizeof(double)) != hipSuccess ||
                    hipMalloc((void**)&h->pn_rnp, (size_t)npanels * h->T * FP_NP * sizeof(double)) != hipSuccess ||
                    hipMalloc((void**)&h->pn_list, (size_t)batch * sizeof(int)) != hipSuccess ||
                    hipMalloc((void**)&h->pn_nuws, (size_t)npanels * FP_NP * h->nb * h->n * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
                h->pn_cap = batch;
            }
            FpParams Q;
            fmpc_panel_params(h, k, Q);
            Q.batch = batch; Q.npanels = npanels; Q.step_ld = fmpc_step_ld(n_newton);
            Q.x0 = x0; Q.x0p = x0_pre; Q.w = w; Q.nu0 = nu0;
            Q.zout = z_out; Q.status = status; Q.iters = iters; Q.step = step;
            Q.nuws = h->pn_nuws; Q.nuout = nu_out;
            Q.gate = h->pn_gate; Q.epsp = h->pn_epsp; Q.handed = h->pn_cnt;
            Q.rnp = h->pn_rnp;
            // The dual solve in its dense form (no dependency chain, fmpc_kernel_inv.hip): always without w (56 columns),
            // with w while the product is cheaper than the two sweeps of a panel (few panels leave the chip empty)
            // (a closed-loop step hands over [B u1 ; B u2], the 2 n numbers its w depends on: 28 k-steps at any batch)
            const bool lowrank = h->lp_hint && w != nullptr && h->lp_v != nullptr;
            const bool dense_form = h->inv_enabled && (w == nullptr || lowrank || batch <= h->inv_max_batch);
            if (dense_form && (!h->inv_valid || h->inv_k != k) && !(h->inv_failed && h->inv_failed_k == k)) {
                if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return FMPC_E_HIP;   // earlier solves may still read J
                const int rcb = fmpc_build_inverse(h, k, (hipStream_t)stream);
                if (rcb != FMPC_OK) return rcb;
                // J or nuc not finite: the sweep form is taken, and the build (a stream synchronisation, allocations and a
                // 5 ms launch) is not retried on every later solve with this k
                h->inv_failed = h->inv_valid ? 0 : 1; h->inv_failed_k = k;
            }
            // Without w, with a budget of 1 and no nu requested the whole step is ONE product on the matrix cores (affine form,
            // fmpc_kernel_affine.hip) + the exact path in flag mode for the problems whose decision is not clear-cut
            if (dense_form && h->inv_valid && w == nullptr && max_iter == 1 && (nu_out == nullptr || !z_null) && !h->fa_disabled && h->n == FP_N) {
                if (!h->fm_valid || h->fm_k != k) {
                    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return FMPC_E_HIP;
                    const int rcf = fmpc_build_first_move(h, k);
                    if (rcf != FMPC_OK && rcf != FMPC_E_UNSUPPORTED) return rcf;
                }
                if (h->fm_valid && h->fm_k == k && h->fa_valid) {
                    if ((size_t)batch > h->fa_need_cap) {
                        (void)hipDeviceSynchronize();
                        if (h->fa_need) (void)hipFree(h->fa_need);
                        h->fa_need = nullptr; h->fa_need_cap = 0;
                        size_t cap = 256;
                        while (cap < (size_t)batch) cap *= 2;
                        if (hipMalloc((void**)&h->fa_need, cap * sizeof(int)) != hipSuccess) return FMPC_E_ALLOC;
                        h->fa_need_cap = cap;
                    }
                    FaParams A = h->fa_P;
                    A.batch = batch; A.step_ld = fmpc_step_ld(n_newton);
                    A.x0 = x0; A.x0p = x0_pre; A.nu0 = nu0; A.zout = z_null ? nullptr : z_out; A.nuout = nu_out; A.u0out = u0_out;
                    A.status = status; A.iters = iters; A.step = step; A.need = h->fa_need; A.handed = h->pn_cnt;
                    int* const nf = h->fa_nflag;                          // [running count of flagged problems | count dealt with | ticket]: device
                                                                          // state only, the same in an eager call and in a recorded graph
                    static const bool no_nflag = [] { const char* e = getenv("FMPC_NO_NFLAG"); return e && e[0] == '1'; }();   // A/B switch
                    A.nflag = no_nflag ? nullptr : nf;
                    A.ldz = zld;                                          // (0: contiguous rows)
                    if (fmpc_launch_affine(A, h->num_cu, (hipStream_t)stream) != hipSuccess) return FMPC_E_HIP;
                    int g3 = grid < 64 ? grid : 64;                    // flag mode: the waves walk over the flags, few are set
                    e = fmpc_launch_wave(h->dev, h->wave, batch, g3, x0, x0_pre, nullptr, nullptr, nu0, 1, k, z_out, nu_out, status, iters, step,
                                         fmpc_step_ld(n_newton), h->ws, stride, h->wave_lds, (hipStream_t)stream, 1, h->sh_fac, h->sh_rs, h->sh_ok,
                                         h->cold_d, nullptr, nullptr, h->pn_cnt, nullptr, u0_out, 3, nullptr, h->fa_need, 0, no_nflag ? nullptr : nf, zld);
                    h->last_path = FMPC_PATH_PANEL; h->inv_last = 2;
                    return e == hipSuccess ? FMPC_OK : FMPC_E_HIP;
                }
            }
            if (zld) return FMPC_E_UNSUPPORTED;                       // (nothing of this solve is enqueued yet)
            const int split = max_iter > 1;                           // budgets > 1: decide, compact, continue (two launches)
            const int pgrid = npanels < h->num_cu ? npanels : h->num_cu;
            // (a small LDS footprint lets a d_z workgroup of another stream share the CU)
            h->inv_last = dense_form && h->inv_valid;
            // without w, without the terminal row and with a budget of 1 nobody but d_z reads nu+: d_z computes it itself
            const bool fuse = h->inv_last && h->inv_fuse && w == nullptr && !h->has_xf && max_iter == 1;
            Q.gate_only = fuse ? 1 : 0;
            const bool u0only = z_null && max_iter == 1 && nu_out == nullptr;       // nothing of z leaves the chip
            Q.u0out = u0_out;
            if (h->inv_last) {
                if (lowrank) { Q.gw = h->lp_v; Q.gwn = 2 * h->n; Q.jimg = h->inv_jimg2; Q.jks = h->inv_jks2; Q.jksp = 16 * ((h->inv_jks2 + 15) / 16 + 3); }
                else { Q.gw = w; Q.gwn = h->T * h->n; }
            }
            if (h->inv_last) e = fmpc_launch_inv(Q, (hipStream_t)stream);
            else e = fmpc_launch_panel(Q, pgrid, fmpc_panel_lds_used(h->nb, h->pn_mp, h->pn_nsf + h->pn_nsb), (hipStream_t)stream);
            if (e != hipSuccess) return FMPC_E_HIP;
            const int ntasks = npanels * h->T;
            // one task per wave, 8 per workgroup; workgroups b, b + 8, ... take the panels b % 8, b % 8 + 8, ...
            const int ppx = (npanels + 7) / 8;                       // panels of the fullest XCD share
            const int dgrid = 8 * ((ppx * h->T + FD_WAVES - 1) / FD_WAVES);
            (void)ntasks;
            e = fmpc_launch_dz(Q, dgrid, split, (hipStream_t)stream, fuse ? 1 : 0, u0only ? 1 : 0);
            if (e != hipSuccess) return FMPC_E_HIP;
            // decides the step length of every problem; solves exactly those whose decision is not clear-cut.  Budget 1:
            // one launch.  Budgets > 1: a decide-only launch that also evaluates the next exit test and COMPACTS the
            // problems that go on (typically a few per cent), then the continuation over that list on as few
            // workgroups as it needs -- the rest of the chip stays free for other streams.
            for (int ph = split ? 1 : 0; ph <= (split ? 2 : 0); ++ph) {
                int g2 = grid;
                if (ph == 2) {
                    // Every workgroup of the continuation needs a CU with its whole LDS free, also those that find the
                    // list empty -- behind other streams' long-running continuations they would wait for a slot.  Size
                    // the grid from the list length an EARLIER call of this handle reported (never waited for); the
                    // kernel loops over the list, so any grid is correct.
                    const int last = ((volatile int*)h->pn_cnt_host)[1];
                    if (last >= 0) {
                        const int wpw2 = fmpc_wave_waves_per_wg();
                        int want = (last + last / 2 + wpw2 - 1) / wpw2 + 4;
                        if (want < 8) want = 8;
                        if (want < g2) g2 = want;
                    }
                }
                if (ph == 2 && h->small_tiled) {
                    // the continuation is a handful of problems per CU at most: the tiled kernel (4 wavefronts per problem)
                    // finishes one in 0.45 ms where the one-wavefront kernel needs 1.0 ms (round 2 measurement)
                    const int last = ((volatile int*)h->pn_cnt_host)[1];
                    const int hint = last >= 0 ? last + last / 2 + 16 : 0;
                    const int rc_t = fmpc_solve_tiled(h, 0, batch, x0, x0_pre, w, nullptr, nu0, n_newton, k, z_out, nu_out, status, iters,
                                                      step, u0_out, (hipStream_t)stream, h->small_nw, h->pn_list, h->pn_cnt + 1, h->pn_nuws, hint);
                    if (rc_t == FMPC_OK) continue;
                    // (no workspace for the tiled kernel: the panel, d_z and decision kernels are already enqueued -- the
                    // one-wavefront kernel below, whose workspace exists, finishes the solve instead of aborting it half-way)
                    if (rc_t != FMPC_E_UNSUPPORTED && rc_t != FMPC_E_ALLOC) return rc_t;
                }
                e = fmpc_launch_wave(h->dev, h->wave, batch, g2, x0, x0_pre, w, z_init, nu0, max_iter, k,
                                     z_out, nu_out, status, iters, step, fmpc_step_ld(n_newton), h->ws, stride,
                                     h->wave_lds, (hipStream_t)stream, mode, h->sh_fac, h->sh_rs, h->sh_ok, h->cold_d,
                                     h->pn_gate, h->pn_epsp, h->pn_cnt, h->pn_nuws, u0_out, ph, h->pn_rnp, h->pn_list, u0only ? 1 : 0);
                if (e != hipSuccess) return FMPC_E_HIP;
            }
            if (split && hipMemcpyAsync(h->pn_cnt_host, h->pn_cnt, 2 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
                return FMPC_E_HIP;
            h->last_path = FMPC_PATH_PANEL;
            return e == hipSuccess ? FMPC_OK : FMPC_E_HIP;
        }
        if (zld) return FMPC_E_UNSUPPORTED;
        if (mode == 0 && h->small_tiled && batch <= 1024) {
            // Every problem factors its own Schur complement and there are at most 4 problems per CU: the tiled kernel's
            // 2 (4) wavefronts per problem finish a problem in 0.5 (0.4) ms where the one-wavefront kernel needs 1.0 ms;
            // beyond 1024 problems the one-wavefront kernel's 8 problems per CU win (measured in round 2)
            const int rc_t = fmpc_solve_tiled(h, 0, batch, x0, x0_pre, w, z_init, nu0, n_newton, k, z_out, nu_out, status, iters, step,
                                              u0_out, (hipStream_t)stream, batch <= 512 ? h->small_nw : 2);
            if (rc_t != FMPC_E_UNSUPPORTED) return rc_t;
        }
        FmpcTiledPlan gplan;
        int gn_hint = 0;
        bool gn_go = mode == 0 && max_iter > 1 && h->small_tiled && h->gn_split;
        if (gn_go) {
            // the continuation's launch is planned (instance, size, workspace) BEFORE the first-step launch is enqueued: if the tiled
            // kernel cannot run, the single launch below solves the batch and nothing is left half done
            const int last = h->gn_cnt_host ? ((volatile int*)h->gn_cnt_host)[1] : -1;   // list length an EARLIER call reported (never waited for)
            gn_hint = last >= 0 ? last + last / 2 + 16 : 0;
            if (fmpc_tiled_plan(h, 0, batch, (hipStream_t)stream, h->small_nw, gn_hint, &gplan) != FMPC_OK) gn_go = false;
        }
        if (gn_go) {
            // Explicit-start batch too large for the tiled kernel, budget > 1: with one problem per wavefront slot a launch lasts
            // as long as its slowest wavefront, so the few per cent of problems that take a second step double it (round 3:
            // 2.0 ms for 2180 problem-iterations, 1.2 ms for the 2000 first ones).  Two launches instead: the one-wavefront
            // kernel takes the first step of every problem and the NEXT exit test (pphase 4), appending the problems that go on
            // to a list; the tiled kernel (two wavefronts per problem: half the latency of one problem) works through that list.
            const size_t nbn = (size_t)h->nb * h->n;
            if ((size_t)batch > h->gn_cap) {
                (void)hipDeviceSynchronize();
                if (h->gn_list) (void)hipFree(h->gn_list);
                if (h->gn_nu) (void)hipFree(h->gn_nu);
                if (h->gn_cnt) (void)hipFree(h->gn_cnt);
                h->gn_list = nullptr; h->gn_nu = nullptr; h->gn_cnt = nullptr; h->gn_cap = 0;
                if (hipMalloc((void**)&h->gn_list, (size_t)batch * sizeof(int)) != hipSuccess || hipMalloc((void**)&h->gn_cnt, 2 * sizeof(int)) != hipSuccess ||
                    hipMalloc((void**)&h->gn_nu, (size_t)batch * nbn * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
                h->gn_cap = batch;
            }
            if (!h->gn_cnt_host) {
                if (hipHostMalloc((void**)&h->gn_cnt_host, 2 * sizeof(int), hipHostMallocDefault) != hipSuccess) return FMPC_E_ALLOC;
                h->gn_cnt_host[0] = -1; h->gn_cnt_host[1] = -1;
            }
            double* nu_arr = nu_out ? nu_out : h->gn_nu;               // nu after the first step, per problem: what the continuation starts from
            // (a kernel, not hipMemsetAsync: a captured graph holding the 8-byte memset node faulted on its SECOND launch --
            // "write access to a read-only page", ROCm 7.2, round 5; with this kernel node instead, and the 8-byte device-to-host
            // copy node below kept, any number of launches is fine: tests/test_gpu_recorded.py)
            hipLaunchKernelGGL(fmpc_zero_ints, dim3(1), dim3(64), 0, (hipStream_t)stream, h->gn_cnt, 2);
            if (hipGetLastError() != hipSuccess) return FMPC_E_HIP;
            e = fmpc_launch_wave(h->dev, h->wave, batch, grid, x0, x0_pre, w, z_init, nu0, max_iter, k,
                                 z_out, nu_arr, status, iters, step, fmpc_step_ld(n_newton), h->ws, stride,
                                 h->wave_lds, (hipStream_t)stream, mode, h->sh_fac, h->sh_rs, h->sh_ok, h->cold_d,
                                 nullptr, nullptr, h->gn_cnt, nullptr, u0_out, 4, nullptr, h->gn_list);
            if (e != hipSuccess) return FMPC_E_HIP;
            h->last_path = FMPC_PATH_WAVE;
            const int rc_t = fmpc_solve_tiled(h, 0, batch, x0, x0_pre, w, nullptr, nu0, n_newton, k, z_out, nu_arr, status, iters, step,
                                              u0_out, (hipStream_t)stream, h->small_nw, h->gn_list, h->gn_cnt + 1, nullptr, gn_hint, &gplan);
            if (rc_t != FMPC_OK) return rc_t;
            if (hipMemcpyAsync(h->gn_cnt_host, h->gn_cnt, 2 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return FMPC_E_HIP;
            return FMPC_OK;
        }
        e = fmpc_launch_wave(h->dev, h->wave, batch, grid, x0, x0_pre, w, z_init, nu0, max_iter, k,
                             z_out, nu_out, status, iters, step, fmpc_step_ld(n_newton), h->ws, stride,
                             h->wave_lds, (hipStream_t)stream, mode, h->sh_fac, h->sh_rs, h->sh_ok, h->cold_d,
                             nullptr, nullptr, nullptr, nullptr, u0_out);
        h->last_path = mode == 1 ? FMPC_PATH_SHARED : FMPC_PATH_WAVE;
    } else {
        h->last_path = FMPC_PATH_GENERIC;
        const int grid = fmpc_grid_for(h, batch);
        size_t stride = 0;
        // full size once: no regrowth (big: a slot holds the whole factor -- as many slots as the launch has workgroups, grown on demand)
        int rc = fmpc_ensure_ws(h, h->generic_big ? grid : h->num_cu * h->wg_per_cu, &stride);
        if (rc != FMPC_OK) return rc;
        e = fmpc_launch_generic(h->dev, batch, grid, x0, x0_pre, w, z_init, nu0, max_iter, k,
                                z_out, nu_out, status, iters, step, fmpc_step_ld(n_newton),
                                h->ws, stride, (hipStream_t)stream, h->generic_big);
        if (e == hipSuccess && u0_out)
            e = fmpc_launch_unpack(h->n, h->m, h->T, batch, z_out, nullptr, nullptr, u0_out, (hipStream_t)stream);
    }
    return e == hipSuccess ? FMPC_OK : FMPC_E_HIP;
}

static int fmpc_solve_device_impl(fmpc_handle h, int batch,
                                  const double* x0, const double* x0_pre, const double* w,
                                  const double* z_init, const double* nu0,
                                  int n_newton, double k,
                                  double* z_out, double* nu_out, int* status, int* iters, double* step,
                                  double* u0_out, void* stream) {
    if (!h || !x0 || (!z_out && !u0_out)) return FMPC_E_NULL;
    if (batch < 0) return FMPC_E_DIM;
    if (batch == 0) return FMPC_OK;
    if (hipSetDevice(h->device) != hipSuccess) return FMPC_E_HIP;
    std::lock_guard<std::mutex> lk(h->mu);
    int rc = fmpc_guard_begin(h, (hipStream_t)stream);
    if (rc != FMPC_OK) return rc;
    rc = fmpc_solve_device_inner(h, batch, x0, x0_pre, w, z_init, nu0, n_newton, k, z_out, nu_out, status, iters, step,
                                 u0_out, stream);
    fmpc_guard_end(h, (hipStream_t)stream);
    return rc;
}

extern "C" int fmpc_solve_device(fmpc_handle h, int batch,
                                 const double* x0, const double* x0_pre, const double* w,
                                 const double* z_init, const double* nu0,
                                 int n_newton, double k,
                                 double* z_out, double* nu_out, int* status, int* iters, double* step,
                                 void* stream) {
    return fmpc_solve_device_impl(h, batch, x0, x0_pre, w, z_init, nu0, n_newton, k, z_out, nu_out, status, iters, step,
                                  nullptr, stream);
}

extern "C" int fmpc_solve_u0_device(fmpc_handle h, int batch,
                                    const double* x0, const double* x0_pre, const double* w,
                                    const double* z_init, const double* nu0,
                                    int n_newton, double k,
                                    double* z_out, double* nu_out, int* status, int* iters, double* step,
                                    double* u0_out, void* stream) {
    if (!u0_out) return FMPC_E_NULL;
    return fmpc_solve_device_impl(h, batch, x0, x0_pre, w, z_init, nu0, n_newton, k, z_out, nu_out, status, iters, step,
                                  u0_out, stream);
}

// fmpc_solve_u0_device with the row distance of z_out as an ARGUMENT of the call (0: contiguous rows) instead of handle state
extern "C" int fmpc_solve_u0_device_ld(fmpc_handle h, int batch,
                                       const double* x0, const double* x0_pre, const double* w,
                                       const double* z_init, const double* nu0,
                                       int n_newton, double k,
                                       double* z_out, double* nu_out, int* status, int* iters, double* step,
                                       double* u0_out, int ldz, void* stream) {
    if (!h) return FMPC_E_NULL;
    if (ldz < 0 || (ldz != 0 && ldz < h->T * (h->n + h->m))) return FMPC_E_DIM;
    fmpc_tl_zld_explicit = ldz;
    const int rc = fmpc_solve_device_impl(h, batch, x0, x0_pre, w, z_init, nu0, n_newton, k, z_out, nu_out, status, iters, step,
                                          u0_out, stream);
    fmpc_tl_zld_explicit = -1;
    return rc;
}

// Closed-loop step of a few realisations in the first-move form (fmpc_kernel_first.hip): ONE launch computes the loop inputs,
// the first moves and the step-length decision; a second one (the exact path in flag mode) returns at once unless a
// realisation was not clear-cut.  FMPC_E_UNSUPPORTED: the caller takes the four-launch path.  Caller holds h->mu.
#define FMPC_FIRST_MOVE_MAX_BATCH 64
#define FMPC_WALK_MAX_BATCH 4096          // realisations of a one-launch walk (fmpc_loop_run_walk): one workgroup each
static int fmpc_first_move_ensure(fmpc_handle h, int batch, double k, hipStream_t stream, size_t* stride_out, int max_batch = FMPC_FIRST_MOVE_MAX_BATCH) {
    if (h->fm_disabled || !h->use_wave || !h->sh_enabled || !h->pn_enabled || !h->inv_enabled || h->n != FP_N || batch > max_batch)
        return FMPC_E_UNSUPPORTED;
    size_t stride = 0;
    int rc = fmpc_ensure_wave_ws(h, &stride);
    if (rc != FMPC_OK) return rc;
    rc = fmpc_ensure_cold(h, k, stride, stream);
    if (rc != FMPC_OK) return rc;
    if (!h->pn_valid) return FMPC_E_UNSUPPORTED;
    if (!h->inv_valid || h->inv_k != k) {
        if (h->inv_failed && h->inv_failed_k == k) return FMPC_E_UNSUPPORTED;
        if (hipStreamSynchronize(stream) != hipSuccess) return FMPC_E_HIP;
        rc = fmpc_build_inverse(h, k, stream);
        if (rc != FMPC_OK) return rc;
        h->inv_failed = h->inv_valid ? 0 : 1; h->inv_failed_k = k;
        if (!h->inv_valid) return FMPC_E_UNSUPPORTED;
    }
    if (!h->fm_valid || h->fm_k != k) {
        if (hipStreamSynchronize(stream) != hipSuccess) return FMPC_E_HIP;
        rc = fmpc_build_first_move(h, k);
        if (rc != FMPC_OK) return rc;
    }
    // scratch iterate for realisations the exact path redoes, one flag per realisation
    const size_t needz = (size_t)batch * h->T * (h->n + h->m);
    if (needz > h->zs_doubles) {
        if (h->zs) { (void)hipDeviceSynchronize(); (void)hipFree(h->zs); h->zs = nullptr; h->zs_doubles = 0; }
        if (hipMalloc((void**)&h->zs, needz * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
        h->zs_doubles = needz;
    }
    if ((size_t)batch > h->fm_need_cap) {
        if (h->fm_need) { (void)hipDeviceSynchronize(); (void)hipFree(h->fm_need); h->fm_need = nullptr; h->fm_need_cap = 0; }
        const size_t cap = batch > FMPC_FIRST_MOVE_MAX_BATCH ? (size_t)batch : (size_t)FMPC_FIRST_MOVE_MAX_BATCH;
        if (hipMalloc((void**)&h->fm_need, cap * sizeof(int)) != hipSuccess) return FMPC_E_ALLOC;
        h->fm_need_cap = cap;
    }
    *stride_out = stride;
    return FMPC_OK;
}

static int fmpc_first_move_step(fmpc_handle h, int batch, const double* a_k, const double* x0_last, const double* u1, const double* u2,
                                double* x0, double* x0_pre, double* w, const double* nu0, double k,
                                int* status, int* iters, double* step, double* u0_out, hipStream_t stream, int given = 0) {
    size_t stride = 0;
    int rc = fmpc_first_move_ensure(h, batch, k, stream, &stride);
    if (rc != FMPC_OK) return rc;
    FmParams P = h->fm_P;
    P.step_ld = fmpc_step_ld(1);
    P.x0_given = given;                            // (a_k IS x0, x0_last IS x0_pre: nothing is written to x0 / x0_pre)
    const double* xs = given ? a_k : x0; const double* xps = given ? x0_last : x0_pre;
    P.a_k = a_k; P.x0_last = x0_last; P.u1 = u1; P.u2 = u2; P.nu0 = nu0;
    P.x0 = x0; P.x0_pre = x0_pre; P.w = w; P.u0out = u0_out; P.status = status; P.iters = iters; P.step = step; P.need = h->fm_need;
    P.forms = h->fm_forms;
    P.handed = h->pn_cnt;                          // zeroed by the kernel; the exact path (next launch) counts what it redoes
    if (fmpc_launch_first_move(P, batch, stream) != hipSuccess) return FMPC_E_HIP;
    // the exact path for flagged realisations (cold start with the shared factor), first moves from its own z
    const int wpw = fmpc_wave_waves_per_wg();
    const int grid = (batch + wpw - 1) / wpw;
    if (fmpc_launch_wave(h->dev, h->wave, batch, grid, xs, xps, w, nullptr, nu0, 1, k, h->zs, nullptr, status, iters, step,
                         fmpc_step_ld(1), h->ws, stride, h->wave_lds, stream, 1, h->sh_fac, h->sh_rs, h->sh_ok, h->cold_d,
                         nullptr, nullptr, h->pn_cnt, nullptr, u0_out, 3, nullptr, h->fm_need) != hipSuccess) return FMPC_E_HIP;
    h->last_path = FMPC_PATH_PANEL; h->inv_last = 1;
    return FMPC_OK;
}

// Closed-loop step of MANY realisations with first moves only: the loop-input kernel, the first-move form as a product over the
// batch (fmpc_kernel_loopu0.hip), the exact path in flag mode -- three launches, none of them over the T stages.
#define FMPC_LOOP_U0_MAX_BATCH 65536
static int fmpc_loop_u0_step(fmpc_handle h, int batch, const double* a_k, const double* x0_last, const double* u1, const double* u2,
                             double* x0, double* x0_pre, double* w, const double* nu0, double k,
                             int* status, int* iters, double* step, double* u0_out, hipStream_t stream, int given = 0) {
    if (h->fl_disabled) return FMPC_E_UNSUPPORTED;
    size_t stride = 0;
    int rc = fmpc_first_move_ensure(h, batch, k, stream, &stride, FMPC_LOOP_U0_MAX_BATCH);
    if (rc != FMPC_OK) return rc;
    if (!h->fl_valid) return FMPC_E_UNSUPPORTED;
    const int wpw = fmpc_wave_waves_per_wg();
    int grid = (batch + wpw - 1) / wpw;
    if (grid > 64) grid = 64;                      // flagged realisations are few: the list is walked by a small grid (as after the affine kernel)
    // ONE launch for the loop inputs, the first moves and the decision -- when the other workgroups may read x0_last while
    // x0 and x0_pre are written, i.e. when the caller does not update x0 in place
    if (h->fs_valid && h->loop_M1 && (given || x0_last == nullptr || (x0_last != x0 && x0_last != x0_pre))) {
        FlParams L = h->fs_P;
        FlStepIn I = h->fs_I;
        I.x0_given = given;
        const double* xs = given ? a_k : x0; const double* xps = given ? x0_last : x0_pre;
        L.batch = batch; L.step_ld = fmpc_step_ld(1);
        L.nu0 = nu0; L.u0out = u0_out; L.status = status; L.iters = iters; L.step = step; L.need = h->fm_need; L.handed = h->pn_cnt;
        L.x0w = x0; L.x0pw = x0_pre;
        I.a = a_k; I.x0_last = x0_last; I.u1 = u1; I.u2 = u2; I.w = w;
        if (fmpc_launch_loop_step27(L, I, stream) != hipSuccess) return FMPC_E_HIP;
        if (fmpc_launch_wave(h->dev, h->wave, batch, grid, xs, xps, w, nullptr, nu0, 1, k, h->zs, nullptr, status, iters, step,
                             fmpc_step_ld(1), h->ws, stride, h->wave_lds, stream, 1, h->sh_fac, h->sh_rs, h->sh_ok, h->cold_d,
                             nullptr, nullptr, h->pn_cnt, nullptr, u0_out, 3, nullptr, h->fm_need) != hipSuccess) return FMPC_E_HIP;
        h->last_path = FMPC_PATH_PANEL; h->inv_last = 4;
        return FMPC_OK;
    }
    if (given) return FMPC_E_UNSUPPORTED;          // (the caller takes the general route)
    if ((size_t)batch > h->lp_cap) {
        (void)hipDeviceSynchronize();
        if (h->lp_v) (void)hipFree(h->lp_v);
        h->lp_v = nullptr; h->lp_cap = 0;
        if (hipMalloc((void**)&h->lp_v, (size_t)batch * 2 * h->n * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
        h->lp_cap = batch;
    }
    if (fmpc_launch_loop_inputs(h->n, h->m, h->T, batch, h->dev.Bt, h->loop_M1, h->loop_M2, a_k, x0_last, u1, u2,
                                x0, x0_pre, w, stream, h->lp_v) != hipSuccess) return FMPC_E_HIP;
    FlParams L = h->fl_P;
    L.batch = batch; L.step_ld = fmpc_step_ld(1);
    L.x0 = x0; L.x0_pre = x0_pre; L.v = h->lp_v; L.nu0 = nu0;
    L.u0out = u0_out; L.status = status; L.iters = iters; L.step = step; L.need = h->fm_need; L.handed = h->pn_cnt;
    if (fmpc_launch_loop_u0(L, stream) != hipSuccess) return FMPC_E_HIP;
    if (fmpc_launch_wave(h->dev, h->wave, batch, grid, x0, x0_pre, w, nullptr, nu0, 1, k, h->zs, nullptr, status, iters, step,
                         fmpc_step_ld(1), h->ws, stride, h->wave_lds, stream, 1, h->sh_fac, h->sh_rs, h->sh_ok, h->cold_d,
                         nullptr, nullptr, h->pn_cnt, nullptr, u0_out, 3, nullptr, h->fm_need) != hipSuccess) return FMPC_E_HIP;
    h->last_path = FMPC_PATH_PANEL; h->inv_last = 3;
    return FMPC_OK;
}

// One closed-loop step: fmpc_loop_inputs_device + fmpc_solve_u0_device under one lock, with the knowledge that
// w = -M1 (B u1) - M2 (B u2) has only 2 n degrees of freedom.
// given = 1 (fmpc_ao_step_device): a_k IS x0 and x0_last IS x0_pre -- the loop with its estimator, where the solver's x0 is the
// estimated residual and not a[k] + B u[k-1] (README.md:482-497); x0 / x0_pre are not written (may be NULL).
static int fmpc_loop_step_impl(fmpc_handle h, int batch, const double* a_k, const double* x0_last,
                               const double* u1, const double* u2, double* x0, double* x0_pre, double* w,
                               const double* nu0, int n_newton, double k,
                               double* z_out, double* nu_out, int* status, int* iters, double* step,
                               double* u0_out, void* stream, int given) {
    if (!h || !a_k || (!given && (!x0 || !x0_pre)) || !w || !u0_out) return FMPC_E_NULL;       // z_out may be NULL: first moves only
    if (given && !x0_last && h->var_order == 2) return FMPC_E_NULL;             // (x0_pre of a VAR(2) model: zeros at the first step, not NULL)
    if (batch < 0) return FMPC_E_DIM;
    if (batch == 0) return FMPC_OK;
    if (hipSetDevice(h->device) != hipSuccess) return FMPC_E_HIP;
    std::lock_guard<std::mutex> lk(h->mu);
    int rc = fmpc_guard_begin(h, (hipStream_t)stream);
    if (rc != FMPC_OK) return rc;
    const bool lr = h->inv_enabled && h->inv_jimg2 != nullptr && h->n == FP_N;
    // (a handle switched to the fp32 factor or to FMPC_TILED=1 takes the kernel it was switched to, whatever outputs are asked for)
    if (lr && !z_out && !nu_out && n_newton == 1 && h->prec == FMPC_PREC_F64 && !h->force_tiled) {
        // first moves only, a few realisations: one launch instead of four (falls through when the form does not apply)
        // few realisations: a workgroup per realisation (fmpc_first_move); from fs_min_batch on (default 65, FMPC_PRODUCT_MIN_BATCH)
        // the product form, whose 16 workgroups per 16 realisations spread the constants over as many CUs
        rc = FMPC_E_UNSUPPORTED;
        if (batch >= h->fs_min_batch)
            rc = fmpc_loop_u0_step(h, batch, a_k, x0_last, u1, u2, x0, x0_pre, w, nu0, k, status, iters, step, u0_out, (hipStream_t)stream, given);
        if (rc == FMPC_E_UNSUPPORTED)
            rc = fmpc_first_move_step(h, batch, a_k, x0_last, u1, u2, x0, x0_pre, w, nu0, k, status, iters, step, u0_out, (hipStream_t)stream, given);
        if (rc == FMPC_E_UNSUPPORTED && batch > FMPC_FIRST_MOVE_MAX_BATCH && batch < h->fs_min_batch)
            rc = fmpc_loop_u0_step(h, batch, a_k, x0_last, u1, u2, x0, x0_pre, w, nu0, k, status, iters, step, u0_out, (hipStream_t)stream, given);
        if (rc != FMPC_E_UNSUPPORTED) { fmpc_guard_end(h, (hipStream_t)stream); return rc; }
        rc = FMPC_OK;
    }
    if (lr && (size_t)batch > h->lp_cap) {
        (void)hipDeviceSynchronize();
        if (h->lp_v) (void)hipFree(h->lp_v);
        h->lp_v = nullptr; h->lp_cap = 0;
        if (hipMalloc((void**)&h->lp_v, (size_t)batch * 2 * h->n * sizeof(double)) != hipSuccess) { fmpc_guard_end(h, (hipStream_t)stream); return FMPC_E_ALLOC; }
        h->lp_cap = batch;
    }
    const double* a_in = a_k; const double* xl_in = x0_last; double* x0_o = x0; double* x0p_o = x0_pre;
    if (given) {
        // the loop-input kernel only has to produce w: a = 0, its x0 / x0_pre go to scratch (3 batch n doubles of the handle)
        if ((size_t)batch > h->ao_cap) {
            (void)hipDeviceSynchronize();
            if (h->ao_scr) (void)hipFree(h->ao_scr);
            h->ao_scr = nullptr; h->ao_cap = 0;
            if (hipMalloc((void**)&h->ao_scr, (size_t)3 * batch * h->n * sizeof(double)) != hipSuccess) { fmpc_guard_end(h, (hipStream_t)stream); return FMPC_E_ALLOC; }
            if (hipMemsetAsync(h->ao_scr, 0, (size_t)batch * h->n * sizeof(double), (hipStream_t)stream) != hipSuccess) { fmpc_guard_end(h, (hipStream_t)stream); return FMPC_E_HIP; }
            h->ao_cap = batch;
        }
        a_in = h->ao_scr; xl_in = nullptr; x0_o = h->ao_scr + (size_t)h->ao_cap * h->n; x0p_o = h->ao_scr + (size_t)2 * h->ao_cap * h->n;
    }
    if (fmpc_launch_loop_inputs(h->n, h->m, h->T, batch, h->dev.Bt, h->loop_M1, h->loop_M2, a_in, xl_in, u1, u2,
                                x0_o, x0p_o, w, (hipStream_t)stream, lr ? h->lp_v : nullptr) != hipSuccess) rc = FMPC_E_HIP;
    if (rc == FMPC_OK) {
        h->lp_hint = lr ? 1 : 0;
        rc = fmpc_solve_device_inner(h, batch, given ? a_k : x0, given ? x0_last : x0_pre, w, nullptr, nu0, n_newton, k, z_out, nu_out, status, iters, step,
                                     u0_out, stream);
        h->lp_hint = 0;
    }
    fmpc_guard_end(h, (hipStream_t)stream);
    return rc;
}

extern "C" int fmpc_loop_step_device(fmpc_handle h, int batch, const double* a_k, const double* x0_last,
                                     const double* u1, const double* u2, double* x0, double* x0_pre, double* w,
                                     const double* nu0, int n_newton, double k,
                                     double* z_out, double* nu_out, int* status, int* iters, double* step,
                                     double* u0_out, void* stream) {
    return fmpc_loop_step_impl(h, batch, a_k, x0_last, u1, u2, x0, x0_pre, w, nu0, n_newton, k, z_out, nu_out, status, iters, step, u0_out, stream, 0);
}

// The MPC part of one timestep of the loop WITH its estimator (README.md:482-497,548-556,589): x0 = ad_est[k], x0_pre = ad_est[k-1]
// come from the estimator, b_ref = -M1 B u[k-1] - M2 B u[k-2] from the moves applied.  See include/fastmpc.h.
extern "C" int fmpc_ao_step_device(fmpc_handle h, int batch, const double* x0, const double* x0_pre,
                                   const double* u1, const double* u2, double* w,
                                   const double* nu0, int n_newton, double k,
                                   double* z_out, double* nu_out, int* status, int* iters, double* step,
                                   double* u0_out, void* stream) {
    return fmpc_loop_step_impl(h, batch, x0, x0_pre, u1, u2, nullptr, nullptr, w, nu0, n_newton, k, z_out, nu_out, status, iters, step, u0_out, stream, 1);
}

// A recorded stretch of the loop in ONE host call: steps consecutive fmpc_loop_step_device calls with the first moves fed back
// on the device (u[k] = U0[k], u[k-1] = U0[k-1], ...).  See include/fastmpc.h.
// The steps [0, upto) of a recorded stretch as ONE launch per walk (fmpc_first_move_run): every realisation walks until a step
// is not clear-cut; the host has that step redone by the one-step call (its exact path) and starts the walk again behind it.
// Synchronises the stream once per walk (it has to see where the walks stopped).  FMPC_E_UNSUPPORTED: not applicable, nothing done.
static int fmpc_loop_run_walk(fmpc_handle h, int batch, int steps, int upto, const double* a, const double* nu0,
                              const double* ub1, const double* ub2, int have_x0_last, double k,
                              double* x0, double* x0_pre, double* w, double* U0, double* X0, int* status, int* iters, hipStream_t stream) {
    std::vector<int> start(batch, 0), stop(batch, 0), idx, stp;      // (host sides of asynchronous copies: alive until the function returns)
    std::vector<std::vector<int>> keep;                              // index lists of the stepwise mode (no synchronisation between its copies)
    // Every walk costs a synchronisation, and every stop a gather + one-step call + scatter + a new batch-wide launch: on a stretch
    // where many steps are not clear-cut that is slower than one call per step.  After `max_restarts` walks, or when more than a
    // tenth of the realisations stop in one walk, the rest of the stretch is done STEPWISE: the realisations that are furthest
    // behind take one step through the one-step call (a compact batch, as for a stopped step) until all have arrived -- at most
    // `upto` such calls, none of them synchronises.
    int max_restarts = 8, restarts = 0;
    { const char* e = getenv("FMPC_WALK_MAX_RESTARTS"); if (e && e[0] >= '0' && e[0] <= '9') max_restarts = atoi(e); }
    bool stepwise = false;
    for (;;) {
        if (stepwise) {
            int smin = upto;
            for (int p = 0; p < batch; ++p) smin = start[p] < smin ? start[p] : smin;
            if (smin >= upto) return FMPC_OK;
            keep.emplace_back(); keep.emplace_back();
            std::vector<int>& ki = keep[keep.size() - 2];
            std::vector<int>& ks = keep[keep.size() - 1];
            for (int p = 0; p < batch; ++p)
                if (start[p] == smin) { ki.push_back(p); ks.push_back(smin); start[p] = smin + 1; }
            idx = ki; stp = ks;
        } else {
            if (hipSetDevice(h->device) != hipSuccess) return FMPC_E_HIP;
            std::lock_guard<std::mutex> lk(h->mu);
            int rc = fmpc_guard_begin(h, stream);
            if (rc != FMPC_OK) return rc;
            size_t stride = 0;
            rc = fmpc_first_move_ensure(h, 1, k, stream, &stride);
            if (rc == FMPC_OK && (size_t)batch > h->fm_walk_cap) {
                (void)hipDeviceSynchronize();
                if (h->fm_walk_i) (void)hipFree(h->fm_walk_i);
                if (h->fm_compact) (void)hipFree(h->fm_compact);
                h->fm_walk_i = nullptr; h->fm_compact = nullptr; h->fm_walk_cap = 0;
                size_t cap = 64;
                while (cap < (size_t)batch) cap *= 2;
                const size_t nd = fmpc_compact_doubles(h->n, h->m, h->T, h->nb, (int)cap);
                if (hipMalloc((void**)&h->fm_walk_i, 2 * cap * sizeof(int)) != hipSuccess || hipMalloc((void**)&h->fm_compact, nd * sizeof(double)) != hipSuccess) rc = FMPC_E_ALLOC;
                else h->fm_walk_cap = cap;
            }
            if (rc != FMPC_OK) { fmpc_guard_end(h, stream); return rc; }
            int* d_start = h->fm_walk_i;
            int* d_stop = h->fm_walk_i + h->fm_walk_cap;
            FmParams P = h->fm_P;
            P.step_ld = fmpc_step_ld(1);
            P.x0 = x0; P.x0_pre = x0_pre; P.status = status; P.iters = iters; P.handed = h->pn_cnt; P.forms = nullptr;
            FmRun R;
            R.steps = upto; R.batch = batch; R.have_x0_last = have_x0_last; R.start = d_start; R.stop = d_stop;
            R.a = a; R.nu0 = nu0; R.U0 = U0; R.X0 = X0; R.ub1 = ub1; R.ub2 = ub2;
            bool ok = hipMemcpyAsync(d_start, start.data(), batch * sizeof(int), hipMemcpyHostToDevice, stream) == hipSuccess;
            ok = ok && fmpc_launch_first_move_run(P, R, stream) == hipSuccess;
            ok = ok && hipMemcpyAsync(stop.data(), d_stop, batch * sizeof(int), hipMemcpyDeviceToHost, stream) == hipSuccess;
            h->last_path = FMPC_PATH_PANEL; h->inv_last = 1;
            fmpc_guard_end(h, stream);
            if (!ok || hipStreamSynchronize(stream) != hipSuccess) return FMPC_E_HIP;
        }
        // the stopped realisations, each at its own step, as ONE compact batch through the one-step call (its exact path redoes
        // them), results scattered back; then the walks go on behind those steps
        if (!stepwise) {
            idx.clear(); stp.clear();
            for (int p = 0; p < batch; ++p) {
                if (stop[p] >= upto) { start[p] = upto; continue; }
                idx.push_back(p); stp.push_back(stop[p]);
                start[p] = stop[p] + 1;
            }
        }
        const bool done = idx.empty();
        if (!done) {
            const int cnt = (int)idx.size();
            const int* hidx = stepwise ? keep[keep.size() - 2].data() : idx.data();
            const int* hstp = stepwise ? keep[keep.size() - 1].data() : stp.data();
            FmCompact C;
            C.cnt = cnt; C.n = h->n; C.m = h->m; C.T = h->T; C.nb = h->nb; C.batch = batch; C.have_x0_last = have_x0_last;
            C.a = a; C.nu0 = nu0; C.U0 = U0; C.X0 = X0; C.ub1 = ub1; C.ub2 = ub2;
            C.x0 = x0; C.x0_pre = x0_pre; C.w = w; C.status = status; C.iters = iters;
            {
                std::lock_guard<std::mutex> lk(h->mu);
                int* d_idx = h->fm_walk_i;                                     // start / stop slots: free between walks
                int* d_stp = h->fm_walk_i + h->fm_walk_cap;
                fmpc_compact_carve(C, h->fm_compact, (int)h->fm_walk_cap);
                C.idx = d_idx; C.stp = d_stp;
                if (hipMemcpyAsync(d_idx, hidx, cnt * sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess ||
                    hipMemcpyAsync(d_stp, hstp, cnt * sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess ||
                    fmpc_launch_walk_gather(C, stream) != hipSuccess) return FMPC_E_HIP;
            }
            const int rc = fmpc_loop_step_device(h, cnt, C.ca, C.cx0, C.cu1, C.cu2, C.cx0, C.cx0p, C.cw, nu0 ? C.cnu : nullptr, 1, k,
                                                 nullptr, nullptr, C.cst, C.cit, nullptr, C.cu0, stream);
            if (rc != FMPC_OK) return rc;
            if (fmpc_launch_walk_scatter(C, stream) != hipSuccess) return FMPC_E_HIP;
        }
        if (getenv("FMPC_DEBUG_WALK")) {
            if (stepwise) fprintf(stderr, "[fastmpc] stepwise: %d realisations take step %d\n", (int)idx.size(), stp.empty() ? -1 : stp[0]);
            else {
                int nstop = 0;
                for (int p = 0; p < batch; ++p) nstop += stop[p] < upto ? 1 : 0;
                fprintf(stderr, "[fastmpc] walk over %d realisations up to step %d: %d stopped\n", batch, upto, nstop);
            }
        }
        if (done) return FMPC_OK;                    // (a restarted walk begins at a step >= 1 of the stretch: x0 holds its predecessor's residual)
        if (!stepwise && (++restarts >= max_restarts || idx.size() * 10 > (size_t)batch)) stepwise = true;
    }
}

extern "C" int fmpc_loop_run_device(fmpc_handle h, int batch, int steps, const double* a, const double* nu0,
                                    const double* u_before1, const double* u_before2, int have_x0_last,
                                    int n_newton, double k, double* x0, double* x0_pre, double* w,
                                    double* U0, double* X0, int* status, int* iters, void* stream) {
    if (!h || !a || !x0 || !x0_pre || !w || !U0) return FMPC_E_NULL;
    if (batch < 0 || steps < 0) return FMPC_E_DIM;
    if (batch == 0 || steps == 0) return FMPC_OK;
    const size_t sn = (size_t)batch * h->n, sm = (size_t)batch * h->m, snu = (size_t)batch * h->nb * h->n;
    int s_begin = 0;
    if (n_newton == 1 && steps >= 3 && batch <= FMPC_WALK_MAX_BATCH && h->n == FP_N && h->inv_jimg2 != nullptr) {
        // all steps but the last in one launch; the last one by the one-step call, which leaves x0, x0_pre, w, status and
        // iters exactly as a step-by-step run does
        const int rc = fmpc_loop_run_walk(h, batch, steps, steps - 1, a, nu0, u_before1, u_before2, have_x0_last, k, x0, x0_pre, w, U0, X0,
                                          status, iters, (hipStream_t)stream);
        if (rc == FMPC_OK) s_begin = steps - 1;
        else if (rc != FMPC_E_UNSUPPORTED) return rc;
    }
    for (int s = s_begin; s < steps; ++s) {
        const double* u1 = s >= 1 ? U0 + (size_t)(s - 1) * sm : u_before1;
        const double* u2 = s >= 2 ? U0 + (size_t)(s - 2) * sm : (s == 1 ? u_before1 : u_before2);
        const int rc = fmpc_loop_step_device(h, batch, a + (size_t)s * sn, (s >= 1 || have_x0_last) ? x0 : nullptr, u1, u2, x0, x0_pre, w,
                                             nu0 ? nu0 + (size_t)s * snu : nullptr, n_newton, k, nullptr, nullptr, status, iters, nullptr,
                                             U0 + (size_t)s * sm, stream);
        if (rc != FMPC_OK) return rc;
        if (X0 && hipMemcpyAsync(X0 + (size_t)s * sn, x0, sn * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return FMPC_E_HIP;
    }
    return FMPC_OK;
}

// Diagnostic (tests): the one-step first-move kernel writes, per realisation, the bounds its decision used -- upper bound of
// ||e||^2, lower bounds of ||r_p||^2 and of rho^2 -- to dev_forms[3 p ..] (device memory, NULL: off).
extern "C" int fmpc_debug_first_move_forms(fmpc_handle h, double* dev_forms) {
    if (!h) return FMPC_E_NULL;
    std::lock_guard<std::mutex> lk(h->mu);
    h->fm_forms = dev_forms;
    return FMPC_OK;
}

extern "C" int fmpc_set_z_ld(fmpc_handle h, int ldz) {
    if (!h) return FMPC_E_NULL;
    if (ldz != 0 && ldz < h->T * (h->n + h->m)) return FMPC_E_DIM;
    std::lock_guard<std::mutex> lk(h->mu);
    h->z_ld = ldz;
    return FMPC_OK;
}

extern "C" int fmpc_set_small_batch_kernel(fmpc_handle h, int tiled) {
    if (!h) return FMPC_E_NULL;
    std::lock_guard<std::mutex> lk(h->mu);
    h->small_tiled = tiled ? 1 : 0;
    if (tiled) h->small_nw = tiled == 2 ? 2 : 4;
    return FMPC_OK;
}

extern "C" int fmpc_last_dual_form(fmpc_handle h) {
    if (!h) return FMPC_E_NULL;
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->last_path == FMPC_PATH_RAMP) return h->rc_last ? 5 : 0;     // 5: the cold-start step with ramp rows in its Woodbury form
    return h->last_path == FMPC_PATH_PANEL ? h->inv_last : 0;
}

extern "C" int fmpc_set_dense_form(fmpc_handle h, int enabled, int max_batch_with_w) {
    if (!h) return FMPC_E_NULL;
    std::lock_guard<std::mutex> lk(h->mu);
    if (enabled && !h->inv_jimg) return FMPC_E_UNSUPPORTED;
    h->inv_enabled = enabled ? 1 : 0;
    if (max_batch_with_w >= 0) h->inv_max_batch = max_batch_with_w;
    return FMPC_OK;
}

extern "C" int fmpc_last_tiled_wavefronts(fmpc_handle h) { return h ? h->tl_last_nw : 0; }

extern "C" int fmpc_last_dispatch(fmpc_handle h, int* path, int* handed_over) {
    if (!h) return FMPC_E_NULL;
    if (hipSetDevice(h->device) != hipSuccess) return FMPC_E_HIP;
    std::lock_guard<std::mutex> lk(h->mu);
    if (path) *path = h->last_path;
    if (handed_over) {
        *handed_over = 0;
        if (h->last_path == FMPC_PATH_PANEL) {
            if (hipDeviceSynchronize() != hipSuccess) return FMPC_E_HIP;
            if (hipMemcpy(handed_over, h->pn_cnt, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return FMPC_E_HIP;
        }
    }
    return FMPC_OK;
}

extern "C" int fmpc_set_ramp(fmpc_handle h, const double* du_min, const double* du_max) {
    if (!h || !du_min || !du_max) return FMPC_E_NULL;
    if (hipSetDevice(h->device) != hipSuccess) return FMPC_E_HIP;
    for (int c = 0; c < h->m; ++c)
        if (!(du_min[c] < du_max[c])) return FMPC_E_DIM;
    const size_t lds = fmpc_ramp_lds_bytes(h->n, h->m, h->nb * h->n);
    if (h->n > 64 || lds > FMPC_LDS_LIMIT || h->denseQ || h->denseR) return FMPC_E_UNSUPPORTED;   // (the ramp kernel keeps Q, Qf as diagonals)
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->ramp_du) {
        if (hipMalloc((void**)&h->ramp_du, 2 * (size_t)h->m * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
        if (fmpc_ramp_prepare(lds) != hipSuccess) return FMPC_E_HIP;
    } else if (hipDeviceSynchronize() != hipSuccess) {
        return FMPC_E_HIP;
    }
    if (hipMemcpy(h->ramp_du, du_min, h->m * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->ramp_du + h->m, du_max, h->m * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return FMPC_E_HIP;
    h->hm_dumin.assign(du_min, du_min + h->m); h->hm_dumax.assign(du_max, du_max + h->m);
    h->rc_valid = 0; h->rc_failed = 0;                              // (the constants of the cold-start form depend on the bounds)
    return FMPC_OK;
}

// Constants of the cold-start step's Woodbury form for barrier weight k (fmpc_host_build_ramp_cold): built on the host in long
// double on first use of a k, uploaded as one pool.  FMPC_E_UNSUPPORTED: the general path takes the solve (LDS, a start point
// at which Phibar or Ybar is not positive definite, FMPC_NO_RAMP_COLD=1).  Caller holds h->mu.
static int fmpc_ensure_ramp_cold(fmpc_handle h, double k) {
    if (h->rc_disabled) return FMPC_E_UNSUPPORTED;
    if (h->rc_valid && h->rc_k == k) return FMPC_OK;
    if (h->rc_failed && h->rc_failed_k == k) return FMPC_E_UNSUPPORTED;
    const size_t lds = fmpc_ramp_cold_lds_bytes(h->n, h->m, h->T, h->nb);
    if (lds > FMPC_LDS_LIMIT || h->n > 64) return FMPC_E_UNSUPPORTED;
    FmpcRampColdIn In;
    In.n = h->n; In.m = h->m; In.T = h->T; In.nb = h->nb; In.var2 = h->var_order == 2 ? 1 : 0; In.has_xf = h->has_xf;
    In.bt = h->hm_bt.data(); In.a1 = h->hm_a1f.data(); In.a2 = h->hm_a2f.data();
    In.umax = h->hm_umax.data(); In.umin = h->hm_umin.data(); In.umid = h->hm_umid.data(); In.xmid = h->hm_xmid.data();
    In.R2 = h->hm_R2.data(); In.rl = h->hm_rl.data(); In.Q2 = h->hm_Q2.data(); In.Qf2 = h->hm_Qf2.data(); In.ql = h->hm_ql.data();
    In.qfl = h->hm_qfl.data(); In.xf = h->hm_xf.data(); In.dumin = h->hm_dumin.data(); In.dumax = h->hm_dumax.data(); In.k = k;
    FmpcRampColdOut O;
    fmpc_host_build_ramp_cold(In, O);
    if (!O.valid) { h->rc_failed = 1; h->rc_failed_k = k; return FMPC_E_UNSUPPORTED; }
    std::vector<double> pool;
    auto push = [&](const std::vector<double>& v) { const size_t o = pool.size(); pool.insert(pool.end(), v.begin(), v.end()); if (pool.size() & 1) pool.push_back(0.0); return o; };
    const size_t o_g0 = push(O.g0), o_Gf = push(O.Gf), o_pu = push(O.phib_u), o_px = push(O.phib_x), o_gu = push(O.gbar_u), o_gx = push(O.gbar_x),
                 o_hd = push(O.hd), o_er = push(O.erb), o_cp = push(O.cpb), o_bb = push(O.betab), o_Yi = push(O.Yinv), o_G = push(O.G),
                 o_Xi = push(O.Xiu0t), o_y0 = push(O.y0c);
    size_t o_ik, o_ib;
    {
        std::vector<double> ik, ib;
        fmpc_host_mfma_images(h->hm_bt.data(), h->m, h->n, (h->n + 3) / 4, ik);
        fmpc_host_mfma_images(h->hm_b.data(), h->n, h->m, (h->m + 3) / 4, ib);
        o_ik = push(ik); o_ib = push(ib);
    }
    size_t o_Gt;
    {   // G in the LDS tile layout of fr_tile_cholesky_lds: upper tile triangle of [G | rhs] packed, 16 x 16 row-major, zero padded
        const int m_ = h->m, NTm = (m_ + 15) / 16, NT1 = NTm + 1;
        std::vector<double> Gt;
        for (int I = 0; I < NTm; ++I)
            for (int J = I; J < NT1; ++J)
                for (int e = 0; e < 256; ++e) {
                    const int row = 16 * I + (e >> 4), col = 16 * J + (e & 15);
                    Gt.push_back(row < m_ && col < m_ ? O.G[(size_t)row * ((m_ + 1) & ~1) + col] : 0.0);
                }
        o_Gt = push(Gt);
    }
    if (h->rc_pool) { (void)hipDeviceSynchronize(); (void)hipFree(h->rc_pool); h->rc_pool = nullptr; }
    h->rc_valid = 0;
    if (hipMalloc((void**)&h->rc_pool, pool.size() * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
    if (hipMemcpy(h->rc_pool, pool.data(), pool.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return FMPC_E_HIP;
    if (fmpc_ramp_cold_prepare(lds) != hipSuccess) return FMPC_E_HIP;
    FrColdParams& P = h->rc_P;
    memset(&P, 0, sizeof(P));
    P.M = h->dev; P.dumin = h->ramp_du; P.dumax = h->ramp_du + h->m; P.kbar = k;
    P.g0 = h->rc_pool + o_g0; P.Gf = h->rc_pool + o_Gf; P.phib_u = h->rc_pool + o_pu; P.phib_x = h->rc_pool + o_px; P.gbar_u = h->rc_pool + o_gu;
    P.gbar_x = h->rc_pool + o_gx; P.hd = h->rc_pool + o_hd; P.erb = h->rc_pool + o_er; P.cpb = h->rc_pool + o_cp; P.betab = h->rc_pool + o_bb;
    P.Yinv = h->rc_pool + o_Yi; P.G = h->rc_pool + o_G; P.Gt = h->rc_pool + o_Gt; P.imgBk = h->rc_pool + o_ik; P.imgBb = h->rc_pool + o_ib; P.Xiu0t = h->rc_pool + o_Xi; P.y0c = h->rc_pool + o_y0;
    h->rc_valid = 1; h->rc_k = k;
    return FMPC_OK;
}

// fmpc_solve_ramp_device (u0_out == NULL) / fmpc_solve_ramp_u0_device: the first moves come from the cold-start kernel itself when
// it takes the whole solve (budget 1), from the unpack kernel otherwise; z_out == NULL (first moves only) works in a scratch array
static int fmpc_solve_ramp_device_impl(fmpc_handle h, int batch,
                                       const double* x0, const double* x0_pre, const double* w, const double* u_prev,
                                       const double* z_init, const double* nu0, int n_newton, double k,
                                       double* z_out, double* nu_out, int* status, int* iters, double* step,
                                       double* u0_out, void* stream) {
    if (!h || !x0 || (!z_out && !u0_out) || !u_prev) return FMPC_E_NULL;
    if (!h->ramp_du) return FMPC_E_UNSUPPORTED;                    // fmpc_set_ramp first
    if (!fmpc_tl_contiguous_z && fmpc_effective_zld(h) > h->T * (h->n + h->m)) return FMPC_E_UNSUPPORTED;   // padded z rows: the cold-start affine step only
    if (batch < 0) return FMPC_E_DIM;
    if (batch == 0) return FMPC_OK;
    if (hipSetDevice(h->device) != hipSuccess) return FMPC_E_HIP;
    std::lock_guard<std::mutex> lk(h->mu);
    const int max_iter = n_newton > 0 ? n_newton : 1000;
    const bool z_null = z_out == nullptr;
    const bool cold_only = z_init == nullptr && max_iter == 1 && !h->rc_disabled;   // (the cold-start kernel then needs no z array at all)
    if (z_null && !cold_only) {
        const size_t need = (size_t)batch * h->T * (h->n + h->m);
        if (need > h->zs_doubles) {
            if (h->zs) { (void)hipDeviceSynchronize(); (void)hipFree(h->zs); h->zs = nullptr; h->zs_doubles = 0; }
            if (hipMalloc((void**)&h->zs, need * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
            h->zs_doubles = need;
        }
        z_out = h->zs;
    }
    // one workgroup per problem in flight; the workspace holds the dense Y of each (nb n)^2 doubles
    const size_t stride = fmpc_ramp_ws_doubles(h->n, h->m, h->T, h->nb);
    // up to one problem per CU: 512-thread workgroups (latency: 0.61 instead of 0.82 ms per Newton step at n = 27,
    // m = 144, T = 10); beyond: 256-thread workgroups, 3 per CU (throughput: 4.3e5 instead of 3.6e5 problems/s)
    int threads = 512;      // (8 wavefronts per problem, one problem per CU: faster than 3 x 256 threads per CU at every batch size)
    { const char* e = getenv("FMPC_RAMP_THREADS"); if (e && e[0]) { const int t = atoi(e); if (t == 256 || t == 512 || t == 1024) threads = t; } }   // experiments
    int cap = (threads == 256 ? 3 : 1) * h->num_cu;
    const size_t budget = (size_t)2 << 30;                         // doubles (16 GB) for all workgroups together
    if ((size_t)cap * stride > budget) cap = (int)(budget / stride);
    if (cap < 1) return FMPC_E_ALLOC;
    const int grid = batch < cap ? batch : cap;
    auto ensure_general_ws = [&]() -> int {                          // (the cold-start form with a budget of 1 never needs it)
        const size_t need = stride * (size_t)grid;
        if (need > h->ramp_ws_doubles) {
            if (h->ramp_ws) { (void)hipDeviceSynchronize(); (void)hipFree(h->ramp_ws); h->ramp_ws = nullptr; h->ramp_ws_doubles = 0; }
            if (hipMalloc((void**)&h->ramp_ws, need * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
            h->ramp_ws_doubles = need;
        }
        return FMPC_OK;
    };
    h->last_path = FMPC_PATH_RAMP;
    h->rc_last = 0;
    // Cold start: the first Newton step in its Woodbury form (one m x m factorisation per problem, fmpc_ramp_cold); a budget > 1
    // continues with the general kernel from the iterate that step leaves (it0 = 1).
    if (z_init == nullptr) {
        const int rcc = fmpc_ensure_ramp_cold(h, k);
        if (rcc != FMPC_OK && rcc != FMPC_E_UNSUPPORTED) return rcc;
        if (rcc == FMPC_OK) {
            const size_t cstride = fmpc_ramp_cold_ws_doubles(h->m);
            const int cgrid = batch < h->num_cu ? batch : h->num_cu;
            if (cstride * (size_t)cgrid > h->rc_ws_doubles) {
                if (h->rc_ws) { (void)hipDeviceSynchronize(); (void)hipFree(h->rc_ws); h->rc_ws = nullptr; h->rc_ws_doubles = 0; }
                size_t want = cstride * (size_t)(batch < h->num_cu ? (batch < 16 ? 16 : batch) : h->num_cu);
                if (want > cstride * (size_t)h->num_cu) want = cstride * (size_t)h->num_cu;
                if (hipMalloc((void**)&h->rc_ws, want * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
                h->rc_ws_doubles = want;
            }
            double* nu_first = nu_out;
            int* st_first = status; int* it_first = iters;
            if (max_iter > 1 && (!nu_out || !status || !iters)) {
                if ((size_t)batch > h->rc_cap) {
                    (void)hipDeviceSynchronize();
                    if (h->rc_nu) (void)hipFree(h->rc_nu);
                    if (h->rc_si) (void)hipFree(h->rc_si);
                    h->rc_nu = nullptr; h->rc_si = nullptr; h->rc_cap = 0;
                    if (hipMalloc((void**)&h->rc_nu, (size_t)batch * h->nb * h->n * sizeof(double)) != hipSuccess ||
                        hipMalloc((void**)&h->rc_si, 2 * (size_t)batch * sizeof(int)) != hipSuccess) return FMPC_E_ALLOC;
                    h->rc_cap = batch;
                }
                if (!nu_out) nu_first = h->rc_nu;
                if (!status) st_first = h->rc_si;
                if (!iters) it_first = h->rc_si + batch;
            }
            if (max_iter > 1) { const int rw = ensure_general_ws(); if (rw != FMPC_OK) return rw; }
            FrColdParams P = h->rc_P;
            P.batch = batch; P.x0 = x0; P.x0p = x0_pre; P.w = w; P.uprev = u_prev; P.nu0 = nu0;
            if (max_iter > 1 && !z_out) {                            // (cold_only was assumed but the budget is larger: cannot happen; guard)
                return FMPC_E_NULL;
            }
            P.zout = z_out; P.nuout = nu_first; P.u0out = max_iter == 1 ? u0_out : nullptr; P.status = st_first; P.iters = it_first; P.step = step;
            P.step_ld = fmpc_step_ld(n_newton); P.ws = h->rc_ws; P.ws_stride = cstride;
            if (fmpc_guard_begin(h, (hipStream_t)stream) != FMPC_OK) return FMPC_E_HIP;
            hipError_t e = fmpc_launch_ramp_cold(P, cgrid, (hipStream_t)stream);
            h->rc_last = 1;
            if (e == hipSuccess && max_iter > 1)
                e = fmpc_launch_ramp(h->dev, h->ramp_du, h->ramp_du + h->m, batch, grid, x0, x0_pre, w, u_prev, z_out,
                                     nu_first, max_iter, k, z_out, nu_out, st_first, it_first, step, fmpc_step_ld(n_newton),
                                     h->ramp_ws, stride, threads, (hipStream_t)stream, 1);
            if (e == hipSuccess && max_iter > 1 && u0_out)
                e = fmpc_launch_unpack(h->n, h->m, h->T, batch, z_out, nullptr, nullptr, u0_out, (hipStream_t)stream);
            fmpc_guard_end(h, (hipStream_t)stream);
            return e == hipSuccess ? FMPC_OK : FMPC_E_HIP;
        }
        if (z_null && !z_out) {                                      // the cold-start form is not available after all: scratch iterate
            const size_t need = (size_t)batch * h->T * (h->n + h->m);
            if (need > h->zs_doubles) {
                if (h->zs) { (void)hipDeviceSynchronize(); (void)hipFree(h->zs); h->zs = nullptr; h->zs_doubles = 0; }
                if (hipMalloc((void**)&h->zs, need * sizeof(double)) != hipSuccess) return FMPC_E_ALLOC;
                h->zs_doubles = need;
            }
            z_out = h->zs;
        }
    }
    { const int rw = ensure_general_ws(); if (rw != FMPC_OK) return rw; }
    if (fmpc_guard_begin(h, (hipStream_t)stream) != FMPC_OK) return FMPC_E_HIP;
    hipError_t e = fmpc_launch_ramp(h->dev, h->ramp_du, h->ramp_du + h->m, batch, grid, x0, x0_pre, w, u_prev, z_init,
                                    nu0, max_iter, k, z_out, nu_out, status, iters, step, fmpc_step_ld(n_newton),
                                    h->ramp_ws, stride, threads, (hipStream_t)stream);
    if (e == hipSuccess && u0_out)
        e = fmpc_launch_unpack(h->n, h->m, h->T, batch, z_out, nullptr, nullptr, u0_out, (hipStream_t)stream);
    fmpc_guard_end(h, (hipStream_t)stream);
    return e == hipSuccess ? FMPC_OK : FMPC_E_HIP;
}

extern "C" int fmpc_solve_ramp_device(fmpc_handle h, int batch,
                                      const double* x0, const double* x0_pre, const double* w, const double* u_prev,
                                      const double* z_init, const double* nu0, int n_newton, double k,
                                      double* z_out, double* nu_out, int* status, int* iters, double* step,
                                      void* stream) {
    if (!z_out) return FMPC_E_NULL;
    return fmpc_solve_ramp_device_impl(h, batch, x0, x0_pre, w, u_prev, z_init, nu0, n_newton, k, z_out, nu_out, status, iters, step, nullptr, stream);
}

extern "C" int fmpc_solve_ramp_u0_device(fmpc_handle h, int batch,
                                         const double* x0, const double* x0_pre, const double* w, const double* u_prev,
                                         const double* z_init, const double* nu0, int n_newton, double k,
                                         double* z_out, double* nu_out, int* status, int* iters, double* step,
                                         double* u0_out, void* stream) {
    if (!u0_out) return FMPC_E_NULL;
    return fmpc_solve_ramp_device_impl(h, batch, x0, x0_pre, w, u_prev, z_init, nu0, n_newton, k, z_out, nu_out, status, iters, step, u0_out, stream);
}

// host-pointer solve; u_prev != NULL selects the ramp-rate path
// Host-pointer entries (fmpc_solve, fmpc_solve_ramp, fmpc_solve_once): stage in, solve, stage out.
// One device block holds the inputs that are present, then the outputs that are asked for.  Up to FMPC_PIN_LIMIT bytes (the
// literal drop-in call: ONE problem per timestep, README.md:548-556) the block has a PINNED host twin kept by the handle: the
// inputs are packed into it and go up in ONE asynchronous copy, the outputs come back in ONE, and the call waits once -- two
// transfers and one synchronisation where round 3 issued up to six blocking hipMemcpy each way (163 us per call).  Larger
// batches are bandwidth-bound and copy straight between the caller's arrays and the device block.
#define FMPC_PIN_LIMIT ((size_t)1 << 20)
#define FMPC_ZC_LIMIT ((size_t)128 << 10)      /* calls this small skip the copies: the kernels work on the pinned block itself */
static int fmpc_solve_host(fmpc_handle h, int batch,
                           const double* x0, const double* x0_pre, const double* w, const double* u_prev,
                           const double* z_init, const double* nu0,
                           int n_newton, double k,
                           double* z_out, double* nu_out, int* status, int* iters, double* step, double* u0_out = nullptr) {
    if (!h || !x0 || (!z_out && !u0_out)) return FMPC_E_NULL;
    if (u0_out && u_prev) return FMPC_E_UNSUPPORTED;                // (first-move output: not with the ramp-rate rows)
    if (batch < 0) return FMPC_E_DIM;
    if (batch == 0) return FMPC_OK;
    if (hipSetDevice(h->device) != hipSuccess) return FMPC_E_HIP;
    std::lock_guard<std::mutex> host_lk(h->host_mu);
    const size_t n = h->n, Nz = (size_t)h->T * (h->n + h->m), nbn = (size_t)h->nb * h->n;
    const size_t Tn = (size_t)h->T * h->n, sld = fmpc_step_ld(n_newton), B = batch;
    size_t off = 0;
    auto take = [&](bool present, size_t bytes) { if (!present) return (size_t)0; size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_x0 = take(true, n * B * 8), o_x0p = take(x0_pre != nullptr, n * B * 8), o_w = take(w != nullptr, Tn * B * 8);
    const size_t o_zi = take(z_init != nullptr, Nz * B * 8), o_nu0 = take(nu0 != nullptr, nbn * B * 8);
    const size_t o_up = take(u_prev != nullptr, (size_t)h->m * B * 8);
    const size_t in_bytes = off;
    const size_t o_out = off;                                       // outputs from here on: one copy down
    const size_t o_z = take(z_out != nullptr, Nz * B * 8), o_nu = take(nu_out != nullptr, nbn * B * 8);
    const size_t o_u0 = take(u0_out != nullptr, (size_t)h->m * B * 8);
    const size_t o_st = take(true, B * 4), o_it = take(true, B * 4), o_step = take(step != nullptr, sld * B * 8);
    const bool pinned = off <= FMPC_PIN_LIMIT;
    char* base;
    char* pin = nullptr;
    char* pin_dev = nullptr;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        if (off > h->stage_bytes) {
            if (h->stage) { (void)hipDeviceSynchronize(); (void)hipFree(h->stage); h->stage = nullptr; h->stage_bytes = 0; }
            if (hipMalloc(&h->stage, off) != hipSuccess) return FMPC_E_ALLOC;
            h->stage_bytes = off;
        }
        base = (char*)h->stage;
        if (pinned) {
            if (off > h->pin_bytes) {
                if (h->pin) { (void)hipDeviceSynchronize(); (void)hipHostFree(h->pin); h->pin = nullptr; h->pin_bytes = 0; }
                size_t want = 64 * 1024;
                while (want < off) want *= 2;
                h->pin_dev = nullptr;
                if (hipHostMalloc(&h->pin, want, hipHostMallocDefault) == hipSuccess) {
                    h->pin_bytes = want;
                    if (hipHostGetDevicePointer(&h->pin_dev, h->pin, 0) != hipSuccess) h->pin_dev = nullptr;   // (the block as the kernels see it)
                } else h->pin = nullptr;                            // (no pinned memory: the blocking copies below still work)
            }
            pin = (char*)h->pin;
            pin_dev = (char*)h->pin_dev;
        }
    }
    // A call of a few problems (the reference's per-timestep call is ONE) is all latency: the two copies and their
    // synchronisation are 23 of its 54 us (round 5, scripts/once_latency.py).  Up to FMPC_ZC_LIMIT bytes the kernels read the
    // inputs straight from the pinned block (it is device-accessible) -- no copy up -- and, when the iterate is written once
    // (cold start, Newton budget 1: no kernel uses z as its working storage except for a flagged problem), write the outputs
    // straight into it -- no copy down.  FMPC_NO_ZEROCOPY=1: the copies as before (A/B).
    static const bool no_zc = [] { const char* e = getenv("FMPC_NO_ZEROCOPY"); return e && e[0] == '1'; }();
    const bool zc_in = pin != nullptr && pin_dev != nullptr && !no_zc && off <= FMPC_ZC_LIMIT;
    const bool zc_out = zc_in && n_newton == 1 && z_init == nullptr && u_prev == nullptr;
    char* const base_in = zc_in ? pin_dev : base;
    char* const base_o = zc_out ? pin_dev : base;
    auto dptr = [&](const void* src, size_t o) -> const double* { return src ? (const double*)(base_in + o) : nullptr; };
    if (pin) {
        auto pack = [&](size_t o, const double* src, size_t cnt) { if (src) memcpy(pin + o, src, cnt * 8); };
        pack(o_x0, x0, n * B); pack(o_x0p, x0_pre, n * B); pack(o_w, w, Tn * B); pack(o_zi, z_init, Nz * B); pack(o_nu0, nu0, nbn * B);
        pack(o_up, u_prev, (size_t)h->m * B);
        if (!zc_in && hipMemcpyAsync(base, pin, in_bytes, hipMemcpyHostToDevice, nullptr) != hipSuccess) return FMPC_E_HIP;
    } else {
        auto up = [&](size_t o, const double* src, size_t cnt) { return !src || hipMemcpy(base + o, src, cnt * 8, hipMemcpyHostToDevice) == hipSuccess; };
        if (!up(o_x0, x0, n * B) || !up(o_x0p, x0_pre, n * B) || !up(o_w, w, Tn * B) || !up(o_zi, z_init, Nz * B) || !up(o_nu0, nu0, nbn * B) ||
            !up(o_up, u_prev, (size_t)h->m * B)) return FMPC_E_HIP;
    }
    int* d_st = (int*)(base_o + o_st);
    int* d_it = (int*)(base_o + o_it);
    double* d_nu = nu_out ? (double*)(base_o + o_nu) : nullptr;
    double* d_step = step ? (double*)(base_o + o_step) : nullptr;
    fmpc_tl_contiguous_z = 1;                                   // (the staging block's rows are contiguous: fmpc_set_z_ld is for device batches)
    double* d_z = z_out ? (double*)(base_o + o_z) : nullptr;
    int rc = u_prev ? fmpc_solve_ramp_device(h, batch, dptr(x0, o_x0), dptr(x0_pre, o_x0p), dptr(w, o_w), dptr(u_prev, o_up), dptr(z_init, o_zi),
                                             dptr(nu0, o_nu0), n_newton, k, d_z, d_nu, d_st, d_it, d_step, nullptr)
            : u0_out ? fmpc_solve_u0_device(h, batch, dptr(x0, o_x0), dptr(x0_pre, o_x0p), dptr(w, o_w), dptr(z_init, o_zi), dptr(nu0, o_nu0),
                                            n_newton, k, d_z, d_nu, d_st, d_it, d_step, (double*)(base_o + o_u0), nullptr)
                    : fmpc_solve_device(h, batch, dptr(x0, o_x0), dptr(x0_pre, o_x0p), dptr(w, o_w), dptr(z_init, o_zi), dptr(nu0, o_nu0),
                                        n_newton, k, d_z, d_nu, d_st, d_it, d_step, nullptr);
    fmpc_tl_contiguous_z = 0;
    if (rc != FMPC_OK) { (void)hipDeviceSynchronize(); return rc; }
    std::vector<int> stv;
    const int* st;
    if (pin) {
        if (!zc_out && hipMemcpyAsync(pin + o_out, base + o_out, off - o_out, hipMemcpyDeviceToHost, nullptr) != hipSuccess) return FMPC_E_HIP;
        if (hipStreamSynchronize(nullptr) != hipSuccess) return FMPC_E_HIP;
        if (z_out) memcpy(z_out, pin + o_z, Nz * B * 8);
        if (u0_out) memcpy(u0_out, pin + o_u0, (size_t)h->m * B * 8);
        if (nu_out) memcpy(nu_out, pin + o_nu, nbn * B * 8);
        if (iters) memcpy(iters, pin + o_it, B * 4);
        if (step) memcpy(step, pin + o_step, sld * B * 8);
        st = (const int*)(pin + o_st);
    } else {
        if (hipDeviceSynchronize() != hipSuccess) return FMPC_E_HIP;
        stv.resize(B);
        if (z_out && hipMemcpy(z_out, base + o_z, Nz * B * 8, hipMemcpyDeviceToHost) != hipSuccess) return FMPC_E_HIP;
        if (u0_out && hipMemcpy(u0_out, base + o_u0, (size_t)h->m * B * 8, hipMemcpyDeviceToHost) != hipSuccess) return FMPC_E_HIP;
        if (nu_out && hipMemcpy(nu_out, base + o_nu, nbn * B * 8, hipMemcpyDeviceToHost) != hipSuccess) return FMPC_E_HIP;
        if (hipMemcpy(stv.data(), d_st, B * 4, hipMemcpyDeviceToHost) != hipSuccess) return FMPC_E_HIP;
        if (iters && hipMemcpy(iters, d_it, B * 4, hipMemcpyDeviceToHost) != hipSuccess) return FMPC_E_HIP;
        if (step && hipMemcpy(step, base + o_step, sld * B * 8, hipMemcpyDeviceToHost) != hipSuccess) return FMPC_E_HIP;
        st = stv.data();
    }
    int worst = FMPC_OK;
    for (size_t i = 0; i < B; ++i) {
        if (status) status[i] = st[i];
        if (st[i] < 0) { if (worst >= 0 || st[i] < worst) worst = st[i]; }
        else if (worst >= 0 && st[i] > worst) worst = st[i];
    }
    return worst;
}

extern "C" int fmpc_solve(fmpc_handle h, int batch,
                          const double* x0, const double* x0_pre, const double* w,
                          const double* z_init, const double* nu0,
                          int n_newton, double k,
                          double* z_out, double* nu_out, int* status, int* iters, double* step) {
    return fmpc_solve_host(h, batch, x0, x0_pre, w, nullptr, z_init, nu0, n_newton, k, z_out, nu_out, status, iters, step);
}

// fmpc_solve with HOST pointers that returns the first moves u0 = z(1:m) (README.md:589: the caller applies U(1:nu) only): m x batch
// doubles come back instead of N_z x batch (2.3 MB instead of 82 MB per 2000 problems at (27,144,30)); z_out may be given as well.
extern "C" int fmpc_solve_u0(fmpc_handle h, int batch,
                             const double* x0, const double* x0_pre, const double* w,
                             const double* z_init, const double* nu0,
                             int n_newton, double k,
                             double* z_out, double* u0_out, int* status, int* iters) {
    if (!u0_out) return FMPC_E_NULL;
    return fmpc_solve_host(h, batch, x0, x0_pre, w, nullptr, z_init, nu0, n_newton, k, z_out, nullptr, status, iters, nullptr, u0_out);
}

extern "C" int fmpc_solve_ramp(fmpc_handle h, int batch,
                               const double* x0, const double* x0_pre, const double* w, const double* u_prev,
                               const double* z_init, const double* nu0,
                               int n_newton, double k,
                               double* z_out, double* nu_out, int* status, int* iters, double* step) {
    if (!u_prev) return FMPC_E_NULL;
    return fmpc_solve_host(h, batch, x0, x0_pre, w, u_prev, z_init, nu0, n_newton, k, z_out, nu_out, status, iters, step);
}

extern "C" int fmpc_unpack_device(fmpc_handle h, int batch, const double* z, double* U, double* X,
                                  double* u0, void* stream) {
    if (!h || !z) return FMPC_E_NULL;
    if (batch < 0) return FMPC_E_DIM;
    if (batch == 0) return FMPC_OK;
    if (hipSetDevice(h->device) != hipSuccess) return FMPC_E_HIP;
    return fmpc_launch_unpack(h->n, h->m, h->T, batch, z, U, X, u0, (hipStream_t)stream) == hipSuccess
               ? FMPC_OK : FMPC_E_HIP;
}

extern "C" int fmpc_loop_inputs_device(fmpc_handle h, int batch, const double* a_k, const double* x0_last,
                                       const double* u1, const double* u2,
                                       double* x0, double* x0_pre, double* w, void* stream) {
    if (!h || !a_k || !x0 || !x0_pre || !w) return FMPC_E_NULL;
    if (batch < 0) return FMPC_E_DIM;
    if (batch == 0) return FMPC_OK;
    if (hipSetDevice(h->device) != hipSuccess) return FMPC_E_HIP;
    return fmpc_launch_loop_inputs(h->n, h->m, h->T, batch, h->dev.Bt, h->loop_M1, h->loop_M2, a_k, x0_last, u1, u2,
                                   x0, x0_pre, w, (hipStream_t)stream) == hipSuccess ? FMPC_OK : FMPC_E_HIP;
}

hipError_t fmpc_launch_phase_residual(int batch, size_t npx, int n, int m, const double* Bt, const double* phase, const double* u,
                                      const double* Z, double* out, hipStream_t stream);
// Residual phase screens of a timestep: README.md:453, 590-601 (see include/fastmpc.h).
extern "C" int fmpc_phase_residual_device(fmpc_handle h, int batch, long long npx, const double* phase, const double* u_prev,
                                          const double* Z, double* out, void* stream) {
    if (!h || !phase || !out || (u_prev && !Z)) return FMPC_E_NULL;
    if (batch < 0 || npx < 0) return FMPC_E_DIM;
    if (batch == 0 || npx == 0) return FMPC_OK;
    if (h->n > 32) return FMPC_E_UNSUPPORTED;
    if (hipSetDevice(h->device) != hipSuccess) return FMPC_E_HIP;
    return fmpc_launch_phase_residual(batch, (size_t)npx, h->n, h->m, h->dev.Bt, phase, u_prev, Z, out, (hipStream_t)stream) == hipSuccess
               ? FMPC_OK : FMPC_E_HIP;
}

extern "C" int fmpc_unpack(fmpc_handle h, int batch, const double* z, double* U, double* X, double* u0) {
    if (!h || !z) return FMPC_E_NULL;
    if (batch < 0) return FMPC_E_DIM;
    if (batch == 0) return FMPC_OK;
    if (hipSetDevice(h->device) != hipSuccess) return FMPC_E_HIP;
    const size_t B = batch, Nz = (size_t)h->T * (h->n + h->m), Tm = (size_t)h->T * h->m, Tn = (size_t)h->T * h->n;
    double *dz = nullptr, *dU = nullptr, *dX = nullptr, *du0 = nullptr;
    int rc = FMPC_OK;
    if (hipMalloc((void**)&dz, Nz * B * 8) != hipSuccess) return FMPC_E_ALLOC;
    if ((U && hipMalloc((void**)&dU, Tm * B * 8) != hipSuccess) ||
        (X && hipMalloc((void**)&dX, Tn * B * 8) != hipSuccess) ||
        (u0 && hipMalloc((void**)&du0, h->m * B * 8) != hipSuccess)) rc = FMPC_E_ALLOC;
    if (rc == FMPC_OK && hipMemcpy(dz, z, Nz * B * 8, hipMemcpyHostToDevice) != hipSuccess) rc = FMPC_E_HIP;
    if (rc == FMPC_OK) rc = fmpc_unpack_device(h, batch, dz, dU, dX, du0, nullptr);
    if (rc == FMPC_OK && hipDeviceSynchronize() != hipSuccess) rc = FMPC_E_HIP;
    if (rc == FMPC_OK && U && hipMemcpy(U, dU, Tm * B * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = FMPC_E_HIP;
    if (rc == FMPC_OK && X && hipMemcpy(X, dX, Tn * B * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = FMPC_E_HIP;
    if (rc == FMPC_OK && u0 && hipMemcpy(u0, du0, h->m * B * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = FMPC_E_HIP;
    (void)hipFree(dz);
    if (dU) (void)hipFree(dU);
    if (dX) (void)hipFree(dX);
    if (du0) (void)hipFree(du0);
    return rc;
}

// The reference rebuilds its (value-class) object at every timestep (README.md:548) with the same model; doing the same
// here would allocate, upload and -- on the panel path -- factor again at every call.  fmpc_solve_once therefore keeps
// the handles of the last few models it has seen, keyed on the exact bytes of every model argument.
namespace {
std::mutex once_mu;
FmpcLru<fmpc_handle> once_cache(4);                            // (fmpc_host.h) the last 4 distinct models
}  // namespace

extern "C" int fmpc_solve_once_cache_clear(void) {
    std::lock_guard<std::mutex> lk(once_mu);
    once_cache.clear([](fmpc_handle hh) { fmpc_destroy(hh); });
    return FMPC_OK;
}

extern "C" int fmpc_solve_once(int n, int m, int T, int var_order,
                               const double* Q, const double* R, const double* S, const double* Qf,
                               const double* q, const double* r, const double* qf,
                               const double* x_min, const double* x_max,
                               const double* u_min, const double* u_max,
                               const double* du_min, const double* du_max,
                               const double* x0, const double* x0_pre, const double* u_prev,
                               const double* A1, const double* A2, const double* B,
                               const double* w, const double* xf, const double* x_init,
                               const double* nu0, int nw, double k, int device,
                               double* x_opt, int* iters) {
    (void)S;                                                  // stored, never used (Fast_MPC2.m:33)
    // VAR_2 ignores du_min, du_max, u_prev (its ramp rows are commented out, D8); VAR_1 builds ramp rows from them
    // (VAR_1/fast_mpc_ineq_const.m:58-76)
    const bool ramp = var_order == 1 && du_min && du_max && u_prev;
    if (!x0 || (var_order == 2 && !x0_pre)) return FMPC_E_DIM;   // fast_mpc_eq_const.m:27-30
    if (n <= 0 || m <= 0 || T <= 0 || (var_order != 1 && var_order != 2)) return FMPC_E_DIM;
    if (!A1 || !B || (var_order == 2 && !A2)) return FMPC_E_NULL;
    if (!Q || !R || !Qf || !x_min || !x_max || !u_min || !u_max) return FMPC_E_NULL;
    std::lock_guard<std::mutex> lk(once_mu);                  // (also keeps an entry alive while it is in use)
    // The model of a call is compared IN PLACE against the stored keys (same record format as fmpc_host_key_push: a count or -1,
    // then the values): a hit -- every call but the first of a simulation -- builds no key (round 3 copied ~220 KB per call,
    // most of it the m x m matrix R, before comparing it).
    const double dims[5] = {(double)n, (double)m, (double)T, (double)var_order, (double)device};
    const size_t nn = (size_t)n * n;
    struct Seg { const double* p; size_t cnt; };
    const Seg segs[15] = {{dims, 5}, {A1, nn}, {var_order == 2 ? A2 : nullptr, nn}, {B, (size_t)n * m}, {Q, nn}, {R, (size_t)m * m}, {Qf, nn},
                          {q, (size_t)n}, {r, (size_t)m}, {qf, (size_t)n}, {x_min, (size_t)n}, {x_max, (size_t)n}, {u_min, (size_t)m},
                          {u_max, (size_t)m}, {xf, (size_t)n}};
    size_t key_len = 0;
    for (const Seg& sg : segs) key_len += 1 + (sg.p ? sg.cnt : 0);
    FmpcLru<fmpc_handle>::Entry* ent = nullptr;
    for (auto& e : once_cache.items) {
        if (e.key.size() != key_len) continue;
        const double* kp = e.key.data();
        bool same = true;
        for (const Seg& sg : segs) {
            if (*kp++ != (sg.p ? (double)sg.cnt : -1.0)) { same = false; break; }
            if (sg.p) { if (memcmp(kp, sg.p, sg.cnt * sizeof(double)) != 0) { same = false; break; } kp += sg.cnt; }
        }
        if (same) { ent = &e; break; }
    }
    int rc;
    if (!ent) {
        std::vector<double> key;
        key.reserve(key_len);
        for (const Seg& sg : segs) fmpc_host_key_push(key, sg.p, sg.cnt);
        fmpc_handle h = nullptr;
        rc = fmpc_create(&h, n, m, T, var_order, A1, A2, B, Q, R, Qf, q, r, qf, x_min, x_max, u_min, u_max, xf, device);
        if (rc != FMPC_OK) return rc;
        ent = once_cache.insert(std::move(key), h, [](fmpc_handle hh) { fmpc_destroy(hh); });   // evicts the least recently used model
    }
    once_cache.touch(ent);
    int st = 0;
    if (ramp) {
        // (the README shifts du_min/du_max by u_prev at every step, README.md:543-544: the bounds are per-call data)
        std::vector<double> rb(du_min, du_min + m);
        rb.insert(rb.end(), du_max, du_max + m);
        rc = FMPC_OK;
        if (rb != ent->ramp) { rc = fmpc_set_ramp(ent->h, du_min, du_max); if (rc == FMPC_OK) ent->ramp = rb; }
        if (rc == FMPC_OK) rc = fmpc_solve_ramp(ent->h, 1, x0, x0_pre, w, u_prev, x_init, nu0, nw, k, x_opt, nullptr, &st, iters, nullptr);
    } else {
        rc = fmpc_solve(ent->h, 1, x0, x0_pre, w, x_init, nu0, nw, k, x_opt, nullptr, &st, iters, nullptr);
    }
    return rc;
}

// VAR(2) identification (README.md:108-130) on the device; see include/fastmpc.h
extern "C" int fmpc_var_identify_device(int n, int num_train, int num_samples, int batch, const double* series,
                                        double* A1, double* A2, int* status, void* stream) {
    if (!series || !A1 || !A2) return FMPC_E_NULL;
    if (n <= 0 || batch < 0 || num_train < 2 * n + 2 || num_samples < num_train) return FMPC_E_DIM;   // fewer rows than unknowns: singular
    if (n > 32) return FMPC_E_UNSUPPORTED;
    if (batch == 0) return FMPC_OK;
    return fmpc_launch_var_identify(n, num_train, num_samples, batch, series, A1, A2, status, (hipStream_t)stream) == hipSuccess
               ? FMPC_OK : FMPC_E_HIP;
}
