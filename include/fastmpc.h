/*
 * fastmpc.h -- C ABI of the MI355X-native fastMPC inner solver.
 *
 * This is the drop-in boundary for ONE path of jinsungkim96/MPC-SensorlessAO: the call
 *     obj   = Fast_MPC2(Q,R,S,Qf,q,r,qf,xmin,xmax,umin,umax,dumin,dumax,T,x0,x0_pre,u_prev,
 *                       A1,A2,B,w,xf,x_init)              Fast_MPC/VAR_2/Fast_MPC2.m:28-55
 *     x_opt = obj.mpc_fixed_log_newton(nw,k)              Fast_MPC/VAR_2/Fast_MPC2.m:124-130
 * as issued once per timestep by the notebook loop (README.md:548,555), plus the caller-side
 * unpack of x_opt (README.md:558-570,589).  Everything behind that call -- fast_mpc_init.m,
 * fast_mpc_objective.m, fast_mpc_eq_const.m, fast_mpc_ineq_const.m, inf_newton_KKT_H.m,
 * inf_newton_solver.m, backtracking_inf_newton.m -- runs as HIP kernels on gfx950.
 *
 * Conventions
 *   - All matrices are fp64 COLUMN-MAJOR with leading dimension = rows, exactly as MATLAB
 *     stores them (M(r,c) = M[r + c*rows]).  "n x batch" arrays are therefore one contiguous
 *     n-vector per problem.
 *   - z layout is the reference's interleaved vector [u0;x1;u1;x2;...;u_{T-1};x_T]
 *     (fast_mpc_init.m:22-25), N_z = T*(n+m).
 *   - nu has `fmpc_nu_len()` = n*(T + (xf given ? 1 : 0)) entries (length(b),
 *     inf_newton_solver.m:2).
 *   - Caller owns every buffer.  The handle copies the shared model to the device once; the model is
 *     immutable afterwards.  No pointer passed to a solve call is retained.
 *   - A handle may be used from several threads and streams: its device workspaces serve one solve at
 *     a time, so the library orders the solves of ONE handle on the device (a solve enqueued on
 *     another stream than the handle's previous one first waits for that one).  Solves that should
 *     overlap need a handle each (mpc-sensorlessao_amd/lanes.py).
 *   - Functions return FMPC_OK (0), a negative error, or a positive warning; nothing throws.
 *   - There is NO CPU fallback: without a usable HIP device fmpc_create fails with
 *     FMPC_E_NO_DEVICE / FMPC_E_HIP.
 */
#ifndef FASTMPC_H
#define FASTMPC_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMPC_VERSION 100  /* 0.1.0 */

/* status codes */
#define FMPC_OK               0
#define FMPC_W_LINESEARCH     1   /* backtracking collapsed to t = 0 (reference: t underflows,
                                     backtracking_inf_newton.m:3-8)         */
#define FMPC_E_NULL          -1   /* required pointer missing (fast_mpc_eq_const.m:19-25)     */
#define FMPC_E_DIM           -2   /* size mismatch (fast_mpc_eq_const.m:27-32,
                                     fast_mpc_ineq_const.m:4-9, fast_mpc_objective.m:17-47,
                                     fast_mpc_init.m:13-14)                                    */
#define FMPC_E_UNSUPPORTED   -3   /* valid for the reference, not implemented on the device
                                     yet (n > 79, sizes beyond the LDS budget, dense R with n > 47) */
#define FMPC_E_NOT_PD_PHI    -4   /* chol(KKT_H) would fail (inf_newton_solver.m:24)          */
#define FMPC_E_NOT_PD_SCHUR  -5   /* chol(Schur) would fail (inf_newton_solver.m:30)          */
#define FMPC_E_HIP           -6   /* HIP runtime error                                         */
#define FMPC_E_ALLOC         -7
#define FMPC_E_NO_DEVICE     -8   /* no HIP device: this library has no CPU path              */

typedef struct fmpc_handle_s* fmpc_handle;

int         fmpc_version(void);
const char* fmpc_strerror(int code);

/*
 * Shared model ("handle").  Replaces the property copy of the Fast_MPC2 constructor for
 * everything that does not change between timesteps (Fast_MPC2.m:30-54) and the constant part
 * of the assembly (fast_mpc_objective.m:50-65, fast_mpc_eq_const.m:38-49,
 * fast_mpc_ineq_const.m:46-56).
 *   var_order  2: VAR(2) (Fast_MPC/VAR_2).  1: VAR(1) intended dynamics = VAR_2 code with
 *              A2 = 0 (A2 may be NULL); ramp-rate rows of VAR_1 are not built.
 *   n          any.  Specialised kernels: n = 27 (the AO configuration), n <= 79 in fp64 and n <= 111 with the fp32 factor + fp64
 *              residuals on the matrix cores (default: fp64; fmpc_set_precision for the fp32 factor); every other size (the reference checks shapes only,
 *              fast_mpc_objective.m:17-47) is solved in fp64 by the generic kernel with its tiles in the HBM workspace -- a size
 *              fallback without a speed claim (tests/test_gpu_any_size.py: n = 83 .. 140).
 *   Q,R,Qf     n x n, m x m, n x n (fast_mpc_objective.m:51-55): any symmetric positive definite matrices, at any size.  Dense Q
 *              or Qf: the tiled kernel (n <= 47 in fp64, <= 79 with the fp32 factor), beyond it the workspace instance.  Dense
 *              R: the u block of Phi is then a dense m x m matrix per stage and Newton step (inf_newton_KKT_H.m:13), factored
 *              per stage (ft_dense_r) -- in LDS by the tiled kernel in fp64 (n <= 47, m (m + 1) / 2 + m (n + 2) doubles:
 *              m = 144 fits), in the workspace beyond; a generality path, ~13 x slower than a diagonal R.  Ramp rows
 *              (fmpc_set_ramp) need diagonal weights and n <= 64.
 *              Not positive definite / not symmetric: FMPC_E_NOT_PD_PHI (the reference's chol(KKT_H) error,
 *              inf_newton_solver.m:24).
 *   q,r,qf     NULL = zeros (fast_mpc_objective.m:26-47).
 *   x_min/max  only used for the cold start (state bounds are not constraints,
 *              fast_mpc_ineq_const.m:25-40).
 *   xf         NULL = no terminal equality (fast_mpc_eq_const.m:67-71).
 *   device     HIP device ordinal (>= 0).
 */
int fmpc_create(fmpc_handle* out, int n, int m, int T, int var_order,
                const double* A1, const double* A2, const double* B,
                const double* Q, const double* R, const double* Qf,
                const double* q, const double* r, const double* qf,
                const double* x_min, const double* x_max,
                const double* u_min, const double* u_max,
                const double* xf, int device);
int fmpc_destroy(fmpc_handle h);

int fmpc_dims(fmpc_handle h, int* n, int* m, int* T, int* nz, int* nu_len);

/*
 * One inf_newton_solver call per problem (inf_newton_solver.m:1-43) for `batch` independent
 * problems; batch = 1 is the reference call.  HOST buffers; synchronous.
 *   x0, x0_pre  n x batch (x0_pre may be NULL = zeros; ignored for var_order 1)
 *   w           (T*n) x batch, NULL = zeros (superset of fast_mpc_eq_const.m:33-35, D7)
 *   z_init      N_z x batch, NULL = cold start at mid-box (fast_mpc_init.m:19-25)
 *   nu0         nu_len x batch, NULL = zeros.  The reference draws nu = rand(...)
 *               (inf_newton_solver.m:2); the caller draws it and passes it here.
 *   n_newton    fixed Newton-step count; <= 0 means the reference's nw = [] mode:
 *               at most 1000 iterations (inf_newton_solver.m:4-8).
 *               The tolerance exit (:19-22) is active in both modes, as in the reference.
 *   k           barrier weight
 *   z_out       N_z x batch
 *   nu_out      nu_len x batch, nullable
 *   status      batch, nullable: per-problem code
 *   iters       batch, nullable: Newton steps taken
 *   step        step_ld x batch, nullable: accepted t of every Newton step (unused tail = -1);
 *               step_ld = fmpc_step_ld(n_newton)
 * Returns the worst per-problem status (most negative error, else largest warning).
 * Staging: the inputs are packed into one pinned block, copied up once, the outputs copied down once.  A call of up to
 * 128 KB in all (the reference's per-timestep call: one problem) skips both copies -- the kernels read the pinned block and,
 * from the cold start with a budget of 1, write z into it directly (round 5: 51 -> 43 us per call; FMPC_NO_ZEROCOPY=1 for A/B).
 */
int fmpc_solve(fmpc_handle h, int batch,
               const double* x0, const double* x0_pre, const double* w,
               const double* z_init, const double* nu0,
               int n_newton, double k,
               double* z_out, double* nu_out, int* status, int* iters, double* step);

int fmpc_step_ld(int n_newton);

/*
 * Same call with DEVICE pointers, asynchronous on `stream` (a hipStream_t passed as void*;
 * NULL = the default stream).  `status`/`iters` are device int arrays.  The return value only
 * covers argument and launch errors; per-problem codes are in `status`.
 */
int fmpc_solve_device(fmpc_handle h, int batch,
                      const double* x0, const double* x0_pre, const double* w,
                      const double* z_init, const double* nu0,
                      int n_newton, double k,
                      double* z_out, double* nu_out, int* status, int* iters, double* step,
                      void* stream);

/*
 * fmpc_solve_device that also leaves the first move u0 = z(1:m) of every problem (u_prev = U(1:nu), README.md:589)
 * in u0_out (m x batch): the solve and the one output a closed loop needs in ONE call.  On the n = 27 paths the last
 * kernel of the solve writes it (no extra launch); otherwise it is fmpc_solve_device + fmpc_unpack_device.
 * Output options: z_out may be NULL -- the caller of the reference only applies U(1:nu) (README.md:558-570,589).  The
 * first moves, status, iters and step are then exactly those of the call with z_out given; on the cold-start panel path
 * with n_newton = 1 and nu_out = NULL nothing of z is written at all (41 KB per problem at (27,144,30) become 1.1 KB),
 * on every other path the iterate lives in a scratch array of the handle.
 */
int fmpc_solve_u0_device(fmpc_handle h, int batch,
                         const double* x0, const double* x0_pre, const double* w,
                         const double* z_init, const double* nu0, int n_newton, double k,
                         double* z_out, double* nu_out, int* status, int* iters, double* step,
                         double* u0_out, void* stream);

/*
 * fmpc_solve_u0_device with the distance between the z rows of consecutive problems as an ARGUMENT of the call: ldz doubles
 * (0 = contiguous rows, else >= N_z; see fmpc_set_z_ld for what padded rows buy and which solves take them).  The handle's
 * persistent fmpc_set_z_ld value is neither read nor changed, so concurrent solves on one handle may use different layouts.
 * u0_out may be NULL here (then it is fmpc_solve_device with an explicit ldz).
 */
int fmpc_solve_u0_device_ld(fmpc_handle h, int batch,
                            const double* x0, const double* x0_pre, const double* w,
                            const double* z_init, const double* nu0, int n_newton, double k,
                            double* z_out, double* nu_out, int* status, int* iters, double* step,
                            double* u0_out, int ldz, void* stream);

/*
 * fmpc_solve (HOST pointers) returning the first moves: u0_out (m x batch) receives u0 = z(1:m) of every problem -- all the
 * reference's loop applies (u_prev = U(1:nu), README.md:589; de-interleave README.md:558-570).  z_out may be NULL: then
 * m x batch doubles come back over PCIe instead of N_z x batch (2.3 MB instead of 82 MB per 2000 problems at (27,144,30))
 * and the cold-start step writes nothing of z on the device either.  status, iters nullable.  Same return value as fmpc_solve.
 */
int fmpc_solve_u0(fmpc_handle h, int batch,
                  const double* x0, const double* x0_pre, const double* w,
                  const double* z_init, const double* nu0, int n_newton, double k,
                  double* z_out, double* u0_out, int* status, int* iters);

/*
 * Caller-side unpack of x_opt (README.md:558-570) and u_prev = U(1:nu) (README.md:589):
 * z (N_z x batch) -> U (T*m x batch), X (T*n x batch), u0 (m x batch); any output may be NULL.
 */
int fmpc_unpack(fmpc_handle h, int batch, const double* z, double* U, double* X, double* u0);
int fmpc_unpack_device(fmpc_handle h, int batch, const double* z, double* U, double* X,
                       double* u0, void* stream);

/*
 * One-shot form taking the reference's full 23-argument constructor set
 * (Fast_MPC/VAR_2/Fast_MPC2.m:28-29) plus (nw, k) of mpc_fixed_log_newton (:124) and nu0.
 * S and x_min/x_max (beyond the cold start) are accepted and ignored as in the reference.
 * du_min, du_max, u_prev: ignored for var_order 2 (the VAR_2 ramp rows are commented out,
 * VAR_2/fast_mpc_ineq_const.m:61-79); for var_order 1 they add the ramp-rate rows of
 * VAR_1/fast_mpc_ineq_const.m:58-76 when all three are given (NULL: box rows only).
 * Empty MATLAB arguments are NULL.  x_opt: N_z.  var_order 1 ignores x0_pre/A2.
 */
int fmpc_solve_once(int n, int m, int T, int var_order,
                    const double* Q, const double* R, const double* S, const double* Qf,
                    const double* q, const double* r, const double* qf,
                    const double* x_min, const double* x_max,
                    const double* u_min, const double* u_max,
                    const double* du_min, const double* du_max,
                    const double* x0, const double* x0_pre, const double* u_prev,
                    const double* A1, const double* A2, const double* B,
                    const double* w, const double* xf, const double* x_init,
                    const double* nu0, int nw, double k, int device,
                    double* x_opt, int* iters);
/*
 * The reference rebuilds its object at every timestep with an unchanged model (README.md:548).  fmpc_solve_once keeps
 * the device handles of the last 4 distinct models (compared byte for byte over every model argument), so only the
 * first call with a model allocates, uploads and factors; fmpc_solve_once_cache_clear releases them.
 */
int fmpc_solve_once_cache_clear(void);

/*
 * Coefficient-space closed loop: the steps either side of the solver in the reference's simulation loop
 * (README.md:482-497, 589-590; M1, M2 of MPC_DesignMatrices, main.mlx "System Matrix design").  From the
 * turbulence coefficients a_k of the current step and the two previous first moves it produces the inputs of
 * the next solve:
 *     x0     = a_k + B u1                  residual after the mirror's correction ad_cor = B u_prev (u1 NULL: a_k)
 *     x0_pre = x0_last                     (NULL: zeros, the first step)
 *     w      = b_ref = -M1 B u1 - M2 B u2  (a NULL input drops its term: steps 1 and 2 of the loop)
 * Device pointers, column-major: a_k, x0_last, x0, x0_pre n x batch; u1, u2 m x batch; w T n x batch.
 * x0 may alias x0_last.  The estimator that produces the residual coefficients in the reference
 * (README.md:456-480) is out of scope: a_k is the caller's.
 */
int fmpc_loop_inputs_device(fmpc_handle h, int batch, const double* a_k, const double* x0_last,
                            const double* u1, const double* u2,
                            double* x0, double* x0_pre, double* w, void* stream);

/*
 * One closed-loop step in one call: fmpc_loop_inputs_device followed by fmpc_solve_u0_device on the inputs it produced
 * (same arguments, same results; x0, x0_pre, w are written as before).  Knowing that w = -M1 (B u1) - M2 (B u2) has
 * only 2 n degrees of freedom, the dense form of the cold-start dual solve (see fmpc_set_dense_form) takes
 * [B u1 ; B u2] in place of the T n entries of w: 28 instead of 217 k-steps per tile at (27, 144, 30), at any batch.
 * z_out may be NULL (first moves only, see fmpc_solve_u0_device).
 * x0 may alias x0_last here too.  With first moves only, a Newton budget of 1 and more than 64 realisations, a caller that
 * does NOT update x0 in place (x0_last different from x0 and from x0_pre, or NULL) gets the loop inputs, the first moves and the
 * step-length decision in ONE launch (fmpc_last_dual_form = 4; with the update in place: three launches, = 3), the same
 * results to 1e-11 (tests/test_gpu_closed_loop.py); FMPC_NO_LOOP_FUSE=1 switches the one-launch form off.
 */
int fmpc_loop_step_device(fmpc_handle h, int batch, const double* a_k, const double* x0_last,
                          const double* u1, const double* u2, double* x0, double* x0_pre, double* w,
                          const double* nu0, int n_newton, double k,
                          double* z_out, double* nu_out, int* status, int* iters, double* step,
                          double* u0_out, void* stream);

/*
 * A recorded stretch of the loop in ONE host call (the reference's simulation knows its turbulence coefficients in advance:
 * README.md:51-93 generates and fits all phase screens before the loop of :444-626 starts): `steps` consecutive
 * fmpc_loop_step_device calls, the first moves fed back on the device.  a: n x batch x steps (one n x batch slab per step),
 * nu0: nu_len x batch x steps or NULL, U0: m x batch x steps receives u[k] = U(1:nu) of every step (README.md:589), X0
 * (nullable): n x batch x steps receives the residuals x0 of every step.  u_before1 / u_before2 (nullable): the first moves of
 * the two steps before this stretch, have_x0_last != 0: x0 holds the previous step's residual (continuing an earlier stretch).
 * z is not produced (first moves only); status / iters hold the last step's values.  The host then spends one call, not one
 * per timestep: a sequential loop of a few realisations is otherwise bound by the host.
 * Up to 4096 realisations with n_newton = 1 (and the first-move form available) all steps but the last run in ONE launch: the
 * workgroup of a realisation keeps its rows of the first-move form in registers and walks through the steps (3 us per step
 * at (27, 144, 30)); a step whose step-length decision is not clear-cut ends that walk, the exact path redoes the step and the
 * walk goes on behind it -- same results as one call per step (bit for bit up to 64 realisations, where the one-step call
 * uses the same form; to rounding, 1e-13, beyond).  In that case the call SYNCHRONISES the stream
 * (the host has to see where the walks stopped): the results are complete when it returns.  The number of walks of a call is
 * capped: after 8 of them (FMPC_WALK_MAX_RESTARTS), or when more than a tenth of the realisations stop in one walk, the rest of
 * the stretch is done stepwise -- the realisations furthest behind take one step through the one-step call until all have
 * arrived: at most `steps` such calls, none synchronises -- so a stretch where most steps are not clear-cut costs about what
 * one call per step does, not a batch-wide launch per stop.
 */
int fmpc_loop_run_device(fmpc_handle h, int batch, int steps, const double* a, const double* nu0,
                         const double* u_before1, const double* u_before2, int have_x0_last,
                         int n_newton, double k, double* x0, double* x0_pre, double* w,
                         double* U0, double* X0, int* status, int* iters, void* stream);

/*
 * Ramp-rate rows of the VAR_1 variant (VAR_1/Fast_MPC2.m:26-27 arguments dumin, dumax, u_prev;
 * VAR_1/fast_mpc_ineq_const.m:58-76): per stage j   du_min <= u_j - u_{j-1} <= du_max,  u_{-1} = u_prev.
 * fmpc_set_ramp stores the bounds (m each, du_min < du_max) in the handle; fmpc_solve_ramp[_device] is
 * fmpc_solve[_device] with the extra per-problem input u_prev (m x batch).  The rows couple consecutive stages, so
 * Y = C Phi^-1 C' is dense across the horizon: this path factors a dense (T n)^2 matrix per problem and Newton
 * step (fmpc_kernel_ramp.hip); it needs diagonal Q, R, Qf like the other device paths.
 * From the COLD START (z_init == NULL, the reference loop's call: Fast_MPC2(..., x_init = []).mpc_fixed_log_newton(1, k))
 * only the ramp rows of stage 0 (u_0 - u_prev) depend on the problem -- every other ramp slack u_j - u_{j-1} is zero at the
 * mid-box start -- so the KKT matrix of the first Newton step is a CONSTANT matrix plus a diagonal term on the m entries of u_0,
 * and the step costs one m x m Cholesky factorisation per problem and two passes through constant operators built once per
 * (handle, k, bounds) on the host (Woodbury form, fmpc_ramp_cold; fmpc_last_dual_form = 5): 0.8 instead of 18.7 MFLOP at
 * (27, 144, 10).  A budget n_newton > 1 continues with the dense factorisation from the iterate that step leaves.  Same
 * results to rounding (tests/test_gpu_ramp.py: both forms against the dense oracle); FMPC_NO_RAMP_COLD=1 at create time keeps
 * the general path.
 * FMPC_E_UNSUPPORTED: fmpc_set_ramp has not been called (or n > 64).
 */
int fmpc_set_ramp(fmpc_handle h, const double* du_min, const double* du_max);
int fmpc_solve_ramp(fmpc_handle h, int batch,
                    const double* x0, const double* x0_pre, const double* w, const double* u_prev,
                    const double* z_init, const double* nu0, int n_newton, double k,
                    double* z_out, double* nu_out, int* status, int* iters, double* step);
int fmpc_solve_ramp_device(fmpc_handle h, int batch,
                           const double* x0, const double* x0_pre, const double* w, const double* u_prev,
                           const double* z_init, const double* nu0, int n_newton, double k,
                           double* z_out, double* nu_out, int* status, int* iters, double* step,
                           void* stream);
/* The ramp twin of fmpc_solve_u0_device: also leaves the first moves u0 = z(1:m) in u0_out (u_prev = U(1:nu), README.md:589).
 * From the cold start with n_newton = 1 the step's own kernel writes them (no further launch) and z_out may be NULL: nothing
 * of z is written then; otherwise z_out == NULL works in a scratch array of the handle. */
int fmpc_solve_ramp_u0_device(fmpc_handle h, int batch,
                              const double* x0, const double* x0_pre, const double* w, const double* u_prev,
                              const double* z_init, const double* nu0, int n_newton, double k,
                              double* z_out, double* nu_out, int* status, int* iters, double* step,
                              double* u0_out, void* stream);

/*
 * VAR(2) model identification (the step that produces A1, A2 for the solver; reference README.md:108-130):
 *     AA(i-2,:) = [ad_acc(i-1,:), ad_acc(i-2,:)],  BB(i-2,:) = ad_acc(i,:),  i = 3..num_train
 *     PARA = (AA'*AA) \ AA'*BB ;  A1 = PARA(1:n,:)' ;  A2 = PARA(n+1:2n,:)'
 * for `batch` coefficient series at once, on the device (Gram matrices on the fp64 matrix cores, Cholesky solve of the
 * normal equations).  Device pointers; series: per realisation n x num_samples column-major (MATLAB: ad_acc', one
 * n-vector per time step), the first num_train samples are used; A1, A2: per realisation n x n column-major;
 * status (nullable): per realisation 0 or FMPC_E_NOT_PD_SCHUR when AA'AA is not positive definite.  n <= 32.
 */
int fmpc_var_identify_device(int n, int num_train, int num_samples, int batch, const double* series,
                             double* A1, double* A2, int* status, void* stream);

/*
 * Arithmetic of the per-problem-factor path (no counterpart in the reference, which is fp64 throughout).
 *   FMPC_PREC_F64        everything in fp64: the default wherever an fp64 kernel on the matrix cores exists (n <= 79; round 5 --
 *                        the reference is fp64 throughout and the literal call fmpc_solve_once has no precision argument) and
 *                        beyond (the generic kernel's workspace instance: exact, slow).
 *   FMPC_PREC_F32_MIXED  "fp32 mixed precision" (BASELINE configs[4]): Y = C Phi^-1 C', its block Cholesky factor
 *                        (inf_newton_solver.m:27,30) and the two triangular sweeps (:31-32) in fp32 on the matrix
 *                        cores; the residuals r_d, r_p (:12-17), the right-hand side (:28-29), d_z, the line search
 *                        and the iterate z, nu stay fp64, so every Newton step refines the fp32 KKT solve of the
 *                        previous one against fp64 residuals.  On request, n <= 111 with a diagonal R: 2.2 x faster than fp64
 *                        at configs[4] (n = 65), 35 x faster than the exact fallback at n = 96; a step differs from the
 *                        fp64 one by ~1e-6.  The default only where it is the only matrix-core kernel that fits (fp64 tiles
 *                        beyond the LDS: very large m).
 * FMPC_E_UNSUPPORTED when the handle's size has no kernel of that type.
 */
#define FMPC_PREC_F64        0
#define FMPC_PREC_F32_MIXED  1
int fmpc_set_precision(fmpc_handle h, int mode);

/*
 * Diagnostic (no counterpart in the reference): which device path the last fmpc_solve[_device] call of
 * this handle took, and how many problems the panel kernel handed to the exact per-problem path because
 * their step-length / exit decision was not clear-cut.  Synchronises the device.
 *   path  0 generic kernel, 1 wave kernel (per-problem factor), 2 wave kernel (shared cold-start factor),
 *         3 panel kernel (+ exact path for `handed_over` problems), 4 ramp-rate kernel.
 */
#define FMPC_PATH_GENERIC 0
#define FMPC_PATH_WAVE    1
#define FMPC_PATH_SHARED  2
#define FMPC_PATH_PANEL   3
#define FMPC_PATH_RAMP    4
#define FMPC_PATH_TILED   5   /* tiled kernel, fp64 factor */
#define FMPC_PATH_TILED_F32 6 /* tiled kernel, fp32 factor + fp64 residuals */
int fmpc_last_dispatch(fmpc_handle h, int* path, int* handed_over);
/* Diagnostic: wavefronts per problem of the last launch of the tiled kernel on this handle (0 = none yet); the kernel runs with
 * 2, 4 or 8 depending on n, the precision and the batch (few problems: more wavefronts each). */
int fmpc_last_tiled_wavefronts(fmpc_handle h);

/* FMPC_PATH_PANEL has two forms of the cold-start dual solve nu+ = Y^-1 (ct - b) (inf_newton_solver.m:27-32 at the
 * constant start of fast_mpc_init.m:19-20): the two sweeps through the shared block factor (one CU per 16 problems,
 * a chain of ~31 dependent steps), and the DENSE FORM nu+ = nuc + J [x0; x0_pre; w] as one product on the matrix
 * cores, J = d nu+ / d data built once per (handle, k) from the same factor.  The dense form is taken when w == NULL
 * (only the 56 columns of [x0; x0_pre] remain) and, with w, for batches of at most max_batch_with_w problems
 * (default 768; environment FMPC_INV_MAX_BATCH, FMPC_NO_INV=1 switches the form off).  Both forms agree to
 * round-off (tests/test_gpu_dense_form.py); a result does not depend on the batch it was solved in as long as the
 * form is the same.
 *   fmpc_set_dense_form: enabled 0/1, max_batch_with_w < 0 keeps the bound.  FMPC_E_UNSUPPORTED if the handle has
 *                        no panel path (n != 27).
 *   fmpc_last_dual_form: 0 = the sweeps, 1 = the dense form of the dual solve, 2 = the AFFINE FORM of the whole step: with
 *                        w == NULL and n_newton == 1 (the reference's replay call, README.md:548-556; nu_out is served too
 *                        when z_out is given: further row tiles of the same product; nu_out WITHOUT z_out takes form 1) the
 *                        step from the cold start is z+ = zc + Kz [x0; x0_pre], one product per batch on the matrix cores
 *                        (Kz built once per (handle, k) with J; the step-length decision from two quadratic forms of the
 *                        data, problems that are not clear-cut redone by the exact path: tests/test_gpu_affine.py).
 *                        FMPC_NO_AFFINE=1 or fmpc_set_dense_form(h, 0, ..) switch it off.
 *                        3 = the first-move form as a product: fmpc_loop_step_device with first moves only (z_out = nu_out =
 *                        NULL, n_newton == 1) and more than 64 realisations: u0 = u0c + K0 [x0; x0_pre; B u1; B u2] and the
 *                        same two-form decision, one product per batch (tests/test_gpu_closed_loop.py; FMPC_NO_LOOP_U0=1
 *                        switches it off).  4 = the same with the loop inputs in the same launch (see fmpc_loop_step_device). */
int fmpc_set_dense_form(fmpc_handle h, int enabled, int max_batch_with_w);

/*
 * Recording solves into a HIP graph.  The device-pointer solves do not allocate, synchronise or read back once the handle's
 * workspaces exist for the batch size and barrier weight of a call (one eager call first), and they leave the handle's
 * cross-stream event alone while their stream is being captured: a stretch of solves on known inputs -- the reference's
 * replay of a realisation, README.md:548-556 -- can be captured once (hipStreamBeginCapture / torch.cuda.graph around the
 * calls; Python: RecordedSolves) and replayed with one host call; inside a graph the launches follow each other more
 * closely than the host can submit them (31.2 against 33.9 us per 2000-problem step).  A graph holds the addresses of the
 * handle's workspaces: fmpc_alloc_generation() changes whenever any handle of the process allocates or releases device
 * memory -- compare it with its value at the recording before every replay, and record again when it has changed.
 */
unsigned long long fmpc_alloc_generation(void);

/*
 * Padded output rows for batches on the device: row p of z_out starts at z_out + p * ldz (ldz >= T (n + m); 0 restores the
 * contiguous rows).  A batch is this library's extension of the reference's one-problem call (Fast_MPC2.m:47-60), so the
 * distance between its rows is ours to offer: with ldz a multiple of 16 (128 bytes) and z_out 128-byte aligned every
 * 128-byte run a tile of the cold-start step writes is one cache line, and the step's 82 MB of output leave with
 * non-temporal stores (34.6 against 39.1 us per 2000-problem step).  Honoured by the affine form of the cold-start step
 * (fmpc_solve_device / fmpc_solve_u0_device with w == NULL, z_init == NULL, n_newton == 1, n = 27) including its exact
 * path for the problems whose step-length decision is not clear-cut; any other solve on a handle with padded rows returns
 * FMPC_E_UNSUPPORTED before anything is enqueued.  z_init, nu, u0, status are not affected.
 */
int fmpc_set_z_ld(fmpc_handle h, int ldz);

/* n = 27: explicit-start batches of at most 1024 problems and the continuation of a Newton budget > 1 (a few hundred problems)
 * run on the tiled kernel (tiled = 1, default: lowest latency of ONE call) or on the one-wavefront kernel (tiled = 0: its single
 * wavefronts share the chip better when many handles have solves in flight at the same time; bench.py `budget5_in_flight_12`).
 * On the tiled kernel batches of at most 512 problems and continuations take FOUR wavefronts per problem (19 % lower latency than
 * two; tiled = 1 or 4), larger ones two; tiled = 2 selects two throughout.  (Round 4 had made two the default after a compile-time
 * sibling instance <double,2,4,11> was miscompiled by round 2's build; that instance is gone since round 5, the run-time
 * four-wavefront form is what every n <= 31 takes up to 512 problems, under the 24-seed stress test and a 192-case sweep.)
 * Environment at create time: FMPC_NO_SMALL_TILED=1 = tiled 0, FMPC_SMALL_TILED_NW=2 = tiled 2. */
int fmpc_set_small_batch_kernel(fmpc_handle h, int tiled);
int fmpc_last_dual_form(fmpc_handle h);

/*
 * The MPC part of one timestep of the reference's loop WITH its estimator (README.md:482-497, 548-556, 589), in one call:
 *     x0 = ad_est[k], x0_pre = ad_est[k-1]        from the estimator (fmpc_est_apply_device), INPUTS here
 *     w  = b_ref = -M1 B u1 - M2 B u2             (README.md:490-497; u1 = u[k-1], u2 = u[k-2], NULL = zeros), written to w
 *     [z, nu] = fastMPC step from the cold start, u0_out = U(1:nu)
 * i.e. fmpc_loop_step_device without the coefficient-space plant x0 = a[k] + B u[k-1].  With first moves only (z_out = nu_out =
 * NULL) and a Newton budget of 1 it takes the same forms: one launch + the exact-path launch for the realisations whose
 * step-length decision is not clear-cut.  x0_pre must not be NULL for a VAR(2) model (zeros at the first timestep).
 * Device pointers as in fmpc_loop_step_device.  tests/test_gpu_estimator.py.
 */
int fmpc_ao_step_device(fmpc_handle h, int batch, const double* x0, const double* x0_pre,
                        const double* u1, const double* u2, double* w,
                        const double* nu0, int n_newton, double k,
                        double* z_out, double* nu_out, int* status, int* iters, double* step,
                        double* u0_out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------------
 * Phase-diversity estimator: replaces the "% Estimator" block of the reference's simulation loop (README.md:456-480)
 *
 *     for k = 1:numel(zd_list)
 *         kW = zd_list(k).*squeeze(Zs(idx2,:,:));   P_defocus = pupil.*exp(1i*(scrn+kW));
 *         I_defocus = fftshift(fft2(fftshift(P_defocus),res,res))*dx^2;   im = abs(I_defocus).^2;
 *         v_im(:,:,k) = im(range_min:range_max,range_min:range_max)*AU;   Y_M = [Y_M; reshape(v_im(:,:,k),[],1)];
 *     end
 *     Y_M = Y_M + Y_M_noise;      ad_est = lsqminnorm((A_s'*A_s),((A_s)'*(Y_M-b_s)));
 *
 * for a batch of residual phase screens at once: the window of each PSF as a partial DFT on the fp64 matrix cores (the
 * reference keeps 31 x 31 samples of a 512 x 512 FFT), ad_est = G (Y_M - b_s) with G = pinv(A_s'A_s) A_s' built once.
 *
 * fmpc_est_create   len: pixels per side (a multiple of 64; the reference: 512, mag = 1 so res = len).
 *                   first, d: the window im(range_min:range_max, ...) as first = range_min - 1 (0-based), d = range_max -
 *                   range_min + 1 <= 32.   ndiv <= 3 diversities;  D_re, D_im: ndiv arrays len x len (column-major, MATLAB
 *                   order) = real / imaginary part of pupil.*exp(1i*zd_list(k)*squeeze(Zs(idx2,:,:))).   scale = dx^4*AU.
 *                   A_s: p x nx column-major, b_s: p, p = ndiv d^2 (model_approx.mat of the reference, piston removed).
 * fmpc_est_apply[_device]   scrn: batch arrays len x len column-major [rad], |scrn| < 1e6 (a pixel beyond that, or a
 *                   non-finite one, makes that screen's outputs NaN: the kernel reduces the phase by pi/2 itself); noise: batch x p or NULL (Y_M_noise);
 *                   ad_est: batch x nx;  Y_out: batch x p or NULL (Y_M, for Y_M_acc of the reference).
 * fmpc_est_dims     any pointer may be NULL; rank = numerical rank of A_s'A_s found when G was built.
 */
typedef struct fmpc_est_s* fmpc_est;
/* The screen the estimator looks at (README.md:453 with :590-601): phase_res = phase_valid(:,:,k) + phase_cor, phase_cor =
 * sum_j ad_cor(j).*Zs(j+1,:,:), ad_cor = B*u_prev, for a batch of screens: out[b] = phase[b] + sum_j (B u_prev[b])_j Z[j].
 * npx = len^2 pixels per screen (any order, the same for phase, Z and out); Z: n maps (piston removed); u_prev: batch x m, NULL at
 * the first step (out = phase, README.md:447).  B is the handle's. */
int fmpc_phase_residual_device(fmpc_handle h, int batch, long long npx, const double* phase, const double* u_prev,
                               const double* Z, double* out, void* stream);
int fmpc_est_create(fmpc_est* out, int len, int first, int d, int ndiv, const double* D_re, const double* D_im,
                    double scale, const double* A_s, const double* b_s, int p, int nx, int device);
int fmpc_est_destroy(fmpc_est e);
int fmpc_est_dims(fmpc_est e, int* len, int* d, int* ndiv, int* nx, int* p, int* rank);
int fmpc_est_apply(fmpc_est e, int batch, const double* scrn, const double* noise, double* ad_est, double* Y_out);
int fmpc_est_apply_device(fmpc_est e, int batch, const double* scrn, const double* noise, double* ad_est, double* Y_out,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FASTMPC_H */
