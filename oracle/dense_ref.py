"""TEST INFRASTRUCTURE -- dense CPU oracle #1 for the fastMPC hot path.  PARITY UNPINNED.

This file is a CPU restatement (numpy) of the reference's MATLAB algorithm for the path
`Fast_MPC/VAR_{1,2}` of jinsungkim96/MPC-SensorlessAO.  It follows the reference op for op,
*including its quirks*, with dense matrices exactly as the reference builds them.

It is a checker, never the product: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  The shipped path (HIP kernels behind
`include/fastmpc.h`) never routes through this file.

PARITY UNPINNED: the reference is MATLAB-only, MATLAB/Octave do not exist in the build image,
and the reference's own `test_fast_mpc.m` holds no assertions, golden vectors or fixtures
(SURVEY.md §4, §8c).  Nothing the reference ships can pin this restatement, so it is pinned
only by (a) line-by-line correspondence with the cited reference lines and (b) independent
algebra (full-KKT solve, stationarity) in `tests/test_oracle_dense.py`.

Reference lines followed (all under /root/reference/Fast_MPC):
  fast_mpc_init          VAR_2/fast_mpc_init.m:12-27
  fast_mpc_objective     VAR_2/fast_mpc_objective.m:17-65
  fast_mpc_eq_const      VAR_2/fast_mpc_eq_const.m:19-71 ; VAR_1/fast_mpc_eq_const.m:17-55
  fast_mpc_ineq_const    VAR_2/fast_mpc_ineq_const.m:4-82 ; VAR_1/fast_mpc_ineq_const.m:42-79
  inf_newton_KKT_H       VAR_2/inf_newton_KKT_H.m:3-13
  inf_newton_solver      VAR_2/inf_newton_solver.m:1-43
  backtracking_inf_newton VAR_2/backtracking_inf_newton.m:1-13
  drivers                VAR_2/Fast_MPC2.m:88-144
MATLAB built-ins (chol, linsolve, norm, mtimes) are LAPACK/BLAS semantics; scipy/numpy call the
same routines (dpotrf, dtrsm, dnrm2, dgemm).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import cholesky, solve_triangular


class RefError(Exception):
    """Stands in for a MATLAB error() raised by the reference."""


def _col(v, name=None):
    """MATLAB column vector; [] -> None."""
    if v is None:
        return None
    a = np.asarray(v, dtype=np.float64)
    if a.size == 0:
        return None
    return a.reshape(-1)


def _mat(v):
    if v is None:
        return None
    a = np.asarray(v, dtype=np.float64)
    if a.size == 0:
        return None
    if a.ndim == 0:
        a = a.reshape(1, 1)
    return a


class DenseFastMPC:
    """Restates the value class `Fast_MPC2` (VAR_2/Fast_MPC2.m:1-55, VAR_1/Fast_MPC2.m:1-51).

    var_order=2: 23-argument constructor of VAR_2; var_order=1: the 21-argument VAR_1 form is
    reached through `DenseFastMPC.var1(...)`.

    bug_compat_var1: reproduce the misplaced row block of VAR_1/fast_mpc_eq_const.m:36
    (SURVEY.md App. B-D1).  Off by default (intended dynamics = VAR_2 code with A2 = 0).
    ramp: VAR_1 has ramp-rate rows (VAR_1/fast_mpc_ineq_const.m:58-76), VAR_2 has them
    commented out (VAR_2/fast_mpc_ineq_const.m:61-79).
    """

    def __init__(self, Q, R, S, Qf, q, r, qf, xmin, xmax, umin, umax, dumin, dumax, T, x0,
                 x0_pre, u_prev, A1, A2, B, w, xf, x_init, *, var_order=2, ramp=None,
                 bug_compat_var1=False):
        self.Q, self.R, self.S, self.Qf = _mat(Q), _mat(R), S, _mat(Qf)
        self.q, self.r, self.qf = _col(q), _col(r), _col(qf)
        self.x_min, self.x_max = _col(xmin), _col(xmax)
        self.u_min, self.u_max = _col(umin), _col(umax)
        self.du_min, self.du_max = _col(dumin), _col(dumax)
        self.T = int(T)
        self.x0, self.x0_pre, self.u_prev = _col(x0), _col(x0_pre), _col(u_prev)
        self.A1, self.A2, self.B = _mat(A1), _mat(A2), _mat(B)
        self.w = _col(w)
        self.x_final = _col(xf)
        self.x_init = _col(x_init)
        self.var_order = int(var_order)
        self.ramp = (self.var_order == 1) if ramp is None else bool(ramp)
        self.bug_compat_var1 = bool(bug_compat_var1)

    @classmethod
    def var1(cls, Q, R, S, Qf, q, r, qf, xmin, xmax, umin, umax, dumin, dumax, T, x0, u_prev,
             A, B, w, xf, x_init, **kw):
        """VAR_1 constructor order (VAR_1/Fast_MPC2.m:26-27)."""
        kw.setdefault("var_order", 1)
        return cls(Q, R, S, Qf, q, r, qf, xmin, xmax, umin, umax, dumin, dumax, T, x0, None,
                   u_prev, A, None, B, w, xf, x_init, **kw)

    # ------------------------------------------------------------------ assembly
    def initialize(self):
        """fast_mpc_init.m:12-27 -- caller's start, or every stage at mid-box."""
        T, n, m = self.T, self.Q.shape[0], self.R.shape[0]
        if self.x_init is not None:
            if self.x_init.shape[0] != T * (n + m):
                raise RefError("Initialization size mismatch (T*(n+m))")
            return self.x_init.copy()
        x_init = (self.x_min + self.x_max) / 2
        u_init = (self.u_min + self.u_max) / 2
        z = np.zeros(T * (m + n))
        for i in range(0, T * (m + n) - (m + n) + 1, m + n):
            z[i:i + m] = u_init
            z[i + m:i + m + n] = x_init
        return z

    def objective_function(self):
        """fast_mpc_objective.m:17-65 -- dense H, g; cost is z'Hz + g'z (no 1/2)."""
        T, n, m = self.T, self.Q.shape[0], self.R.shape[0]
        Q, R, Qf = self.Q, self.R, self.Qf
        if Q.shape[0] != Q.shape[1] or Qf.shape[0] != Qf.shape[1]:
            raise RefError("State stage cost must a square matrix")
        if R.shape[0] != R.shape[1]:
            raise RefError("Control stage cost must a square matrix")
        q, r, qf = self.q, self.r, self.qf
        if q is not None:
            if q.shape[0] != n:
                raise RefError("Linear state cost needs to be a vector of size n")
        else:
            q = np.zeros(n)
        if r is not None:
            if r.shape[0] != m:
                raise RefError("Linear control cost needs to be a vector of size n")
        else:
            r = np.zeros(m)
        if qf is not None:
            if qf.shape[0] != n:
                raise RefError("State terminal linear cost needs to be a vector of size n")
        else:
            qf = np.zeros(n)
        N = T * (n + m)
        H = np.zeros((N, N))
        blk = np.block([[Q, np.zeros((n, m))], [np.zeros((m, n)), R]])
        # MATLAB: for i=m+1:(n+m):size(H,1)-n   (1-based)  -> 0-based start m
        for i in range(m, N - n, n + m):
            H[i:i + n + m, i:i + n + m] = blk
        H[0:m, 0:m] = R
        H[N - n:, N - n:] = Qf
        g = np.zeros(N)
        for i in range(m, N, n + m):
            if i == N - n:
                g[N - n:] = qf
            else:
                g[i:i + n + m] = np.concatenate([q, r])
        g[0:m] = r
        return H, g

    def equality_const(self):
        """fast_mpc_eq_const.m (VAR_2 :19-71, VAR_1 :17-55) -- dense C, b."""
        if self.var_order == 2:
            return self._eq_var2()
        return self._eq_var1()

    def _eq_var2(self):
        A1, A2, B, w = self.A1, self.A2, self.B, self.w
        if A1 is None or A2 is None:
            raise RefError("Define the state dynamics/equality constrained matrix")
        if B is None:
            raise RefError("Define the control dynamics/equality constrained matrix")
        n, m, T = A1.shape[1], B.shape[1], self.T
        x0, x0_pre = self.x0, self.x0_pre
        if x0 is None or A1.shape[1] != x0.shape[0]:
            raise RefError("The equality state dynamics matrix size does not match")
        if x0_pre is None or A2.shape[1] != x0_pre.shape[0]:
            raise RefError("The equality state dynamics matrix size does not match")
        if B.shape[1] != self.R.shape[1]:
            raise RefError("The equality control dynamics matrix size does not match")
        if w is None:
            w = np.zeros(n)          # :33-35 -- only n long (quirk D7)
        s = n + m
        C = np.zeros((T * n, T * s))
        b = np.zeros(T * n)
        I = np.eye(n)
        C[0:n, 0:s] = np.hstack([-B, I])                                   # :38
        b[0:n] = A1 @ x0 + A2 @ x0_pre + w[0:n]                            # :39
        for i in range(1, T):                                              # :41
            if (i + 1) * n > w.shape[0]:
                raise RefError("Index exceeds the number of array elements (w)")  # D7
            if i == 1:
                C[n:2 * n, m:2 * s] = np.hstack([-A1, -B, I])               # :43
                b[n:2 * n] = A2 @ x0 + w[n:2 * n]                          # :44
            else:
                c0 = m + s * (i - 2)
                C[n * i:n * (i + 1), c0:s * (i + 1)] = np.hstack(
                    [-A2, np.zeros((n, m)), -A1, -B, I])                   # :46
                b[n * i:n * (i + 1)] = w[n * i:n * (i + 1)]                # :47
        xf = self.x_final
        if xf is not None:                                                 # :67-70
            b = np.concatenate([b, xf])
            C = np.vstack([C, np.zeros((n, C.shape[1]))])
        C[-n:, -n:] = np.eye(n)                                            # :71
        return C, b

    def _eq_var1(self):
        A, B, w = self.A1, self.B, self.w
        if A is None:
            raise RefError("Define the state dynamics/equality constrained matrix")
        if B is None:
            raise RefError("Define the control dynamics/equality constrained matrix")
        n, m, T = A.shape[1], B.shape[1], self.T
        x = self.x0
        if x is None or A.shape[1] != x.shape[0]:
            raise RefError("The equality state dynamics matrix size does not match")
        if B.shape[1] != self.R.shape[1]:
            raise RefError("The equality control dynamics matrix size does not match")
        if w is None:
            w = np.zeros(n)
        s = n + m
        C = np.zeros((T * n, T * s))
        b = np.zeros(T * n)
        I = np.eye(n)
        C[0:n, 0:s] = np.hstack([-B, I])                                   # VAR_1 :32
        blk = np.hstack([-A, -B, I])
        for j in range(1, T):                                              # i = j*n, :34
            if (j + 1) * n > w.shape[0]:
                raise RefError("Index exceeds the number of array elements (w)")
            if j == 1:
                # :36 writes at 1-based column n (intended: m+1).  D1.
                c0 = (n - 1) if self.bug_compat_var1 else m
                if c0 + 2 * n + m > C.shape[1]:
                    raise RefError("bug_compat_var1: MATLAB would grow C; not emulated")
                C[n:2 * n, c0:c0 + 2 * n + m] = blk
            else:
                c0 = (j - 1) * s + m                                       # :40
                C[n * j:n * (j + 1), c0:c0 + 2 * n + m] = blk
            b[n * j:n * (j + 1)] = w[n * j:n * (j + 1)]                    # :37,:41
        b[0:n] = A @ x + w[0:n]                                            # :48
        xf = self.x_final
        if xf is not None:
            b = np.concatenate([b, xf])
            C = np.vstack([C, np.zeros((n, C.shape[1]))])
        C[-n:, -n:] = np.eye(n)                                            # :55
        return C, b

    def inequality_const(self):
        """fast_mpc_ineq_const.m -- box rows on u only (VAR_2 :42-56); VAR_1 appends ramp rows
        (VAR_1 :58-76).  State bounds are NOT constraints (VAR_2 :25-40 commented out)."""
        n0 = self.Q.shape[0]
        if self.x_min is None or self.x_max is None or \
                self.x_min.shape[0] != n0 or self.x_max.shape[0] != n0:
            raise RefError("Check the state inequality constraints dimensions")
        m0 = self.R.shape[0]
        if self.u_min is None or self.u_max is None or \
                self.u_min.shape[0] != m0 or self.u_max.shape[0] != m0:
            raise RefError("Check cotrol iequality constraint dimension")
        T, n, m = self.T, self.x_min.shape[0], self.u_min.shape[0]
        s = n + m
        Im = np.eye(m)
        P_box = np.zeros((2 * T * m, T * s))
        h_box = np.zeros(2 * T * m)
        for j in range(T):                       # row block j <-> u_j  (:46-52)
            P_box[2 * m * j:2 * m * (j + 1), s * j:s * j + m] = np.vstack([Im, -Im])
            h_box[2 * m * j:2 * m * (j + 1)] = np.concatenate([self.u_max, -self.u_min])
        if not self.ramp:
            return P_box, h_box
        P_ramp = np.zeros((2 * T * m, T * s))
        h_ramp = np.zeros(2 * T * m)
        for j in range(T):                       # VAR_1 :62-76
            rows = slice(2 * m * j, 2 * m * (j + 1))
            if j == 0:
                P_ramp[rows, 0:m] = np.vstack([Im, -Im])
                h_ramp[rows] = np.concatenate([self.u_prev + self.du_max,
                                               -self.u_prev - self.du_min])
            else:
                c0 = s * (j - 1)
                P_ramp[rows, c0:c0 + s + m] = np.block(
                    [[-Im, np.zeros((m, n)), Im], [Im, np.zeros((m, n)), -Im]])
                h_ramp[rows] = np.concatenate([self.du_max, -self.du_min])
        return np.vstack([P_box, P_ramp]), np.concatenate([h_box, h_ramp])

    # ------------------------------------------------------------------ drivers
    def _assemble(self):
        z = self.initialize()
        H, g = self.objective_function()
        P, h = self.inequality_const()
        C, b = self.equality_const()
        return z, H, g, P, h, C, b

    def mpc_fixed_log_newton(self, nw, k, *, nu0=None, rng=None, info=None, literal_D=False):
        """Fast_MPC2.m:124-130."""
        z, H, g, P, h, C, b = self._assemble()
        return inf_newton_solver(H, g, P, h, C, b, k, z, nw, nu0=nu0, rng=rng, info=info,
                                 literal_D=literal_D)

    def mpc_fixed_log(self, k, *, nu0=None, rng=None, info=None):
        """Fast_MPC2.m:116-123 (nw = [] -> up to 1000 iterations with the tolerance exit)."""
        z, H, g, P, h, C, b = self._assemble()
        return inf_newton_solver(H, g, P, h, C, b, k, z, None, nu0=nu0, rng=rng, info=info)

    def _k_schedule(self, nw, nu0s, rng, infos):
        """Shared body of mpc_solve_full (:100-115) and mpc_fixed_newton (:131-144)."""
        z, H, g, P, h, C, b = self._assemble()
        k, mu = 1.0, 1.0 / 10
        x_opt = z
        it = 0
        while k * z.shape[0] >= 10e-3:
            nu0 = None if nu0s is None else nu0s[it]
            info = {} if infos is not None else None
            x_opt = inf_newton_solver(H, g, P, h, C, b, k, z, nw, nu0=nu0, rng=rng, info=info)
            if infos is not None:
                info["k"] = k
                infos.append(info)
            k = mu * k
            z = x_opt
            it += 1
        return x_opt

    def mpc_solve_full(self, *, nu0s=None, rng=None, infos=None):
        return self._k_schedule(None, nu0s, rng, infos)

    def mpc_fixed_newton(self, nw, *, nu0s=None, rng=None, infos=None):
        return self._k_schedule(nw, nu0s, rng, infos)

    def mpc_solve_check(self, k_min, k_max, *, nu0s=None, rng=None, infos=None):
        """Fast_MPC2.m:88-99 -- five barrier weights linspace(k_max,k_min,5), warm-started."""
        z, H, g, P, h, C, b = self._assemble()
        ks = np.linspace(k_max, k_min, 5)
        x_opt = z
        for i, k in enumerate(ks):
            nu0 = None if nu0s is None else nu0s[i]
            info = {} if infos is not None else None
            x_opt = inf_newton_solver(H, g, P, h, C, b, float(k), z, None, nu0=nu0, rng=rng,
                                      info=info)
            if infos is not None:
                info["k"] = float(k)
                infos.append(info)
            z = x_opt
        return x_opt


# ---------------------------------------------------------------------- Newton kernel
def inf_newton_KKT_H(H, P, h, z, k, literal_D=False):
    """inf_newton_KKT_H.m:3-13.  Returns (Phi, d).

    literal_D=True builds the dense diagonal matrix D and evaluates k*P'*D*P by two dense
    products as the reference does (:5-9,:13); the default scales rows instead.  Both give the
    same Phi bit for bit here because P holds only 0/+-1 and D is diagonal (each entry of the
    product is one non-zero term plus exact zeros) -- checked in tests/test_oracle_dense.py.
    """
    d_inv = h - P @ z
    d = 1.0 / d_inv
    if literal_D:
        nn = d_inv.shape[0]
        D = np.zeros((nn, nn))
        for i in range(nn):
            D[i, i] = (1.0 / d_inv[i]) ** 2
        Phi = 2 * H + k * P.T @ D @ P
    else:
        Dv = (1.0 / d_inv) ** 2
        Phi = 2 * H + k * (P.T * Dv) @ P
    return Phi, d


def backtracking_inf_newton(z, nu, del_z, del_nu, rp, rd, al, bt, info=None):
    """backtracking_inf_newton.m:2-11.  The counter is never decremented (:3,:6), so a failing
    search halves t until it underflows to 0 and then leaves with t = 0 (quirk D2)."""
    t = 1.0
    n0 = np.linalg.norm(np.concatenate([rp(z), rd(z, nu)]))
    halvings = 0
    while np.linalg.norm(np.concatenate([rp(z + t * del_z),
                                         rd(z + t * del_z, nu + t * del_nu)])) > (1 - al * t) * n0:
        t = bt * t
        halvings += 1
    if info is not None:
        info.setdefault("t", []).append(t)
        info.setdefault("halvings", []).append(halvings)
        # diagnostics for the property tests: the norm the literal test compared at the accepted t, and the one at the start
        info.setdefault("n_before", []).append(n0)
        info.setdefault("n_after", []).append(np.linalg.norm(np.concatenate([rp(z + t * del_z), rd(z + t * del_z, nu + t * del_nu)])))
    return z + t * del_z, nu + t * del_nu


def inf_newton_solver(H, g, P, h, C, b, k, z, newton, *, nu0=None, rng=None, info=None,
                      literal_D=False):
    """inf_newton_solver.m:1-43.

    nu0 replaces `nu = rand(length(b),1)` (:2); with nu0=None the draw comes from `rng`
    (numpy Generator, U(0,1)) which mirrors the reference's use of the global stream.
    info (dict) collects per-call diagnostics: iters (Newton steps taken), nu, t, residual norms.
    """
    if nu0 is None:
        rng = np.random.default_rng() if rng is None else rng
        nu = rng.random(b.shape[0])
    else:
        nu = np.asarray(nu0, dtype=np.float64).reshape(-1).copy()
        if nu.shape[0] != b.shape[0]:
            raise RefError("nu0 size mismatch")
    z = z.copy()
    max_iter = 1000 if newton is None else int(newton)                     # :4-8
    tol = 1e-6
    steps = 0
    if info is not None:
        info.update({"n_r": [], "n_g": []})
    for _ in range(max_iter):                                              # :10
        KKT_H, d = inf_newton_KKT_H(H, P, h, z, k, literal_D=literal_D)    # :11
        kPd = k * (P.T @ d)

        def rd(zz, v, kPd=kPd):                                            # :12 (d frozen)
            return 2 * (H @ zz) + g + kPd + C.T @ v

        def rp(zz):                                                        # :13
            return C @ zz - b

        tol_g = C @ z - b                                                  # :15
        n_r = np.linalg.norm(np.concatenate([-rd(z, nu), -rp(z)]))         # :14,:16
        n_g = np.linalg.norm(tol_g)                                        # :17
        if info is not None:
            info["n_r"].append(n_r)
            info["n_g"].append(n_g)
        if n_r <= tol and n_g <= 1e-8:                                     # :19-22
            break
        try:
            L = cholesky(KKT_H, lower=True)                                # :24
        except np.linalg.LinAlgError as e:
            raise RefError("chol: Phi not positive definite") from e
        Schur = C @ solve_triangular(L.T, solve_triangular(L, C.T, lower=True), lower=False)
        rdz = rd(z, nu)
        phi_inv_rd = solve_triangular(L.T, solve_triangular(L, rdz, lower=True), lower=False)
        Beta = -rp(z) + C @ phi_inv_rd                                     # :29
        try:
            SL = cholesky(Schur, lower=True)                               # :30
        except np.linalg.LinAlgError as e:
            raise RefError("chol: Schur complement not positive definite") from e
        int_nu = solve_triangular(SL, -Beta, lower=True)                   # :31
        del_nu = solve_triangular(SL.T, int_nu, lower=False)               # :32
        int_z = solve_triangular(L, -rdz - C.T @ del_nu, lower=True)       # :34
        del_z = solve_triangular(L.T, int_z, lower=False)                  # :35
        z, nu = backtracking_inf_newton(z, nu, del_z, del_nu, rp, rd, 1e-4, 0.5, info)  # :36-38
        steps += 1
    if info is not None:
        info["iters"] = steps
        info["nu"] = nu
    return z


def deinterleave(z, n, m, T):
    """Caller-side unpack, README.md:558-570: z=[u0;x1;u1;x2;...] -> (U (T*m), X (T*n))."""
    Z = np.asarray(z).reshape(T, m + n)
    return Z[:, :m].reshape(-1).copy(), Z[:, m:].reshape(-1).copy()
