"""Host-side mirror of the reference's solver facade, `Fast_MPC2`
(Fast_MPC/VAR_2/Fast_MPC2.m:1-146; VAR_1/Fast_MPC2.m for the 21-argument form).

Same constructor argument order, same driver names, same meaning of `[]` (here: None or an empty
array), same return value (x_opt = interleaved z).  Every solve goes through the C ABI
(include/fastmpc.h) to the HIP kernels; nothing is solved on the host.

What is different from the reference, on purpose:
  * `nu = rand(length(b),1)` (inf_newton_solver.m:2) is drawn here (numpy Generator passed as
    `rng`, or an explicit `nu0`) and handed to the device, so the caller controls the stream.
  * The object is cheap to rebuild every timestep as the notebook does (README.md:548): device
    handles are cached on the content of the shared model.
  * `matlab_solve` / `fomulate_mpc` (fmincon, Fast_MPC2.m:68-87) are outside the path.
  * The dense builders `objective_function` / `inequality_const` / `equality_const`
    (Fast_MPC2.m:56-64) are not provided: the device path never forms H, P or C.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import FastMPCError
from .handle import FastMPCHandle

_HANDLE_CACHE: "OrderedDict[str, FastMPCHandle]" = OrderedDict()
_HANDLE_CACHE_MAX = 8


def _empty(v):
    return v is None or np.asarray(v).size == 0


def _vec(v):
    return None if _empty(v) else np.asarray(v, dtype=np.float64).reshape(-1)


def _mat(v):
    if _empty(v):
        return None
    a = np.asarray(v, dtype=np.float64)
    return a.reshape(1, 1) if a.ndim == 0 else a


def _digest(*arrays):
    hsh = hashlib.blake2b(digest_size=16)
    for a in arrays:
        if a is None:
            hsh.update(b"\x00none")
        else:
            a = np.ascontiguousarray(a)
            hsh.update(str(a.shape).encode())
            hsh.update(a.tobytes())
    return hsh.hexdigest()


def deinterleave(z, n, m, T):
    """Caller-side unpack (README.md:558-570): z=[u0;x1;u1;x2;...] -> (U, X)."""
    Z = np.asarray(z).reshape(T, m + n)
    return Z[:, :m].reshape(-1).copy(), Z[:, m:].reshape(-1).copy()


class Fast_MPC2:
    """obj = Fast_MPC2(Q,R,S,Qf,q,r,qf,xmin,xmax,umin,umax,dumin,dumax,T,x0,x0_pre,u_prev,
                       A1,A2,B,w,xf,x_init)                       (VAR_2/Fast_MPC2.m:28-29)"""

    var_order = 2

    def __init__(self, Q, R, S, Qf, q, r, qf, xmin, xmax, umin, umax, dumin, dumax, T, x0, x0_pre,
                 u_prev, A1, A2, B, w, xf, x_init, *, device=0, rng=None):
        self.Q, self.R, self.S, self.Qf = _mat(Q), _mat(R), S, _mat(Qf)
        self.q, self.r, self.qf = _vec(q), _vec(r), _vec(qf)
        self.x_min, self.x_max = _vec(xmin), _vec(xmax)
        self.u_min, self.u_max = _vec(umin), _vec(umax)
        self.du_min, self.du_max = _vec(dumin), _vec(dumax)      # unused in VAR_2 (D8)
        self.T = int(T)
        self.x0, self.x0_pre, self.u_prev = _vec(x0), _vec(x0_pre), _vec(u_prev)
        self.A1, self.A2, self.B = _mat(A1), _mat(A2), _mat(B)
        self.w = _vec(w)
        self.x_final = _vec(xf)
        self.x_init = _vec(x_init)
        self.device = int(device)
        self.rng = rng
        self.last_info = None

    # ---------------------------------------------------------------- validation
    def _check(self):
        """The reference's error() calls, in the order its drivers hit them."""
        E = lambda msg: FastMPCError(_lib.FMPC_E_DIM, msg)
        Q, R, Qf = self.Q, self.R, self.Qf
        if Q is None or R is None or Qf is None:
            raise FastMPCError(_lib.FMPC_E_NULL, "Q, R, Qf are required")
        n, m, T = Q.shape[0], R.shape[0], self.T
        # fast_mpc_init.m:13-14
        if self.x_init is not None and self.x_init.shape[0] != T * (n + m):
            raise E("Initialization size mismatch (T*(n+m))")
        # fast_mpc_objective.m:17-47
        if Q.shape[0] != Q.shape[1] or Qf.shape[0] != Qf.shape[1]:
            raise E("State stage cost must a square matrix")
        if R.shape[0] != R.shape[1]:
            raise E("Control stage cost must a square matrix")
        if self.q is not None and self.q.shape[0] != n:
            raise E("Linear state cost needs to be a vector of size n")
        if self.r is not None and self.r.shape[0] != m:
            raise E("Linear control cost needs to be a vector of size n")
        if self.qf is not None and self.qf.shape[0] != n:
            raise E("State terminal linear cost needs to be a vector of size n")
        # fast_mpc_ineq_const.m:4-9
        if self.x_min is None or self.x_max is None or self.x_min.shape[0] != n or self.x_max.shape[0] != n:
            raise E("Check the state inequality constraints dimensions")
        if self.u_min is None or self.u_max is None or self.u_min.shape[0] != m or self.u_max.shape[0] != m:
            raise E("Check cotrol iequality constraint dimension")
        # fast_mpc_eq_const.m:19-32
        if self.A1 is None or (self.var_order == 2 and self.A2 is None):
            raise FastMPCError(_lib.FMPC_E_NULL, "Define the state dynamics/equality constrained matrix")
        if self.B is None:
            raise FastMPCError(_lib.FMPC_E_NULL, "Define the control dynamics/equality constrained matrix")
        if self.x0 is None or self.A1.shape[1] != self.x0.shape[0]:
            raise E("The equality state dynamics matrix size does not match")
        if self.var_order == 2 and (self.x0_pre is None or self.A2.shape[1] != self.x0_pre.shape[0]):
            raise E("The equality state dynamics matrix size does not match")
        if self.B.shape[1] != R.shape[1]:
            raise E("The equality control dynamics matrix size does not match")
        if self.A1.shape != (n, n) or self.B.shape[0] != n:
            raise E("The equality state dynamics matrix size does not match")
        if self.w is not None and self.w.shape[0] != T * n:
            # the reference indexes w(n*i+1:n*(i+1)) for every stage (fast_mpc_eq_const.m:44,47)
            raise E("Index exceeds the number of array elements (w must have T*n entries)")
        if self.w is None and T > 1:
            # fast_mpc_eq_const.m:33-35 makes w only n long and then indexes past it (D7);
            # zeros(T*n) is the documented superset.
            pass
        if self.x_final is not None and self.x_final.shape[0] != n:
            raise E("Terminal state size mismatch")
        return n, m, T

    def _handle(self):
        n, m, T = self._check()
        key = _digest(np.array([n, m, T, self.var_order, self.device]), self.A1,
                      self.A2 if self.var_order == 2 else None, self.B, self.Q, self.R, self.Qf,
                      self.q, self.r, self.qf, self.x_min, self.x_max, self.u_min, self.u_max,
                      self.x_final)
        h = _HANDLE_CACHE.get(key)
        if h is None:
            h = FastMPCHandle(self.A1, self.A2 if self.var_order == 2 else None, self.B, self.Q,
                              self.R, self.Qf, self.u_min, self.u_max, self.x_min, self.x_max, T,
                              q=self.q, r=self.r, qf=self.qf, xf=self.x_final,
                              var_order=self.var_order, device=self.device)
            _HANDLE_CACHE[key] = h
            while len(_HANDLE_CACHE) > _HANDLE_CACHE_MAX:
                _HANDLE_CACHE.popitem(last=False)[1].close()
        else:
            _HANDLE_CACHE.move_to_end(key)
        return h

    def _draw_nu0(self, h, nu0):
        if nu0 is not None:
            return np.asarray(nu0, dtype=np.float64).reshape(-1)
        rng = self.rng if self.rng is not None else np.random.default_rng()
        return rng.random(h.nu_len)                 # nu = rand(length(b),1), inf_newton_solver.m:2

    def _u_prev_for_solve(self):
        return None                                  # VAR_2: no ramp rows (fast_mpc_ineq_const.m:61-79 commented out)

    def _solve(self, h, z_init, nw, k, nu0):
        z, info = h.solve(self.x0, self.x0_pre if self.var_order == 2 else None, self.w,
                          z_init=z_init, nu0=self._draw_nu0(h, nu0), n_newton=nw, k=k,
                          return_info=True, u_prev=self._u_prev_for_solve())
        st = int(info["status"][0])
        if st < 0:                                   # chol() error in the reference
            raise FastMPCError(st, "inf_newton_solver")
        return z, info

    # ---------------------------------------------------------------- reference API
    def initialize(self):
        """fast_mpc_init.m:12-27."""
        n, m, T = self._check()
        if self.x_init is not None:
            return self.x_init.copy()
        z = np.zeros((T, m + n))
        z[:, :m] = (self.u_min + self.u_max) / 2
        z[:, m:] = (self.x_min + self.x_max) / 2
        return z.reshape(-1)

    def mpc_fixed_log_newton(self, nw, k, nu0=None):
        """Fast_MPC2.m:124-130 -- the hot entry (README.md:555)."""
        h = self._handle()
        z, self.last_info = self._solve(h, self.x_init, int(nw), float(k), nu0)
        return z

    def mpc_fixed_log(self, k, nu0=None):
        """Fast_MPC2.m:116-123 -- nw = []: <= 1000 iterations, tolerance exit."""
        h = self._handle()
        z, self.last_info = self._solve(h, self.x_init, 0, float(k), nu0)
        return z

    def _k_schedule(self, nw, nu0s):
        h = self._handle()
        z = self.initialize()
        k, mu = 1.0, 1.0 / 10
        infos = []
        x_opt = z
        it = 0
        while k * z.shape[0] >= 10e-3:               # Fast_MPC2.m:108,138
            nu0 = None if nu0s is None else nu0s[it]
            x_opt, info = self._solve(h, z, nw, k, nu0)
            info["k"] = k
            infos.append(info)
            k = mu * k
            z = x_opt
            it += 1
        self.last_info = infos
        return x_opt

    def mpc_solve_full(self, nu0s=None):
        """Fast_MPC2.m:100-115."""
        return self._k_schedule(0, nu0s)

    def mpc_fixed_newton(self, nw, nu0s=None):
        """Fast_MPC2.m:131-144."""
        return self._k_schedule(int(nw), nu0s)

    def mpc_solve_check(self, k_min, k_max, nu0s=None):
        """Fast_MPC2.m:88-99."""
        h = self._handle()
        z = self.initialize()
        infos = []
        x_opt = z
        for i, k in enumerate(np.linspace(k_max, k_min, 5)):
            nu0 = None if nu0s is None else nu0s[i]
            x_opt, info = self._solve(h, z, 0, float(k), nu0)
            info["k"] = float(k)
            infos.append(info)
            z = x_opt
        self.last_info = infos
        return x_opt

    # ---------------------------------------------------------------- dense builders of the class (host side)
    # The device never forms H, P, C; these exist because they are public methods of the reference class
    # (VAR_2/Fast_MPC2.m:56-67) that a caller may use for its own purposes (e.g. `fomulate_mpc`).  Written from the
    # index maps of SURVEY App. A.1-A.3, checked against oracle/dense_ref.py in tests/test_host_api.py.
    def objective_function(self):
        """[H, g]: cost z'Hz + g'z (no 1/2), H = blkdiag(R, [Q 0; 0 R] x (T-1), Qf)   (fast_mpc_objective.m:50-65)."""
        n, m, T = self._check()
        s = n + m
        H = np.zeros((T * s, T * s)); g = np.zeros(T * s)
        zr = lambda v, k: np.zeros(k) if v is None else v
        for j in range(T):
            H[j * s:j * s + m, j * s:j * s + m] = self.R
            H[j * s + m:(j + 1) * s, j * s + m:(j + 1) * s] = self.Qf if j == T - 1 else self.Q
            g[j * s:j * s + m] = zr(self.r, m)
            g[j * s + m:(j + 1) * s] = zr(self.qf, n) if j == T - 1 else zr(self.q, n)
        return H, g

    def inequality_const(self):
        """[P, h]: P z <= h, per stage u_j <= u_max, -u_j <= -u_min (fast_mpc_ineq_const.m:46-56); the VAR_1 class appends
        the ramp rows (VAR_1/fast_mpc_ineq_const.m:58-76)."""
        n, m, T = self._check()
        s = n + m
        P = np.zeros((2 * T * m, T * s)); h = np.zeros(2 * T * m)
        for j in range(T):
            P[2 * j * m:(2 * j + 1) * m, j * s:j * s + m] = np.eye(m)
            P[(2 * j + 1) * m:(2 * j + 2) * m, j * s:j * s + m] = -np.eye(m)
            h[2 * j * m:(2 * j + 1) * m] = self.u_max
            h[(2 * j + 1) * m:(2 * j + 2) * m] = -self.u_min
        return P, h

    def equality_const(self):
        """[C, b]: row block i (0-based) x_{i+1} - B u_i - [i>=1] A1 x_i - [i>=2] A2 x_{i-1} = b_i with the prediction in
        b_0 = A1 x0 + A2 x0_pre + w_0, b_1 = A2 x0 + w_1 (fast_mpc_eq_const.m:38-49); terminal rows x_T = xf (:67-71).
        The VAR_1 class has A2 = 0 (intended dynamics)."""
        n, m, T = self._check()
        s = n + m
        A1 = self.A1
        A2 = self.A2 if (self.var_order == 2 and self.A2 is not None) else np.zeros((n, n))
        x0p = self.x0_pre if (self.var_order == 2 and self.x0_pre is not None) else np.zeros(n)
        w = np.zeros(T * n) if self.w is None else self.w
        nb = T + (1 if self.x_final is not None else 0)
        C = np.zeros((nb * n, T * s)); b = np.zeros(nb * n)
        for i in range(T):
            rows = slice(i * n, (i + 1) * n)
            C[rows, i * s:i * s + m] = -self.B
            C[rows, i * s + m:(i + 1) * s] = np.eye(n)
            if i >= 1:
                C[rows, (i - 1) * s + m:i * s] = -A1
            if i >= 2:
                C[rows, (i - 2) * s + m:(i - 1) * s] = -A2
            b[rows] = w[i * n:(i + 1) * n]
        b[:n] += A1 @ self.x0 + A2 @ x0p
        if T > 1:
            b[n:2 * n] += A2 @ self.x0
        if self.x_final is not None:
            C[T * n:, (T - 1) * s + m:] = np.eye(n)
            b[T * n:] = self.x_final
        return C, b

    def matlab_solve(self):
        raise NotImplementedError("fmincon path (Fast_MPC2.m:76-87) is outside the fastMPC hot path")

    fomulate_mpc = matlab_solve


class Fast_MPC2_VAR1(Fast_MPC2):
    """VAR(1) form: Fast_MPC2(Q,R,S,Qf,q,r,qf,xmin,xmax,umin,umax,dumin,dumax,T,x0,u_prev,
    A,B,w,xf,x_init)  (VAR_1/Fast_MPC2.m:26-27).

    The device path solves the *intended* VAR(1) dynamics (VAR_2 code with A2 = 0; the reference's
    misplaced row block VAR_1/fast_mpc_eq_const.m:36 is not reproduced).
    VAR_1's ramp-rate rows (VAR_1/fast_mpc_ineq_const.m:58-76: du_min <= u_j - u_{j-1} <= du_max with
    u_{-1} = u_prev) are on by default, as in the reference (the ramp-rate kernel, fmpc_kernel_ramp.hip);
    ramp=False solves with the box rows only."""

    var_order = 1

    def __init__(self, Q, R, S, Qf, q, r, qf, xmin, xmax, umin, umax, dumin, dumax, T, x0, u_prev,
                 A, B, w, xf, x_init, *, ramp=True, device=0, rng=None):
        super().__init__(Q, R, S, Qf, q, r, qf, xmin, xmax, umin, umax, dumin, dumax, T, x0, None,
                         u_prev, A, None, B, w, xf, x_init, device=device, rng=rng)
        self.ramp = bool(ramp)

    def _check(self):
        n, m, T = super()._check()
        if self.ramp:
            E = lambda msg: FastMPCError(_lib.FMPC_E_DIM, msg)
            if self.du_min is None or self.du_max is None or self.du_min.shape[0] != m or self.du_max.shape[0] != m:
                raise E("Check ramp rate constraint dimension")
            if self.u_prev is None or self.u_prev.shape[0] != m:
                raise E("Check u_prev dimension")
        return n, m, T

    def _handle(self):
        h = super()._handle()
        if self.ramp:
            h.set_ramp(self.du_min, self.du_max)     # (bounds live in the cached handle; set per call, m doubles)
        return h

    def inequality_const(self):
        """Box rows as VAR_2, then per stage +-(u_j - u_{j-1}) <= +-du with u_{-1} = u_prev (VAR_1/fast_mpc_ineq_const.m:58-76)
        when the ramp rows are on."""
        P, h = super().inequality_const()
        if not getattr(self, "ramp", True) or self.du_min is None or self.du_max is None or self.u_prev is None:
            return P, h
        n, m, T = self._check()
        s = n + m
        Pr = np.zeros((2 * T * m, T * s)); hr = np.zeros(2 * T * m)
        for j in range(T):
            Pr[2 * j * m:(2 * j + 1) * m, j * s:j * s + m] = np.eye(m)
            Pr[(2 * j + 1) * m:(2 * j + 2) * m, j * s:j * s + m] = -np.eye(m)
            if j >= 1:
                Pr[2 * j * m:(2 * j + 1) * m, (j - 1) * s:(j - 1) * s + m] = -np.eye(m)
                Pr[(2 * j + 1) * m:(2 * j + 2) * m, (j - 1) * s:(j - 1) * s + m] = np.eye(m)
            up = self.u_prev if j == 0 else np.zeros(m)
            hr[2 * j * m:(2 * j + 1) * m] = up + self.du_max
            hr[(2 * j + 1) * m:(2 * j + 2) * m] = -up - self.du_min
        return np.vstack([P, Pr]), np.concatenate([h, hr])

    def _u_prev_for_solve(self):
        return self.u_prev if self.ramp else None
