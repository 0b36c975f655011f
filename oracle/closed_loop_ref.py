"""TEST INFRASTRUCTURE -- CPU oracle of the coefficient-space closed loop around the fastMPC solve
(SURVEY.md §8(f) rank 1).  PARITY UNPINNED, like the solver oracles it calls: the reference is MATLAB only and
ships no vectors.

Restates the steps either side of the solver in the reference's simulation loop, with the phase-screen
estimator (README.md:456-480, out of scope) replaced by the true residual coefficients:

  M1, M2                         main.mlx §"System Matrix design" (MPC_DesignMatrices): M1_0 = A1, M2_0 = A2,
                                 M1_1 = A1^2 + A2, M2_1 = A1 A2, M1_i = A1 M1_{i-1} + A2 M1_{i-2}, M2_i = M1_{i-1} A2
  x0 = residual coefficients     README.md:482-483 (ad_est; here a[k] + B u[k-1], the turbulence plus the mirror's
                                 correction ad_cor = B u_prev of README.md:589-590)
  x0_pre                         README.md:484-488 (zeros at the first step, then the previous x0)
  b_ref                          README.md:490-497 (0; -M1 B u[k-1]; -M1 B u[k-1] - M2 B u[k-2])
  w = b_ref, the solver call     README.md:547-555 (Fast_MPC2(..., x_init = []).mpc_fixed_log_newton(n_fix, k_fix))
  u_prev = U(1:nu)               README.md:589

Checker only: imported by tests/ and bench.py's checks, never by the package.
"""
from __future__ import annotations

import numpy as np

from .banded_ref import BandedFastMPC


def design_matrices(A1, A2, T):
    """M1, M2 as (T*n) x n stacks (MPC_DesignMatrices)."""
    n = A1.shape[0]
    M1 = np.zeros((T * n, n)); M2 = np.zeros((T * n, n))
    blk = lambda M, i: M[i * n:(i + 1) * n]
    for i in range(T):
        if i == 0:
            blk(M1, 0)[:] = A1; blk(M2, 0)[:] = A2
        elif i == 1:
            blk(M1, 1)[:] = A1 @ A1 + A2; blk(M2, 1)[:] = A1 @ A2
        else:
            blk(M1, i)[:] = A1 @ blk(M1, i - 1) + A2 @ blk(M1, i - 2)
            blk(M2, i)[:] = blk(M1, i - 1) @ A2
    return M1, M2


def closed_loop(model, a, n_newton=1, k=1e-2, nu0=None, ramp=None):
    """a: (steps, n) turbulence coefficients of ONE realisation.  Returns dict of per-step x0, u (first moves),
    w, status.  nu0: (steps, nb*n) or None (zeros).  ramp = (du_min, du_max): the VAR_1 variant's ramp-rate rows
    against the previous first move (u_prev = U(1:nu), README.md:589; zeros at the first step), solved by the DENSE
    oracle (the structured one has no ramp rows)."""
    A1, A2, B, T = model["A1"], model["A2"], model["B"], model["T"]
    n, m = B.shape
    solver = BandedFastMPC(A1, A2, B, model["Q"], model["R"], model["Qf"], model["u_min"], model["u_max"],
                           model["x_min"], model["x_max"], T)
    if ramp is not None:
        from .dense_ref import DenseFastMPC
        var1 = model.get("var_order", 2) == 1

        class _Dense:
            def solve(self, x0, x0_pre, w, nw, kk, nu0=None):
                info = {}
                nu = np.zeros(T * n) if nu0 is None else nu0
                if var1:
                    d = DenseFastMPC.var1(model["Q"], model["R"], None, model["Qf"], None, None, None, model["x_min"],
                                          model["x_max"], model["u_min"], model["u_max"], ramp[0], ramp[1], T, x0, self.u_prev,
                                          A1, B, w, None, None, ramp=True)
                else:
                    d = DenseFastMPC(model["Q"], model["R"], None, model["Qf"], None, None, None, model["x_min"],
                                     model["x_max"], model["u_min"], model["u_max"], ramp[0], ramp[1], T, x0, x0_pre,
                                     self.u_prev, A1, A2, B, w, None, None, ramp=True)
                z = d.mpc_fixed_log_newton(nw, kk, nu0=nu, info=info)
                return z, info["nu"], info["iters"], 0
        solver = _Dense()
    M1, M2 = design_matrices(A1, A2, T)
    steps = a.shape[0]
    X0 = np.zeros((steps, n)); U0 = np.zeros((steps, m)); W = np.zeros((steps, T * n)); ST = np.zeros(steps, dtype=int)
    u1 = np.zeros(m); u2 = np.zeros(m); x0_last = np.zeros(n)
    for s in range(steps):
        x0 = a[s] + (B @ u1 if s >= 1 else 0.0)
        x0_pre = x0_last if s >= 1 else np.zeros(n)
        w = np.zeros(T * n)
        if s >= 1:
            w -= M1 @ (B @ u1)
        if s >= 2:
            w -= M2 @ (B @ u2)
        if ramp is not None:
            solver.u_prev = u1.copy()
        z, _, _, st = solver.solve(x0, x0_pre, w, n_newton, k, nu0=None if nu0 is None else nu0[s])
        u0 = z[:m].copy()
        X0[s], U0[s], W[s], ST[s] = x0, u0, w, st
        u2, u1, x0_last = u1, u0, x0
    return {"x0": X0, "u0": U0, "w": W, "status": ST}
