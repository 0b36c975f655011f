"""TEST INFRASTRUCTURE -- structured CPU oracle #2 for the fastMPC hot path.  PARITY UNPINNED.

Same Newton iteration as `oracle/dense_ref.py` (which follows the reference's MATLAB op for op),
but with the structure of the KKT system exploited exactly as the HIP kernels do
(SURVEY.md App. A.4/A.5): Phi is block-diagonal, the Schur complement Y = C Phi^-1 C' is
block-penta-diagonal with n x n blocks, the line search is the closed form of the frozen-d
residual.  It exists so that GPU results can be checked at sizes where the dense oracle needs
seconds per Newton step.  It is pinned against the dense oracle in
`tests/test_oracle_banded.py` (<= 1e-10 relative); the dense oracle itself is PARITY UNPINNED
(see its header), hence so is this file.

Checker only: imported by `tests/`, `__graft_entry__.smoke()`, `bench.py`'s cpu_baseline leg.

Reference lines restated (under /root/reference/Fast_MPC/VAR_2):
  residuals r_d, r_p, early exit     inf_newton_solver.m:11-22
  Phi = 2H + k P'DP                  inf_newton_KKT_H.m:3-13
  Y = C Phi^-1 C', chol, dnu, dz     inf_newton_solver.m:24-35
  line search                        backtracking_inf_newton.m:2-11
  problem data                       fast_mpc_eq_const.m:38-71, fast_mpc_objective.m:50-65,
                                     fast_mpc_ineq_const.m:46-56, fast_mpc_init.m:19-25
"""
from __future__ import annotations

import numpy as np

MAX_HALVINGS = 64      # quirk D2: the reference halves ~1075 times until t underflows to 0

ST_OK = 0
ST_W_LINESEARCH = 1
ST_E_NOT_PD_PHI = -4
ST_E_NOT_PD_SCHUR = -5


def _is_diag(M):
    return np.count_nonzero(M - np.diag(np.diagonal(M))) == 0


class BandedFastMPC:
    """Shared model data ("handle"): everything that does not change between MPC steps."""

    def __init__(self, A1, A2, B, Q, R, Qf, u_min, u_max, x_min, x_max, T, q=None, r=None,
                 qf=None, xf=None):
        self.A1 = np.asarray(A1, dtype=np.float64)
        n = self.A1.shape[0]
        self.A2 = np.zeros((n, n)) if A2 is None else np.asarray(A2, dtype=np.float64)
        self.B = np.asarray(B, dtype=np.float64)
        m = self.B.shape[1]
        self.n, self.m, self.T = n, m, int(T)
        self.Q, self.R, self.Qf = (np.asarray(v, dtype=np.float64) for v in (Q, R, Qf))
        self.q = np.zeros(n) if q is None else np.asarray(q, dtype=np.float64).reshape(-1)
        self.r = np.zeros(m) if r is None else np.asarray(r, dtype=np.float64).reshape(-1)
        self.qf = np.zeros(n) if qf is None else np.asarray(qf, dtype=np.float64).reshape(-1)
        self.u_min = np.asarray(u_min, dtype=np.float64).reshape(-1)
        self.u_max = np.asarray(u_max, dtype=np.float64).reshape(-1)
        self.x_min = np.asarray(x_min, dtype=np.float64).reshape(-1)
        self.x_max = np.asarray(x_max, dtype=np.float64).reshape(-1)
        self.xf = None if xf is None else np.asarray(xf, dtype=np.float64).reshape(-1)
        self.nb = self.T + (1 if self.xf is not None else 0)     # block rows of C / Y
        self.R_diag = _is_diag(self.R)
        # inverse of the (constant) state part of Phi: X = (2Q)^-1, Xf = (2Qf)^-1
        self.X = np.linalg.inv(2 * self.Q)
        self.Xf = np.linalg.inv(2 * self.Qf)
        self._const_blocks()

    def _Xj(self, j):
        """Phi^-1 block of x_j, j = 1..T (x_T carries Qf: fast_mpc_objective.m:55)."""
        return self.Xf if j == self.T else self.X

    def _const_blocks(self):
        """Iteration-invariant parts of Y."""
        T, A1, A2 = self.T, self.A1, self.A2
        nb = self.nb
        self.Yd = [None] * nb      # Y_ii without the B Rt^-1 B' term
        self.Y1 = [None] * nb      # Y_{i,i+1}
        self.Y2 = [None] * nb      # Y_{i,i+2}
        for i in range(T):
            Yd = self._Xj(i + 1).copy()
            if i >= 1:
                Yd += A1 @ self._Xj(i) @ A1.T
            if i >= 2:
                Yd += A2 @ self._Xj(i - 1) @ A2.T
            self.Yd[i] = Yd
            if i + 1 < T:
                Y1 = -self._Xj(i + 1) @ A1.T
                if i >= 1:
                    Y1 += A1 @ self._Xj(i) @ A2.T
                self.Y1[i] = Y1
            if i + 2 < T:
                self.Y2[i] = -self._Xj(i + 1) @ A2.T
        if self.xf is not None:
            self.Yd[T] = self.Xf.copy()
            self.Y1[T - 1] = self.Xf.copy()      # rows T-1 and T share x_T only

    # -------------------------------------------------------------- per-problem pieces
    def cold_start(self):
        """fast_mpc_init.m:19-25."""
        n, m, T = self.n, self.m, self.T
        z = np.zeros((T, m + n))
        z[:, :m] = (self.u_min + self.u_max) / 2
        z[:, m:] = (self.x_min + self.x_max) / 2
        return z.reshape(-1)

    def rhs_b(self, x0, x0_pre, w):
        """fast_mpc_eq_const.m:39,44,47 (+ :68 for xf)."""
        n, T = self.n, self.T
        b = np.zeros((self.nb, n))
        if w is not None:
            b[:T] = np.asarray(w, dtype=np.float64).reshape(T, n)
        b[0] += self.A1 @ x0 + self.A2 @ x0_pre
        if T > 1:
            b[1] += self.A2 @ x0
        if self.xf is not None:
            b[T] = self.xf
        return b

    def solve(self, x0, x0_pre, w, n_newton, k, z_init=None, nu0=None, info=None):
        """One `inf_newton_solver` call (n_newton <= 0 -> 1000 iterations + tolerance exit).

        Returns (z, nu, iters, status)."""
        n, m, T, nb = self.n, self.m, self.T, self.nb
        A1, A2, B = self.A1, self.A2, self.B
        x0 = np.asarray(x0, dtype=np.float64).reshape(-1)
        x0_pre = np.zeros(n) if x0_pre is None else np.asarray(x0_pre, dtype=np.float64).reshape(-1)
        b = self.rhs_b(x0, x0_pre, w)
        z = (self.cold_start() if z_init is None else np.asarray(z_init, dtype=np.float64)).copy()
        Z = z.reshape(T, m + n)
        U, Xs = Z[:, :m].copy(), Z[:, m:].copy()         # u_j ; x_{j+1}
        nu = np.zeros(nb * n) if nu0 is None else np.asarray(nu0, dtype=np.float64).reshape(-1).copy()
        NU = nu.reshape(nb, n)
        max_iter = 1000 if (n_newton is None or n_newton <= 0) else int(n_newton)
        status = ST_OK
        steps = 0
        Rd = np.diagonal(self.R)
        for _ in range(max_iter):
            # ---- barrier pieces (inf_newton_KKT_H.m:3-13; slacks are NOT checked, quirk D3)
            sp = self.u_max[None, :] - U
            sm = U - self.u_min[None, :]
            dp, dm = 1.0 / sp, 1.0 / sm
            hess = k * (dp * dp + dm * dm)              # k * diag(P'DP) on the u entries
            bar = k * (dp - dm)                         # k * P'd
            # ---- residuals (inf_newton_solver.m:12-17)
            if self.R_diag:
                rdu = 2 * Rd[None, :] * U + self.r[None, :] + bar - NU[:T] @ B
            else:
                rdu = 2 * U @ self.R.T + self.r[None, :] + bar - NU[:T] @ B
            rdx = np.empty((T, n))
            for j in range(1, T + 1):                   # x_j lives in Xs[j-1]
                Qj = self.Qf if j == T else self.Q
                qj = self.qf if j == T else self.q
                v = 2 * Qj @ Xs[j - 1] + qj + NU[j - 1]
                if j < T:
                    v -= A1.T @ NU[j]
                if j + 1 < T:
                    v -= A2.T @ NU[j + 1]
                if j == T and self.xf is not None:
                    v += NU[T]
                rdx[j - 1] = v
            rp = np.empty((nb, n))
            for i in range(T):
                v = Xs[i] - B @ U[i] - b[i]
                if i >= 1:
                    v -= A1 @ Xs[i - 1]
                if i >= 2:
                    v -= A2 @ Xs[i - 2]
                rp[i] = v
            if self.xf is not None:
                rp[T] = Xs[T - 1] - b[T]
            rho2 = float(np.sum(rdu * rdu) + np.sum(rdx * rdx) + np.sum(rp * rp))
            n_r, n_g = np.sqrt(rho2), np.sqrt(float(np.sum(rp * rp)))
            if info is not None:
                info.setdefault("n_r", []).append(n_r)
                info.setdefault("n_g", []).append(n_g)
            if n_r <= 1e-6 and n_g <= 1e-8:             # :19-22, tested before the step
                break
            # ---- Phi^-1 r_d and the rhs  -beta = r_p - C Phi^-1 r_d   (:28-29)
            if self.R_diag:
                rt = 2 * Rd[None, :] + hess             # diag of Rt_j
                if np.any(rt <= 0) or not np.all(np.isfinite(rt)):
                    status = ST_E_NOT_PD_PHI
                    break
                winv = 1.0 / rt
                phu = rdu * winv
            else:
                Rt_inv = []
                phu = np.empty((T, m))
                try:
                    for j in range(T):
                        Rt = 2 * self.R + np.diag(hess[j])
                        np.linalg.cholesky(Rt)
                        Ri = np.linalg.inv(Rt)
                        Rt_inv.append(Ri)
                        phu[j] = Ri @ rdu[j]
                except np.linalg.LinAlgError:
                    status = ST_E_NOT_PD_PHI
                    break
            phx = np.empty((T, n))
            for j in range(1, T + 1):
                phx[j - 1] = self._Xj(j) @ rdx[j - 1]
            rhs = np.empty((nb, n))
            for i in range(T):
                cv = phx[i] - B @ phu[i]
                if i >= 1:
                    cv -= A1 @ phx[i - 1]
                if i >= 2:
                    cv -= A2 @ phx[i - 2]
                rhs[i] = rp[i] - cv
            if self.xf is not None:
                rhs[T] = rp[T] - phx[T - 1]
            # ---- block-penta-diagonal Cholesky of Y fused with the forward sweep (:27,:30-31)
            Ld = [None] * nb
            L1 = [None] * nb      # L_{i+1,i}
            L2 = [None] * nb      # L_{i+2,i}
            y = np.empty((nb, n))
            ok = True
            for i in range(nb):
                S = self.Yd[i].copy()
                if i < T:
                    if self.R_diag:
                        S += (B * winv[i][None, :]) @ B.T
                    else:
                        S += B @ Rt_inv[i] @ B.T
                s = rhs[i].copy()
                if i >= 1:
                    S -= L1[i - 1] @ L1[i - 1].T
                    s -= L1[i - 1] @ y[i - 1]
                if i >= 2 and L2[i - 2] is not None:
                    S -= L2[i - 2] @ L2[i - 2].T
                    s -= L2[i - 2] @ y[i - 2]
                try:
                    L = np.linalg.cholesky(S)
                except np.linalg.LinAlgError:
                    ok = False
                    break
                Ld[i] = L
                y[i] = np.linalg.solve(L, s)
                if i + 1 < nb:
                    M = self.Y1[i].T.copy()             # Y_{i+1,i}
                    if i >= 1 and L2[i - 1] is not None:
                        M -= L2[i - 1] @ L1[i - 1].T    # L_{i+1,i-1} L_{i,i-1}'
                    L1[i] = np.linalg.solve(L, M.T).T
                if i + 2 < nb and self.Y2[i] is not None:
                    L2[i] = np.linalg.solve(L, self.Y2[i]).T   # Y_{i+2,i} L^-T
            if not ok:
                status = ST_E_NOT_PD_SCHUR
                break
            # ---- backward sweep (:32)
            dnu = np.empty((nb, n))
            for i in range(nb - 1, -1, -1):
                v = y[i].copy()
                if i + 1 < nb:
                    v -= L1[i].T @ dnu[i + 1]
                if i + 2 < nb and L2[i] is not None:
                    v -= L2[i].T @ dnu[i + 2]
                dnu[i] = np.linalg.solve(Ld[i].T, v)
            # ---- dz = Phi^-1 (-r_d - C' dnu)   (:34-35)
            tu = -rdu + dnu[:T] @ B
            du = tu * winv if self.R_diag else np.stack([Rt_inv[j] @ tu[j] for j in range(T)])
            dx = np.empty((T, n))
            for j in range(1, T + 1):
                v = -rdx[j - 1] - dnu[j - 1]
                if j < T:
                    v += A1.T @ dnu[j]
                if j + 1 < T:
                    v += A2.T @ dnu[j + 1]
                if j == T and self.xf is not None:
                    v -= dnu[T]
                dx[j - 1] = self._Xj(j) @ v
            # ---- line search, closed form of backtracking_inf_newton.m:2-11 (App. A.5)
            e = hess * du                                # k P'DP dz (zero on x entries)
            beta_e = float(np.sum(rdu * e))
            eps2 = float(np.sum(e * e))
            t, halv = 1.0, 0
            al = 1e-4
            while True:
                # ||r(t)||^2 - ((1-al t) rho)^2 = t * gq(t)
                gq = (t - 2 + 2 * al - al * al * t) * rho2 - 2 * (1 - t) * beta_e + t * eps2
                if gq <= 0:
                    break
                t *= 0.5
                halv += 1
                if halv >= MAX_HALVINGS:
                    t = 0.0
                    status = ST_W_LINESEARCH
                    break
            if info is not None:
                info.setdefault("t", []).append(t)
                info.setdefault("halvings", []).append(halv)
                info.setdefault("eps2", []).append(eps2)          # the sums the step-length decision rests on
                info.setdefault("rp2", []).append(float(np.sum(rp * rp)))
                info.setdefault("rho2", []).append(rho2)
            U += t * du
            Xs += t * dx
            NU += t * dnu
            steps += 1
        Z[:, :m], Z[:, m:] = U, Xs
        return z, NU.reshape(-1).copy(), steps, status

    # dense blocks for structure tests
    def dense_Y(self, winv):
        """Assemble Y densely from the block formulas (tests only)."""
        n, nb, T = self.n, self.nb, self.T
        Y = np.zeros((nb * n, nb * n))
        for i in range(nb):
            D = self.Yd[i].copy()
            if i < T:
                D += (self.B * winv[i][None, :]) @ self.B.T
            Y[i * n:(i + 1) * n, i * n:(i + 1) * n] = D
            if i + 1 < nb and self.Y1[i] is not None:
                Y[i * n:(i + 1) * n, (i + 1) * n:(i + 2) * n] = self.Y1[i]
                Y[(i + 1) * n:(i + 2) * n, i * n:(i + 1) * n] = self.Y1[i].T
            if i + 2 < nb and self.Y2[i] is not None:
                Y[i * n:(i + 1) * n, (i + 2) * n:(i + 3) * n] = self.Y2[i]
                Y[(i + 2) * n:(i + 3) * n, i * n:(i + 1) * n] = self.Y2[i].T
        return Y
