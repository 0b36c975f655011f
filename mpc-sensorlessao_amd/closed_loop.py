"""Device-resident coefficient-space closed loop around the fastMPC solve.

Mirrors the steps either side of the solver in the reference's simulation loop (README.md:444-622) for R
independent realisations at once, with the phase-screen estimator (README.md:456-480, out of scope) replaced by
the caller's residual-free turbulence coefficients a[k]:

    x0 = a[k] + B u[k-1]                      residual after the mirror's correction (README.md:482-483, 589-590)
    x0_pre = previous x0                      README.md:484-488
    w = b_ref = -M1 B u[k-1] - M2 B u[k-2]    README.md:490-497
    z = Fast_MPC2(..., w, [], []).mpc_fixed_log_newton(n_fix, k_fix)      README.md:547-555
    u[k] = U(1:nu)                            README.md:589

Everything stays in HBM between the steps: one call per step on torch's current stream (`fmpc_loop_step_device` =
`fmpc_loop_inputs_device` + `fmpc_solve_u0_device`; the two separate calls with `fused=False` or ramp rows), no host
round trip.
"""
from __future__ import annotations


class ClosedLoop:
    """ramp=True (after `handle.set_ramp`): every solve carries the VAR_1 variant's ramp-rate rows against the
    previous first move, u_prev = U(1:nu) (README.md:589; VAR_1/fast_mpc_ineq_const.m:58-76), zeros at the first step."""

    def __init__(self, handle, batch, n_newton=1, k=1e-2, device=None, ramp=False, fused=True, keep_z=True):
        import torch
        self.h, self.batch, self.n_newton, self.k = handle, int(batch), int(n_newton), float(k)
        dev = torch.device("cuda", handle.device) if device is None else device
        f64 = dict(dtype=torch.float64, device=dev)
        n, m, T = handle.n, handle.m, handle.T
        self.x0 = torch.zeros((batch, n), **f64)
        self.x0_pre = torch.zeros((batch, n), **f64)
        self.w = torch.zeros((batch, T * n), **f64)
        self.u = [torch.zeros((batch, m), **f64) for _ in range(3)]     # ring: u[k], u[k-1], u[k-2]
        # keep_z=False: only the first moves leave the solve (z_out = NULL at the C ABI; README.md:589 applies U(1:nu) only)
        self.z = torch.empty((batch, handle.nz), **f64) if (keep_z or ramp) else None
        self.status = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.iters = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.steps_done = 0
        self.ramp = bool(ramp)
        self.fused = bool(fused)        # one C call per step (fmpc_loop_step_device) instead of two

    def step(self, a_k, nu0=None):
        """One closed-loop step for all realisations.  a_k: (batch, n) device tensor.  Returns u[k] (batch, m),
        a view into the loop's ring buffer (valid until two further steps)."""
        s = self.steps_done
        u_new, u1, u2 = self.u[s % 3], self.u[(s - 1) % 3], self.u[(s - 2) % 3]
        if self.ramp or not self.fused:
            self.h.loop_inputs_device(a_k, self.x0 if s >= 1 else None, u1 if s >= 1 else None, u2 if s >= 2 else None,
                                      self.x0, self.x0_pre, self.w)
            self.h.solve_device(self.x0, self.x0_pre, self.w, None, nu0, self.n_newton, self.k, z_out=self.z,
                                status=self.status, iters=self.iters, u_prev=u1 if self.ramp else None, u0_out=u_new,
                                want_z=self.z is not None)
        else:
            self.h.loop_step_device(a_k, self.x0 if s >= 1 else None, u1 if s >= 1 else None, u2 if s >= 2 else None,
                                    self.x0, self.x0_pre, self.w, nu0, self.n_newton, self.k, z_out=self.z,
                                    status=self.status, iters=self.iters, u0_out=u_new)
        self.steps_done = s + 1
        return u_new

    def run(self, a, nu0=None):
        """a: (steps, batch, n) device tensor.  Returns (U0 (steps, batch, m), X0 (steps, batch, n))."""
        import torch
        steps = a.shape[0]
        U0 = torch.empty((steps, self.batch, self.h.m), dtype=torch.float64, device=a.device)
        X0 = torch.empty((steps, self.batch, self.h.n), dtype=torch.float64, device=a.device)
        for s in range(steps):
            u = self.step(a[s], None if nu0 is None else nu0[s])
            U0[s].copy_(u); X0[s].copy_(self.x0)
        return U0, X0
