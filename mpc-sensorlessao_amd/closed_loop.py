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
        # x0 alternates between two buffers (the fused step reads x0[k-1] while it writes x0[k]: with the update in place the
        # library cannot do loop inputs, first moves and decision in ONE launch -- fmpc_loop_step_device in include/fastmpc.h)
        self._xb = [torch.zeros((batch, n), **f64) for _ in range(2)]
        self._cur = 0
        self.x0_pre = torch.zeros((batch, n), **f64)
        self.w = torch.zeros((batch, T * n), **f64)
        self.u = [torch.zeros((batch, m), **f64) for _ in range(3)]     # ring: u[k], u[k-1], u[k-2]
        # keep_z=False: only the first moves leave the solve (z_out = NULL at the C ABI; README.md:589 applies U(1:nu) only)
        self.z = torch.empty((batch, handle.nz), **f64) if keep_z else None
        self.status = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.iters = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.steps_done = 0
        self.ramp = bool(ramp)
        self.fused = bool(fused)        # one C call per step (fmpc_loop_step_device) instead of two
        # The fused step is one C call whose arguments, apart from a[k] and nu0, are this object's own buffers: their
        # pointers are built once (a sequential loop of one realisation is bound by the HOST time per step otherwise:
        # tensor checks and ctypes conversions cost more than the two launches of the first-move form).
        import ctypes as C
        self._C, self._torch, self._dev = C, torch, dev
        vp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        self._p = dict(xb=[vp(b) for b in self._xb], x0_pre=vp(self.x0_pre), w=vp(self.w), u=[vp(u) for u in self.u], z=vp(self.z),
                       status=vp(self.status), iters=vp(self.iters))
        self._fn = handle._lib.fmpc_loop_step_device
        self._n_newton_c, self._k_c = int(self.n_newton), float(self.k)

    @property
    def x0(self):
        """x0 of the last step (batch, n)."""
        return self._xb[self._cur]

    def _step_fused(self, a_k, nu0, s):
        torch, C = self._torch, self._C
        if a_k.dtype != torch.float64 or not a_k.is_cuda or not a_k.is_contiguous() or a_k.shape[0] != self.batch or a_k.shape[-1] != self.h.n:
            raise ValueError("a_k: need a contiguous float64 HIP tensor of shape (batch, n)")
        if nu0 is not None and (nu0.dtype != torch.float64 or not nu0.is_contiguous() or nu0.numel() != self.batch * self.h.nu_len):
            raise ValueError("nu0: need a contiguous float64 HIP tensor of shape (batch, nu_len)")
        P = self._p
        u = P["u"]
        cur = self._cur
        rc = self._fn(self.h._h, self.batch, C.c_void_p(a_k.data_ptr()), P["xb"][cur] if s >= 1 else None, u[(s - 1) % 3] if s >= 1 else None,
                      u[(s - 2) % 3] if s >= 2 else None, P["xb"][1 - cur], P["x0_pre"], P["w"], None if nu0 is None else C.c_void_p(nu0.data_ptr()),
                      self._n_newton_c, self._k_c, P["z"], None, P["status"], P["iters"], None, u[s % 3],
                      C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream))
        if rc != 0:
            from ._lib import FastMPCError
            raise FastMPCError(rc, "fmpc_loop_step_device")
        self._cur = 1 - cur

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
            self._step_fused(a_k, nu0, s)
        self.steps_done = s + 1
        return u_new

    def run_recorded(self, a, nu0=None, want_x0=True):
        """The whole stretch a (steps, batch, n) in ONE C call (fmpc_loop_run_device): first moves only, fed back on the device.
        Continues from this object's state.  Returns (U0 (steps, batch, m), X0 (steps, batch, n) or None)."""
        torch, C = self._torch, self._C
        if self.ramp or not self.fused:
            raise ValueError("run_recorded: the fused step without ramp rows only")
        steps = a.shape[0]
        assert a.is_cuda and a.dtype == torch.float64 and a.is_contiguous() and tuple(a.shape[1:]) == (self.batch, self.h.n)
        assert nu0 is None or (nu0.is_contiguous() and nu0.dtype == torch.float64 and nu0.numel() == steps * self.batch * self.h.nu_len)
        U0 = torch.empty((steps, self.batch, self.h.m), dtype=torch.float64, device=a.device)
        X0 = torch.empty((steps, self.batch, self.h.n), dtype=torch.float64, device=a.device) if want_x0 else None
        s = self.steps_done
        P = self._p
        vp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        rc = self.h._lib.fmpc_loop_run_device(self.h._h, self.batch, steps, vp(a), vp(nu0), P["u"][(s - 1) % 3] if s >= 1 else None,
                                              P["u"][(s - 2) % 3] if s >= 2 else None, 1 if s >= 1 else 0, self._n_newton_c, self._k_c,
                                              P["xb"][self._cur], P["x0_pre"], P["w"], vp(U0), vp(X0), P["status"], P["iters"],
                                              C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream))
        if rc != 0:
            from ._lib import FastMPCError
            raise FastMPCError(rc, "fmpc_loop_run_device")
        # keep the ring consistent for a following step(): u[k-1], u[k-2] of the next step
        for j in range(min(steps, 2)):
            self.u[(s + steps - 1 - j) % 3].copy_(U0[steps - 1 - j])
        self.steps_done = s + steps
        return U0, X0

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


class AOLoop:
    """The reference's simulation loop WITH its estimator on the device (README.md:444-626) for `batch` realisations at once:
    residual screen = phase_valid[k] + sum_j (B u[k-1])_j Z_j (`fmpc_phase_residual_device`), the phase-diversity estimator
    (`PhaseDiversityEstimator`: three PSF windows, ad_est = G (Y_M - b_s)), x0 = ad_est, x0_pre = the previous ad_est,
    b_ref from the last two first moves (`fmpc_loop_inputs_device`), the fastMPC solve, u[k] = U(1:nu).
    Z: (n, len, len) mode maps indexed [j, row, column] (piston removed).  Screens are handed over indexed [b, row, column]."""

    def __init__(self, handle, estimator, Z, batch, n_newton=1, k=1e-2, one_call=True):
        import ctypes as C
        import numpy as np
        import torch
        self.h, self.est, self.batch, self.n_newton, self.k = handle, estimator, int(batch), int(n_newton), float(k)
        self.one_call = bool(one_call)      # b_ref + fastMPC step as ONE C call (fmpc_ao_step_device); False: loop inputs and solve as two
        dev = torch.device("cuda", handle.device)
        f64 = dict(dtype=torch.float64, device=dev)
        n, m, T = handle.n, handle.m, handle.T
        assert Z.shape[0] == n and Z.shape[1] == estimator.len
        self.Z = torch.from_numpy(np.ascontiguousarray(np.swapaxes(np.asarray(Z, dtype=np.float64), -1, -2))).to(dev)   # column-major planes
        self.npx = estimator.len ** 2
        self.scrn = torch.empty((batch, estimator.len, estimator.len), **f64)
        self._xb = [torch.zeros((batch, n), **f64) for _ in range(2)]          # ad_est of this and of the previous step, in turn
        self.x0, self.x0_pre = self._xb[0], self._xb[1]
        self.w = torch.zeros((batch, T * n), **f64)
        self._scr_x = torch.zeros((batch, n), **f64); self._scr_xp = torch.zeros((batch, n), **f64); self._zero_a = torch.zeros((batch, n), **f64)
        self.u = [torch.zeros((batch, m), **f64) for _ in range(3)]
        self.status = torch.zeros(batch, dtype=torch.int32, device=dev); self.iters = torch.zeros(batch, dtype=torch.int32, device=dev)
        self.steps_done = 0
        self._C, self._torch, self._dev = C, torch, dev

    def step(self, phase_k, noise=None, colmajor=False):
        """phase_k: (batch, len, len) float64 HIP tensor indexed [b, row, column] (phase_valid(:,:,k) of every realisation);
        colmajor=True: indexed [b, column, row] already -- the order MATLAB keeps phase_valid(:,:,k) in -- and no copy is made.
        Returns (u[k] (batch, m), ad_est (batch, n)); both views valid until two further steps / the next step."""
        torch, C = self._torch, self._C
        s = self.steps_done
        u_new, u1, u2 = self.u[s % 3], self.u[(s - 1) % 3], self.u[(s - 2) % 3]
        ph = phase_k if colmajor else phase_k.transpose(-1, -2).contiguous()          # column-major planes, as the mode maps
        assert ph.is_contiguous() and ph.dtype == torch.float64
        vp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)
        rc = self.h._lib.fmpc_phase_residual_device(self.h._h, self.batch, self.npx, vp(ph), vp(u1) if s >= 1 else None, vp(self.Z), vp(self.scrn), stream)
        if rc != 0:
            from ._lib import FastMPCError
            raise FastMPCError(rc, "fmpc_phase_residual_device")
        # x0 = ad_est, x0_pre = the previous ad_est (README.md:482-488): two buffers in turn, the estimator writes into this step's
        self.x0, self.x0_pre = self._xb[s % 2], self._xb[(s + 1) % 2]
        if s == 0:
            self.x0_pre.zero_()
        self.est.apply_device(self.scrn, noise, colmajor=True, out=self.x0)
        # b_ref = -M1 B u[k-1] - M2 B u[k-2] (README.md:490-497) and the fastMPC step on (x0, x0_pre, b_ref) in one call
        if self.one_call:
            rc = self.h._lib.fmpc_ao_step_device(self.h._h, self.batch, vp(self.x0), vp(self.x0_pre), vp(u1) if s >= 1 else None, vp(u2) if s >= 2 else None,
                                                 vp(self.w), None, self.n_newton, self.k, None, None, vp(self.status), vp(self.iters), None, vp(u_new), stream)
            if rc != 0:
                from ._lib import FastMPCError
                raise FastMPCError(rc, "fmpc_ao_step_device")
        else:
            self.h.loop_inputs_device(self._zero_a, None, u1 if s >= 1 else None, u2 if s >= 2 else None, self._scr_x, self._scr_xp, self.w)
            self.h.solve_device(self.x0, self.x0_pre, self.w, None, None, self.n_newton, self.k, z_out=None, status=self.status, iters=self.iters,
                                u0_out=u_new, want_z=False)
        self.steps_done = s + 1
        return u_new, self.x0
