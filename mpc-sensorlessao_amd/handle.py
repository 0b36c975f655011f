"""Shared-model handle: the batched entry point above the C ABI.

`FastMPCHandle` owns one `fmpc_handle` (include/fastmpc.h).  `solve` takes host numpy arrays
(problem-major: shape (batch, len), which is MATLAB's len x batch column-major layout);
`solve_device` takes torch tensors that already live in HBM and launches on torch's current
stream, so inputs stay resident between MPC steps and `torch.cuda.Event` timing sees the kernel.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FastMPCError


def _f64(a, shape=None, name="array"):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None and a.size != int(np.prod(shape)):
        raise FastMPCError(_lib.FMPC_E_DIM, f"{name}: expected {shape}, got {a.shape}")
    return a


def _colmajor(M, rows, cols, name):
    """Flatten a matrix to MATLAB's column-major order."""
    if M is None:
        return None
    M = np.asarray(M, dtype=np.float64)
    if M.ndim == 0:
        M = M.reshape(1, 1)
    if M.shape != (rows, cols):
        raise FastMPCError(_lib.FMPC_E_DIM, f"{name}: expected {(rows, cols)}, got {M.shape}")
    return np.ascontiguousarray(M.T).reshape(-1)      # C-order of M' == column-major of M


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class FastMPCHandle:
    """Everything that does not change between timesteps (Fast_MPC2.m:30-54 minus x0, x0_pre,
    w, x_init): A1, A2, B, Q, R, Qf, q, r, qf, bounds, xf, T."""

    def __init__(self, A1, A2, B, Q, R, Qf, u_min, u_max, x_min, x_max, T, q=None, r=None,
                 qf=None, xf=None, var_order=2, device=0):
        lib = _lib.load()
        A1 = np.asarray(A1, dtype=np.float64)
        B = np.asarray(B, dtype=np.float64)
        if A1.ndim != 2 or A1.shape[0] != A1.shape[1] or B.ndim != 2 or B.shape[0] != A1.shape[0]:
            raise FastMPCError(_lib.FMPC_E_DIM, "A1/B")
        n, m = A1.shape[0], B.shape[1]
        self.n, self.m, self.T = n, m, int(T)
        self.var_order = int(var_order)
        self.has_xf = xf is not None
        self.nz = self.T * (n + m)
        self.nu_len = n * (self.T + (1 if self.has_xf else 0))
        self.device = int(device)
        args = [
            _colmajor(A1, n, n, "A1"), _colmajor(A2, n, n, "A2"), _colmajor(B, n, m, "B"),
            _colmajor(Q, n, n, "Q"), _colmajor(R, m, m, "R"), _colmajor(Qf, n, n, "Qf"),
            _f64(q, (n,), "q"), _f64(r, (m,), "r"), _f64(qf, (n,), "qf"),
            _f64(x_min, (n,), "x_min"), _f64(x_max, (n,), "x_max"),
            _f64(u_min, (m,), "u_min"), _f64(u_max, (m,), "u_max"), _f64(xf, (n,), "xf"),
        ]
        h = C.c_void_p()
        rc = lib.fmpc_create(C.byref(h), n, m, self.T, self.var_order, *[_ptr(a) for a in args],
                             self.device)
        if rc != _lib.FMPC_OK:
            raise FastMPCError(rc, "fmpc_create")
        self._h = h
        self._lib = lib

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fmpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ host buffers
    def set_ramp(self, du_min, du_max):
        """Ramp-rate bounds of the VAR_1 variant (VAR_1/fast_mpc_ineq_const.m:58-76): fmpc_set_ramp.
        Afterwards `solve(..., u_prev=...)` / `solve_device(..., u_prev=...)` add the ramp rows."""
        du_min = _f64(du_min, (self.m,), "du_min"); du_max = _f64(du_max, (self.m,), "du_max")
        rc = self._lib.fmpc_set_ramp(self._h, _ptr(du_min), _ptr(du_max))
        if rc != _lib.FMPC_OK:
            raise FastMPCError(rc, "fmpc_set_ramp")

    def set_precision(self, mode):
        """fmpc_set_precision: 'f64' or 'f32' (fp32 factor + fp64 residuals, BASELINE configs[4])."""
        code = {"f64": _lib.FMPC_PREC_F64, "f32": _lib.FMPC_PREC_F32_MIXED}.get(mode, mode)
        rc = self._lib.fmpc_set_precision(self._h, int(code))
        if rc != _lib.FMPC_OK:
            raise FastMPCError(rc, "fmpc_set_precision")

    def solve(self, x0, x0_pre=None, w=None, z_init=None, nu0=None, n_newton=1, k=1e-2,
              return_info=False, check=True, u_prev=None, z_out=None):
        """One `inf_newton_solver` per problem.  x0: (batch, n) or (n,).  Returns z (batch, N_z)
        (or (N_z,) for a single vector input); with return_info also a dict nu/status/iters/step.
        u_prev (batch, m): solve WITH the ramp-rate rows (after `set_ramp`).
        z_out: a (batch, N_z) float64 C-contiguous array to receive z (a caller that solves batch after batch keeps ONE output
        array: the first touch of a fresh 82 MB array costs more than the solve and both transfers); nu is then only produced
        with return_info."""
        single = np.asarray(x0).ndim == 1
        x0 = _f64(x0)
        batch = 1 if single else x0.shape[0]
        x0 = _f64(x0, (batch, self.n), "x0")
        x0_pre = _f64(x0_pre, (batch, self.n), "x0_pre")
        w = _f64(w, (batch, self.T * self.n), "w")
        z_init = _f64(z_init, (batch, self.nz), "z_init")
        nu0 = _f64(nu0, (batch, self.nu_len), "nu0")
        n_newton = 0 if n_newton is None else int(n_newton)
        sld = self._lib.fmpc_step_ld(n_newton)
        if z_out is not None:
            if not (isinstance(z_out, np.ndarray) and z_out.dtype == np.float64 and z_out.flags.c_contiguous and z_out.shape == (batch, self.nz)):
                raise FastMPCError(_lib.FMPC_E_DIM, "z_out must be a C-contiguous float64 array of shape (batch, N_z)")
            z = z_out
        else:
            z = np.empty((batch, self.nz))
        nu = np.empty((batch, self.nu_len)) if (return_info or z_out is None) else None
        status = np.zeros(batch, dtype=np.int32)
        iters = np.zeros(batch, dtype=np.int32)
        step = np.empty((batch, sld))
        if u_prev is not None:
            u_prev = _f64(u_prev, (batch, self.m), "u_prev")
            rc = self._lib.fmpc_solve_ramp(self._h, batch, _ptr(x0), _ptr(x0_pre), _ptr(w), _ptr(u_prev),
                                           _ptr(z_init), _ptr(nu0), n_newton, float(k), _ptr(z), _ptr(nu),
                                           _ptr(status), _ptr(iters), _ptr(step))
        else:
            rc = self._lib.fmpc_solve(self._h, batch, _ptr(x0), _ptr(x0_pre), _ptr(w), _ptr(z_init),
                                      _ptr(nu0), n_newton, float(k), _ptr(z), _ptr(nu), _ptr(status),
                                      _ptr(iters), _ptr(step))
        if rc < 0 and check:
            raise FastMPCError(rc, "fmpc_solve_ramp" if u_prev is not None else "fmpc_solve")
        zr = z[0] if single else z
        if return_info:
            return zr, {"nu": nu[0] if single else nu, "status": status, "iters": iters,
                        "step": step, "rc": rc}
        return zr

    def unpack(self, z):
        """README.md:558-570,589: z -> (U, X, u0)."""
        z = _f64(z)
        single = z.ndim == 1
        batch = 1 if single else z.shape[0]
        z = _f64(z, (batch, self.nz), "z")
        U = np.empty((batch, self.T * self.m))
        X = np.empty((batch, self.T * self.n))
        u0 = np.empty((batch, self.m))
        rc = self._lib.fmpc_unpack(self._h, batch, _ptr(z), _ptr(U), _ptr(X), _ptr(u0))
        if rc != _lib.FMPC_OK:
            raise FastMPCError(rc, "fmpc_unpack")
        return (U[0], X[0], u0[0]) if single else (U, X, u0)

    def last_dispatch(self):
        """(path, handed_over) of the last solve: see fmpc_last_dispatch in include/fastmpc.h."""
        path = C.c_int(); cnt = C.c_int()
        rc = self._lib.fmpc_last_dispatch(self._h, C.byref(path), C.byref(cnt))
        if rc != _lib.FMPC_OK:
            raise FastMPCError(rc, "fmpc_last_dispatch")
        return path.value, cnt.value

    def last_tiled_wavefronts(self):
        """Wavefronts per problem of the last tiled-kernel launch (fmpc_last_tiled_wavefronts; 0 = none yet)."""
        return int(self._lib.fmpc_last_tiled_wavefronts(self._h))

    def set_dense_form(self, enabled, max_batch_with_w=-1):
        """fmpc_set_dense_form: the cold-start dual solve as one product with J = d nu+ / d [x0; x0_pre; w] (always
        without w; with w up to `max_batch_with_w` problems) instead of the two sweeps through the block factor."""
        rc = self._lib.fmpc_set_dense_form(self._h, int(bool(enabled)), int(max_batch_with_w))
        if rc != _lib.FMPC_OK:
            raise FastMPCError(rc, "fmpc_set_dense_form")

    def set_small_batch_kernel(self, tiled):
        """fmpc_set_small_batch_kernel: tiled kernel (True, default: lowest latency of one call; four wavefronts per problem up to 512
        problems) or the one-wavefront kernel (False: better when many handles have solves in flight) for small per-problem-factor
        batches and budget continuations; 2: the tiled kernel with two wavefronts per problem throughout (include/fastmpc.h)."""
        rc = self._lib.fmpc_set_small_batch_kernel(self._h, 2 if (tiled == 2 and tiled is not True) else (4 if tiled == 4 and tiled is not True else int(bool(tiled))))
        if rc != _lib.FMPC_OK:
            raise FastMPCError(rc, "fmpc_set_small_batch_kernel")

    def last_dual_form(self):
        """1 if the last solve took the dense form of the dual solve (fmpc_last_dual_form)."""
        return int(self._lib.fmpc_last_dual_form(self._h))

    # ------------------------------------------------------------------ device tensors
    def solve_device(self, x0, x0_pre=None, w=None, z_init=None, nu0=None, n_newton=1, k=1e-2,
                     z_out=None, nu_out=None, status=None, iters=None, step=None, u_prev=None, u0_out=None, want_z=True):
        """Asynchronous solve on torch CUDA(HIP) tensors, on torch's current stream.
        Returns (z_out, status, iters).  Nothing is copied through the host.
        u_prev (batch, m): solve WITH the ramp-rate rows (after `set_ramp`).
        u0_out (batch, m): also receives the first moves z[:, :m] (fmpc_solve_u0_device: no separate unpack).
        want_z=False (with u0_out): first moves only -- z_out = NULL at the C ABI, the returned z is None (README.md:589
        uses nothing but U(1:nu)).
        z_out may be a view with padded rows, e.g. torch.empty((batch, ldz))[:, :N_z] (fmpc_set_z_ld for this call): with ldz
        a multiple of 16 the cold-start step writes whole cache lines with non-temporal stores; only the cold-start step
        without w, z_init and budget > 1 takes such rows (FMPC_E_UNSUPPORTED otherwise)."""
        import torch

        def chk(t, cols, name, dtype=torch.float64):
            if t is None:
                return None
            if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
                raise FastMPCError(_lib.FMPC_E_DIM, f"{name}: need a contiguous {dtype} HIP tensor")
            if t.numel() != batch * cols:
                raise FastMPCError(_lib.FMPC_E_DIM, f"{name}: expected {(batch, cols)}")
            return t

        batch = x0.shape[0]
        dev = x0.device
        chk(x0, self.n, "x0"); chk(x0_pre, self.n, "x0_pre"); chk(w, self.T * self.n, "w")
        chk(z_init, self.nz, "z_init"); chk(nu0, self.nu_len, "nu0"); chk(u_prev, self.m, "u_prev")
        n_newton = 0 if n_newton is None else int(n_newton)
        if not want_z and u0_out is None:
            raise FastMPCError(_lib.FMPC_E_NULL, "want_z=False needs u0_out")
        if z_out is None and want_z:
            ld_ = getattr(self, "_z_ld", 0)
            z_out = (torch.empty((batch, ld_), dtype=torch.float64, device=dev)[:, :self.nz] if ld_ > self.nz
                     else torch.empty((batch, self.nz), dtype=torch.float64, device=dev))
        if not want_z:
            z_out = None
        if status is None:
            status = torch.empty(batch, dtype=torch.int32, device=dev)
        if iters is None:
            iters = torch.empty(batch, dtype=torch.int32, device=dev)
        # Row distance of z_out: an ARGUMENT of this call (fmpc_solve_u0_device_ld), never handle state.  A view of a wider array
        # (stride(0) > N_z) is taken as it is; a stride set with set_z_ld applies to the array this method allocates itself, and a
        # z_out whose rows do not have that distance is solved with ITS distance (round 4 wrote such rows at the handle's).
        zld = 0
        if (z_out is not None and z_out.dim() == 2 and tuple(z_out.shape) == (batch, self.nz) and z_out.is_cuda
                and z_out.dtype == torch.float64 and z_out.stride(1) == 1 and z_out.stride(0) > self.nz):
            zld = int(z_out.stride(0))                             # padded rows: a view of a wider array
        else:
            chk(z_out, self.nz, "z_out")
        chk(nu_out, self.nu_len, "nu_out"); chk(u0_out, self.m, "u0_out")
        chk(status, 1, "status", torch.int32); chk(iters, 1, "iters", torch.int32)
        if step is not None:
            chk(step, self._lib.fmpc_step_ld(n_newton), "step")
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if zld and u_prev is not None:
            raise FastMPCError(_lib.FMPC_E_UNSUPPORTED, "padded z rows: not with the ramp-rate rows")
        rc = self._solve_device_call(batch, p, x0, x0_pre, w, z_init, nu0, n_newton, k, z_out, nu_out, status, iters, step,
                                     u_prev, u0_out, stream, zld)
        if rc != _lib.FMPC_OK:
            raise FastMPCError(rc, "fmpc_solve_ramp_device" if u_prev is not None else "fmpc_solve_device")
        return z_out, status, iters

    def _solve_device_call(self, batch, p, x0, x0_pre, w, z_init, nu0, n_newton, k, z_out, nu_out, status, iters, step, u_prev, u0_out,
                           stream, zld=0):
        if u_prev is not None and u0_out is not None:
            rc = self._lib.fmpc_solve_ramp_u0_device(self._h, batch, p(x0), p(x0_pre), p(w), p(u_prev), p(z_init),
                                                     p(nu0), n_newton, float(k), p(z_out), p(nu_out), p(status),
                                                     p(iters), p(step), p(u0_out), stream)
        elif u_prev is not None:
            rc = self._lib.fmpc_solve_ramp_device(self._h, batch, p(x0), p(x0_pre), p(w), p(u_prev), p(z_init),
                                                  p(nu0), n_newton, float(k), p(z_out), p(nu_out), p(status),
                                                  p(iters), p(step), stream)
        else:
            rc = self._lib.fmpc_solve_u0_device_ld(self._h, batch, p(x0), p(x0_pre), p(w), p(z_init), p(nu0),
                                                   n_newton, float(k), p(z_out), p(nu_out), p(status),
                                                   p(iters), p(step), p(u0_out), int(zld), stream)
        return rc

    def solve_u0(self, x0, x0_pre=None, w=None, z_init=None, nu0=None, n_newton=1, k=1e-2, u0_out=None, return_info=False, check=True):
        """fmpc_solve_u0 (host arrays): the first moves u0 = U(1:nu) of every problem (README.md:589) and nothing else --
        m x batch doubles come back from the device instead of N_z x batch.  Returns u0 (batch, m) [, dict status/iters/rc]."""
        single = np.asarray(x0).ndim == 1
        x0 = _f64(x0)
        batch = 1 if single else x0.shape[0]
        x0 = _f64(x0, (batch, self.n), "x0")
        x0_pre = _f64(x0_pre, (batch, self.n), "x0_pre")
        w = _f64(w, (batch, self.T * self.n), "w")
        z_init = _f64(z_init, (batch, self.nz), "z_init")
        nu0 = _f64(nu0, (batch, self.nu_len), "nu0")
        if u0_out is not None:
            if not (isinstance(u0_out, np.ndarray) and u0_out.dtype == np.float64 and u0_out.flags.c_contiguous and u0_out.shape == (batch, self.m)):
                raise FastMPCError(_lib.FMPC_E_DIM, "u0_out must be a C-contiguous float64 array of shape (batch, m)")
            u0 = u0_out
        else:
            u0 = np.empty((batch, self.m))
        status = np.zeros(batch, dtype=np.int32)
        iters = np.zeros(batch, dtype=np.int32)
        rc = self._lib.fmpc_solve_u0(self._h, batch, _ptr(x0), _ptr(x0_pre), _ptr(w), _ptr(z_init), _ptr(nu0),
                                     0 if n_newton is None else int(n_newton), float(k), None, _ptr(u0), _ptr(status), _ptr(iters))
        if rc < 0 and check:
            raise FastMPCError(rc, "fmpc_solve_u0")
        ur = u0[0] if single else u0
        return (ur, {"status": status, "iters": iters, "rc": rc}) if return_info else ur

    def set_z_ld(self, ldz):
        """fmpc_set_z_ld: rows of z_out of the device-pointer solves ldz doubles apart (0: contiguous)."""
        rc = self._lib.fmpc_set_z_ld(self._h, int(ldz))
        if rc != _lib.FMPC_OK:
            raise FastMPCError(rc, "fmpc_set_z_ld")
        self._z_ld = int(ldz)

    def loop_inputs_device(self, a_k, x0_last, u1, u2, x0, x0_pre, w):
        """fmpc_loop_inputs_device: x0 = a_k + B u1, x0_pre = x0_last (None: zeros), w = -M1 B u1 - M2 B u2 (torch
        tensors on the device; None drops a term; x0 may be x0_last)."""
        import torch
        batch = a_k.shape[0]
        for t, cols, name in ((a_k, self.n, "a_k"), (x0_last, self.n, "x0_last"), (u1, self.m, "u1"), (u2, self.m, "u2"),
                              (x0, self.n, "x0"), (x0_pre, self.n, "x0_pre"), (w, self.T * self.n, "w")):
            if t is None:
                continue
            if not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous() or t.numel() != batch * cols:
                raise FastMPCError(_lib.FMPC_E_DIM, f"{name}: need a contiguous float64 HIP tensor of {(batch, cols)}")
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(a_k.device).cuda_stream)
        rc = self._lib.fmpc_loop_inputs_device(self._h, batch, p(a_k), p(x0_last), p(u1), p(u2), p(x0), p(x0_pre), p(w), stream)
        if rc != _lib.FMPC_OK:
            raise FastMPCError(rc, "fmpc_loop_inputs_device")

    def loop_step_device(self, a_k, x0_last, u1, u2, x0, x0_pre, w, nu0=None, n_newton=1, k=1e-2,
                         z_out=None, status=None, iters=None, u0_out=None):
        """fmpc_loop_step_device: loop inputs + solve with first-move output in one call (same results as
        `loop_inputs_device` followed by `solve_device(..., u0_out=...)`).  z_out=None: first moves only."""
        import torch
        batch = a_k.shape[0]
        for t, cols, name in ((a_k, self.n, "a_k"), (x0_last, self.n, "x0_last"), (u1, self.m, "u1"), (u2, self.m, "u2"),
                              (x0, self.n, "x0"), (x0_pre, self.n, "x0_pre"), (w, self.T * self.n, "w"),
                              (nu0, self.nu_len, "nu0"), (z_out, self.nz, "z_out"), (u0_out, self.m, "u0_out")):
            if t is None:
                continue
            if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and tuple(t.shape) == (batch, cols)):
                raise ValueError(f"{name}: need a contiguous float64 CUDA tensor of shape ({batch}, {cols})")
        for t, name in ((status, "status"), (iters, "iters")):
            if t is not None and not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() and tuple(t.shape) == (batch,)):
                raise ValueError(f"{name}: need a contiguous int32 CUDA tensor of shape ({batch},)")
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(a_k.device).cuda_stream)
        rc = self._lib.fmpc_loop_step_device(self._h, batch, p(a_k), p(x0_last), p(u1), p(u2), p(x0), p(x0_pre), p(w), p(nu0),
                                             int(n_newton), float(k), p(z_out), None, p(status), p(iters), None, p(u0_out), stream)
        if rc != _lib.FMPC_OK:
            raise FastMPCError(rc, "fmpc_loop_step_device")

    def unpack_device(self, z, U=None, X=None, u0=None):
        import torch
        batch = z.shape[0]
        dev = z.device
        if u0 is None:
            u0 = torch.empty((batch, self.m), dtype=torch.float64, device=dev)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = self._lib.fmpc_unpack_device(self._h, batch, p(z), p(U), p(X), p(u0), stream)
        if rc != _lib.FMPC_OK:
            raise FastMPCError(rc, "fmpc_unpack_device")
        return U, X, u0
