"""Phase-diversity estimator on the device (include/fastmpc.h: fmpc_est_*): the "% Estimator" block of the reference's
simulation loop, README.md:456-480, for a batch of residual phase screens at once.

    est = PhaseDiversityEstimator(pupil, W, zd_list, dx, range_min, range_max, A_s, b_s)   # MATLAB's 1-based range_min/max
    ad_est = est.apply_device(scrn)            # scrn: (batch, len, len) HIP tensor, scrn[b, i, j] = MATLAB's scrn(i, j)

Arrays are handed over as MATLAB has them (column-major); numpy / torch arrays indexed [row, column] are transposed here."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FastMPCError


def _colmajor(a):
    """(len, len) or (k, len, len) indexed [.., row, column] -> contiguous column-major planes."""
    a = np.asarray(a, dtype=np.float64)
    return np.ascontiguousarray(np.swapaxes(a, -1, -2))


class PhaseDiversityEstimator:
    def __init__(self, pupil, W, zd_list, dx, range_min, range_max, A_s, b_s, AU=1e12, device=0):
        """range_min, range_max: the reference's 1-based window bounds (README.md:378-379); A_s (p, nx), b_s (p)."""
        self._lib = _lib.load()
        pupil = np.asarray(pupil, dtype=np.float64)
        W = np.asarray(W, dtype=np.float64)
        zd = np.asarray(zd_list, dtype=np.float64).reshape(-1)
        length = pupil.shape[0]
        D = pupil[None] * np.exp(1j * zd[:, None, None] * W[None])       # README.md:464-465 without the screen
        Dre, Dim = _colmajor(D.real), _colmajor(D.imag)
        A_s = np.asfortranarray(np.asarray(A_s, dtype=np.float64))
        b_s = np.ascontiguousarray(np.asarray(b_s, dtype=np.float64))
        p, nx = A_s.shape
        d = int(range_max) - int(range_min) + 1
        h = C.c_void_p()
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = self._lib.fmpc_est_create(C.byref(h), length, int(range_min) - 1, d, len(zd), ptr(Dre), ptr(Dim), float(dx) ** 4 * float(AU),
                                       ptr(A_s), ptr(b_s), p, nx, int(device))
        if rc != 0:
            raise FastMPCError(rc, "fmpc_est_create")
        self._h, self.len, self.d, self.ndiv, self.nx, self.p, self.device = h, length, d, len(zd), nx, p, int(device)
        rk = C.c_int()
        self._lib.fmpc_est_dims(self._h, None, None, None, None, None, C.byref(rk))
        self.rank = rk.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fmpc_est_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def apply_device(self, scrn, noise=None, want_Y=False, colmajor=False, out=None):
        """scrn: (batch, len, len) float64 HIP tensor indexed [b, row, column] (colmajor=True: already [b, column, row], no copy).
        Returns ad_est (batch, nx) [, Y_M (batch, p)] on torch's current stream; out: a (batch, nx) tensor to write ad_est into."""
        import torch
        if not colmajor:
            scrn = scrn.transpose(-1, -2).contiguous()
        assert scrn.is_cuda and scrn.dtype == torch.float64 and scrn.is_contiguous() and tuple(scrn.shape[1:]) == (self.len, self.len)
        batch = scrn.shape[0]
        if out is not None:
            assert out.is_cuda and out.dtype == torch.float64 and out.is_contiguous() and tuple(out.shape) == (batch, self.nx)
        ad = out if out is not None else torch.empty((batch, self.nx), dtype=torch.float64, device=scrn.device)
        Y = torch.empty((batch, self.p), dtype=torch.float64, device=scrn.device) if want_Y else None
        if noise is not None:
            assert noise.is_cuda and noise.dtype == torch.float64 and noise.is_contiguous() and tuple(noise.shape) == (batch, self.p)
        vp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        rc = self._lib.fmpc_est_apply_device(self._h, batch, vp(scrn), vp(noise), vp(ad), vp(Y),
                                             C.c_void_p(torch.cuda.current_stream(scrn.device).cuda_stream))
        if rc != 0:
            raise FastMPCError(rc, "fmpc_est_apply_device")
        return (ad, Y) if want_Y else ad

    def apply(self, scrn, noise=None, want_Y=False):
        """Host arrays: scrn (batch, len, len) indexed [b, row, column]."""
        scrn = _colmajor(scrn)
        batch = scrn.shape[0]
        ad = np.empty((batch, self.nx)); Y = np.empty((batch, self.p)) if want_Y else None
        nz = None if noise is None else np.ascontiguousarray(np.asarray(noise, dtype=np.float64))
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        rc = self._lib.fmpc_est_apply(self._h, batch, ptr(scrn), ptr(nz), ptr(ad), ptr(Y))
        if rc != 0:
            raise FastMPCError(rc, "fmpc_est_apply")
        return (ad, Y) if want_Y else ad
