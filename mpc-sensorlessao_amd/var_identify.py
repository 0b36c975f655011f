"""VAR(2) identification on the device: the step of the reference notebook that produces A1, A2 from an open-loop
series of Zernike coefficients (README.md:108-130), above `fmpc_var_identify_device` (include/fastmpc.h)."""
from __future__ import annotations

import ctypes as C

from . import _lib
from ._lib import FastMPCError


def identify_var2_device(series, num_train=None):
    """series: torch float64 HIP tensor (batch, num_samples, n) or (num_samples, n) -- ad_acc with the piston column
    removed, one row per time step.  Returns (A1, A2, status): (batch, n, n) tensors with A[b, i, j] = A_b(i, j)."""
    import torch
    lib = _lib.load()
    single = series.dim() == 2
    s = series.unsqueeze(0) if single else series
    if not s.is_cuda or s.dtype != torch.float64 or not s.is_contiguous():
        raise FastMPCError(_lib.FMPC_E_DIM, "series: need a contiguous float64 HIP tensor")
    batch, ns, n = s.shape
    nt = ns if num_train is None else int(num_train)
    A1 = torch.empty((batch, n, n), dtype=torch.float64, device=s.device)       # filled column-major: transposed below
    A2 = torch.empty_like(A1)
    st = torch.zeros(batch, dtype=torch.int32, device=s.device)
    stream = C.c_void_p(torch.cuda.current_stream(s.device).cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.fmpc_var_identify_device(n, nt, ns, batch, p(s), p(A1), p(A2), p(st), stream)
    if rc != _lib.FMPC_OK:
        raise FastMPCError(rc, "fmpc_var_identify_device")
    A1, A2 = A1.transpose(1, 2), A2.transpose(1, 2)                              # column-major n x n -> [i, j]
    return (A1[0], A2[0], st[0]) if single else (A1, A2, st)
