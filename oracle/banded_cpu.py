"""TEST / BASELINE INFRASTRUCTURE -- ctypes binding of oracle/banded_cpu.c (structured CPU solver, OpenMP over the batch).
PARITY UNPINNED like every oracle here.  Used by tests/ (pinned to oracle/banded_ref.py) and by bench.py's
cpu_baseline leg; the product package never imports this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbanded_cpu.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.banded_cpu_solve_batch.restype = C.c_int
        _lib.banded_cpu_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def solve_batch(model, data, n_newton, k, z_init=None, threads=0, out=None):
    """model / data as in mpc-sensorlessao_amd.synthetic.  Returns z, nu, iters, status, step (batch x step_ld).
    out: the tuple a previous call of the same shape returned -- its arrays are written again (a timing loop then does
    not pay the first touch of 41 KB of z per problem on every repetition)."""
    lib = load()
    n, m, T = model["n"], model["m"], model["T"]
    var2 = model.get("var_order", 2) == 2
    xf = model.get("xf")
    nb = T + (1 if xf is not None else 0)
    for name in ("Q", "R", "Qf"):
        M = np.asarray(model[name])
        assert np.count_nonzero(M - np.diag(np.diagonal(M))) == 0, "banded_cpu: diagonal weights only"
    A1 = _c(model["A1"]); A2 = _c(model["A2"]) if var2 else None; B = _c(model["B"])
    Q2 = _c(2 * np.diagonal(model["Q"])); R2 = _c(2 * np.diagonal(model["R"])); Qf2 = _c(2 * np.diagonal(model["Qf"]))
    x0 = _c(data["x0"]); batch = x0.shape[0]
    x0p = _c(data.get("x0_pre")); w = _c(data.get("w")); nu0 = _c(data.get("nu0")); zi = _c(z_init)
    sld = n_newton if n_newton and n_newton > 0 else 1000
    if out is not None:
        z, nu, iters, status, step = out
        assert z.shape == (batch, T * (n + m)) and nu.shape == (batch, nb * n) and step.shape == (batch, sld)
    else:
        z = np.empty((batch, T * (n + m))); nu = np.empty((batch, nb * n))
        iters = np.zeros(batch, dtype=np.int32); status = np.zeros(batch, dtype=np.int32); step = np.empty((batch, sld))
    keep = [_c(model.get("q")), _c(model.get("r")), _c(model.get("qf")), _c(model["u_min"]), _c(model["u_max"]),
            _c(model["x_min"]), _c(model["x_max"]), _c(xf)]
    rc = lib.banded_cpu_solve_batch(C.c_int(n), C.c_int(m), C.c_int(T), _p(A1), _p(A2), _p(B), _p(Q2), _p(R2), _p(Qf2),
                                    *[_p(a) for a in keep], C.c_int(batch), _p(x0), _p(x0p), _p(w), _p(zi), _p(nu0),
                                    C.c_int(int(n_newton or 0)), C.c_double(k), _p(z), _p(nu), _p(iters), _p(status),
                                    _p(step), C.c_int(sld), C.c_int(threads))
    assert rc == 0
    return z, nu, iters, status, step


def max_threads():
    return int(load().banded_cpu_max_threads())
