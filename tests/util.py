"""Shared helpers for the parity tests (checker side: these may import oracle/)."""
import numpy as np

from oracle.banded_ref import BandedFastMPC
from oracle.dense_ref import DenseFastMPC


def banded_from_model(model):
    return BandedFastMPC(model["A1"], model["A2"] if model.get("var_order", 2) == 2 else None,
                         model["B"], model["Q"], model["R"], model["Qf"], model["u_min"],
                         model["u_max"], model["x_min"], model["x_max"], model["T"],
                         q=model.get("q"), r=model.get("r"), qf=model.get("qf"), xf=model.get("xf"))


def dense_from_model(model, x0, x0_pre, w, x_init=None, **kw):
    n, m = model["n"], model["m"]
    if model.get("var_order", 2) == 2:
        return DenseFastMPC(model["Q"], model["R"], None, model["Qf"], model.get("q"), model.get("r"),
                            model.get("qf"), model["x_min"], model["x_max"], model["u_min"],
                            model["u_max"], None, None, model["T"], x0, x0_pre, np.zeros(m),
                            model["A1"], model["A2"], model["B"], w, model.get("xf"), x_init, **kw)
    kw.setdefault("ramp", False)
    return DenseFastMPC.var1(model["Q"], model["R"], None, model["Qf"], model.get("q"), model.get("r"),
                             model.get("qf"), model["x_min"], model["x_max"], model["u_min"],
                             model["u_max"], np.zeros(m), np.zeros(m), model["T"], x0, np.zeros(m),
                             model["A1"], model["B"], w, model.get("xf"), x_init, **kw)


def handle_from_model(pkg, model, device=0):
    return pkg.FastMPCHandle(model["A1"], model["A2"] if model.get("var_order", 2) == 2 else None,
                             model["B"], model["Q"], model["R"], model["Qf"], model["u_min"],
                             model["u_max"], model["x_min"], model["x_max"], model["T"],
                             q=model.get("q"), r=model.get("r"), qf=model.get("qf"),
                             xf=model.get("xf"), var_order=model.get("var_order", 2), device=device)


def oracle_batch(model, data, n_newton, k, z_init=None):
    """Structured oracle over a batch -> z, nu, iters, status, steps (list of t lists)."""
    b = banded_from_model(model)
    B = data["x0"].shape[0]
    nz = model["T"] * (model["n"] + model["m"])
    z = np.empty((B, nz)); nu = np.empty((B, b.nb * b.n))
    iters = np.zeros(B, dtype=int); status = np.zeros(B, dtype=int); steps = []
    for p in range(B):
        info = {}
        zz, nn, it, st = b.solve(data["x0"][p], None if data.get("x0_pre") is None else data["x0_pre"][p],
                                 None if data.get("w") is None else data["w"][p], n_newton, k,
                                 z_init=None if z_init is None else z_init[p],
                                 nu0=None if data.get("nu0") is None else data["nu0"][p], info=info)
        z[p], nu[p], iters[p], status[p] = zz, nn, it, st
        steps.append(info.get("t", []))
    return z, nu, iters, status, steps


def rel_err(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def canon_steps(t):
    """Line-search steps with the collapse case canonicalised: the reference's literal loop leaves a
    failing search at t ~ 2^-50 (where (1 - al*t) rounds to 1), the closed form at exactly 0 after 64
    halvings (FMPC_W_LINESEARCH); both mean "no move" (SURVEY App. B-D2)."""
    t = np.asarray(t, dtype=np.float64).copy()
    t[(t >= 0) & (t < 1e-12)] = 0.0
    return t
