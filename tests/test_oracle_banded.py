"""Structured oracle (oracle/banded_ref.py) pinned against the dense op-for-op oracle (<= 1e-10)."""
import importlib

import numpy as np
import pytest

from oracle.banded_ref import BandedFastMPC
from oracle.dense_ref import inf_newton_KKT_H
from tests.util import banded_from_model, canon_steps, dense_from_model, rel_err

S = importlib.import_module("mpc-sensorlessao_amd").synthetic


def _dense_weights(md, seed):
    rng = np.random.default_rng(seed); n, m = md["n"], md["m"]
    G = rng.standard_normal((n, n)); md["Q"] = G @ G.T / n + np.eye(n)
    G = rng.standard_normal((n, n)); md["Qf"] = G @ G.T / n + 5 * np.eye(n)
    G = rng.standard_normal((m, m)); md["R"] = G @ G.T / m + np.eye(m)
    return md


@pytest.mark.parametrize("xf", [False, True])
@pytest.mark.parametrize("var_order", [1, 2])
def test_structure_of_Phi_and_Y_T2(xf, var_order):
    md, data = S.make_test_problem(6, 4, 5, seed=11, xf=xf, var_order=var_order)
    d = dense_from_model(md, data["x0"][0], data["x0_pre"][0], data["w"][0])
    z, H, g, P, h, C, b = d._assemble()
    z = z + 0.05 * np.random.default_rng(3).standard_normal(z.size)
    k = 0.2
    Phi, _ = inf_newton_KKT_H(H, P, h, z, k)
    Y = C @ np.linalg.solve(Phi, C.T)
    n, m = 6, 4
    # Phi is diagonal here (diagonal Q, R)
    assert np.count_nonzero(Phi - np.diag(np.diag(Phi))) == 0
    bm = banded_from_model(md)
    U = z.reshape(5, 10)[:, :4]
    winv = 1.0 / (2 * np.diag(md["R"])[None, :] + k * (1 / (md["u_max"] - U) ** 2 + 1 / (U - md["u_min"]) ** 2))
    assert rel_err(bm.dense_Y(winv), Y) < 1e-13
    nb = bm.nb
    for i in range(nb):
        for j in range(nb):
            blk = Y[i * n:(i + 1) * n, j * n:(j + 1) * n]
            if abs(i - j) > (2 if var_order == 2 else 1):
                assert np.count_nonzero(blk) == 0


@pytest.mark.parametrize("case", [
    dict(n=8, m=5, T=10, xf=True, nw=5, k=0.01), dict(n=8, m=5, T=10, xf=False, nw=1, k=0.01),
    dict(n=8, m=5, T=10, xf=True, nw=5, k=0.01, dense=True), dict(n=4, m=3, T=1, nw=3, k=0.1),
    dict(n=4, m=3, T=2, xf=True, nw=3, k=0.1), dict(n=8, m=5, T=10, nw=0, k=0.01),
    dict(n=8, m=5, T=10, nw=5, k=0.01, var_order=1), dict(n=27, m=144, T=2, nw=1, k=0.01, ao=True),
    dict(n=27, m=144, T=10, nw=5, k=0.01, ao=True)])
def test_banded_equals_dense_T11(case):
    if case.get("ao"):
        md = S.make_model(case["n"], case["m"], case["T"]); data = S.make_replay_batch(md, 0, 1)
        data["w"] = np.zeros((1, case["T"] * case["n"]))
    else:
        md, data = S.make_test_problem(case["n"], case["m"], case["T"], seed=1, xf=case.get("xf", False),
                                       var_order=case.get("var_order", 2))
    if case.get("dense"):
        _dense_weights(md, 5)
    d = dense_from_model(md, data["x0"][0], data["x0_pre"][0], data["w"][0])
    i1 = {}
    nw = case["nw"]
    zd = d.mpc_fixed_log_newton(nw, case["k"], nu0=data["nu0"][0], info=i1) if nw else d.mpc_fixed_log(case["k"], nu0=data["nu0"][0], info=i1)
    b = banded_from_model(md)
    i2 = {}
    zb, nub, it, st = b.solve(data["x0"][0], data["x0_pre"][0], data["w"][0], nw, case["k"], nu0=data["nu0"][0], info=i2)
    assert st == 0 and it == i1["iters"] and i1.get("t", []) == i2.get("t", [])
    assert rel_err(zb, zd) <= 1e-10 and rel_err(nub, i1["nu"]) <= 1e-9


def test_line_search_closed_form_T4():
    """Closed form (App. A.5) == the literal loop of backtracking_inf_newton.m for t = 1, t < 1 and
    the collapse case (quirk D2: the reference ends at t = 0 by underflow)."""
    md, data = S.make_test_problem(8, 5, 10, seed=7, umax=0.3)
    rng = np.random.default_rng(11)
    z0 = np.zeros((10, 13)); z0[:, :5] = rng.uniform(-0.25, 0.25, (10, 5)); z0[:, 5:] = rng.uniform(-1, 1, (10, 8))
    d = dense_from_model(md, data["x0"][0], data["x0_pre"][0], data["w"][0], x_init=z0.reshape(-1))
    i1, i2 = {}, {}
    zd = d.mpc_fixed_log_newton(8, 10.0, nu0=data["nu0"][0], info=i1)
    zb, _, it, st = banded_from_model(md).solve(data["x0"][0], data["x0_pre"][0], data["w"][0], 8, 10.0,
                                                z_init=z0.reshape(-1), nu0=data["nu0"][0], info=i2)
    assert i1["t"] == i2["t"] and min(i1["t"]) < 1.0
    assert rel_err(zb, zd) <= 1e-9
    # collapse: the dense loop underflows to t = 0, the closed form stops after 64 halvings at t = 0
    md2 = S.make_model(27, 144, 10); md2["u_min"] = -0.05 * np.ones(144); md2["u_max"] = 0.05 * np.ones(144)
    data2 = S.make_replay_batch(md2, r=5, steps=2)
    w = np.zeros(10 * 27)
    d2 = dense_from_model(md2, data2["x0"][1], data2["x0_pre"][1], w)
    j1, j2 = {}, {}
    zd2 = d2.mpc_fixed_log_newton(6, 1e-2, nu0=data2["nu0"][1], info=j1)
    zb2, _, _, st2 = banded_from_model(md2).solve(data2["x0"][1], data2["x0_pre"][1], w, 6, 1e-2, nu0=data2["nu0"][1], info=j2)
    assert np.array_equal(canon_steps(j1["t"]), canon_steps(j2["t"]))
    assert 0.0 in canon_steps(j1["t"]) and st2 == 1 and max(j1["halvings"]) >= 40
    assert rel_err(zb2, zd2) <= 1e-9


def test_closed_loop_design_matrices_predict_the_var2_recursion():
    """oracle/closed_loop_ref.design_matrices (MPC_DesignMatrices, main.mlx): with no control the stacked
    prediction M1 x0 + M2 x0_pre must equal iterating x[i+1] = A1 x[i] + A2 x[i-1] (the recursion the solver's
    equality constraints encode, fast_mpc_eq_const.m:38-47)."""
    from oracle.closed_loop_ref import design_matrices
    rng = np.random.default_rng(7)
    n, T = 5, 6
    A1 = 0.5 * rng.standard_normal((n, n)); A2 = 0.3 * rng.standard_normal((n, n))
    M1, M2 = design_matrices(A1, A2, T)
    x0 = rng.standard_normal(n); xp = rng.standard_normal(n)
    pred = (M1 @ x0 + M2 @ xp).reshape(T, n)
    a, b = x0, xp
    for i in range(T):
        nxt = A1 @ a + A2 @ b
        assert np.allclose(pred[i], nxt, rtol=1e-13, atol=1e-13)
        a, b = nxt, a
