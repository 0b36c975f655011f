"""GPU parity of the ramp-rate path (fmpc_kernel_ramp.hip; SURVEY.md §8 row a6', (f) rank 3) against the dense oracle
with the VAR_1 ramp rows (oracle/dense_ref.py `ramp=True`, VAR_1/fast_mpc_ineq_const.m:58-76).  Tolerance: 1e-9
relative on z (fp64), 1e-7 on nu, identical iteration counts and step lengths."""
import ctypes as C

import numpy as np
import pytest

from oracle.dense_ref import DenseFastMPC
from tests.util import canon_steps, handle_from_model, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _oracle(md, x0, x0_pre, w, u_prev, du_min, du_max, nw, k, nu0, x_init=None):
    m = md["m"]
    info = {}
    if md.get("var_order", 2) == 1:
        d = DenseFastMPC.var1(md["Q"], md["R"], None, md["Qf"], md.get("q"), md.get("r"), md.get("qf"), md["x_min"], md["x_max"],
                              md["u_min"], md["u_max"], du_min, du_max, md["T"], x0, u_prev, md["A1"], md["B"], w,
                              md.get("xf"), x_init, ramp=True)
    else:
        d = DenseFastMPC(md["Q"], md["R"], None, md["Qf"], md.get("q"), md.get("r"), md.get("qf"), md["x_min"], md["x_max"],
                         md["u_min"], md["u_max"], du_min, du_max, md["T"], x0, x0_pre, u_prev, md["A1"], md["A2"], md["B"], w,
                         md.get("xf"), x_init, ramp=True)
    assert d.inequality_const()[0].shape[0] == 4 * md["T"] * m
    z = d.mpc_fixed_log_newton(nw, k, nu0=nu0, info=info)
    return z, info


def _ramp_inputs(md, batch, seed, width=0.4):
    rng = np.random.default_rng(seed)
    m = md["m"]
    du_min = -width * (0.5 + rng.random(m)); du_max = width * (0.5 + rng.random(m))
    # the mid-box start has u_0 = umid: keep u_prev close enough for a positive first-stage ramp slack
    umid = 0.5 * (md["u_min"] + md["u_max"])
    u_prev = umid + 0.5 * (du_min + (du_max - du_min) * rng.random((batch, m)))
    return du_min, du_max, u_prev


@pytest.mark.parametrize("n,m,T,var_order,xf,nw", [(8, 5, 6, 1, False, 1), (8, 5, 6, 1, False, 5), (8, 5, 10, 1, True, 3),
                                                  (5, 8, 4, 1, False, 4), (8, 5, 6, 2, False, 3), (6, 4, 1, 1, False, 2)])
def test_ramp_matches_dense_oracle(pkg, gpu, n, m, T, var_order, xf, nw):
    md, data = pkg.synthetic.make_test_problem(n, m, T, seed=11 + n + T, xf=xf, var_order=var_order, batch=5)
    du_min, du_max, u_prev = _ramp_inputs(md, 5, 3)
    h = handle_from_model(pkg, md)
    h.set_ramp(du_min, du_max)
    x0p = data["x0_pre"] if var_order == 2 else None
    z, info = h.solve(data["x0"], x0p, data["w"], nu0=data["nu0"], n_newton=nw, k=0.01, return_info=True, u_prev=u_prev)
    assert h.last_dispatch()[0] == pkg.FMPC_PATH_RAMP
    for p in range(5):
        zo, io = _oracle(md, data["x0"][p], None if x0p is None else x0p[p], data["w"][p], u_prev[p], du_min, du_max, nw, 0.01,
                         data["nu0"][p])
        # a collapsed line search (oracle: t ~ 2^-46, "no move"; device: t = 0 and FMPC_W_LINESEARCH) is canonicalised
        collapsed = bool((canon_steps(io["t"]) == 0).any())
        assert info["status"][p] == (pkg.FMPC_W_LINESEARCH if collapsed else 0) and info["iters"][p] == io["iters"]
        assert np.array_equal(canon_steps(info["step"][p][:io["iters"]]), canon_steps(io["t"][:io["iters"]]))
        assert rel_err(z[p], zo) <= TOL, (p, rel_err(z[p], zo))
        assert rel_err(info["nu"][p], io["nu"]) <= 1e-7
    # the ramp rows matter: without them the result differs
    z_box = h.solve(data["x0"], x0p, data["w"], nu0=data["nu0"], n_newton=nw, k=0.01)
    assert rel_err(z_box[0], z[0]) > 1e-6
    h.close()


def test_ramp_config0_size(pkg, gpu):
    """BASELINE configs[0]: VAR(1), n = 27, m = 144, T = 10, ramp rows on (README.md:355-356: du = +-0.2121)."""
    md = pkg.synthetic.make_model(27, 144, 10, var_order=1)
    data = pkg.synthetic.make_replay_batch(md, r=2, steps=3)
    du = 0.2121 * np.ones(144)
    rng = np.random.default_rng(8)
    u_prev = 0.1 * rng.standard_normal((3, 144))
    h = handle_from_model(pkg, md)
    h.set_ramp(-du, du)
    z, info = h.solve(data["x0"], None, None, nu0=data["nu0"][:, :270], n_newton=2, k=0.01, return_info=True, u_prev=u_prev)
    for p in range(2):
        zo, io = _oracle(md, data["x0"][p], None, np.zeros(270), u_prev[p], -du, du, 2, 0.01, data["nu0"][p, :270])    # replay: w = 0
        assert info["status"][p] == 0 and info["iters"][p] == io["iters"] == 2
        assert np.array_equal(info["step"][p][:2], np.asarray(io["t"][:2]))
        assert rel_err(z[p], zo) <= TOL, rel_err(z[p], zo)
    h.close()


def test_ramp_warm_start_batch_and_device_entry(pkg, gpu):
    """x_init given (fast_mpc_init.m:12-15), more problems than workgroups in flight, torch entry point == host entry."""
    import torch
    md, data = pkg.synthetic.make_test_problem(8, 5, 6, seed=5, var_order=1, batch=700)
    du_min, du_max, u_prev = _ramp_inputs(md, 700, 9)
    rng = np.random.default_rng(2)
    z_init = np.tile(np.concatenate([np.concatenate([0.2 * rng.standard_normal(5), rng.standard_normal(8)]) for _ in range(6)]), (700, 1))
    z_init[:, :5] = u_prev + 0.5 * (du_min + du_max)            # first-stage ramp slack positive
    for j in range(1, 6):
        z_init[:, 13 * j:13 * j + 5] = z_init[:, 13 * (j - 1):13 * (j - 1) + 5] + 0.25 * (du_min + du_max)
    h = handle_from_model(pkg, md)
    h.set_ramp(du_min, du_max)
    z, info = h.solve(data["x0"], None, data["w"], z_init=z_init, nu0=data["nu0"], n_newton=3, k=0.05, return_info=True, u_prev=u_prev)
    for p in (0, 333, 699):
        zo, io = _oracle(md, data["x0"][p], None, data["w"][p], u_prev[p], du_min, du_max, 3, 0.05, data["nu0"][p], x_init=z_init[p])
        assert info["iters"][p] == io["iters"] and rel_err(z[p], zo) <= TOL
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    zd, st, it = h.solve_device(t(data["x0"]), None, t(data["w"]), t(z_init), t(data["nu0"]), 3, 0.05, u_prev=t(u_prev))
    torch.cuda.synchronize()
    assert np.array_equal(zd.cpu().numpy(), z) and np.array_equal(st.cpu().numpy(), info["status"])
    assert (info["status"] >= 0).all()
    assert np.array_equal(it.cpu().numpy(), info["iters"])
    h.close()


def test_ramp_class_one_shot_and_errors(pkg, gpu):
    md, data = pkg.synthetic.make_test_problem(8, 5, 10, seed=39, var_order=1)
    du_min, du_max, u_prev = _ramp_inputs(md, 1, 4)
    args = (md["Q"], md["R"], [], md["Qf"], [], [], [], md["x_min"], md["x_max"], md["u_min"], md["u_max"], du_min, du_max, 10,
            data["x0"][0], u_prev[0], md["A1"], md["B"], data["w"][0], [], [])
    zo, io = _oracle(md, data["x0"][0], None, data["w"][0], u_prev[0], du_min, du_max, 5, 0.01, data["nu0"][0])
    v1 = pkg.Fast_MPC2_VAR1(*args)                                   # ramp rows on by default, as in the reference
    z = v1.mpc_fixed_log_newton(5, 0.01, nu0=data["nu0"][0])
    assert rel_err(z, zo) <= TOL
    # one-shot C entry with the reference's full argument set (var_order 1 -> ramp rows from du_min, du_max, u_prev)
    lib = pkg.load()
    cm = lambda M: np.ascontiguousarray(np.asarray(M, dtype=np.float64).T).reshape(-1)
    P = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.c_void_p)
    keep = [cm(md["Q"]), cm(md["R"]), cm(md["Qf"]), md["x_min"], md["x_max"], md["u_min"], md["u_max"], du_min, du_max,
            data["x0"][0], u_prev[0], cm(md["A1"]), cm(md["B"]), data["w"][0], data["nu0"][0]]
    x_opt = np.empty(10 * 13); iters = C.c_int(0)
    rc = lib.fmpc_solve_once(8, 5, 10, 1, P(keep[0]), P(keep[1]), None, P(keep[2]), None, None, None, P(keep[3]), P(keep[4]),
                             P(keep[5]), P(keep[6]), P(keep[7]), P(keep[8]), P(keep[9]), None, P(keep[10]), P(keep[11]), None,
                             P(keep[12]), P(keep[13]), None, None, P(keep[14]), 5, 0.01, 0, P(x_opt), C.byref(iters))
    assert rc == 0 and iters.value == io["iters"] and rel_err(x_opt, zo) <= TOL
    # u_prev without fmpc_set_ramp: refused
    h = handle_from_model(pkg, md)
    with pytest.raises(pkg.FastMPCError) as e:
        h.solve(data["x0"], None, data["w"], nu0=data["nu0"], n_newton=1, k=0.01, u_prev=u_prev)
    assert e.value.code == pkg.FMPC_E_UNSUPPORTED
    with pytest.raises(pkg.FastMPCError) as e:
        h.set_ramp(du_max, du_min)                                   # empty ramp interval
    assert e.value.code == pkg.FMPC_E_DIM
    h.close()


@pytest.mark.parametrize("n,m,T,var_order,xf,nw,batch,k,edge", [
    (27, 144, 10, 1, False, 1, 6, 0.01, 0.5),       # BASELINE configs[0], the reference loop's call (one step from the cold start)
    (27, 144, 10, 1, False, 3, 5, 0.01, 0.5),       # first step in the Woodbury form, two more by the dense factorisation
    (27, 144, 4, 2, True, 2, 4, 0.01, 0.5),         # VAR(2) dynamics + terminal rows
    (8, 5, 6, 1, False, 1, 700, 0.01, 0.5),         # more problems than workgroups in flight
    (8, 5, 6, 1, False, 4, 40, 1.0, 0.9),           # heavy barrier, u_prev next to the ramp bound: the line search backtracks (t = 1/2 .. 1/8)
    (8, 5, 6, 1, False, 3, 12, 10.0, 0.97),         # ... and collapses for some problems ("no move", FMPC_W_LINESEARCH)
    (5, 8, 1, 1, False, 2, 3, 0.01, 0.5),           # T = 1: no ramp row between stages at all
    (20, 33, 5, 1, False, 1, 9, 0.05, 0.8),         # m not a multiple of 16: masked edge tiles of the m x m factorisation
])
def test_ramp_cold_start_woodbury_form(pkg, gpu, n, m, T, var_order, xf, nw, batch, k, edge):
    """fmpc_ramp_cold (the cold-start step with the ramp rows as a constant KKT matrix + a diagonal term on u_0: one m x m
    factorisation per problem) against the general path of the same library (FMPC_NO_RAMP_COLD=1: dense (T n)^2 factorisation;
    1e-10 on z, identical iteration counts / status / step lengths) and against the dense oracle (1e-9)."""
    import os
    if n == 27:
        md = pkg.synthetic.make_model(n, m, T, var_order=var_order)
        data = pkg.synthetic.make_replay_batch(md, r=4, steps=batch)
        data["nu0"] = data["nu0"][:, :T * n]
        data["w"] = 0.01 * np.random.default_rng(1).standard_normal((batch, T * n))
        if xf:
            md["xf"] = 0.01 * np.random.default_rng(2).standard_normal(n)
            data["nu0"] = np.random.default_rng(3).random((batch, (T + 1) * n))
    else:
        md, data = pkg.synthetic.make_test_problem(n, m, T, seed=17 + n + T, xf=xf, var_order=var_order, batch=batch)
    rng = np.random.default_rng(5)
    du_min = -0.4 * (0.5 + rng.random(m)); du_max = 0.4 * (0.5 + rng.random(m))
    umid = 0.5 * (md["u_min"] + md["u_max"])
    frac = 0.5 + edge * (rng.random((batch, m)) - 0.5)                   # u_0 - u_prev at `frac` of the way from du_min to du_max
    u_prev = umid - (du_min + frac * (du_max - du_min))
    x0p = data["x0_pre"] if var_order == 2 else None
    h = handle_from_model(pkg, md)
    h.set_ramp(du_min, du_max)
    z, info = h.solve(data["x0"], x0p, data["w"], nu0=data["nu0"], n_newton=nw, k=k, return_info=True, u_prev=u_prev)
    assert h.last_dispatch()[0] == pkg.FMPC_PATH_RAMP and h.last_dual_form() == 5
    os.environ["FMPC_NO_RAMP_COLD"] = "1"
    try:
        hg = handle_from_model(pkg, md)
    finally:
        del os.environ["FMPC_NO_RAMP_COLD"]
    hg.set_ramp(du_min, du_max)
    zg, ig = hg.solve(data["x0"], x0p, data["w"], nu0=data["nu0"], n_newton=nw, k=k, return_info=True, u_prev=u_prev)
    assert hg.last_dual_form() == 0
    assert np.array_equal(info["iters"], ig["iters"]) and np.array_equal(info["status"], ig["status"])
    assert np.array_equal(canon_steps(info["step"]), canon_steps(ig["step"]))
    gerr = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-3))
    assert max(gerr(z[p], zg[p]) for p in range(batch)) <= 1e-10
    assert max(gerr(info["nu"][p], ig["nu"][p]) for p in range(batch)) <= 1e-8
    if k >= 1.0:
        assert (info["step"][:, 0] < 1.0).any(), "no backtracking: the case does not test the line search"
    aerr = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-3))     # (a collapsed search leaves z at the zero start point)
    for p in sorted({0, batch // 2, batch - 1}):
        zo, io = _oracle(md, data["x0"][p], None if x0p is None else x0p[p], data["w"][p], u_prev[p], du_min, du_max, nw, k, data["nu0"][p])
        assert info["iters"][p] == io["iters"] and np.array_equal(canon_steps(info["step"][p][:io["iters"]]), canon_steps(io["t"][:io["iters"]]))
        assert aerr(z[p], zo) <= TOL and aerr(info["nu"][p], io["nu"]) <= 1e-7
    # an explicit start point takes the general path; a NaN in u_prev ends in the same status on both paths, the others unharmed
    zi = np.tile(np.concatenate([umid, 0.5 * (md["x_min"] + md["x_max"])]), (batch, T))
    zw = h.solve(data["x0"], x0p, data["w"], z_init=zi, nu0=data["nu0"], n_newton=1, k=k, u_prev=u_prev)
    assert h.last_dual_form() == 0 and max(gerr(zw[p], (z if nw == 1 else zw)[p]) for p in range(batch)) <= 1e-10
    up_bad = u_prev.copy(); up_bad[0, 1] = np.nan
    zb, ib = h.solve(data["x0"], x0p, data["w"], nu0=data["nu0"], n_newton=nw, k=k, return_info=True, u_prev=up_bad, check=False)
    zc, ic = hg.solve(data["x0"], x0p, data["w"], nu0=data["nu0"], n_newton=nw, k=k, return_info=True, u_prev=up_bad, check=False)
    assert ib["status"][0] == ic["status"][0] < 0 and ib["iters"][0] == ic["iters"][0] == 0
    if batch > 1:
        assert np.array_equal(ib["status"][1:], info["status"][1:]) and all(np.array_equal(zb[p], z[p]) for p in range(1, batch))
    h.close(); hg.close()
