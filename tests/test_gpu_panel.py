"""GPU tests of the panel path (fmpc_kernel_panel.hip + fmpc_kernel_dz.hip): the cold-start Newton step of the
AO configuration (n = 27) on panels of 16 problems, the host-built twisted factorisation and sweep schedules, the
step-length decision with its hand-over to the exact one-wave-per-problem path, and the continuation of a Newton
budget > 1 after the panel step.  Checker: the structured oracle (oracle/banded_ref.py) and the exact path
(FMPC_NO_PANEL=1 at create time).  Tolerance: 1e-9 relative on z, 1e-7 on nu, as everywhere (fp64)."""
import os

import numpy as np
import pytest

from tests.util import canon_steps, handle_from_model, oracle_batch, rel_err

pytestmark = pytest.mark.gpu
TOL_Z, TOL_NU = 1e-9, 1e-7


def _case(pkg, T, batch, xf, use_w, use_nu, seed, tight=False):
    md = pkg.synthetic.make_model(27, 144, T)
    if tight:
        md["u_min"] = -0.05 * np.ones(144); md["u_max"] = 0.05 * np.ones(144)
    rng = np.random.default_rng(seed)
    if xf:
        md["xf"] = 0.01 * rng.standard_normal(27)
    data = pkg.synthetic.make_replay_batch(md, r=seed, steps=batch)
    data["w"] = 0.01 * rng.standard_normal((batch, T * 27)) if use_w else None
    data["nu0"] = rng.standard_normal((batch, (T + (1 if xf else 0)) * 27)) if use_nu else None
    return md, data


def _exact_handle(pkg, md):
    os.environ["FMPC_NO_PANEL"] = "1"
    try:
        return handle_from_model(pkg, md)
    finally:
        del os.environ["FMPC_NO_PANEL"]


@pytest.mark.parametrize("T,batch,xf,use_w,use_nu", [
    (30, 37, False, False, True),       # ragged last panel, the bench configuration
    (30, 16, True, True, True),         # terminal equality row, disturbance, nu0: every input present
    (30, 5, False, True, False),        # fewer problems than a panel, nu0 = NULL
    (10, 33, True, False, False),       # short horizon with xf
    (3, 20, False, False, True),        # horizon too short for the twisted order (natural elimination order)
    (1, 4, True, True, True),           # one stage + terminal row
    (2, 17, False, False, True),        # the horizon the README itself runs (README.md:338)
    (8, 16, False, True, True),         # nb = 8: the shortest horizon with the twisted order
    (7, 9, True, False, True),          # nb = 8 through the terminal row
    (9, 16, False, False, False),       # odd number of block rows
])
def test_panel_step_matches_oracle_and_exact_path(pkg, gpu, T, batch, xf, use_w, use_nu):
    md, data = _case(pkg, T, batch, xf, use_w, use_nu, seed=11)
    hp = handle_from_model(pkg, md)
    hw = _exact_handle(pkg, md)
    zp, ip = hp.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=1, k=1e-2, return_info=True)
    path, handed = hp.last_dispatch()
    assert path == pkg.FMPC_PATH_PANEL and handed == 0          # the panel kernels produced every result
    zw, iw = hw.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=1, k=1e-2, return_info=True)
    assert hw.last_dispatch()[0] == pkg.FMPC_PATH_SHARED
    zo, nuo, ito, sto, steps = oracle_batch(md, data, 1, 1e-2)
    assert np.array_equal(ip["iters"], ito) and np.array_equal(ip["status"], sto)
    assert np.array_equal(ip["iters"], iw["iters"]) and np.array_equal(ip["step"], iw["step"])
    assert np.array_equal(canon_steps(ip["step"][:, 0]), canon_steps([s[0] for s in steps]))
    for p in range(batch):
        assert rel_err(zp[p], zo[p]) <= TOL_Z and rel_err(ip["nu"][p], nuo[p]) <= TOL_NU
        assert rel_err(zp[p], zw[p]) <= 1e-11 and rel_err(ip["nu"][p], iw["nu"][p]) <= 1e-11
    hp.close(); hw.close()


def test_panel_step_var1_and_odd_actuator_count(pkg, gpu):
    """VAR(1) (no lag-2 blocks: Y is block-tridiagonal, the schedule has fewer edges) and m not a multiple of 16
    (partial last column block of B')."""
    for var_order, m in [(1, 144), (2, 97)]:
        md = pkg.synthetic.make_model(27, m, 12, var_order=var_order)
        data = pkg.synthetic.make_replay_batch(md, r=4, steps=19)
        hp = handle_from_model(pkg, md)
        z, info = hp.solve(data["x0"], data["x0_pre"] if var_order == 2 else None, None, nu0=data["nu0"], n_newton=1,
                           k=1e-2, return_info=True)
        assert hp.last_dispatch() == (pkg.FMPC_PATH_PANEL, 0)
        zo, nuo, ito, sto, _ = oracle_batch(md, data, 1, 1e-2)
        assert np.array_equal(info["iters"], ito) and np.array_equal(info["status"], sto)
        assert max(rel_err(z[p], zo[p]) for p in range(19)) <= TOL_Z
        assert max(rel_err(info["nu"][p], nuo[p]) for p in range(19)) <= TOL_NU
        hp.close()


def test_panel_step_with_linear_costs_and_asymmetric_box(pkg, gpu):
    """Every host constant of the panel path that vanishes in the symmetric AO model: linear costs q, r, qf,
    a box that is not centred on zero (ubar, xbar != 0), a terminal state, per-entry different bounds."""
    rng = np.random.default_rng(21)
    md = pkg.synthetic.make_model(27, 144, 12)
    md["q"] = 50.0 * rng.standard_normal(27); md["r"] = 0.3 * rng.standard_normal(144); md["qf"] = 80.0 * rng.standard_normal(27)
    md["u_min"] = -28.0 + 6.0 * rng.random(144); md["u_max"] = 20.0 + 10.0 * rng.random(144)
    md["x_min"] = -90.0 + 20.0 * rng.random(27); md["x_max"] = 70.0 + 40.0 * rng.random(27)
    md["xf"] = 0.5 * rng.standard_normal(27)
    data = pkg.synthetic.make_replay_batch(md, r=6, steps=21)
    data["w"] = 0.05 * rng.standard_normal((21, 12 * 27))
    data["nu0"] = rng.standard_normal((21, 13 * 27))
    hp = handle_from_model(pkg, md)
    for k in (1e-2, 3.0):
        z, info = hp.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=1, k=k, return_info=True)
        assert hp.last_dispatch()[0] == pkg.FMPC_PATH_PANEL
        zo, nuo, ito, sto, steps = oracle_batch(md, data, 1, k)
        assert np.array_equal(info["iters"], ito) and np.array_equal(info["status"], sto)
        assert np.array_equal(canon_steps(info["step"][:, 0]), canon_steps([s_[0] for s_ in steps]))
        assert max(rel_err(z[p], zo[p]) for p in range(21)) <= TOL_Z
        assert max(rel_err(info["nu"][p], nuo[p]) for p in range(21)) <= TOL_NU
    hp.close()


def test_unclear_step_length_goes_to_the_exact_path(pkg, gpu):
    """Tight bounds: the barrier is active at the cold start, ||e||^2 is not small against rho^2, and the panel
    path must not decide the step length: every problem is redone by the exact path (bit-identical to it)."""
    md, data = _case(pkg, 30, 40, False, False, True, seed=5, tight=True)
    hp = handle_from_model(pkg, md)
    hw = _exact_handle(pkg, md)
    zp, ip = hp.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=1, k=1e-2, return_info=True)
    path, handed = hp.last_dispatch()
    assert path == pkg.FMPC_PATH_PANEL and handed == 40
    zw, iw = hw.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=1, k=1e-2, return_info=True)
    assert np.array_equal(zp, zw) and np.array_equal(ip["nu"], iw["nu"]) and np.array_equal(ip["step"], iw["step"])
    zo, _, ito, _, _ = oracle_batch(md, {k: (None if v is None else v[:8]) for k, v in data.items()}, 1, 1e-2)
    assert max(rel_err(zp[p], zo[p]) for p in range(8)) <= TOL_Z
    hp.close(); hw.close()


def test_bad_input_is_confined_to_its_problem(pkg, gpu):
    """A NaN state poisons one column of a panel only; that problem is handed to the exact path, which reports
    it (the shared factor is fine, so the failure shows as a collapsed line search), and its panel neighbours are
    untouched."""
    md, data = _case(pkg, 30, 24, False, False, True, seed=3)
    x0 = data["x0"].copy(); x0[5, 2] = np.nan
    hp = handle_from_model(pkg, md)
    z, info = hp.solve(x0, data["x0_pre"], None, nu0=data["nu0"], n_newton=1, k=1e-2, return_info=True, check=False)
    assert hp.last_dispatch() == (pkg.FMPC_PATH_PANEL, 1)
    assert info["status"][5] != 0 and (np.delete(info["status"], 5) == 0).all()
    zg, _ = hp.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=1, k=1e-2, return_info=True)
    good = np.delete(np.arange(24), 5)
    assert np.array_equal(z[good], zg[good])                      # bit-identical: columns of a panel do not mix
    hp.close()


def test_barrier_weights_on_one_handle_and_budget_continuation(pkg, gpu):
    """The factorisation, the images and the schedules are rebuilt per barrier weight k; a Newton budget > 1 runs
    the first step on the panels and the remaining iterations on the exact path (fixed-log-Newton schedule)."""
    md, data = _case(pkg, 30, 24, False, False, True, seed=9)
    md["u_min"] = -0.5 * np.ones(144); md["u_max"] = 0.5 * np.ones(144)      # some problems need > 1 step
    hp = handle_from_model(pkg, md)
    for k, nw in [(1e-2, 1), (1.0, 1), (1e-2, 4), (1e-1, 3)]:
        z, info = hp.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=nw, k=k, return_info=True)
        assert hp.last_dispatch()[0] == pkg.FMPC_PATH_PANEL
        zo, nuo, ito, sto, steps = oracle_batch(md, data, nw, k)
        assert np.array_equal(info["iters"], ito) and np.array_equal(info["status"], sto)
        for p in range(24):
            assert rel_err(z[p], zo[p]) <= TOL_Z and rel_err(info["nu"][p], nuo[p]) <= TOL_NU
            assert np.array_equal(canon_steps(info["step"][p, :len(steps[p])]), canon_steps(steps[p]))
    hp.close()


def test_panel_path_on_device_tensors_full_size(pkg, gpu):
    """Batch 2000 through the device-pointer entry point: deterministic, position independent (a problem's
    result does not depend on its panel or column), nullable outputs, and the dynamics residual of the result."""
    import torch
    md = pkg.synthetic.make_model(27, 144, 30)
    data = pkg.synthetic.make_replay_batch(md, r=2, steps=2000)
    h = handle_from_model(pkg, md)
    dev = torch.device("cuda:0")
    x0 = torch.from_numpy(data["x0"]).to(dev); x0p = torch.from_numpy(data["x0_pre"]).to(dev)
    nu0 = torch.from_numpy(data["nu0"]).to(dev)
    z1, st, it = h.solve_device(x0, x0p, None, None, nu0, 1, 1e-2)
    z2, _, _ = h.solve_device(x0, x0p, None, None, nu0, 1, 1e-2)
    torch.cuda.synchronize()
    assert h.last_dispatch() == (pkg.FMPC_PATH_PANEL, 0)
    assert torch.equal(z1, z2) and int(st.abs().sum()) == 0 and int((it - 1).abs().sum()) == 0
    perm = torch.from_numpy(np.random.default_rng(0).permutation(2000)).to(dev)
    z3, _, _ = h.solve_device(x0[perm].contiguous(), x0p[perm].contiguous(), None, None, nu0[perm].contiguous(), 1, 1e-2)
    torch.cuda.synchronize()
    assert torch.equal(z3, z1[perm])
    # the full step solves the linearised dynamics exactly: C (z + d_z) = b
    zz = z1.cpu().numpy()[:64]
    U, X, _ = h.unpack(zz)
    U = U.reshape(64, 30, 144); X = X.reshape(64, 30, 27)
    A1, A2, B = md["A1"], md["A2"], md["B"]
    r0 = X[:, 0] - U[:, 0] @ B.T - data["x0"][:64] @ A1.T - data["x0_pre"][:64] @ A2.T
    r1 = X[:, 1] - U[:, 1] @ B.T - X[:, 0] @ A1.T - data["x0"][:64] @ A2.T
    r2 = X[:, 2:] - U[:, 2:] @ B.T - X[:, 1:-1] @ A1.T - X[:, :-2] @ A2.T
    assert max(np.abs(r0).max(), np.abs(r1).max(), np.abs(r2).max()) <= 1e-11 * max(np.abs(X).max(), 1.0)
    zo, _, _, _, _ = oracle_batch(md, {k: (None if v is None else v[:6]) for k, v in data.items()}, 1, 1e-2)
    assert max(rel_err(zz[p], zo[p]) for p in range(6)) <= TOL_Z
    h.close()


@pytest.mark.parametrize("tight,bad", [(False, False), (True, False), (False, True)])
def test_budget_over_one_decide_compact_continue(pkg, gpu, tight, bad):
    """Newton budgets > 1 on the panel path run as decide + compacted continuation (two launches): iteration counts,
    step lengths and results against the oracle for problems that stop after one step, problems that go on, problems
    handed to the exact path (tight bounds), a NaN problem among healthy ones, and a ragged batch."""
    md, data = _case(pkg, 10, 53, False, False, True, seed=21)
    if tight:
        md = dict(md); md["u_min"] = -0.05 * np.ones(144); md["u_max"] = 0.05 * np.ones(144)
    x0 = data["x0"].copy()
    if bad:
        x0[17, 3] = np.nan
    hp = handle_from_model(pkg, md)
    for nw in (2, 5):
        z, info = hp.solve(x0, data["x0_pre"], None, nu0=data["nu0"], n_newton=nw, k=1e-2, return_info=True, check=False)
        path, handed = hp.last_dispatch()
        assert path == pkg.FMPC_PATH_PANEL and (handed == 53 if tight else handed == (1 if bad else 0))
        sub = {k: (None if v is None else v[:53]) for k, v in data.items()}
        sub["x0"] = x0
        zo, nuo, ito, sto, steps = oracle_batch(md, sub, nw, 1e-2)
        for p in range(53):
            if bad and p == 17:
                assert info["status"][p] != 0
                continue
            assert info["status"][p] == sto[p] and info["iters"][p] == ito[p], (p, info["iters"][p], ito[p])
            assert np.array_equal(canon_steps(info["step"][p][:ito[p]]), canon_steps(steps[p][:ito[p]]))
            assert rel_err(z[p], zo[p]) <= TOL_Z and rel_err(info["nu"][p], nuo[p]) <= TOL_NU
        if not tight and not bad:
            assert 1 <= info["iters"].min() and info["iters"].max() <= nw
    hp.close()


def test_budget5_bench_workload_against_the_oracle(pkg, gpu):
    """BASELINE configs[1] at the reference test's Newton budget of 5 (test_fast_mpc.m:53,59): every problem's iteration
    count, status, step lengths and z against the structured oracle (most stop after one step: the exit test of
    iteration 2 is decided from the d_z kernel's residual sums; the rest goes through the compacted continuation)."""
    md = pkg.synthetic.make_model(27, 144, 30)
    data = pkg.synthetic.make_replay_batch(md, r=0, steps=2000)          # the bench's batch
    h = handle_from_model(pkg, md)
    z, info = h.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=5, k=1e-2, return_info=True)
    assert h.last_dispatch() == (pkg.FMPC_PATH_PANEL, 0)
    # checked against the oracle: every problem that took more than one step, and every 5th of the others
    idx = np.array(sorted(set(np.nonzero(info["iters"] >= 2)[0]) | set(range(0, 2000, 5))))
    assert (info["iters"] >= 2).sum() > 20 and (info["iters"] == 1).sum() > 1000          # both kinds are present
    sub = {k: (None if v is None else v[idx]) for k, v in data.items()}
    zo, nuo, ito, sto, steps = oracle_batch(md, sub, 5, 1e-2)
    assert np.array_equal(info["iters"][idx], ito) and np.array_equal(info["status"][idx], sto)
    for q, p in enumerate(idx):
        assert np.array_equal(canon_steps(info["step"][p][:ito[q]]), canon_steps(steps[q][:ito[q]]))
        assert rel_err(z[p], zo[q]) <= TOL_Z and rel_err(info["nu"][p], nuo[q]) <= TOL_NU
    h.close()
