"""HIP path (through the C ABI) vs the structured oracle on the same seeded inputs.
Tolerance: 1e-9 relative on z (fp64; north-star "stated fp64 tolerance"), iteration counts,
status codes and line-search steps identical."""
import numpy as np
import pytest

from tests.util import canon_steps as canon,  handle_from_model, oracle_batch, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.fixture(autouse=True, params=["default", "generic"])
def kernel_choice(request, monkeypatch):
    """Every case of this file runs twice: on the kernel the library picks (n = 27: wave / panel / affine forms; other n: the tiled
    kernel where it exists, round 5) and with FMPC_FORCE_GENERIC=1 on the generic kernel (the LDS instance of fmpc_newton_generic)."""
    if request.param == "generic":
        monkeypatch.setenv("FMPC_FORCE_GENERIC", "1")
    return request.param


def _run(pkg, model, data, nw, k, z_init=None):
    h = handle_from_model(pkg, model)
    z, info = h.solve(data["x0"], data.get("x0_pre"), data.get("w"), z_init=z_init,
                      nu0=data.get("nu0"), n_newton=nw, k=k, return_info=True, check=False)
    h.close()
    return z, info


def _compare(pkg, model, data, nw, k, z_init=None, tol=TOL):
    z, info = _run(pkg, model, data, nw, k, z_init)
    zo, nuo, ito, sto, steps = oracle_batch(model, data, nw, k, z_init)
    assert np.array_equal(info["status"], sto), (info["status"], sto)
    assert np.array_equal(info["iters"], ito), (info["iters"], ito)
    for p in range(z.shape[0]):
        assert rel_err(z[p], zo[p]) <= tol, (p, rel_err(z[p], zo[p]))
        assert rel_err(info["nu"][p], nuo[p]) <= 1e-7
        got = info["step"][p][:ito[p]]
        assert np.array_equal(got, np.array(steps[p])), (p, got, steps[p])
        assert np.all(info["step"][p][ito[p]:] == -1.0)
    return z, info


@pytest.mark.parametrize("xf", [False, True])
@pytest.mark.parametrize("var_order", [2, 1])
@pytest.mark.parametrize("nw", [1, 5, 0])
def test_reference_demo_config(pkg, gpu, xf, var_order, nw):
    """test_fast_mpc.m:8-37 configuration (n=8, m=5, T=10, k=0.01)."""
    model, data = pkg.synthetic.make_test_problem(8, 5, 10, seed=3, xf=xf, var_order=var_order, batch=4)
    _compare(pkg, model, data, nw, 0.01)


@pytest.mark.parametrize("T", [1, 2, 3])
def test_short_horizons(pkg, gpu, T):
    model, data = pkg.synthetic.make_test_problem(4, 3, T, seed=5, xf=(T == 2), batch=3)
    _compare(pkg, model, data, 4, 0.1)


def test_active_barrier_and_backtracking(pkg, gpu):
    """Tight box, strong barrier, off-centre start: t < 1 steps."""
    model, data = pkg.synthetic.make_test_problem(8, 5, 10, seed=7, umax=0.3, batch=6)
    rng = np.random.default_rng(11)
    nz = 10 * 13
    z0 = np.zeros((6, nz)).reshape(6, 10, 13)
    z0[:, :, :5] = rng.uniform(-0.25, 0.25, (6, 10, 5))
    z0[:, :, 5:] = rng.uniform(-1, 1, (6, 10, 8))
    z0 = z0.reshape(6, nz)
    z, info = _compare(pkg, model, data, 8, 10.0, z_init=z0)
    assert np.any((info["step"] > 0) & (info["step"] < 1)), "case must exercise backtracking"


def test_linesearch_collapse_warning(pkg, gpu):
    """k=100, u_max=0.2: the search collapses; reference ends at t=0 (quirk D2)."""
    model, data = pkg.synthetic.make_test_problem(8, 5, 10, seed=9, umax=0.2, batch=2)
    z, info = _run(pkg, model, data, 3, 100.0)
    zo, nuo, ito, sto, steps = oracle_batch(model, data, 3, 100.0)
    assert np.array_equal(info["status"], sto)
    for p in range(2):
        assert rel_err(z[p], zo[p]) <= TOL


@pytest.mark.parametrize("T,nw", [(2, 1), (10, 5), (30, 1), (30, 5)])
def test_ao_config_batch(pkg, gpu, T, nw):
    """n=27 Zernike modes, m=144 actuators, README weights; replay batch of 64 problems."""
    model = pkg.synthetic.make_model(27, 144, T)
    data = pkg.synthetic.make_replay_batch(model, r=1, steps=64)
    _compare(pkg, model, data, nw, 1e-2)


def test_ao_config_tight_bounds(pkg, gpu):
    """Same model with a box the optimum presses against, so several Newton steps are real."""
    model = pkg.synthetic.make_model(27, 144, 10)
    model["u_min"] = -0.05 * np.ones(144); model["u_max"] = 0.05 * np.ones(144)
    data = pkg.synthetic.make_replay_batch(model, r=2, steps=16)
    z, info = _compare(pkg, model, data, 6, 1e-2)
    assert info["iters"].min() >= 3


def test_deterministic_and_position_independent(pkg, gpu):
    model = pkg.synthetic.make_model(27, 144, 10)
    data = pkg.synthetic.make_replay_batch(model, r=3, steps=48)
    z1, _ = _run(pkg, model, data, 3, 1e-2)
    z2, _ = _run(pkg, model, data, 3, 1e-2)
    assert np.array_equal(z1, z2), "two runs must agree bitwise"
    perm = np.random.default_rng(0).permutation(48)
    dperm = {k: (None if v is None else v[perm]) for k, v in data.items()}
    z3, _ = _run(pkg, model, dperm, 3, 1e-2)
    assert np.array_equal(z3, z1[perm]), "a problem's result must not depend on its batch slot"


def test_unpack(pkg, gpu):
    model, data = pkg.synthetic.make_test_problem(8, 5, 10, seed=1, batch=5)
    h = handle_from_model(pkg, model)
    z = h.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=2, k=0.01)
    U, X, u0 = h.unpack(z)
    Z = z.reshape(5, 10, 13)
    assert np.array_equal(U, Z[:, :, :5].reshape(5, -1))
    assert np.array_equal(X, Z[:, :, 5:].reshape(5, -1))
    assert np.array_equal(u0, Z[:, 0, :5])
    h.close()


@pytest.mark.parametrize("nw", [1, 4])
def test_per_problem_factor_dispatch_by_batch_size(pkg, gpu, nw, kernel_choice):
    """From an explicit start every problem factors its own Schur complement: up to 1024 problems the tiled kernel (2 or 4
    wavefronts per problem), beyond that the one-wavefront kernel (8 problems per CU); FMPC_NO_SMALL_TILED=1 keeps the
    one-wavefront kernel.  All three against the oracle, and against each other to 1e-11."""
    if kernel_choice == "generic":
        pytest.skip("the dispatch among the n = 27 kernels")
    import os
    model = pkg.synthetic.make_model(27, 144, 10)
    model["u_min"] = -0.1 * np.ones(144); model["u_max"] = 0.1 * np.ones(144)
    data = pkg.synthetic.make_replay_batch(model, r=5, steps=24)
    rng = np.random.default_rng(2)
    z0 = np.tile(np.concatenate([np.zeros(144), np.zeros(27)]), 10)[None, :] + np.zeros((24, 1))
    z0 = z0.reshape(24, 10, 171); z0[:, :, :144] = rng.uniform(-0.06, 0.06, (24, 10, 144)); z0 = z0.reshape(24, -1)
    h = handle_from_model(pkg, model)
    zt, it_ = h.solve(data["x0"], data["x0_pre"], None, z_init=z0, nu0=data["nu0"], n_newton=nw, k=1e-2, return_info=True)
    assert h.last_dispatch()[0] == pkg._lib.FMPC_PATH_TILED
    h.close()
    os.environ["FMPC_NO_SMALL_TILED"] = "1"
    try:
        h = handle_from_model(pkg, model)
    finally:
        del os.environ["FMPC_NO_SMALL_TILED"]
    zw, iw = h.solve(data["x0"], data["x0_pre"], None, z_init=z0, nu0=data["nu0"], n_newton=nw, k=1e-2, return_info=True)
    assert h.last_dispatch()[0] == pkg.FMPC_PATH_WAVE
    # 1100 problems (the 24 repeated): the one-wavefront kernel without any switch
    rep = lambda a: np.ascontiguousarray(np.tile(a, (46, 1))[:1100])
    zb, ib = h.solve(rep(data["x0"]), rep(data["x0_pre"]), None, z_init=rep(z0), nu0=rep(data["nu0"]), n_newton=nw, k=1e-2, return_info=True)
    h.close()
    h = handle_from_model(pkg, model)
    zc, ic = h.solve(rep(data["x0"]), rep(data["x0_pre"]), None, z_init=rep(z0), nu0=rep(data["nu0"]), n_newton=nw, k=1e-2, return_info=True)
    assert h.last_dispatch()[0] == pkg.FMPC_PATH_WAVE
    h.close()
    # (budget > 1: the default handle takes the first step with the one-wavefront kernel and the steps behind it with the tiled
    #  kernel over the compacted list of problems that go on -- round 4 -- so it agrees with the single launch to rounding only)
    assert np.array_equal(zb[:24], zw)
    if nw == 1:
        assert np.array_equal(zc, zb)
    else:
        assert max(rel_err(zc[p], zb[p]) for p in range(1100)) <= 1e-11
        assert np.array_equal(ic["iters"], ib["iters"]) and np.array_equal(ic["status"], ib["status"]) and np.array_equal(canon(ic["step"]), canon(ib["step"]))
    zo, nuo, ito, sto, steps = oracle_batch(model, data, nw, 1e-2, z0)
    for z_, i_ in ((zt, it_), (zw, iw)):
        assert np.array_equal(i_["status"], sto) and np.array_equal(i_["iters"], ito)
        for p in range(24):
            assert rel_err(z_[p], zo[p]) <= TOL and rel_err(i_["nu"][p], nuo[p]) <= 1e-7
            assert np.array_equal(i_["step"][p][:ito[p]], np.array(steps[p]))
    assert max(rel_err(zt[p], zw[p]) for p in range(24)) <= 1e-11



@pytest.mark.parametrize("want_nu", [True, False])
def test_explicit_start_budget5_large_batch_first_step_then_compacted_continuation(pkg, gpu, want_nu, kernel_choice):
    """Explicit start, more problems than the tiled kernel takes (> 1024), Newton budget 5 (test_fast_mpc.m:53,59): the
    one-wavefront kernel takes the first step of every problem and the next exit test (inf_newton_solver.m:19-22), the problems
    that go on are compacted into a list and finished by the tiled kernel.  Starts near the bounds for some problems: 2 to 5
    steps per problem, and first steps whose line search collapses.  Against the oracle problem by problem (z, nu, iteration
    counts, status, step lengths) and against the single launch (FMPC_NO_GENERAL_SPLIT=1)."""
    if kernel_choice == "generic":
        pytest.skip("the dispatch among the n = 27 kernels")
    import os
    import torch
    model = pkg.synthetic.make_model(27, 144, 8)
    nb_ = 40
    data = pkg.synthetic.make_replay_batch(model, r=9, steps=nb_)
    rng = np.random.default_rng(4)
    z0 = np.zeros((nb_, 8, 171)); z0[:, :, :144] = rng.uniform(-3, 3, (nb_, 8, 144)); z0[:, :, 144:] = 0.3 * rng.standard_normal((nb_, 8, 27))
    for p in range(0, nb_, 5):                                    # starts close to the bounds (u in [-28, 28]): more steps, and a first
        z0[p, :, :144:7] = 27.99 * np.sign(rng.standard_normal((8, 21)))      # step whose line search collapses (status 1 must survive
    for p in range(2, nb_, 9):                                    # the continuation)
        z0[p, :, :144:5] = 27.995
    for p in range(3, nb_, 9):
        z0[p, :, :144:5] = 27.9999
    z0 = z0.reshape(nb_, -1)
    B = 1100
    rep = lambda a: np.ascontiguousarray(np.tile(a, ((B + nb_ - 1) // nb_, 1))[:B])
    zo, nuo, ito, sto, steps = oracle_batch(model, data, 5, 1e-2, z0)
    assert len(set(ito.tolist())) >= 4 and (sto == 1).any() and (sto == 0).any()      # the case exercises what it claims
    outs = {}
    for tag, env in (("split", {}), ("one_launch", {"FMPC_NO_GENERAL_SPLIT": "1"})):
        for k_, v_ in env.items():
            os.environ[k_] = v_
        try:
            h = handle_from_model(pkg, model)
        finally:
            for k_ in env:
                del os.environ[k_]
        t = lambda a: torch.from_numpy(rep(a)).to(gpu)
        nu = torch.empty((B, h.nu_len), dtype=torch.float64, device=gpu) if want_nu else None
        stp = torch.full((B, 5), -7.0, dtype=torch.float64, device=gpu)
        u0 = torch.empty((B, 144), dtype=torch.float64, device=gpu)
        z, st, it = h.solve_device(t(data["x0"]), t(data["x0_pre"]), None, t(z0), t(data["nu0"]), 5, 1e-2, nu_out=nu, step=stp, u0_out=u0)
        torch.cuda.synchronize()
        assert h.last_dispatch()[0] == pkg.FMPC_PATH_WAVE
        outs[tag] = (z.cpu().numpy(), None if nu is None else nu.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), stp.cpu().numpy(), u0.cpu().numpy())
        h.close()
    for tag, (z, nu, st, it, stp, u0) in outs.items():
        for p in range(B):
            q = p % nb_
            assert it[p] == ito[q] and st[p] == sto[q], (tag, p, it[p], ito[q], st[p], sto[q])
            assert rel_err(z[p], zo[q]) <= TOL, (tag, p)
            assert nu is None or rel_err(nu[p], nuo[q]) <= 1e-7
            ts = np.full(5, -1.0); ts[:len(steps[q])] = steps[q]
            assert np.array_equal(canon(stp[p]), canon(ts)), (tag, p, stp[p], ts)
        assert np.array_equal(u0, z[:, :144])
    assert max(rel_err(outs["split"][0][p], outs["one_launch"][0][p]) for p in range(B)) <= 1e-11


def test_small_sizes_beyond_512_problems(pkg, gpu, kernel_choice):
    """n <= 31 (other than 27): up to 512 problems take four wavefronts each on the tiled kernel, more take two -- both sides of that
    switch in one handle (520 and 40 problems of the reference demo's configuration)."""
    model, data = pkg.synthetic.make_test_problem(8, 5, 10, seed=21, batch=520)
    h = handle_from_model(pkg, model)
    zo, nuo, ito, sto, steps = oracle_batch(model, data, 3, 0.01)
    for sl in (slice(0, 520), slice(100, 140)):
        z, info = h.solve(data["x0"][sl], data["x0_pre"][sl], data["w"][sl], nu0=data["nu0"][sl], n_newton=3, k=0.01, return_info=True, check=False)
        assert np.array_equal(info["status"], sto[sl]) and np.array_equal(info["iters"], ito[sl])
        assert max(rel_err(z[p], zo[sl][p]) for p in range(z.shape[0])) <= TOL
    h.close()
