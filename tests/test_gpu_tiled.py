"""Tiled kernel (fmpc_kernel_tiled.hip: workgroup per problem, 16 x 16 tiles on the matrix cores) through the C ABI
vs the structured oracle on the same seeded inputs.
  fp64 factor:  1e-9 relative on z (the parity bar of every fp64 path), iteration counts, status, line-search steps equal.
  fp32 factor + fp64 residuals (BASELINE configs[4], "fp32 mixed precision"): 1e-4 relative on z."""
import os

import numpy as np
import pytest

from tests.util import canon_steps, handle_from_model, oracle_batch, rel_err

pytestmark = pytest.mark.gpu
TOL64 = 1e-9
TOL32 = 1e-4


@pytest.fixture()
def tiled_env():
    old = os.environ.get("FMPC_TILED")
    os.environ["FMPC_TILED"] = "1"          # read by fmpc_create: every solve of the handle takes the tiled kernel
    yield
    if old is None:
        os.environ.pop("FMPC_TILED", None)
    else:
        os.environ["FMPC_TILED"] = old


def _solve(pkg, model, data, nw, k, z_init=None, prec=None):
    h = handle_from_model(pkg, model)
    if prec is not None:
        h.set_precision(prec)
    z, info = h.solve(data["x0"], data.get("x0_pre"), data.get("w"), z_init=z_init, nu0=data.get("nu0"),
                      n_newton=nw, k=k, return_info=True, check=False)
    path, _ = h.last_dispatch()
    h.close()
    return z, info, path


def _compare64(pkg, model, data, nw, k, z_init=None):
    z, info, path = _solve(pkg, model, data, nw, k, z_init)
    assert path == pkg._lib.FMPC_PATH_TILED
    zo, nuo, ito, sto, steps = oracle_batch(model, data, nw, k, z_init=z_init)
    B = data["x0"].shape[0]
    assert np.array_equal(info["status"], sto), (info["status"], sto)
    assert np.array_equal(info["iters"], ito), (info["iters"], ito)
    for p in range(B):
        assert rel_err(z[p], zo[p]) <= TOL64, (p, rel_err(z[p], zo[p]))
        assert rel_err(info["nu"][p], nuo[p]) <= 1e-7
        t = canon_steps(info["step"][p][:ito[p]])
        assert np.allclose(t, canon_steps(steps[p]), rtol=0, atol=0), (p, t, steps[p])
    return z, info


@pytest.mark.parametrize("n,m,T,xf,var", [(8, 5, 10, False, 2), (8, 5, 10, True, 2), (8, 5, 10, False, 1),
                                          (16, 7, 6, False, 2), (16, 7, 6, True, 2), (20, 33, 5, False, 2),
                                          (40, 21, 7, False, 2), (5, 3, 1, False, 2), (5, 3, 2, True, 2)])
def test_tiled_fp64_demo_sizes(pkg, gpu, tiled_env, n, m, T, xf, var):
    """NB = 1, 2, 3 tiles per block incl. n a multiple of 16 (the rhs column then has a tile column of its own)."""
    model, data = pkg.synthetic.make_test_problem(n, m, T, seed=n + T, xf=xf, var_order=var, batch=3)
    _compare64(pkg, model, data, 5, 0.01)


@pytest.mark.parametrize("waves", [2, 4])
@pytest.mark.parametrize("T,nw", [(2, 1), (10, 5), (30, 1), (30, 5)])
def test_tiled_fp64_ao_config(pkg, gpu, tiled_env, T, nw, waves):
    """n = 27 (compile-time block structure 16 + 11) with 2 and with 4 wavefronts per problem: different instances of the
    kernel (FMPC_TILED_NW is read when the handle is created)."""
    old = os.environ.get("FMPC_TILED_NW")
    os.environ["FMPC_TILED_NW"] = str(waves)
    try:
        model = pkg.synthetic.make_model(27, 144, T)
        data = pkg.synthetic.make_replay_batch(model, r=1, steps=40)
        _compare64(pkg, model, data, nw, 1e-2)
    finally:
        if old is None:
            os.environ.pop("FMPC_TILED_NW", None)
        else:
            os.environ["FMPC_TILED_NW"] = old


def test_tiled_fp64_tight_bounds_and_warm_start(pkg, gpu, tiled_env):
    model = pkg.synthetic.make_model(27, 144, 10)
    model["u_min"] = -0.05 * np.ones(144); model["u_max"] = 0.05 * np.ones(144)
    data = pkg.synthetic.make_replay_batch(model, r=2, steps=16)
    z, info = _compare64(pkg, model, data, 6, 1e-2)
    assert info["iters"].min() >= 3
    rng = np.random.default_rng(7)
    z0 = np.tile(np.concatenate([0.04 * rng.uniform(-1, 1, 144), rng.standard_normal(27)]), (16, 10))
    _compare64(pkg, model, data, 4, 1e-2, z_init=z0)


def test_tiled_fp64_backtracking_and_collapse(pkg, gpu, tiled_env):
    model, data = pkg.synthetic.make_test_problem(8, 5, 10, seed=3, umax=0.3, batch=4)
    rng = np.random.default_rng(11)
    z0 = np.tile(np.concatenate([0.25 * rng.uniform(-1, 1, 5), rng.standard_normal(8)]), (4, 10))
    _compare64(pkg, model, data, 8, 10.0, z_init=z0)
    model, data = pkg.synthetic.make_test_problem(8, 5, 10, seed=9, umax=0.2, batch=2)
    z, info, _ = _solve(pkg, model, data, 3, 100.0)
    zo, nuo, ito, sto, steps = oracle_batch(model, data, 3, 100.0)
    assert np.array_equal(info["status"], sto)
    for p in range(2):
        assert rel_err(z[p], zo[p]) <= TOL64


def test_tiled_fp64_batch_beyond_grid_is_position_independent(pkg, gpu, tiled_env):
    """More problems than resident workgroups (grid-stride loop), bitwise reproducible and slot independent."""
    model = pkg.synthetic.make_model(27, 144, 4)
    data = pkg.synthetic.make_replay_batch(model, r=3, steps=700)
    z1, _, _ = _solve(pkg, model, data, 2, 1e-2)
    z2, _, _ = _solve(pkg, model, data, 2, 1e-2)
    assert np.array_equal(z1, z2)
    perm = np.random.default_rng(0).permutation(700)
    dperm = {k: (None if v is None else v[perm]) for k, v in data.items()}
    z3, _, _ = _solve(pkg, model, dperm, 2, 1e-2)
    assert np.array_equal(z3, z1[perm])
    zo, *_ = oracle_batch(model, {k: (None if v is None else v[:5]) for k, v in data.items()}, 2, 1e-2)
    for p in range(5):
        assert rel_err(z1[p], zo[p]) <= TOL64


# ------------------------------------------------------------------ fp32 factor + fp64 residuals
def _compare32(pkg, model, data, nw, k, z_init=None, expect_path=True):
    z, info, path = _solve(pkg, model, data, nw, k, z_init, prec="f32")
    assert path == pkg._lib.FMPC_PATH_TILED_F32
    zo, nuo, ito, sto, steps = oracle_batch(model, data, nw, k, z_init=z_init)
    B = data["x0"].shape[0]
    assert np.array_equal(info["status"], sto)
    assert np.all(info["iters"] >= ito), (info["iters"], ito)
    errs = [rel_err(z[p], zo[p]) for p in range(B)]
    assert max(errs) <= TOL32, errs
    return max(errs)


@pytest.mark.parametrize("nw", [1, 3])
def test_tiled_fp32_ao_config(pkg, gpu, nw):
    model = pkg.synthetic.make_model(27, 144, 30)
    data = pkg.synthetic.make_replay_batch(model, r=1, steps=24)
    err = _compare32(pkg, model, data, nw, 1e-2)
    print(f"fp32 factor, n=27 T=30 nw={nw}: max rel err on z {err:.2e}")


@pytest.mark.parametrize("waves", [8, 4])
@pytest.mark.parametrize("nw", [1, 3])
def test_tiled_fp32_config4_n65_T60(pkg, gpu, nw, waves, monkeypatch):
    """BASELINE configs[4]: VAR(2), n = 65 (radial order 10), T = 60, fp32 mixed precision."""
    monkeypatch.setenv("FMPC_TILED_NW", str(waves))
    model = pkg.synthetic.make_model(65, 144, 60)
    data = pkg.synthetic.make_replay_batch(model, r=4, steps=6)
    h = handle_from_model(pkg, model)
    h.set_precision("f32")                                # (the default at n = 65 is fp64 since round 5: the fp32 factor is a request)
    z, info = h.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=nw, k=1e-2, return_info=True)
    path, _ = h.last_dispatch()
    h.close()
    assert path == pkg._lib.FMPC_PATH_TILED_F32
    zo, nuo, ito, sto, _ = oracle_batch(model, data, nw, 1e-2)
    assert np.array_equal(info["status"], sto) and np.all(info["iters"] >= ito), (info["iters"], ito)
    errs = [rel_err(z[p], zo[p]) for p in range(6)]
    print(f"fp32 factor, n=65 T=60 nw={nw}: max rel err on z {max(errs):.2e}")
    assert max(errs) <= TOL32, errs


def test_tiled_fp32_tight_bounds_n65(pkg, gpu):
    """Active barrier, several real Newton steps: the fp64 residuals keep the fp32 solves on the oracle's trajectory."""
    model = pkg.synthetic.make_model(65, 144, 12)
    model["u_min"] = -0.05 * np.ones(144); model["u_max"] = 0.05 * np.ones(144)
    data = pkg.synthetic.make_replay_batch(model, r=5, steps=4)
    _compare32(pkg, model, data, 5, 1e-2)


@pytest.mark.parametrize("n,m,T,var,xf", [(33, 20, 6, 2, False), (40, 150, 5, 2, True), (47, 60, 4, 1, False),      # 3 blocks of 16 (NB = 3)
                                          (48, 30, 5, 2, False), (63, 64, 4, 2, True),                                 # NB = 4, last block 0 / 15 rows
                                          (64, 33, 4, 2, False), (66, 70, 4, 1, True), (79, 150, 3, 2, False)])       # NB = 5: 0, 2, 15 live rows
def test_tiled_fp32_random_models_over_the_block_sizes(pkg, gpu, n, m, T, var, xf):
    """The fp32 instances besides the AO sizes (block structure at run time: n = 64 has NO live row in its fifth block, 79 fills
    it), on random stable models (tests/test_property_random.py: the generator of test_fast_mpc.m:8-37), two Newton steps."""
    from tests.test_property_random import random_problem
    model, data = random_problem(7000 + n, n, m, T, var, False, False, xf and m >= n, False, batch=4)
    err = _compare32(pkg, model, data, 2, 1e-1)
    print(f"fp32 factor, random model n={n} m={m} T={T}: max rel err on z {err:.2e}")


@pytest.mark.parametrize("n,m,T,var,xf", [(50, 30, 4, 2, False), (65, 70, 3, 1, True), (79, 40, 3, 2, False), (27, 144, 6, 2, False)])
def test_tiled_fp32_dense_state_weights(pkg, gpu, n, m, T, var, xf):
    """Dense Q, Qf with the fp32 factor (the default arithmetic for 47 < n <= 79; round 5: no test had this combination): three
    Newton steps against the structured oracle at the fp32 tolerance, and against the fp64 answer of the same library (the generic
    kernel's workspace instance)."""
    from tests.test_property_random import random_problem
    model, data = random_problem(8000 + n, n, m, T, var, True, False, xf and m >= n, True, batch=4)
    err = _compare32(pkg, model, data, 3, 1e-1)
    h = handle_from_model(pkg, model)
    h.set_precision("f32")
    z32 = h.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=3, k=1e-1, check=False)
    if n > 47:
        h.set_precision("f64")
        z64 = h.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=3, k=1e-1, check=False)
        assert h.last_dispatch()[0] == pkg._lib.FMPC_PATH_TILED          # (the fp64 instance of 4 / 5 blocks, round 5)
        assert rel_err(z32, z64) <= TOL32
    h.close()
    print(f"fp32 factor, dense Q, n={n}: max rel err on z {err:.2e}")


def test_precision_switch_errors(pkg, gpu):
    """The fp32 factor exists for diagonal R only: refused cleanly with a dense R.  (fp64: every size and every weight is solved --
    beyond the matrix-core kernels by the generic kernel's workspace instance, tests/test_gpu_any_size.py.)"""
    model = pkg.synthetic.make_model(65, 144, 4)
    Rd = np.array(model["R"], dtype=float)
    Rd[0, 1] = Rd[1, 0] = 1e-3 * Rd[0, 0]
    h = handle_from_model(pkg, dict(model, R=Rd))
    with pytest.raises(pkg.FastMPCError) as e:
        h.set_precision("f32")
    assert e.value.code == pkg._lib.FMPC_E_UNSUPPORTED
    h.set_precision("f64")
    h.close()
    h = handle_from_model(pkg, model)
    h.set_precision("f64"); h.set_precision("f32"); h.set_precision("f64")     # diagonal weights: both, back and forth
    with pytest.raises(pkg.FastMPCError) as e:
        h.set_precision(7)
    assert e.value.code == pkg._lib.FMPC_E_DIM
    h.close()
    handle_from_model(pkg, pkg.synthetic.make_model(80, 16, 4)).close()


def test_tiled_repeated_solves_alternating_wave_counts(pkg, gpu, tiled_env, monkeypatch):
    """Fresh handles (fresh, dirty workspaces) again and again, alternating the kernel instance: every solve must agree with
    the oracle -- catches state carried between launches and sporadic races that one solve per test would miss."""
    for T in (2, 10):
        model = pkg.synthetic.make_model(27, 144, T)
        data = pkg.synthetic.make_replay_batch(model, r=1, steps=40)
        zo, _, ito, sto, _ = oracle_batch(model, data, 2, 1e-2)
        for rep in range(4):
            for waves in ("2", "4"):
                monkeypatch.setenv("FMPC_TILED_NW", waves)
                z, info, path = _solve(pkg, model, data, 2, 1e-2)
                assert path == pkg._lib.FMPC_PATH_TILED
                assert np.array_equal(info["status"], sto) and np.array_equal(info["iters"], ito), (T, rep, waves)
                assert max(rel_err(z[p], zo[p]) for p in range(40)) <= TOL64, (T, rep, waves)


# ------------------------------------------------------------------ dense state weights (fast_mpc_objective.m:52-55)
def _spd(n, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n))
    return scale * (G @ G.T / n + np.eye(n))


@pytest.mark.parametrize("n,m,T,xf,nw,umax", [(8, 5, 10, False, 5, 2.0), (8, 5, 10, True, 5, 2.0), (8, 5, 10, False, 8, 0.3),
                                              (27, 144, 10, False, 3, None), (20, 33, 5, False, 4, 2.0),
                                              (40, 30, 4, False, 3, 2.0), (47, 50, 3, True, 3, 2.0)])          # three blocks of 16 (round 5)
def test_dense_spd_state_weights(pkg, gpu, n, m, T, xf, nw, umax):
    """Random symmetric positive definite Q and Qf (R diagonal): Phi and Phi^-1 on the states are dense blocks, the
    constant Y blocks carry (2Q)^-1; handled by the tiled kernel whatever n.  Oracle: the structured restatement with the
    same dense weights (pinned to the dense one in tests/test_oracle_banded.py)."""
    if umax is None:
        model = pkg.synthetic.make_model(n, m, T)
        data = pkg.synthetic.make_replay_batch(model, r=6, steps=5)
        model["Q"] = _spd(n, 1, 1.5e4); model["Qf"] = _spd(n, 2, 1.5e4)
    else:
        model, data = pkg.synthetic.make_test_problem(n, m, T, seed=n + T, umax=umax, xf=xf, batch=5)
        model["Q"] = _spd(n, 1); model["Qf"] = _spd(n, 2, 50.0)
    z, info, path = _solve(pkg, model, data, nw, 0.01)
    assert path == pkg._lib.FMPC_PATH_TILED
    zo, nuo, ito, sto, steps = oracle_batch(model, data, nw, 0.01)
    assert np.array_equal(info["status"], sto) and np.array_equal(info["iters"], ito)
    for p in range(5):
        assert rel_err(z[p], zo[p]) <= TOL64, (p, rel_err(z[p], zo[p]))
        assert np.array_equal(canon_steps(info["step"][p][:ito[p]]), canon_steps(steps[p]))


@pytest.mark.parametrize("n,m,T,xf,nw,umax,denseq", [(8, 5, 10, False, 5, 2.0, False), (8, 5, 10, True, 5, 2.0, True),
                                                     (8, 5, 10, False, 8, 0.3, False), (27, 144, 10, False, 3, None, False),
                                                     (27, 144, 4, False, 2, None, True), (20, 33, 5, False, 4, 2.0, False),
                                                     (40, 30, 4, False, 3, 2.0, False), (47, 20, 3, False, 3, 2.0, True)])   # n + 1 > 32 columns (round 5 fix)
def test_dense_spd_input_weight(pkg, gpu, n, m, T, xf, nw, umax, denseq):
    """Random symmetric positive definite R (fast_mpc_objective.m:51-54 takes any square R): the u block of Phi,
    2R + k diag(1/s+^2 + 1/s-^2), is a dense m x m matrix per stage and Newton step (inf_newton_KKT_H.m:13), factored in LDS
    by the tiled kernel; alone and together with dense Q, Qf.  Oracle: the structured restatement with the same weights."""
    if umax is None:
        model = pkg.synthetic.make_model(n, m, T)
        data = pkg.synthetic.make_replay_batch(model, r=6, steps=5)
        model["R"] = _spd(m, 7)
        if denseq:
            model["Q"] = _spd(n, 1, 1.5e4); model["Qf"] = _spd(n, 2, 1.5e4)
    else:
        model, data = pkg.synthetic.make_test_problem(n, m, T, seed=n + T, umax=umax, xf=xf, batch=5)
        model["R"] = _spd(m, 7)
        if denseq:
            model["Q"] = _spd(n, 1); model["Qf"] = _spd(n, 2, 50.0)
    z, info, path = _solve(pkg, model, data, nw, 0.01)
    assert path == pkg._lib.FMPC_PATH_TILED
    zo, nuo, ito, sto, steps = oracle_batch(model, data, nw, 0.01)
    assert np.array_equal(info["status"], sto) and np.array_equal(info["iters"], ito)
    for p in range(5):
        assert rel_err(z[p], zo[p]) <= TOL64, (p, rel_err(z[p], zo[p]))
        assert rel_err(info["nu"][p], nuo[p]) <= 1e-7
        assert np.array_equal(canon_steps(info["step"][p][:ito[p]]), canon_steps(steps[p]))


def test_dense_weights_error_codes(pkg, gpu):
    model, data = pkg.synthetic.make_test_problem(8, 5, 4, seed=1, batch=1)
    bad = dict(model); bad["R"] = _spd(5, 3); bad["R"][0, 1] += 1e-3          # not symmetric
    with pytest.raises(pkg.FastMPCError) as e:
        handle_from_model(pkg, bad)
    assert e.value.code == pkg._lib.FMPC_E_NOT_PD_PHI
    ind = _spd(5, 5); ind[0, 0] = 1e-3; ind[0, 1] = ind[1, 0] = 5.0          # symmetric, positive diagonal, indefinite
    bad = dict(model); bad["R"] = ind
    with pytest.raises(pkg.FastMPCError) as e:
        handle_from_model(pkg, bad)
    assert e.value.code == pkg._lib.FMPC_E_NOT_PD_PHI
    ok = dict(model); ok["R"] = _spd(5, 3)
    h = handle_from_model(pkg, ok)
    with pytest.raises(pkg.FastMPCError) as e:
        h.set_precision("f32")                           # dense R is solved in fp64 only
    assert e.value.code == pkg._lib.FMPC_E_UNSUPPORTED
    h.close()
    bad = dict(model); bad["Q"] = -_spd(8, 4)
    with pytest.raises(pkg.FastMPCError) as e:
        handle_from_model(pkg, bad)
    assert e.value.code == pkg._lib.FMPC_E_NOT_PD_PHI
    ind = _spd(8, 5); ind[0, 0] = 1e-3; ind[0, 1] = ind[1, 0] = 5.0      # symmetric, positive diagonal, indefinite
    bad = dict(model); bad["Qf"] = ind
    with pytest.raises(pkg.FastMPCError) as e:
        handle_from_model(pkg, bad)
    assert e.value.code == pkg._lib.FMPC_E_NOT_PD_PHI


def test_many_seed_stress_of_the_production_instances(pkg, gpu, monkeypatch):
    """ADVICE (round 2): one instance of the tiled kernel (<double,2,4,11>) came out wrong in several builds and is not
    instantiated; the instances that ARE in production for n = 27 -- <double,2,4> (run-time block structure: explicit-start
    batches <= 512 and budget continuations), <double,2,2,11> (batches <= 1024) and the one-wavefront kernel -- are compared
    here on many seeds, explicit off-centre starts with active barrier terms, 2 Newton steps: any pair must agree to 1e-12
    relative on z and nu with identical iteration counts and step lengths.  A race or undefined behaviour that depends on
    timing shows up as a seed-dependent outlier."""
    import torch
    dev = torch.device("cuda:0")
    n, m, T, B = 27, 144, 30, 96
    md = pkg.synthetic.make_model(n, m, T)
    md["u_min"] = -0.6 * np.ones(m); md["u_max"] = 0.6 * np.ones(m)

    def make(env):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        h_ = handle_from_model(pkg, md)
        for k_ in env:
            monkeypatch.delenv(k_, raising=False)
        return h_
    # (a handle created with FMPC_SMALL_TILED_NW=4 takes the tiled kernel with FOUR wavefronts per problem for explicit-start batches <= 512; FMPC_TILED=1
    #  forces the tiled kernel with its default of two.  FMPC_TILED_NW is read at the first tiled solve, not at create time.)
    # (Round 5: the compile-time instance <double,2,4,11> and its switch are gone; four wavefronts per problem = the run-time form.)
    # Round 5: the default for small batches is FOUR wavefronts per problem again ("default": what production takes, the run-time form);
    # two are FMPC_SMALL_TILED_NW=2 at create time / fmpc_set_small_batch_kernel(h, 2).  Both stay under this stress test.
    hs = {"default": make({}), "tiled_nw4": make({"FMPC_SMALL_TILED_NW": "2"}),
          "tiled_nw2": make({"FMPC_TILED": "1"}), "wave": make({"FMPC_NO_SMALL_TILED": "1"})}
    want_nw = {"default": 4, "tiled_nw4": 2, "tiled_nw2": 2}
    worst = 0.0
    for seed in range(24):
        rng = np.random.default_rng(1000 + seed)
        data = pkg.synthetic.make_replay_batch(md, r=50 + seed, steps=B)
        z0 = np.zeros((B, T, n + m))
        z0[:, :, :m] = rng.uniform(-0.5, 0.5, (B, T, m))
        z0[:, :, m:] = rng.standard_normal((B, T, n))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        x0, x0p, nu0, zi = t(data["x0"]), t(data["x0_pre"]), t(data["nu0"]), t(z0.reshape(B, -1))
        out = {}
        for name, h in hs.items():
            nu = torch.empty((B, T * n), dtype=torch.float64, device=dev)
            stp = torch.empty((B, 2), dtype=torch.float64, device=dev)
            z, st, it = h.solve_device(x0, x0p, None, zi, nu0, 2, 1e-2, nu_out=nu, step=stp)
            torch.cuda.synchronize()
            out[name] = (z.cpu().numpy(), nu.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), stp.cpu().numpy())
        want = {"default": pkg._lib.FMPC_PATH_TILED, "tiled_nw4": pkg._lib.FMPC_PATH_TILED,
                "tiled_nw2": pkg._lib.FMPC_PATH_TILED, "wave": pkg.FMPC_PATH_WAVE}
        for name, h in hs.items():
            assert h.last_dispatch()[0] == want[name]
            assert name == "wave" or h.last_tiled_wavefronts() == want_nw[name], (name, h.last_tiled_wavefronts())
        ref = out["wave"]
        assert int((ref[2] < 0).sum()) == 0
        for name in ("default", "tiled_nw4", "tiled_nw2"):
            o = out[name]
            assert np.array_equal(o[2], ref[2]) and np.array_equal(o[3], ref[3]), (seed, name)
            assert np.array_equal(canon_steps(o[4]), canon_steps(ref[4])), (seed, name)
            ez = max(rel_err(o[0][p], ref[0][p]) for p in range(B)); en = max(rel_err(o[1][p], ref[1][p]) for p in range(B))
            worst = max(worst, ez, en)
            assert ez <= 1e-12 and en <= 1e-12, (seed, name, ez, en)
    for h in hs.values():
        h.close()
    print("worst relative difference between the three kernels over 24 seeds x 96 problems: %.2e" % worst)


@pytest.mark.parametrize("n,m,T,var,xf,dq", [(83, 40, 4, 2, False, False), (96, 144, 6, 2, False, False), (111, 30, 3, 1, False, True), (100, 120, 3, 2, True, False)])
def test_tiled_fp32_instances_of_six_and_seven_blocks(pkg, gpu, n, m, T, var, xf, dq):
    """79 < n <= 111: the fp32 factor on request (fmpc_set_precision; the default there is the exact fp64 fallback,
    tests/test_gpu_any_size.py) -- fmpc_newton_tiled<float, 6 | 7, 8> -- against the oracle at the fp32 tolerance after three Newton
    steps, and against the fp64 answer of the same library."""
    from tests.test_property_random import random_problem
    model, data = random_problem(400 + n, n, m, T, var, dq, False, xf, True, batch=3)
    err = _compare32(pkg, model, data, 3, 1e-1)
    h = handle_from_model(pkg, model)
    z64 = h.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=3, k=1e-1, check=False)
    assert h.last_dispatch()[0] == pkg.FMPC_PATH_GENERIC
    h.set_precision("f32")
    z32 = h.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=3, k=1e-1, check=False)
    assert h.last_dispatch()[0] == pkg._lib.FMPC_PATH_TILED_F32 and rel_err(z32, z64) <= TOL32
    h.close()
    print(f"fp32 factor, n={n}: max rel err on z {err:.2e}")
