"""Property tests on random problems: random stable A1/A2, random B, symmetric positive definite diagonal
or dense Q / Qf / R, random boxes (the cold start is the mid-box, an explicit start lies strictly inside), optional linear
costs, terminal rows, disturbances; barrier weights and Newton budgets drawn per case.  The generator is modelled on the
reference's own random problem, Fast_MPC/VAR_2/test_fast_mpc.m:8-37 (A = rand / spectral radius, B = rand, w = rand,
x0 = rand), widened over sizes and weights.

  CPU (hypothesis, derandomised):   dense restatement == structured restatement; the literal line search's accepted step
                                    satisfies its own test and the trial before it did not (backtracking_inf_newton.m:2-11)
  GPU (seeded sweep, >= 200 cases): HIP == structured oracle through the C ABI, n in 3..40, m in 1..150, T in 1..12."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from tests.util import banded_from_model, canon_steps, dense_from_model, handle_from_model, rel_err


def _spd(rng, size, dense, scale):
    if not dense:
        return np.diag(scale * rng.uniform(0.5, 2.0, size))
    G = rng.standard_normal((size, size))
    M = G @ G.T / size + 0.5 * np.eye(size)
    return scale * (M + M.T) / 2


def random_problem(seed, n, m, T, var_order=2, dense_q=False, dense_r=False, xf=False, lin=False, batch=1, umax=None):
    """A model + `batch` problems.  Stability: the companion matrix of (A1, A2) has spectral radius <= 1
    (test_fast_mpc.m:29 normalises A the same way)."""
    rng = np.random.default_rng(seed)
    A1 = rng.random((n, n))
    A2 = (0.4 * rng.standard_normal((n, n)) / np.sqrt(n)) if var_order == 2 else np.zeros((n, n))
    comp = np.block([[A1, A2], [np.eye(n), np.zeros((n, n))]])
    sr = np.max(np.abs(np.linalg.eigvals(comp)))
    A1, A2 = A1 / sr, A2 / sr ** 2                                  # companion of (A1/s, A2/s^2) has the eigenvalues / s
    B = rng.random((n, m))
    umax = float(rng.uniform(0.8, 6.0)) if umax is None else umax
    lo = -umax * rng.uniform(0.5, 1.0, m); hi = umax * rng.uniform(0.5, 1.0, m)
    model = dict(n=n, m=m, T=T, var_order=var_order, A1=A1, A2=A2, B=B,
                 Q=_spd(rng, n, dense_q, rng.uniform(0.5, 20.0)), Qf=_spd(rng, n, dense_q, rng.uniform(5.0, 60.0)),
                 R=_spd(rng, m, dense_r, rng.uniform(0.5, 2.0)), u_min=lo, u_max=hi,
                 x_min=-10.0 * np.ones(n), x_max=10.0 * np.ones(n),
                 xf=rng.random(n) if xf else None)
    if lin:
        model.update(q=0.3 * rng.standard_normal(n), r=0.3 * rng.standard_normal(m), qf=0.3 * rng.standard_normal(n))
    nb = T + (1 if xf else 0)
    data = dict(x0=rng.random((batch, n)), x0_pre=rng.random((batch, n)) if var_order == 2 else None,
                w=rng.random((batch, T * n)), nu0=rng.random((batch, nb * n)))
    return model, data


def random_interior_start(seed, model, batch):
    """x_init strictly inside the box on the u entries (fast_mpc_init.m:12-14 takes any N_z vector)."""
    rng = np.random.default_rng(seed + 77)
    n, m, T = model["n"], model["m"], model["T"]
    z = np.empty((batch, T, n + m))
    z[:, :, :m] = model["u_min"] + (model["u_max"] - model["u_min"]) * rng.uniform(0.1, 0.9, (batch, T, m))
    z[:, :, m:] = rng.uniform(-1.0, 1.0, (batch, T, n))
    return z.reshape(batch, -1)


# --------------------------------------------------------------------------------------------------------------- CPU
@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 10 ** 6), n=st.integers(3, 9), m=st.integers(1, 12), T=st.integers(1, 6),
       var_order=st.sampled_from([1, 2]), dense_q=st.booleans(), dense_r=st.booleans(), xf=st.booleans(), lin=st.booleans(),
       k=st.sampled_from([1e-2, 1e-1, 1.0]), nw=st.integers(1, 5), warm=st.booleans())
def test_dense_equals_structured_on_random_problems(seed, n, m, T, var_order, dense_q, dense_r, xf, lin, k, nw, warm):
    model, data = random_problem(seed, n, m, T, var_order, dense_q, dense_r, xf, lin)
    zi = random_interior_start(seed, model, 1)[0] if warm else None
    d = dense_from_model(model, data["x0"][0], None if var_order == 1 else data["x0_pre"][0], data["w"][0], x_init=zi)
    info_d = {}
    zd = d.mpc_fixed_log_newton(nw, k, nu0=data["nu0"][0], info=info_d)
    b = banded_from_model(model)
    info_b = {}
    zb, nub, itb, stb = b.solve(data["x0"][0], None if var_order == 1 else data["x0_pre"][0], data["w"][0], nw, k, z_init=zi,
                                nu0=data["nu0"][0], info=info_b)
    assert itb == info_d["iters"]
    assert np.array_equal(canon_steps(info_b.get("t", [])), canon_steps(info_d.get("t", [])))
    assert rel_err(zb, zd) <= 1e-8 and rel_err(nub, info_d["nu"]) <= 1e-7


@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 10 ** 6), n=st.integers(3, 8), m=st.integers(1, 10), T=st.integers(1, 5),
       k=st.sampled_from([1e-2, 1.0, 10.0]), umax=st.sampled_from([0.3, 1.0, 4.0]))
def test_literal_line_search_accepts_a_sufficient_decrease(seed, n, m, T, k, umax):
    """backtracking_inf_newton.m:4: the accepted t satisfies ||r(t)|| <= (1 - alpha t) ||r(0)|| with the barrier gradient
    frozen (quirk D4), so that norm decreases monotonically in the literal test's own measure; t is the FIRST of 1, 1/2, ...
    that does (checked through the closed form of the structured restatement, which must pick the same t)."""
    model, data = random_problem(seed, n, m, T, umax=umax)
    d = dense_from_model(model, data["x0"][0], data["x0_pre"][0], data["w"][0])
    info = {}
    d.mpc_fixed_log_newton(6, k, nu0=data["nu0"][0], info=info)
    t = np.asarray(info.get("t", []))
    for s in range(len(t)):
        if t[s] >= 1e-12:
            assert info["n_after"][s] <= (1 - 1e-4 * t[s]) * info["n_before"][s] * (1 + 1e-12)
            assert info["n_after"][s] < info["n_before"][s]
        assert np.isclose(info["n_before"][s], info["n_r"][s], rtol=1e-12)          # the norm of the exit test is the search's start
    ib = {}
    banded_from_model(model).solve(data["x0"][0], data["x0_pre"][0], data["w"][0], 6, k, nu0=data["nu0"][0], info=ib)
    assert np.array_equal(canon_steps(ib.get("t", [])), canon_steps(t))


# --------------------------------------------------------------------------------------------------------------- GPU
def _gpu_cases():
    """~45 models x 5-6 problems: >= 200 (model, problem) cases per run, sizes over the whole supported range."""
    rng = np.random.default_rng(20240604)
    cases = []
    for i in range(46):
        n = int(rng.integers(3, 41)); m = int(rng.integers(1, 151)); T = int(rng.integers(1, 13))
        if i == 0:
            n, m, T = 27, 144, 12                     # the AO sizes take the specialised kernels
        if i == 1:
            n, m, T = 40, 150, 12                     # the corner of the range
        if i == 2:
            n, m, T = 3, 1, 1
        dense_q = bool(rng.integers(0, 4) == 0); dense_r = bool(rng.integers(0, 5) == 0) and m <= 64
        # (terminal rows x_T = xf only where ONE stage has the controls to reach them, m >= n: with fewer the equality rows are
        #  rank deficient or nearly so -- n = 36, m = 7, T = 11 gave cond(Y) = 2.5e14, the two CPU restatements then differ by
        #  3e-7 -- chol(Y) may fail in the reference too, and WHICH problems fail is decided by round-off)
        xf_ok = m >= n
        cases.append(dict(seed=1000 + i, n=n, m=m, T=T, var_order=int(rng.integers(1, 3)), dense_q=dense_q, dense_r=dense_r,
                          xf=bool(rng.integers(0, 3) == 0) and xf_ok, lin=bool(rng.integers(0, 2)), k=float(rng.choice([1e-2, 1e-1, 1.0])),
                          nw=int(rng.integers(1, 6)), warm=bool(rng.integers(0, 2)), batch=int(rng.integers(5, 7))))
    return cases


@pytest.mark.gpu
def test_hip_equals_structured_oracle_on_random_problems(pkg, gpu):
    from tests.util import oracle_batch
    solved = 0
    worst = 0.0
    for c in _gpu_cases():
        model, data = random_problem(c["seed"], c["n"], c["m"], c["T"], c["var_order"], c["dense_q"], c["dense_r"], c["xf"], c["lin"],
                                     batch=c["batch"])
        zi = random_interior_start(c["seed"], model, c["batch"]) if c["warm"] else None
        try:
            h = handle_from_model(pkg, model)
        except pkg.FastMPCError as e:
            assert e.code == pkg._lib.FMPC_E_UNSUPPORTED, (c, e)          # e.g. a dense R whose factor does not fit the LDS
            continue
        z, info = h.solve(data["x0"], data["x0_pre"], data["w"], z_init=zi, nu0=data["nu0"], n_newton=c["nw"], k=c["k"], return_info=True,
                          check=False)
        zo, nuo, ito, sto, steps = oracle_batch(model, data, c["nw"], c["k"], z_init=zi)
        assert np.array_equal(info["status"], sto), (c, info["status"], sto)
        assert (sto >= 0).all(), (c, sto)                              # the generator keeps the problems well posed
        for p in range(c["batch"]):
            e = rel_err(z[p], zo[p])
            worst = max(worst, e)
            assert e <= 1e-9, (c, p, e)
        assert np.array_equal(info["iters"], ito), c
        solved += c["batch"]
        h.close()
    assert solved >= 200, solved
    print("random problems solved on the device: %d, worst relative error on z %.2e" % (solved, worst))


@pytest.mark.gpu
def test_hip_equals_structured_oracle_on_random_problems_of_any_size(pkg, gpu):
    """The same property beyond the specialised kernels' sizes (n = 80 .. 140, diagonal weights): the generic kernel with its tiles
    in the HBM workspace (tests/test_gpu_any_size.py), 12 random models, cold and explicit starts, linear costs, terminal rows."""
    from tests.util import oracle_batch
    rng = np.random.default_rng(20261005)
    worst, solved = 0.0, 0
    for i in range(12):
        n = int(rng.integers(80, 141)); m = int(rng.integers(1, 201)); T = int(rng.integers(1, 7))
        c = dict(seed=5000 + i, n=n, m=m, T=T, var_order=int(rng.integers(1, 3)), xf=bool(rng.integers(0, 2)) and m >= n, lin=bool(rng.integers(0, 2)),
                 k=float(rng.choice([1e-2, 1e-1, 1.0])), nw=int(rng.integers(1, 5)), warm=bool(rng.integers(0, 2)), batch=int(rng.integers(2, 4)))
        model, data = random_problem(c["seed"], n, m, T, c["var_order"], False, False, c["xf"], c["lin"], batch=c["batch"])
        zi = random_interior_start(c["seed"], model, c["batch"]) if c["warm"] else None
        h = handle_from_model(pkg, model)
        z, info = h.solve(data["x0"], data["x0_pre"], data["w"], z_init=zi, nu0=data["nu0"], n_newton=c["nw"], k=c["k"], return_info=True, check=False)
        assert h.last_dispatch()[0] == pkg.FMPC_PATH_GENERIC
        h.close()
        zo, nuo, ito, sto, steps = oracle_batch(model, data, c["nw"], c["k"], z_init=zi)
        assert np.array_equal(info["status"], sto) and (sto >= 0).all(), (c, info["status"], sto)
        assert np.array_equal(info["iters"], ito), c
        for p in range(c["batch"]):
            e = rel_err(z[p], zo[p])
            worst = max(worst, e)
            assert e <= 1e-9, (c, p, e)
        solved += c["batch"]
    print("random problems of any size solved on the device: %d, worst relative error on z %.2e" % (solved, worst))
