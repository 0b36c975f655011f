"""Sizes no specialised kernel takes (n > 79 with diagonal weights; the reference checks shapes only, fast_mpc_objective.m:17-47):
the generic kernel's instance with its tiles in the HBM workspace ("big").  Same bar as tests/test_gpu_parity.py: 1e-9 on z
against the structured oracle, identical iteration counts, status codes and line-search steps."""
import numpy as np
import pytest

from tests.util import handle_from_model, oracle_batch, rel_err

pytestmark = pytest.mark.gpu


def _compare(pkg, model, data, nw, k, z_init=None, tol=1e-9, expect_path=None, prec=None):
    h = handle_from_model(pkg, model)
    if prec:
        h.set_precision(prec)
    z, info = h.solve(data["x0"], data.get("x0_pre"), data.get("w"), z_init=z_init, nu0=data.get("nu0"), n_newton=nw, k=k,
                      return_info=True, check=False)
    path = h.last_dispatch()[0]
    h.close()
    if expect_path is not None:
        assert path == expect_path, path
    zo, nuo, ito, sto, steps = oracle_batch(model, data, nw, k, z_init)
    assert np.array_equal(info["status"], sto), (info["status"], sto)
    assert np.array_equal(info["iters"], ito), (info["iters"], ito)
    for p in range(z.shape[0]):
        assert rel_err(z[p], zo[p]) <= tol, (p, rel_err(z[p], zo[p]))
        assert rel_err(info["nu"][p], nuo[p]) <= 1e-7
        assert np.array_equal(info["step"][p][:ito[p]], np.array(steps[p])), (p, info["step"][p], steps[p])
    return z, info


@pytest.mark.parametrize("xf", [False, True])
@pytest.mark.parametrize("var_order", [2, 1])
def test_big_instance_on_the_reference_demo_config(pkg, gpu, monkeypatch, xf, var_order):
    """The instance forced on a size the LDS instance takes too (test_fast_mpc.m:8-37: n = 8, m = 5, T = 10): cold start with
    budgets 1, 5 and 0 (= until convergence), and an off-centre start under a strong barrier (backtracking)."""
    monkeypatch.setenv("FMPC_GENERIC_BIG", "1")
    model, data = pkg.synthetic.make_test_problem(8, 5, 10, seed=3, xf=xf, var_order=var_order, batch=4)
    for nw in (1, 5, 0):
        _compare(pkg, model, data, nw, 0.01, expect_path=pkg.FMPC_PATH_GENERIC)
    model, data = pkg.synthetic.make_test_problem(8, 5, 10, seed=7, umax=0.3, xf=xf, var_order=var_order, batch=6)
    rng = np.random.default_rng(11)
    z0 = np.zeros((6, 10, 13))
    z0[:, :, :5] = rng.uniform(-0.25, 0.25, (6, 10, 5)); z0[:, :, 5:] = rng.uniform(-1, 1, (6, 10, 8))
    z, info = _compare(pkg, model, data, 8, 10.0, z_init=z0.reshape(6, 130), expect_path=pkg.FMPC_PATH_GENERIC)
    assert np.any((info["step"] > 0) & (info["step"] < 1)), "case must exercise backtracking"


@pytest.mark.parametrize("n,m,T,var_order,xf", [(100, 40, 5, 2, True), (130, 150, 4, 1, False), (83, 7, 3, 2, False), (70, 300, 2, 2, False)])
def test_sizes_beyond_the_specialised_kernels(pkg, gpu, n, m, T, var_order, xf):
    """n > 79 (the tiled kernel's limit) with diagonal weights: solved, by the generic path, to the parity bar; w and an explicit
    dual start included.  n = 70 with m = 300 (the fp64 tiles of the matrix-core kernel do not fit the LDS: the fp32 factor is the default
    there) takes the same path when fp64 is asked for."""
    model, data = pkg.synthetic.make_test_problem(n, m, T, seed=n + m, xf=xf, var_order=var_order, batch=3)
    for nw in (1, 4):
        # (n = 70: fp64 on request takes the eight-wavefront fp64 instance of the tiled kernel where its tiles fit the LDS -- m = 300 does not)
        _compare(pkg, model, data, nw, 0.01, expect_path=pkg.FMPC_PATH_GENERIC, prec="f64" if n <= 79 else None)


def test_first_moves_and_device_entry_at_a_big_size(pkg, gpu):
    import torch
    dev = torch.device("cuda:0")
    model, data = pkg.synthetic.make_test_problem(96, 20, 4, seed=5, batch=5)
    h = handle_from_model(pkg, model)
    z = h.solve(data["x0"], data.get("x0_pre"), data.get("w"), n_newton=3, k=0.01)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    u0 = torch.empty((5, 20), dtype=torch.float64, device=dev)
    zd = h.solve_device(t(data["x0"]), t(data.get("x0_pre")), t(data.get("w")), None, None, 3, 0.01, u0_out=u0)[0]
    torch.cuda.synchronize()
    assert np.array_equal(zd.cpu().numpy(), z) and np.array_equal(u0.cpu().numpy(), z[:, :20])
    h.close()


def test_dense_state_weights_at_any_size(pkg, gpu, monkeypatch):
    """Dense symmetric positive definite Q, Qf (fast_mpc_objective.m:52-55) beyond the tiled kernel's sizes: the workspace
    instance applies 2Q, 2Qf and their inverses as matrices.  Also forced on a small model; fp64 on request at n = 60 (tiled instance and, forced, the workspace instance)."""
    from tests.test_property_random import random_problem
    for (seed, n, m, T, var, xf, lin) in ((11, 90, 30, 4, 2, False, True), (12, 84, 100, 3, 1, True, False)):
        model, data = random_problem(seed, n, m, T, var, True, False, xf, lin, batch=3)
        _compare(pkg, model, data, 3, 0.1, expect_path=pkg.FMPC_PATH_GENERIC)
    model, data = random_problem(13, 60, 20, 4, 2, True, False, False, True, batch=2)
    _compare(pkg, model, data, 3, 0.1, expect_path=pkg._lib.FMPC_PATH_TILED, prec="f64")        # (n = 60: the fp64 tiled instance of 4 blocks)
    monkeypatch.setenv("FMPC_GENERIC_BIG", "1")
    _compare(pkg, model, data, 3, 0.1, expect_path=pkg.FMPC_PATH_GENERIC)                        # (the same in the workspace instance)
    model, data = random_problem(14, 9, 6, 5, 2, True, False, True, True, batch=4)
    _compare(pkg, model, data, 4, 0.01, expect_path=pkg.FMPC_PATH_GENERIC)


def test_dense_input_weight_at_any_size(pkg, gpu, monkeypatch):
    """Dense symmetric positive definite R (fast_mpc_objective.m:51-54; the u block of Phi is then a dense m x m matrix per stage and
    Newton step, inf_newton_KKT_H.m:13) beyond the tiled kernel's n <= 47: factored per stage in the workspace (ft_dense_r).  With
    dense Q too, with backtracking (tight bounds, off-centre start), and forced on a small model."""
    from tests.test_property_random import random_problem, random_interior_start
    for (seed, n, m, T, var, dq, xf, lin) in ((21, 90, 30, 3, 2, False, False, True), (22, 60, 70, 3, 1, True, True, False)):
        model, data = random_problem(seed, n, m, T, var, dq, True, xf, lin, batch=3)
        _compare(pkg, model, data, 3, 0.1, expect_path=pkg.FMPC_PATH_GENERIC)
    model, data = random_problem(23, 85, 12, 4, 2, False, True, False, True, batch=3, umax=0.4)
    zi = random_interior_start(23, model, 3)
    z, info = _compare(pkg, model, data, 6, 5.0, z_init=zi, expect_path=pkg.FMPC_PATH_GENERIC)
    monkeypatch.setenv("FMPC_GENERIC_BIG", "1")
    model, data = random_problem(24, 9, 6, 5, 2, True, True, True, True, batch=4)
    _compare(pkg, model, data, 4, 0.01, expect_path=pkg.FMPC_PATH_GENERIC)


def test_closed_loop_step_at_a_big_size(pkg, gpu):
    """The coefficient-space loop entry (README.md:482-497) at n = 96: loop inputs by the any-size kernel + the solve by the generic
    path, against the same two steps done by hand (numpy inputs + host-pointer solve)."""
    import torch
    from tests.test_gpu_closed_loop import design_matrices
    dev = torch.device("cuda:0")
    n, m, T, R = 96, 20, 4, 6
    model, _ = pkg.synthetic.make_test_problem(n, m, T, seed=9, batch=1)
    rng = np.random.default_rng(3)
    a, xl, u1, u2 = 0.1 * rng.standard_normal((R, n)), 0.1 * rng.standard_normal((R, n)), 0.1 * rng.standard_normal((R, m)), 0.1 * rng.standard_normal((R, m))
    h = handle_from_model(pkg, model)
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
    f = dict(dtype=torch.float64, device=dev)
    x0, x0p, w = torch.empty((R, n), **f), torch.empty((R, n), **f), torch.empty((R, T * n), **f)
    z = torch.empty((R, h.nz), **f)
    u0 = torch.empty((R, m), **f)
    h.loop_step_device(t(a), t(xl), t(u1), t(u2), x0, x0p, w, n_newton=3, k=0.01, z_out=z, u0_out=u0)
    torch.cuda.synchronize()
    M1, M2 = design_matrices(model["A1"], model["A2"], T)
    bu1, bu2 = (model["B"] @ u1.T).T, (model["B"] @ u2.T).T
    w_ref = -(M1 @ bu1.T).T - (M2 @ bu2.T).T
    assert np.abs(w.cpu().numpy() - w_ref).max() <= 1e-12 * max(1.0, np.abs(w_ref).max())
    z_ref = h.solve(a + bu1, xl, w_ref, n_newton=3, k=0.01)
    assert rel_err(z.cpu().numpy(), z_ref) <= 1e-10 and np.array_equal(u0.cpu().numpy(), z.cpu().numpy()[:, :m])
    h.close()


@pytest.mark.parametrize("n,m,T,var_order,xf,dq", [(50, 30, 4, 2, False, False), (65, 144, 6, 2, False, False), (79, 40, 3, 1, False, True),
                                                   (64, 80, 3, 2, True, False), (50, 20, 5, 2, False, False)])
def test_fp64_tiled_instances_of_four_and_five_blocks(pkg, gpu, n, m, T, var_order, xf, dq):
    """47 < n <= 79: fp64 (the default since round 5; also asked for explicitly here) runs on the matrix cores too --
    fmpc_newton_tiled<double, 4 | 5, 8> -- to the fp64 parity bar; n = 50 with m = 20 (the generic kernel's LDS tiles fit: fp64 is
    the default) takes it without asking."""
    from tests.test_property_random import random_problem, random_interior_start
    model, data = random_problem(300 + n, n, m, T, var_order, dq, False, xf, True, batch=4)
    zi = random_interior_start(n, model, 4)
    _compare(pkg, model, data, 1, 0.1, expect_path=pkg._lib.FMPC_PATH_TILED, prec=None if (n, m) == (50, 20) else "f64")
    _compare(pkg, model, data, 4, 1.0, z_init=zi, expect_path=pkg._lib.FMPC_PATH_TILED, prec=None if (n, m) == (50, 20) else "f64")
