"""Dense oracle (oracle/dense_ref.py) against hand-built expectations and independent algebra.
PARITY UNPINNED by the reference (MATLAB-only, no vectors): these tests pin the restatement to the
cited reference lines through small hand-written cases and to the mathematics through a different
algorithm (full KKT solve, stationarity)."""
import numpy as np
import pytest

from oracle.dense_ref import DenseFastMPC, RefError, inf_newton_KKT_H, inf_newton_solver, deinterleave
from tests.util import dense_from_model


def small():
    """(n, m, T) = (2, 1, 3) with distinct entries so that every block position is visible."""
    A1 = np.array([[1.0, 2.0], [3.0, 4.0]]); A2 = np.array([[5.0, 6.0], [7.0, 8.0]])
    B = np.array([[9.0], [10.0]])
    Q = np.diag([11.0, 12.0]); R = np.array([[13.0]]); Qf = np.diag([14.0, 15.0])
    return dict(n=2, m=1, T=3, A1=A1, A2=A2, B=B, Q=Q, R=R, Qf=Qf, u_min=np.array([-1.0]), u_max=np.array([3.0]),
                x_min=np.array([-2.0, -4.0]), x_max=np.array([6.0, 8.0]))


def test_assembly_T1():
    """Index maps of fast_mpc_objective.m:50-65, fast_mpc_eq_const.m:38-49, fast_mpc_ineq_const.m:46-56,
    fast_mpc_init.m:19-25 on (n,m,T)=(2,1,3); z = [u0; x1; u1; x2; u2; x3]."""
    md = small()
    x0 = np.array([0.5, -0.5]); x0p = np.array([0.25, 0.75]); w = np.arange(1.0, 7.0)
    d = dense_from_model(md, x0, x0p, w)
    H, g = d.objective_function()
    expH = np.zeros((9, 9)); expH[0, 0] = 13
    expH[1:3, 1:3] = md["Q"]; expH[3, 3] = 13; expH[4:6, 4:6] = md["Q"]; expH[6, 6] = 13; expH[7:9, 7:9] = md["Qf"]
    assert np.array_equal(H, expH) and np.array_equal(g, np.zeros(9))
    C, b = d.equality_const()
    I2 = np.eye(2); expC = np.zeros((6, 9))
    expC[0:2, 0:1] = -md["B"]; expC[0:2, 1:3] = I2
    expC[2:4, 1:3] = -md["A1"]; expC[2:4, 3:4] = -md["B"]; expC[2:4, 4:6] = I2
    expC[4:6, 1:3] = -md["A2"]; expC[4:6, 4:6] = -md["A1"]; expC[4:6, 6:7] = -md["B"]; expC[4:6, 7:9] = I2
    assert np.array_equal(C, expC)
    expb = np.concatenate([md["A1"] @ x0 + md["A2"] @ x0p + w[0:2], md["A2"] @ x0 + w[2:4], w[4:6]])
    assert np.allclose(b, expb, rtol=0, atol=0)
    P, h = d.inequality_const()
    expP = np.zeros((6, 9))
    for j, col in enumerate([0, 3, 6]):
        expP[2 * j, col] = 1; expP[2 * j + 1, col] = -1
    assert np.array_equal(P, expP) and np.array_equal(h, np.tile([3.0, 1.0], 3))
    z0 = d.initialize()
    assert np.array_equal(z0, np.tile([1.0, 2.0, 2.0], 3))


def test_linear_terms_xf_and_xinit():
    md = small()
    d = DenseFastMPC(md["Q"], md["R"], None, md["Qf"], [1.0, 2.0], [3.0], [4.0, 5.0], md["x_min"], md["x_max"],
                     md["u_min"], md["u_max"], None, None, 3, [0.0, 0.0], [0.0, 0.0], [0.0], md["A1"], md["A2"],
                     md["B"], np.zeros(6), [7.0, 8.0], np.arange(9.0))
    _, g = d.objective_function()
    assert np.array_equal(g, [3, 1, 2, 3, 1, 2, 3, 4, 5])
    C, b = d.equality_const()
    assert C.shape == (8, 9) and np.array_equal(C[6:8, 7:9], np.eye(2)) and np.array_equal(b[6:8], [7.0, 8.0])
    assert np.count_nonzero(C[6:8, :7]) == 0
    assert np.array_equal(d.initialize(), np.arange(9.0))
    d.x_init = np.arange(8.0)
    with pytest.raises(RefError):
        d.initialize()


def test_reference_error_paths():
    md = small()
    base = lambda **kw: dense_from_model({**md, **kw}, np.zeros(2), np.zeros(2), np.zeros(6))
    with pytest.raises(RefError):
        dense_from_model(md, np.zeros(3), np.zeros(2), np.zeros(6)).equality_const()
    with pytest.raises(RefError):
        base(u_min=np.zeros(2)).inequality_const()
    with pytest.raises(RefError):       # quirk D7: w=[] becomes n long and is indexed past its end
        dense_from_model(md, np.zeros(2), np.zeros(2), None).equality_const()
    d = dense_from_model(md, np.zeros(2), np.zeros(2), np.zeros(6)); d.A2 = None
    with pytest.raises(RefError):
        d.equality_const()


def test_literal_D_is_bitwise_equal():
    md = small(); rng = np.random.default_rng(0)
    d = dense_from_model(md, rng.random(2), rng.random(2), rng.random(6))
    z, H, g, P, h, C, b = d._assemble()
    z = z + 0.1 * rng.standard_normal(9)
    Ph1, d1 = inf_newton_KKT_H(H, P, h, z, 0.37, literal_D=True)
    Ph2, d2 = inf_newton_KKT_H(H, P, h, z, 0.37, literal_D=False)
    assert np.array_equal(Ph1, Ph2) and np.array_equal(d1, d2)


@pytest.mark.parametrize("xf", [False, True])
def test_one_newton_step_equals_full_KKT_solve_T3(xf):
    """inf_newton_solver.m:24-35 (block elimination) == solve([[Phi, C'],[C, 0]]) (different route)."""
    import importlib
    S = importlib.import_module("mpc-sensorlessao_amd").synthetic
    md, data = S.make_test_problem(8, 5, 6, seed=2, xf=xf)
    d = dense_from_model(md, data["x0"][0], data["x0_pre"][0], data["w"][0])
    z, H, g, P, h, C, b = d._assemble()
    k = 0.3; nu0 = data["nu0"][0]
    info = {}
    z1 = inf_newton_solver(H, g, P, h, C, b, k, z, 1, nu0=nu0, info=info)
    assert info["t"] == [1.0]
    Phi, dd = inf_newton_KKT_H(H, P, h, z, k)
    rd = 2 * H @ z + g + k * P.T @ dd + C.T @ nu0
    rp = C @ z - b
    K = np.block([[Phi, C.T], [C, np.zeros((C.shape[0],) * 2)]])
    sol = np.linalg.solve(K, -np.concatenate([rd, rp]))
    assert np.allclose(z1, z + sol[:z.size], rtol=1e-11, atol=1e-12)
    assert np.allclose(info["nu"], nu0 + sol[z.size:], rtol=1e-10, atol=1e-11)


def test_converges_to_stationary_point_and_early_exit_T7():
    import importlib
    S = importlib.import_module("mpc-sensorlessao_amd").synthetic
    md, data = S.make_test_problem(8, 5, 10, seed=4)
    d = dense_from_model(md, data["x0"][0], data["x0_pre"][0], data["w"][0])
    info = {}
    z = d.mpc_fixed_log(0.01, nu0=data["nu0"][0], info=info)
    _, H, g, P, h, C, b = d._assemble()
    dd = 1.0 / (h - P @ z)                                   # barrier gradient RE-evaluated at the result
    assert np.linalg.norm(2 * H @ z + g + 0.01 * P.T @ dd + C.T @ info["nu"]) < 1e-5
    assert np.linalg.norm(C @ z - b) < 1e-8
    assert info["iters"] < 20 and len(info["n_r"]) == info["iters"] + 1   # exit tested before the step
    d2 = dense_from_model(md, data["x0"][0], data["x0_pre"][0], data["w"][0], x_init=z)
    info2 = {}
    z2 = d2.mpc_fixed_log_newton(5, 0.01, nu0=info["nu"], info=info2)
    assert info2["iters"] == 0 and np.array_equal(z2, z)      # converged input: returned unchanged


def test_no_feasibility_safeguard_T5():
    """Quirk D3: iterates may leave the box (D = 1/s^2 stays positive) and the solver carries on."""
    import importlib
    S = importlib.import_module("mpc-sensorlessao_amd").synthetic
    md, data = S.make_test_problem(8, 5, 10, seed=7, umax=0.3)
    d = dense_from_model(md, data["x0"][0], data["x0_pre"][0], data["w"][0])
    z = d.mpc_fixed_log_newton(1, 0.01, nu0=data["nu0"][0])
    U, _ = deinterleave(z, 8, 5, 10)
    assert np.abs(U).max() > 0.3


def test_nu0_only_matters_through_the_line_search_T6():
    import importlib
    S = importlib.import_module("mpc-sensorlessao_amd").synthetic
    md, data = S.make_test_problem(8, 5, 10, seed=9)
    d = dense_from_model(md, data["x0"][0], data["x0_pre"][0], data["w"][0])
    i1, i2 = {}, {}
    za = d.mpc_fixed_log_newton(1, 0.01, nu0=data["nu0"][0], info=i1)
    zb = d.mpc_fixed_log_newton(1, 0.01, nu0=np.random.default_rng(1).random(80), info=i2)
    assert i1["t"] == [1.0] and i2["t"] == [1.0]
    assert np.allclose(za, zb, rtol=1e-12, atol=1e-13)


def test_driver_schedules_T8():
    import importlib
    S = importlib.import_module("mpc-sensorlessao_amd").synthetic
    md, data = S.make_test_problem(4, 3, 3, seed=1)
    d = dense_from_model(md, data["x0"][0], data["x0_pre"][0], data["w"][0])
    rng = np.random.default_rng(0)
    infos = []
    d.mpc_fixed_newton(2, rng=rng, infos=infos)
    ks = [i["k"] for i in infos]                 # k = 1, 0.1, ... while k*N_z >= 10e-3 (N_z = 21)
    assert len(ks) == 4 and np.allclose(ks, [1, .1, .01, .001])     # 1e-4 * 21 < 10e-3 stops the loop
    assert all(i["iters"] <= 2 for i in infos)
    infos = []
    d.mpc_solve_check(1e-3, 1.0, rng=rng, infos=infos)
    assert np.allclose([i["k"] for i in infos], np.linspace(1.0, 1e-3, 5))


def test_var1_bug_compat_T9_and_ramp_T10():
    import importlib
    S = importlib.import_module("mpc-sensorlessao_amd").synthetic
    md, data = S.make_test_problem(4, 3, 3, seed=3, var_order=1)          # n = m + 1: the misplaced block lands right
    ok = dense_from_model(md, data["x0"][0], None, data["w"][0])
    bug = dense_from_model(md, data["x0"][0], None, data["w"][0], bug_compat_var1=True)
    assert np.array_equal(ok.equality_const()[0], bug.equality_const()[0])
    md2, data2 = S.make_test_problem(8, 5, 3, seed=3, var_order=1)         # n != m + 1: it does not
    ok2 = dense_from_model(md2, data2["x0"][0], None, data2["w"][0])
    bug2 = dense_from_model(md2, data2["x0"][0], None, data2["w"][0], bug_compat_var1=True)
    C_ok, C_bug = ok2.equality_const()[0], bug2.equality_const()[0]
    assert not np.array_equal(C_ok, C_bug)
    assert np.array_equal(C_bug[8:16, 7:7 + 21], np.hstack([-md2["A1"], -md2["B"], np.eye(8)]))
    # intended VAR(1) == VAR_2 code with A2 = 0
    md3 = dict(md2); md3["var_order"] = 2; md3["A2"] = np.zeros((8, 8))
    v2 = dense_from_model(md3, data2["x0"][0], np.zeros(8), data2["w"][0])
    assert np.array_equal(v2.equality_const()[0], C_ok)
    # ramp rows (VAR_1/fast_mpc_ineq_const.m:58-76)
    r = DenseFastMPC.var1(md2["Q"], md2["R"], None, md2["Qf"], None, None, None, md2["x_min"], md2["x_max"],
                          md2["u_min"], md2["u_max"], -0.2 * np.ones(5), 0.3 * np.ones(5), 3, data2["x0"][0],
                          0.1 * np.ones(5), md2["A1"], md2["B"], data2["w"][0], None, None)
    P, h = r.inequality_const()
    assert P.shape == (4 * 3 * 5, 3 * 13)
    Pr, hr = P[30:], h[30:]
    assert np.array_equal(Pr[0:5, 0:5], np.eye(5)) and np.allclose(hr[0:5], 0.1 + 0.3) and np.allclose(hr[5:10], -0.1 + 0.2)
    assert np.array_equal(Pr[10:15, 0:5], -np.eye(5)) and np.array_equal(Pr[10:15, 13:18], np.eye(5))
    assert np.allclose(hr[10:15], 0.3) and np.allclose(hr[15:20], 0.2)
    z = r.mpc_fixed_log_newton(3, 0.01, nu0=np.zeros(24))          # dense path runs with the ramp rows
    assert np.all(np.isfinite(z))
