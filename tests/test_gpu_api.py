"""GPU tests of the boundary and of the host mirror: raw C ABI (column-major, NULL optionals, status
codes), the Fast_MPC2 drivers against the dense oracle's drivers, golden fixtures, both kernels
(generic and one-wave-per-problem), the device-pointer entry point, and properties at full size."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from tests.test_golden import FILES, load_case
from tests.util import canon_steps, dense_from_model, handle_from_model, oracle_batch, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _p(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.c_void_p)


def test_raw_c_abi_column_major_and_null_optionals(pkg, gpu):
    """Pass MATLAB-layout (column-major) matrices by hand; A1, A2, B are not symmetric, so a row/column
    mix-up cannot pass.  x0_pre = NULL means zeros, w = NULL zeros, z_init = NULL cold start."""
    lib = pkg.load()
    md, data = pkg.synthetic.make_test_problem(8, 5, 10, seed=31, batch=2)
    F = lambda M: np.asfortranarray(M).ravel(order="K").copy()      # column-major flat copy
    h = C.c_void_p()
    rc = lib.fmpc_create(C.byref(h), 8, 5, 10, 2, _p(F(md["A1"])), _p(F(md["A2"])), _p(F(md["B"])), _p(F(md["Q"])),
                         _p(F(md["R"])), _p(F(md["Qf"])), None, None, None, _p(md["x_min"]), _p(md["x_max"]),
                         _p(md["u_min"]), _p(md["u_max"]), None, 0)
    assert rc == 0
    n = C.c_int(); m = C.c_int(); T = C.c_int(); nz = C.c_int(); nul = C.c_int()
    assert lib.fmpc_dims(h, C.byref(n), C.byref(m), C.byref(T), C.byref(nz), C.byref(nul)) == 0
    assert (n.value, m.value, T.value, nz.value, nul.value) == (8, 5, 10, 130, 80)
    z = np.empty((2, 130)); nu = np.empty((2, 80)); st = np.zeros(2, np.int32); it = np.zeros(2, np.int32)
    step = np.empty((2, 3))
    rc = lib.fmpc_solve(h, 2, _p(data["x0"]), None, None, None, _p(data["nu0"]), 3, 0.01, _p(z), _p(nu),
                        st.ctypes.data_as(C.c_void_p), it.ctypes.data_as(C.c_void_p), _p(step))
    assert rc == 0
    d0 = dict(x0=data["x0"], x0_pre=np.zeros((2, 8)), w=None, nu0=data["nu0"])
    zo, nuo, ito, sto, steps = oracle_batch(md, d0, 3, 0.01)
    assert np.array_equal(it, ito) and np.array_equal(st, sto)
    assert max(rel_err(z[p], zo[p]) for p in range(2)) <= TOL
    # nullable outputs
    rc = lib.fmpc_solve(h, 2, _p(data["x0"]), None, None, None, None, 1, 0.01, _p(z), None, None, None, None)
    assert rc == 0
    U = np.empty((2, 50)); u0 = np.empty((2, 5))
    assert lib.fmpc_unpack(h, 2, _p(z), _p(U), None, _p(u0)) == 0
    assert np.array_equal(u0, z[:, :5])
    assert lib.fmpc_solve(h, 2, None, None, None, None, None, 1, 0.01, _p(z), None, None, None, None) == pkg.FMPC_E_NULL
    assert lib.fmpc_solve(h, 0, _p(data["x0"]), None, None, None, None, 1, 0.01, _p(z), None, None, None, None) == 0
    assert lib.fmpc_destroy(h) == 0


def test_status_codes_on_device(pkg, gpu):
    md, data = pkg.synthetic.make_test_problem(8, 5, 10, seed=33, batch=3)
    h = handle_from_model(pkg, md)
    x0 = data["x0"].copy(); x0[1, 2] = np.nan                      # a NaN state: chol(Schur) fails for that problem only
    z, info = h.solve(x0, data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=2, k=0.01, return_info=True, check=False)
    assert info["status"][1] < 0 and info["status"][0] == 0 and info["status"][2] == 0
    assert info["rc"] == info["status"][1]                         # worst status is returned
    with pytest.raises(pkg.FastMPCError):
        h.solve(x0, data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=2, k=0.01)
    h.close()
    md2 = pkg.synthetic.make_model(27, 144, 10); md2["u_min"] = -0.05 * np.ones(144); md2["u_max"] = 0.05 * np.ones(144)
    d2 = pkg.synthetic.make_replay_batch(md2, r=5, steps=2)
    h2 = handle_from_model(pkg, md2)
    z, info = h2.solve(d2["x0"], d2["x0_pre"], None, nu0=d2["nu0"], n_newton=6, k=1e-2, return_info=True)
    assert info["status"][1] == pkg.FMPC_W_LINESEARCH and info["rc"] == pkg.FMPC_W_LINESEARCH
    h2.close()


def test_solve_once_reference_signature(pkg, gpu):
    """fmpc_solve_once: the reference's 23 constructor arguments + (nw, k) in one call."""
    lib = pkg.load()
    md, data = pkg.synthetic.make_test_problem(8, 5, 10, seed=35, xf=True)
    F = lambda M: np.asfortranarray(M).ravel(order="K").copy()
    z = np.empty(130); it = C.c_int()
    rc = lib.fmpc_solve_once(8, 5, 10, 2, _p(F(md["Q"])), _p(F(md["R"])), None, _p(F(md["Qf"])), None, None, None,
                             _p(md["x_min"]), _p(md["x_max"]), _p(md["u_min"]), _p(md["u_max"]), None, None,
                             _p(data["x0"][0]), _p(data["x0_pre"][0]), _p(np.zeros(5)), _p(F(md["A1"])), _p(F(md["A2"])),
                             _p(F(md["B"])), _p(data["w"][0]), _p(md["xf"]), None, _p(data["nu0"][0]), 5, 0.01, 0,
                             _p(z), C.byref(it))
    assert rc == 0
    zo, _, ito, _, _ = oracle_batch(md, {k: v[:1] for k, v in data.items()}, 5, 0.01)
    assert it.value == ito[0] and rel_err(z, zo[0]) <= TOL


def test_solve_once_caches_the_handle_per_model(pkg, gpu):
    """The notebook rebuilds the object at every timestep (README.md:548) and the MATLAB shim therefore calls
    fmpc_solve_once with the full argument set every time: the second call with an unchanged model must reuse the device
    handle (bit-identical result, no re-allocation / re-factorisation), a changed model must not."""
    import time
    lib = pkg.load()
    lib.fmpc_solve_once_cache_clear()
    md = pkg.synthetic.make_model(27, 144, 30)
    d = pkg.synthetic.make_replay_batch(md, r=7, steps=3)
    F = lambda M: np.asfortranarray(M).ravel(order="K").copy()
    args = lambda B, k: (27, 144, 30, 2, _p(F(md["Q"])), _p(F(md["R"])), None, _p(F(md["Qf"])), None, None, None,
                         _p(md["x_min"]), _p(md["x_max"]), _p(md["u_min"]), _p(md["u_max"]), None, None,
                         _p(d["x0"][k]), _p(d["x0_pre"][k]), _p(np.zeros(144)), _p(F(md["A1"])), _p(F(md["A2"])),
                         _p(F(B)), None, None, None, _p(d["nu0"][k]), 1, 1e-2, 0)

    def call(B, k):
        z = np.empty(30 * 171); it = C.c_int()
        t0 = time.perf_counter()
        rc = lib.fmpc_solve_once(*args(B, k), _p(z), C.byref(it))
        return rc, z, time.perf_counter() - t0
    rc, z1, t_first = call(md["B"], 0)
    assert rc == 0
    times = []
    for _ in range(5):
        rc, z2, t = call(md["B"], 0)
        assert rc == 0 and np.array_equal(z1, z2)
        times.append(t)
    assert min(times) * 10 <= t_first, (t_first, times)          # no create / upload / factorisation any more
    zo, *_ = oracle_batch(md, {k: (None if v is None else v[:2]) for k, v in d.items()}, 1, 1e-2)
    assert rel_err(z1, zo[0]) <= TOL
    rc, z3, _ = call(md["B"], 1)                                  # same model, next timestep
    assert rc == 0 and rel_err(z3, zo[1]) <= TOL
    B2 = md["B"].copy(); B2[3, 5] += 0.01                         # another model: must not hit the cached handle
    rc, z4, _ = call(B2, 0)
    md2 = dict(md); md2["B"] = B2
    zo2, *_ = oracle_batch(md2, {k: (None if v is None else v[:1]) for k, v in d.items()}, 1, 1e-2)
    assert rc == 0 and rel_err(z4, zo2[0]) <= TOL and not np.array_equal(z4, z1)
    for i in range(5):                                            # more models than cache slots: eviction keeps working
        Bi = md["B"].copy(); Bi[0, 0] += 0.001 * (i + 1)
        assert call(Bi, 0)[0] == 0
    rc, z5, _ = call(md["B"], 0)
    assert rc == 0 and np.array_equal(z5, z1)
    assert lib.fmpc_solve_once_cache_clear() == 0


@pytest.mark.parametrize("xf", [False, True])
def test_fast_mpc2_drivers_match_dense_oracle_drivers(pkg, gpu, xf):
    """Fast_MPC2.m:88-144: every driver, same nu0 sequence on both sides."""
    md, data = pkg.synthetic.make_test_problem(8, 5, 10, seed=37, xf=xf)
    x0, x0p, w = data["x0"][0], data["x0_pre"][0], data["w"][0]
    mk = lambda: pkg.Fast_MPC2(md["Q"], md["R"], [], md["Qf"], [], [], [], md["x_min"], md["x_max"], md["u_min"],
                               md["u_max"], [], [], 10, x0, x0p, np.zeros(5), md["A1"], md["A2"], md["B"], w,
                               md["xf"] if xf else [], [])
    d = dense_from_model(md, x0, x0p, w)
    rng = np.random.default_rng(3)
    nu0s = rng.random((8, (10 + xf) * 8))
    assert rel_err(mk().mpc_fixed_log_newton(5, 0.01, nu0=nu0s[0]), d.mpc_fixed_log_newton(5, 0.01, nu0=nu0s[0])) <= TOL
    assert rel_err(mk().mpc_fixed_log(0.01, nu0=nu0s[1]), d.mpc_fixed_log(0.01, nu0=nu0s[1])) <= TOL
    a = mk(); za = a.mpc_fixed_newton(5, nu0s=nu0s); infos = []
    zd = d.mpc_fixed_newton(5, nu0s=nu0s, infos=infos)
    assert len(a.last_info) == len(infos) == 5 and rel_err(za, zd) <= 1e-8
    assert [int(i["iters"][0]) for i in a.last_info] == [i["iters"] for i in infos]
    assert rel_err(mk().mpc_solve_full(nu0s=nu0s), d.mpc_solve_full(nu0s=nu0s)) <= 1e-8
    assert rel_err(mk().mpc_solve_check(1e-3, 1.0, nu0s=nu0s), d.mpc_solve_check(1e-3, 1.0, nu0s=nu0s)) <= 1e-8
    # the notebook rebuilds the object every timestep (README.md:548): the device handle is reused
    from importlib import import_module
    cache = import_module("mpc-sensorlessao_amd.fast_mpc2")._HANDLE_CACHE
    before = len(cache); mk().mpc_fixed_log_newton(1, 0.01, nu0=nu0s[0]); assert len(cache) == before
    # caller-side unpack, README.md:558-570
    U, X = pkg.deinterleave(za, 8, 5, 10)
    assert U.shape == (50,) and X.shape == (80,)


def test_var1_without_ramp(pkg, gpu):
    md, data = pkg.synthetic.make_test_problem(8, 5, 10, seed=39, var_order=1)
    v1 = pkg.Fast_MPC2_VAR1(md["Q"], md["R"], [], md["Qf"], [], [], [], md["x_min"], md["x_max"], md["u_min"], md["u_max"],
                            -np.ones(5), np.ones(5), 10, data["x0"][0], np.zeros(5), md["A1"], md["B"], data["w"][0], [], [],
                            ramp=False)
    z = v1.mpc_fixed_log_newton(5, 0.01, nu0=data["nu0"][0])
    d = dense_from_model(md, data["x0"][0], None, data["w"][0])
    assert rel_err(z, d.mpc_fixed_log_newton(5, 0.01, nu0=data["nu0"][0])) <= TOL


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_golden_fixtures_on_gpu(pkg, gpu, path):
    md, data, nw, k, z_init, f = load_case(path)
    h = handle_from_model(pkg, md)
    z, info = h.solve(data["x0"], data["x0_pre"], data["w"], z_init=z_init, nu0=data["nu0"], n_newton=nw, k=k,
                      return_info=True, check=False)
    h.close()
    assert np.array_equal(info["iters"], f["iters"])
    for p in range(z.shape[0]):
        it = int(f["iters"][p])
        assert np.array_equal(canon_steps(info["step"][p][:it]), canon_steps(f["steps"][p][:it]))
        assert rel_err(z[p], f["z"][p]) <= TOL, (p, rel_err(z[p], f["z"][p]))
        assert rel_err(info["nu"][p], f["nu"][p]) <= 1e-7


def test_generic_kernel_matches_wave_kernel_for_n27(pkg, gpu):
    """Both device kernels serve n = 27; FMPC_FORCE_GENERIC=1 selects the generic one at create time."""
    md = pkg.synthetic.make_model(27, 144, 10)
    data = pkg.synthetic.make_replay_batch(md, r=7, steps=12)
    hw = handle_from_model(pkg, md)
    os.environ["FMPC_FORCE_GENERIC"] = "1"
    try:
        hg = handle_from_model(pkg, md)
    finally:
        del os.environ["FMPC_FORCE_GENERIC"]
    zw, iw = hw.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=3, k=1e-2, return_info=True)
    zg, ig = hg.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=3, k=1e-2, return_info=True)
    hw.close(); hg.close()
    assert np.array_equal(iw["iters"], ig["iters"])
    assert max(rel_err(zw[p], zg[p]) for p in range(12)) <= 1e-10


def test_device_pointer_entry_point(pkg, gpu):
    import torch
    md = pkg.synthetic.make_model(27, 144, 10)
    data = pkg.synthetic.make_replay_batch(md, r=8, steps=10)
    h = handle_from_model(pkg, md)
    zh, info = h.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=2, k=1e-2, return_info=True)
    t = lambda a: torch.from_numpy(a).to(gpu)
    z, st, it = h.solve_device(t(data["x0"]), t(data["x0_pre"]), None, None, t(data["nu0"]), 2, 1e-2)
    U, X, u0 = h.unpack_device(z, torch.empty((10, 1440), dtype=torch.float64, device=gpu),
                               torch.empty((10, 270), dtype=torch.float64, device=gpu))
    torch.cuda.synchronize()
    assert np.array_equal(z.cpu().numpy(), zh) and np.array_equal(it.cpu().numpy(), info["iters"])
    assert np.array_equal(u0.cpu().numpy(), zh[:, :144])
    Uh, Xh, _ = h.unpack(zh)
    assert np.array_equal(U.cpu().numpy(), Uh) and np.array_equal(X.cpu().numpy(), Xh)
    with pytest.raises(pkg.FastMPCError):
        h.solve_device(t(data["x0"]).float(), None, None, None, None, 1, 1e-2)     # wrong dtype is refused
    h.close()


def test_full_size_properties_batch_2000(pkg, gpu):
    """BASELINE configs[1] size (n=27, m=144, T=30, 2000 timesteps): size-independent properties.
    (a) bitwise determinism, (b) a problem's result does not depend on the batch around it,
    (c) after a full step (t = 1) the dynamics hold: C z = b to round-off, (d) a random subset
    agrees with the oracle."""
    md = pkg.synthetic.make_model(27, 144, 30)
    data = pkg.synthetic.make_replay_batch(md, r=0, steps=2000)
    h = handle_from_model(pkg, md)
    z1, i1 = h.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=5, k=1e-2, return_info=True)
    z2, i2 = h.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=5, k=1e-2, return_info=True)
    assert np.array_equal(z1, z2) and np.array_equal(i1["iters"], i2["iters"])
    assert (i1["status"] == 0).all() and i1["iters"].min() >= 1 and i1["iters"].max() <= 5
    sel = np.random.default_rng(0).choice(2000, 16, replace=False)
    sub = {k: (None if v is None else v[sel]) for k, v in data.items()}
    zs, _ = h.solve(sub["x0"], sub["x0_pre"], None, nu0=sub["nu0"], n_newton=5, k=1e-2, return_info=True)
    assert np.array_equal(zs, z1[sel])
    Z = z1.reshape(2000, 30, 171); U, X = Z[:, :, :144], Z[:, :, 144:]
    A1, A2, B = md["A1"], md["A2"], md["B"]
    r0 = X[:, 0] - U[:, 0] @ B.T - data["x0"] @ A1.T - data["x0_pre"] @ A2.T
    r1 = X[:, 1] - U[:, 1] @ B.T - X[:, 0] @ A1.T - data["x0"] @ A2.T
    r2 = X[:, 2:] - U[:, 2:] @ B.T - X[:, 1:-1] @ A1.T - X[:, :-2] @ A2.T
    scale = np.abs(X).max()
    assert max(np.abs(r0).max(), np.abs(r1).max(), np.abs(r2).max()) <= 1e-11 * max(scale, 1.0)
    zo, _, ito, _, _ = oracle_batch(md, sub, 5, 1e-2)
    assert np.array_equal(i1["iters"][sel], ito)
    assert max(rel_err(zs[p], zo[p]) for p in range(16)) <= TOL
    h.close()


def test_shared_cold_start_factor_equals_per_problem_factor(pkg, gpu):
    """Cold start: the first Newton step of every problem uses ONE factor kept by the handle (it
    depends on the model and k only).  FMPC_NO_SHARED=1 at create time forces the per-problem path;
    both must agree to round-off, for several k on the same handle, and for later iterations."""
    md = pkg.synthetic.make_model(27, 144, 30)
    md["u_min"] = -0.5 * np.ones(144); md["u_max"] = 0.5 * np.ones(144)      # some problems need > 1 step
    data = pkg.synthetic.make_replay_batch(md, r=9, steps=40)
    hs = handle_from_model(pkg, md)
    os.environ["FMPC_NO_SHARED"] = "1"
    try:
        hp = handle_from_model(pkg, md)
    finally:
        del os.environ["FMPC_NO_SHARED"]
    for k, nw in [(1e-2, 1), (1e-1, 4), (1e-2, 4)]:
        zs, i_s = hs.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=nw, k=k, return_info=True)
        zp, i_p = hp.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=nw, k=k, return_info=True)
        assert np.array_equal(i_s["iters"], i_p["iters"]) and np.array_equal(i_s["status"], i_p["status"])
        assert np.array_equal(canon_steps(i_s["step"]), canon_steps(i_p["step"]))
        assert max(rel_err(zs[p], zp[p]) for p in range(40)) <= 1e-11
    zo, _, ito, _, _ = oracle_batch(md, data, 4, 1e-2)
    assert np.array_equal(i_s["iters"], ito) and max(rel_err(zs[p], zo[p]) for p in range(40)) <= TOL
    hs.close(); hp.close()


@pytest.mark.parametrize("case", ["panel", "panel_budget3", "handed_over", "warm_start", "warm_start_wave", "no_shared", "generic_n8"])
def test_solve_with_first_move_output(pkg, gpu, case):
    """fmpc_solve_u0_device: u0_out == z[:, :m] on every device path (the fused write of the n = 27 kernels, the
    unpack launch elsewhere), and z itself is what fmpc_solve_device gives."""
    import torch
    dev = torch.device("cuda:0")
    n, m, T = (8, 5, 6) if case == "generic_n8" else (27, 144, 10)
    if case == "generic_n8":
        md, data = pkg.synthetic.make_test_problem(n, m, T, seed=3, batch=40)
    else:
        md = pkg.synthetic.make_model(n, m, T)
        if case == "handed_over":
            md["u_min"] = -0.05 * np.ones(m); md["u_max"] = 0.05 * np.ones(m)
        data = pkg.synthetic.make_replay_batch(md, r=4, steps=40)
    if case == "no_shared":
        os.environ["FMPC_NO_SHARED"] = "1"
    if case in ("no_shared", "warm_start_wave"):
        os.environ["FMPC_NO_SMALL_TILED"] = "1"          # (few problems that factor their own Y go to the tiled kernel otherwise)
    try:
        h = handle_from_model(pkg, md)
    finally:
        os.environ.pop("FMPC_NO_SHARED", None); os.environ.pop("FMPC_NO_SMALL_TILED", None)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x0, x0p, w, nu0 = t(data["x0"]), t(data["x0_pre"]), t(data.get("w")), t(data["nu0"])
    nw = 3 if case == "panel_budget3" else 1
    z_init = None
    if case in ("warm_start", "warm_start_wave"):
        z_init = h.solve_device(x0, x0p, w, None, nu0, 1, 1e-2)[0].clone()
    z_ref, st_ref, it_ref = h.solve_device(x0, x0p, w, z_init, nu0, nw, 1e-2)
    z_ref, st_ref, it_ref = z_ref.clone(), st_ref.clone(), it_ref.clone()
    u0 = torch.full((40, m), float("nan"), dtype=torch.float64, device=dev)
    z, st, it = h.solve_device(x0, x0p, w, z_init, nu0, nw, 1e-2, u0_out=u0)
    torch.cuda.synchronize()
    path, handed = h.last_dispatch()
    want = {"panel": pkg.FMPC_PATH_PANEL, "panel_budget3": pkg.FMPC_PATH_PANEL, "handed_over": pkg.FMPC_PATH_PANEL,
            "warm_start": pkg._lib.FMPC_PATH_TILED, "warm_start_wave": pkg.FMPC_PATH_WAVE, "no_shared": pkg.FMPC_PATH_WAVE,
            "generic_n8": pkg._lib.FMPC_PATH_TILED}[case]          # (n = 8: the tiled kernel is the default wherever it exists, round 5)
    assert path == want and (handed > 0) == (case == "handed_over")
    assert torch.equal(z, z_ref) and torch.equal(st, st_ref) and torch.equal(it, it_ref)
    assert torch.equal(u0, z[:, :m])
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["panel", "panel_sweeps_w", "panel_budget3", "handed_over", "warm_start", "warm_start_wave", "no_shared",
                                  "generic_n8", "tiled_f32_n65", "panel_ragged"])
def test_first_moves_only_output(pkg, gpu, case):
    """Output options of the caller (README.md:558-570,589 applies U(1:nu) only): z_out = NULL with u0_out given.  The first
    moves must be BITWISE what the full solve leaves in z[:, :m], with identical status / iters, on every device path --
    the cold-start panel path then writes nothing of z (fmpc_cold_dz<.., .., true>), the others iterate in a scratch array."""
    import torch
    dev = torch.device("cuda:0")
    B = 37 if case == "panel_ragged" else 40
    n, m, T = (8, 5, 6) if case == "generic_n8" else (65, 144, 6) if case == "tiled_f32_n65" else (27, 144, 10)
    if case == "generic_n8":
        md, data = pkg.synthetic.make_test_problem(n, m, T, seed=3, batch=B)
    else:
        md = pkg.synthetic.make_model(n, m, T)
        if case == "handed_over":
            md["u_min"] = -0.05 * np.ones(m); md["u_max"] = 0.05 * np.ones(m)
        data = pkg.synthetic.make_replay_batch(md, r=4, steps=B)
        if case == "panel_sweeps_w":
            data["w"] = 0.01 * np.random.default_rng(5).standard_normal((B, T * n))
    if case == "no_shared":
        os.environ["FMPC_NO_SHARED"] = "1"
    if case in ("no_shared", "warm_start_wave"):
        os.environ["FMPC_NO_SMALL_TILED"] = "1"
    if case == "panel_sweeps_w":
        os.environ["FMPC_NO_INV"] = "1"
    try:
        h = handle_from_model(pkg, md)
    finally:
        for v in ("FMPC_NO_SHARED", "FMPC_NO_SMALL_TILED", "FMPC_NO_INV"):
            os.environ.pop(v, None)
    if case == "tiled_f32_n65":
        h.set_precision("f32")
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x0, x0p, w, nu0 = t(data["x0"]), t(data["x0_pre"]), t(data.get("w")), t(data["nu0"])
    nw = 3 if case == "panel_budget3" else 2 if case == "tiled_f32_n65" else 1
    z_init = None
    if case in ("warm_start", "warm_start_wave"):
        z_init = h.solve_device(x0, x0p, w, None, nu0, 1, 1e-2)[0].clone()
    z_ref, st_ref, it_ref = h.solve_device(x0, x0p, w, z_init, nu0, nw, 1e-2)
    z_ref, st_ref, it_ref = z_ref.clone(), st_ref.clone(), it_ref.clone()
    path_ref = h.last_dispatch()
    u0 = torch.full((B, m), float("nan"), dtype=torch.float64, device=dev)
    z, st, it = h.solve_device(x0, x0p, w, z_init, nu0, nw, 1e-2, u0_out=u0, want_z=False)
    torch.cuda.synchronize()
    assert z is None
    assert h.last_dispatch() == path_ref
    assert (path_ref[1] > 0) == (case == "handed_over")
    assert torch.equal(st, st_ref) and torch.equal(it, it_ref)
    assert torch.equal(u0, z_ref[:, :m])
    # raw C ABI: z_out == NULL without u0_out is refused
    import ctypes as C
    rc = pkg.load().fmpc_solve_device(h._h, B, C.c_void_p(x0.data_ptr()), C.c_void_p(x0p.data_ptr()), None, None, None, 1, 1e-2,
                                      None, None, None, None, None, None)
    assert rc == pkg._lib.FMPC_E_NULL
    h.close()


@pytest.mark.gpu
def test_host_pointer_entry_pinned_and_direct_staging_and_output_reuse(pkg, gpu):
    """fmpc_solve with host pointers: up to 1 MB of staging the inputs / outputs travel through the handle's pinned twin block in
    one asynchronous copy each way (the literal per-timestep call, README.md:548-556), beyond that straight from / to the
    caller's arrays.  Both must give what the device-pointer entry gives, with every combination of optional inputs and
    outputs (absent inputs take no room in the block), and z_out= must write into the caller's array."""
    import torch
    md = pkg.synthetic.make_model(27, 144, 6)
    h = handle_from_model(pkg, md)
    rng = np.random.default_rng(3)
    for B in (1, 3, 40):                               # 40 problems x 8.2 KB of z + nu ... > 1 MB? no: (27+144)*6*8 = 8.2 KB -> pinned; see below
        d = pkg.synthetic.make_replay_batch(md, r=7, steps=B)
        w = 0.01 * rng.standard_normal((B, 6 * 27))
        zi = np.tile(np.concatenate([np.zeros(144), np.zeros(27)]), (B, 6)) + 0.1 * rng.standard_normal((B, 6 * 171))
        for kw in (dict(), dict(w=w), dict(z_init=zi), dict(w=w, z_init=zi, nu0=None)):
            nu0 = kw.pop("nu0", d["nu0"])
            z, info = h.solve(d["x0"], d["x0_pre"], kw.get("w"), z_init=kw.get("z_init"), nu0=nu0, n_newton=2, k=1e-2, return_info=True)
            t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
            nu_d = torch.empty((B, h.nu_len), dtype=torch.float64, device=gpu)
            zd, st, it = h.solve_device(t(d["x0"]), t(d["x0_pre"]), t(kw.get("w")), t(kw.get("z_init")), t(nu0), 2, 1e-2, nu_out=nu_d)
            torch.cuda.synchronize()
            assert np.array_equal(z, zd.cpu().numpy()) and np.array_equal(info["nu"], nu_d.cpu().numpy())
            assert np.array_equal(info["iters"], it.cpu().numpy()) and np.array_equal(info["status"], st.cpu().numpy())
            keep = np.full_like(z, np.nan)
            z2 = h.solve(d["x0"], d["x0_pre"], kw.get("w"), z_init=kw.get("z_init"), nu0=nu0, n_newton=2, k=1e-2, z_out=keep)
            assert z2 is keep and np.array_equal(keep, z)
    h.close()
    # beyond the pinned block: T = 30, 40 problems = 1.6 MB of z
    md = pkg.synthetic.make_model(27, 144, 30)
    h = handle_from_model(pkg, md)
    d = pkg.synthetic.make_replay_batch(md, r=8, steps=40)
    z, info = h.solve(d["x0"], d["x0_pre"], None, nu0=d["nu0"], n_newton=1, k=1e-2, return_info=True)
    z1 = np.stack([h.solve(d["x0"][p], d["x0_pre"][p], None, nu0=d["nu0"][p], n_newton=1, k=1e-2) for p in range(40)])   # pinned, one by one
    assert np.array_equal(z, z1)
    with pytest.raises(pkg.FastMPCError):
        h.solve(d["x0"], d["x0_pre"], None, nu0=d["nu0"], z_out=np.empty((39, h.nz)))
    h.close()


def test_host_pointer_first_move_entry(pkg, gpu):
    """fmpc_solve_u0 (README.md:558-570,589: the caller applies U(1:nu) only): host pointers in, the first moves
    out -- equal bit for bit to z[:, :m] of fmpc_solve on every path (cold start / affine form, Newton budget, explicit start, w,
    pinned and direct staging), status / iters as fmpc_solve reports them, z_out optional beside u0_out, raw ctypes with NULLs."""
    lib = pkg.load()
    for (T, B) in ((6, 3), (30, 1), (30, 300)):
        md = pkg.synthetic.make_model(27, 144, T)
        h = handle_from_model(pkg, md)
        d = pkg.synthetic.make_replay_batch(md, r=11, steps=B)
        rng = np.random.default_rng(T)
        w = 0.01 * rng.standard_normal((B, T * 27))
        zi = 0.1 * rng.standard_normal((B, T * 171))
        for kw in (dict(n_newton=1), dict(n_newton=3), dict(n_newton=1, w=w), dict(n_newton=2, z_init=zi), dict(n_newton=1, nu0=None)):
            nu0 = kw.get("nu0", d["nu0"])
            z, info = h.solve(d["x0"], d["x0_pre"], kw.get("w"), z_init=kw.get("z_init"), nu0=nu0, n_newton=kw["n_newton"], k=1e-2, return_info=True)
            u0, iu = h.solve_u0(d["x0"], d["x0_pre"], kw.get("w"), z_init=kw.get("z_init"), nu0=nu0, n_newton=kw["n_newton"], k=1e-2, return_info=True)
            assert u0.shape == (B, 144) and np.array_equal(u0, z[:, :144]), kw.keys()
            assert np.array_equal(iu["iters"], info["iters"]) and np.array_equal(iu["status"], info["status"]) and iu["rc"] == info["rc"]
        # raw C ABI: z_out beside u0_out, NULL status / iters, NULL u0_out refused
        zr = np.empty((B, T * 171)); ur = np.full((B, 144), np.nan)
        assert lib.fmpc_solve_u0(h._h, B, _p(d["x0"]), _p(d["x0_pre"]), None, None, _p(d["nu0"]), 1, 1e-2, _p(zr), _p(ur), None, None) == 0
        z1 = h.solve(d["x0"], d["x0_pre"], None, nu0=d["nu0"], n_newton=1, k=1e-2)
        assert np.array_equal(zr, z1) and np.array_equal(ur, z1[:, :144])
        assert lib.fmpc_solve_u0(h._h, B, _p(d["x0"]), _p(d["x0_pre"]), None, None, None, 1, 1e-2, _p(zr), None, None, None) == pkg.FMPC_E_NULL
        assert lib.fmpc_solve_u0(h._h, -1, _p(d["x0"]), None, None, None, None, 1, 1e-2, None, _p(ur), None, None) == pkg.FMPC_E_DIM
        h.close()


def test_small_host_calls_work_on_the_pinned_block(pkg, gpu, monkeypatch):
    """A host-pointer call of a few problems skips both copies (the kernels read the inputs from, and -- cold start, budget 1 --
    write the outputs into, the pinned block itself).  Same numbers, bit for bit, as the same call with the copies
    (FMPC_NO_ZEROCOPY=1, read once per process: a child process) and as the device entry -- also when problems are flagged and
    the exact path then uses z in host memory as its working iterate, with w, with a budget, from an explicit start."""
    import subprocess, sys, os, tempfile
    md = pkg.synthetic.make_model(27, 144, 30)
    md["u_min"] = -0.1 * np.ones(144); md["u_max"] = 0.1 * np.ones(144)          # tight bounds: some decisions are not clear-cut
    d = pkg.synthetic.make_replay_batch(md, r=20, steps=2)
    x0 = d["x0"] * np.array([[0.05], [5.0]]); x0p = d["x0_pre"] * np.array([[0.05], [5.0]])
    rng = np.random.default_rng(1)
    w = 0.01 * rng.standard_normal((2, 30 * 27)); zi = 0.02 * rng.standard_normal((2, 30 * 171))
    cases = (dict(n_newton=1), dict(n_newton=1, w=w), dict(n_newton=3), dict(n_newton=2, z_init=zi))
    h = handle_from_model(pkg, md)
    got = []
    for kw in cases:
        z, info = h.solve(x0, x0p, kw.get("w"), z_init=kw.get("z_init"), nu0=d["nu0"], n_newton=kw["n_newton"], k=1e-2, return_info=True, check=False)
        u0 = h.solve_u0(x0, x0p, kw.get("w"), z_init=kw.get("z_init"), nu0=d["nu0"], n_newton=kw["n_newton"], k=1e-2, check=False)
        assert np.array_equal(u0, z[:, :144])
        got.append((z, info["nu"], info["status"], info["iters"], info["step"]))
    assert h.last_dispatch()[0] is not None
    z1, i1 = h.solve(x0, x0p, None, nu0=d["nu0"], n_newton=1, k=1e-2, return_info=True, check=False)
    assert h.last_dispatch()[1] > 0, "no problem was flagged: the case does not exercise the exact path on host memory"
    h.close()
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), x0=x0, x0p=x0p, w=w, zi=zi, nu0=d["nu0"])
        code = (
            "import importlib, sys, numpy as np\n"
            "sys.path.insert(0, %r)\n"
            "pkg = importlib.import_module('mpc-sensorlessao_amd')\n"
            "from tests.util import handle_from_model\n"
            "md = pkg.synthetic.make_model(27, 144, 30); md['u_min'] = -0.1 * np.ones(144); md['u_max'] = 0.1 * np.ones(144)\n"
            "a = np.load(%r); h = handle_from_model(pkg, md); out = {}\n"
            "cases = (dict(n_newton=1), dict(n_newton=1, w=a['w']), dict(n_newton=3), dict(n_newton=2, z_init=a['zi']))\n"
            "for i, kw in enumerate(cases):\n"
            "    z, info = h.solve(a['x0'], a['x0p'], kw.get('w'), z_init=kw.get('z_init'), nu0=a['nu0'], n_newton=kw['n_newton'], k=1e-2, return_info=True, check=False)\n"
            "    out['z%%d' %% i] = z; out['nu%%d' %% i] = info['nu']; out['st%%d' %% i] = info['status']; out['it%%d' %% i] = info['iters']; out['sp%%d' %% i] = info['step']\n"
            "np.savez(%r, **out)\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(td, "in.npz"), os.path.join(td, "out.npz"))
        env = dict(os.environ, FMPC_NO_ZEROCOPY="1")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        ref = np.load(os.path.join(td, "out.npz"))
        for i, (z, nu, st, it, sp) in enumerate(got):
            assert np.array_equal(z, ref["z%d" % i]) and np.array_equal(nu, ref["nu%d" % i]), i
            assert np.array_equal(st, ref["st%d" % i]) and np.array_equal(it, ref["it%d" % i]) and np.array_equal(sp, ref["sp%d" % i], equal_nan=True), i
