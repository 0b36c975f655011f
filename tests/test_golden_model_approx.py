"""The reference's own estimator model, `model_approx.mat` (A_s, b_s; loaded at README.md:294, used at README.md:478
`ad_est = lsqminnorm(A_s'*A_s, A_s'*(Y_M-b_s))`), as a committed fixture: tests/golden/model_approx_As_bs.npz, written by
tests/golden/read_mat73.py (a minimal HDF5 reader; the file is MATLAB v7.3).  It is the one piece of reference-held
numerical data on a SURVEY §8 row (f.4), so the estimator's LINEAR half is pinned to it here:
  * CPU: the fixture's shape is the README's (p = 3 x 31^2 window samples, 28 modes of which the piston column is removed at
    README.md:289); the checker oracle/estimator_ref.estimate agrees with two independent routes on it; the library's host
    builder of G (csrc/fmpc_host.cpp, built with g++ under ASan/UBSan) reproduces the minimum-norm solution, also when a
    column is repeated (the rank cut-off of lsqminnorm);
  * GPU: fmpc_est_create with the REAL A_s, b_s, ad_est against numpy on the device's own Y_M."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import estimator_ref as er

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpc-sensorlessao_amd", "csrc")
FIX = os.path.join(ROOT, "tests", "golden", "model_approx_As_bs.npz")


@pytest.fixture(scope="module")
def model():
    v = np.load(FIX)
    return v["A_s"], v["b_s"].reshape(-1)


def _no_piston(A_s):
    return np.ascontiguousarray(A_s[:, 1:])           # README.md:289  A_s(:,1) = [];


def test_fixture_has_the_readmes_shape(model):
    A_s, b_s = model
    d = 31                                             # range_max - range_min + 1 on the README's grid (oracle window_range)
    lo, hi = er.window_range(512, _dx())
    assert hi - lo + 1 == d
    assert A_s.shape == (3 * d * d, 28) and b_s.shape == (3 * d * d,)
    assert np.isfinite(A_s).all() and np.isfinite(b_s).all()
    assert np.linalg.matrix_rank(_no_piston(A_s)) == 27
    ev = np.linalg.eigvalsh(_no_piston(A_s).T @ _no_piston(A_s))
    assert ev[0] > 27 * np.finfo(float).eps * ev[-1]          # far above lsqminnorm's cut-off: G = inv(A'A) A'
    assert ev[-1] / ev[0] < 1e7


def _dx():
    import importlib
    pkg = importlib.import_module("mpc-sensorlessao_amd")
    return pkg.synthetic.estimator_optics(64)["dx"]


def test_fixture_can_be_regenerated_from_the_reference_when_it_is_present(model, tmp_path):
    src = "/root/reference/model_approx.mat"
    if not os.path.exists(src):
        pytest.skip("the reference is not on this box (the fixture is what travels)")
    import importlib.util
    spec = importlib.util.spec_from_file_location("read_mat73", os.path.join(ROOT, "tests", "golden", "read_mat73.py"))
    rd = importlib.util.module_from_spec(spec); spec.loader.exec_module(rd)
    v = rd.read_mat73(src)
    assert sorted(v) == ["A_s", "b_s"]
    assert np.array_equal(v["A_s"], model[0]) and np.array_equal(v["b_s"].reshape(-1), model[1])


def test_checker_on_the_real_model_against_independent_routes(model):
    A = _no_piston(model[0]); b = model[1]
    rng = np.random.default_rng(11)
    x_true = 0.3 * rng.standard_normal(27)
    Y = A @ x_true + b + 1e-3 * rng.standard_normal(A.shape[0])
    x = er.estimate(A, b, Y)
    x_qr = np.linalg.lstsq(A, Y - b, rcond=None)[0]                          # least squares on A_s itself (QR/SVD route)
    x_ch = np.linalg.solve(A.T @ A, A.T @ (Y - b))                           # normal equations by LU
    sc = np.linalg.norm(x)
    assert np.linalg.norm(x - x_qr) <= 1e-9 * sc and np.linalg.norm(x - x_ch) <= 1e-9 * sc
    assert np.linalg.norm(x - x_true) <= 5e-2 * np.linalg.norm(x_true)       # the model is informative: noise 1e-3 -> small error


@pytest.fixture(scope="module")
def gain_binary(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    out = str(tmp_path_factory.mktemp("est_gain") / "est_gain_main")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", CSRC,
           os.path.join(ROOT, "tests", "host_san", "est_gain_main.cpp"), os.path.join(CSRC, "fmpc_host.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def _host_gain(binary, A, tmp_path):
    p, nx = A.shape
    fa, fg = str(tmp_path / "A.bin"), str(tmp_path / "G.bin")
    np.asfortranarray(A).ravel(order="F").tofile(fa)
    r = subprocess.run([binary, fa, str(p), str(nx), fg], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "runtime error" not in r.stderr, r.stdout + r.stderr
    return int(r.stdout.strip()), np.fromfile(fg).reshape(nx, p)


def test_library_host_gain_on_the_real_model(model, gain_binary, tmp_path):
    A = _no_piston(model[0]); b = model[1]
    rank, G = _host_gain(gain_binary, A, tmp_path)
    assert rank == 27
    rng = np.random.default_rng(12)
    for _ in range(3):
        Y = A @ (0.5 * rng.standard_normal(27)) + b + 1e-2 * rng.standard_normal(A.shape[0])
        x = er.estimate(A, b, Y)
        assert np.linalg.norm(G @ (Y - b) - x) <= 1e-8 * np.linalg.norm(x)
    assert np.abs(G @ A - np.eye(27)).max() <= 1e-9


def test_library_host_gain_rank_cut_off_on_the_real_model(model, gain_binary, tmp_path):
    """lsqminnorm's minimum-norm solution when A_s'A_s is singular: the real model with the piston column kept AND one
    mode column repeated -- the two equal columns get equal coefficients, the rank drops by one."""
    A = np.concatenate([model[0], model[0][:, 5:6]], axis=1)                 # 29 columns, rank 28
    b = model[1]
    rank, G = _host_gain(gain_binary, A, tmp_path)
    assert rank == 28
    rng = np.random.default_rng(13)
    Y = model[0] @ (0.2 * rng.standard_normal(28)) + b
    x = G @ (Y - b)
    x_ref = np.linalg.pinv(A.T @ A, rcond=29 * np.finfo(float).eps, hermitian=True) @ (A.T @ (Y - b))
    assert np.linalg.norm(x - x_ref) <= 1e-7 * np.linalg.norm(x_ref)
    assert abs(x[5] - x[28]) <= 1e-9 * np.linalg.norm(x)


@pytest.mark.gpu
@pytest.mark.parametrize("length,rmin", [(64, 17), (512, None)])
def test_device_estimator_with_the_real_model(pkg, gpu, model, length, rmin):
    """fmpc_est_create with the reference's A_s (piston removed: nx = 27 pins the mode count, p = 3 x 31^2 pins d = 31) and b_s;
    ad_est against numpy's least squares on the Y_M the device formed (the PSF half keeps its own tests on synthetic optics)."""
    A = _no_piston(model[0]); b = model[1]
    op = pkg.synthetic.estimator_optics(length)
    if rmin is None:
        rmin, rmax = op["range_min"] + 1, op["range_max"] + 1               # the README's window on its own grid: 31 samples
    else:
        rmax = rmin + 30                                                    # a 31-sample window on a small grid (cheap case)
    assert rmax - rmin + 1 == 31
    est = pkg.PhaseDiversityEstimator(op["pupil"], op["W"], op["zd_list"], op["dx"], rmin, rmax, A, b)
    assert (est.nx, est.p, est.d, est.rank) == (27, 2883, 31, 27)
    rng = np.random.default_rng(21)
    scr = 0.25 * rng.standard_normal((3, length, length))
    noise = 1e-2 * rng.standard_normal((3, 2883))
    ad, Y = est.apply(scr, noise=noise, want_Y=True)
    for s in range(3):
        ref = np.linalg.lstsq(A.T @ A, A.T @ (Y[s] - b), rcond=None)[0]     # README.md:478
        assert np.linalg.norm(ad[s] - ref) <= 1e-8 * max(np.linalg.norm(ref), 1e-300), (s, np.linalg.norm(ad[s] - ref), np.linalg.norm(ref))
    est.close()
