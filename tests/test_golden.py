"""Both oracles reproduce the committed fixtures (tests/golden/*.npz, written by make_golden.py from
the dense oracle).  The fixtures are NOT reference outputs (PARITY UNPINNED, see their generator)."""
import glob
import os

import numpy as np
import pytest

from tests.util import banded_from_model, canon_steps, dense_from_model, rel_err

FILES = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
               if os.path.basename(p) != "model_approx_As_bs.npz")      # (the reference's estimator model: tests/test_golden_model_approx.py)


def load_case(path):
    f = np.load(path)
    n, m, T, var_order, nw, has_xf = [int(v) for v in f["meta"]]
    md = dict(n=n, m=m, T=T, var_order=var_order)
    for key in f.files:
        if key.startswith("model_"):
            md[key[6:]] = f[key]
    data = dict(x0=f["x0"], x0_pre=f["x0_pre"], nu0=f["nu0"], w=f["w"] if "w" in f.files else None)
    return md, data, nw, float(f["k"][0]), (f["z_init"] if "z_init" in f.files else None), f


def test_fixtures_present():
    assert len(FILES) >= 6


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_banded_oracle_reproduces_fixture(path):
    md, data, nw, k, z_init, f = load_case(path)
    b = banded_from_model(md)
    for p in range(data["x0"].shape[0]):
        info = {}
        z, nu, it, st = b.solve(data["x0"][p], data["x0_pre"][p], None if data["w"] is None else data["w"][p],
                                nw, k, z_init=None if z_init is None else z_init[p], nu0=data["nu0"][p], info=info)
        assert it == f["iters"][p]
        assert np.array_equal(canon_steps(info.get("t", [])), canon_steps(f["steps"][p][:it]))
        assert rel_err(z, f["z"][p]) <= 1e-9 and rel_err(nu, f["nu"][p]) <= 1e-8


@pytest.mark.parametrize("path", [p for p in FILES if "T30" not in p], ids=lambda p: os.path.basename(p)[:-4])
def test_dense_oracle_reproduces_fixture(path):
    md, data, nw, k, z_init, f = load_case(path)
    p = 0
    w = data["w"][p] if data["w"] is not None else np.zeros(md["T"] * md["n"])
    d = dense_from_model(md, data["x0"][p], data["x0_pre"][p], w, x_init=None if z_init is None else z_init[p])
    z = d.mpc_fixed_log_newton(nw, k, nu0=data["nu0"][p])
    assert rel_err(z, f["z"][p]) <= 1e-12
