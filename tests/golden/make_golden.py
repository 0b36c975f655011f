"""Generates tests/golden/*.npz from the dense oracle (oracle/dense_ref.py).

The reference is MATLAB-only and cannot run here, and its own test script holds no vectors
(SURVEY.md §4), so these fixtures are outputs of the op-for-op restatement, NOT of the reference:
PARITY UNPINNED.  They pin the restatement against regressions and give the GPU tests fixed data.
Run from the repo root:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.util import dense_from_model  # noqa: E402

pkg = importlib.import_module("mpc-sensorlessao_amd")
S = pkg.synthetic
OUT = os.path.dirname(os.path.abspath(__file__))


def run_case(name, model, data, nw, k, z_init=None):
    B = data["x0"].shape[0]
    nz = model["T"] * (model["n"] + model["m"])
    nb = model["T"] + (1 if model.get("xf") is not None else 0)
    z = np.empty((B, nz)); nu = np.empty((B, nb * model["n"]))
    iters = np.zeros(B, dtype=np.int64); steps = -np.ones((B, max(nw, 1) if nw else 64))
    for p in range(B):
        w = data["w"][p] if data.get("w") is not None else np.zeros(model["T"] * model["n"])
        d = dense_from_model(model, data["x0"][p], data["x0_pre"][p], w,
                             x_init=None if z_init is None else z_init[p])
        info = {}
        if nw:
            z[p] = d.mpc_fixed_log_newton(nw, k, nu0=data["nu0"][p], info=info)
        else:
            z[p] = d.mpc_fixed_log(k, nu0=data["nu0"][p], info=info)
        nu[p] = info["nu"]; iters[p] = info["iters"]
        t = np.array(info.get("t", []))
        steps[p, :len(t)] = t
    arrays = {("model_" + kk): v for kk, v in model.items() if isinstance(v, np.ndarray)}
    arrays.update(x0=data["x0"], x0_pre=data["x0_pre"], nu0=data["nu0"], z=z, nu=nu, iters=iters, steps=steps,
                  meta=np.array([model["n"], model["m"], model["T"], model.get("var_order", 2), nw or 0,
                                 1 if model.get("xf") is not None else 0]), k=np.array([k]))
    if data.get("w") is not None:
        arrays["w"] = data["w"]
    if z_init is not None:
        arrays["z_init"] = z_init
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print(name, "iters", iters.tolist())


if __name__ == "__main__":
    m, d = S.make_test_problem(8, 5, 10, seed=21, xf=True, batch=3)
    run_case("demo_n8_m5_T10_xf_nw5", m, d, 5, 0.01)
    m, d = S.make_test_problem(8, 5, 10, seed=22, xf=False, var_order=1, batch=3)
    run_case("demo_var1_n8_m5_T10_nw5", m, d, 5, 0.01)
    m, d = S.make_test_problem(8, 5, 10, seed=23, umax=0.3, batch=3)
    rng = np.random.default_rng(5)
    z0 = np.zeros((3, 10, 13)); z0[:, :, :5] = rng.uniform(-0.25, 0.25, (3, 10, 5)); z0[:, :, 5:] = rng.uniform(-1, 1, (3, 10, 8))
    run_case("demo_backtracking_k10_nw8", m, d, 8, 10.0, z_init=z0.reshape(3, -1))
    m = S.make_model(27, 144, 2); d = S.make_replay_batch(m, r=4, steps=3)
    run_case("ao_n27_m144_T2_nw1", m, d, 1, 1e-2)
    m = S.make_model(27, 144, 10); m["u_min"] = -0.05 * np.ones(144); m["u_max"] = 0.05 * np.ones(144)
    d = S.make_replay_batch(m, r=5, steps=2)
    run_case("ao_tight_n27_m144_T10_nw6", m, d, 6, 1e-2)
    m = S.make_model(27, 144, 30); d = S.make_replay_batch(m, r=6, steps=1)
    run_case("ao_n27_m144_T30_nw5", m, d, 5, 1e-2)
