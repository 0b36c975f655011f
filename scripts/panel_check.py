"""Development check of the panel kernel (fmpc_kernel_panel.hip): panel path vs the exact
one-wave-per-problem path (FMPC_NO_PANEL=1) vs the structured oracle.  Run on the GPU box."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mpc-sensorlessao_amd")
from tests.util import handle_from_model, oracle_batch, rel_err  # noqa: E402


def run(xf, use_w, use_nu, batch, k, T=30, tight=False, seed=5):
    md = pkg.synthetic.make_model(27, 144, T)
    if tight:
        md["u_min"] = -0.05 * np.ones(144); md["u_max"] = 0.05 * np.ones(144)
    rng = np.random.default_rng(seed)
    if xf:
        md["xf"] = 0.01 * rng.standard_normal(27)
    data = pkg.synthetic.make_replay_batch(md, r=9, steps=batch)
    if use_w:
        data["w"] = 0.01 * rng.standard_normal((batch, T * 27))
    data["nu0"] = rng.standard_normal((batch, (T + (1 if xf else 0)) * 27)) if use_nu else None
    hp = handle_from_model(pkg, md)
    os.environ["FMPC_NO_PANEL"] = "1"
    try:
        hw = handle_from_model(pkg, md)
    finally:
        del os.environ["FMPC_NO_PANEL"]
    zp, ip = hp.solve(data["x0"], data["x0_pre"], data.get("w"), nu0=data["nu0"], n_newton=1, k=k, return_info=True, check=False)
    disp = hp.last_dispatch()
    zw, iw = hw.solve(data["x0"], data["x0_pre"], data.get("w"), nu0=data["nu0"], n_newton=1, k=k, return_info=True, check=False)
    nb = min(batch, 12)
    sub = {kk: (None if v is None else v[:nb]) for kk, v in data.items() if kk in ("x0", "x0_pre", "w", "nu0")}
    zo, nuo, ito, sto, steps = oracle_batch(md, sub, 1, k)
    ez = max(rel_err(zp[p], zw[p]) for p in range(batch))
    en = max(rel_err(ip["nu"][p], iw["nu"][p]) for p in range(batch))
    eo = max(rel_err(zp[p], zo[p]) for p in range(nb))
    eno = max(rel_err(ip["nu"][p], nuo[p]) for p in range(nb))
    ewo = max(rel_err(zw[p], zo[p]) for p in range(nb))
    same = (np.array_equal(ip["iters"], iw["iters"]) and np.array_equal(ip["status"], iw["status"])
            and np.array_equal(ip["step"], iw["step"]))
    print(f"xf={xf} w={use_w} nu={use_nu} B={batch} k={k} T={T} tight={tight}: panel-vs-wave z {ez:.2e} nu {en:.2e} | "
          f"panel-vs-oracle z {eo:.2e} nu {eno:.2e} (wave-vs-oracle {ewo:.2e}) | info equal {same} "
          f"dispatch {disp} iters {np.bincount(ip['iters'])} steps {np.unique(ip['step'])}", flush=True)
    hp.close(); hw.close()


if __name__ == "__main__":
    run(False, False, True, 37, 1e-2)
    run(True, True, True, 16, 1e-2)
    run(False, True, False, 5, 1e-2)
    run(True, False, False, 33, 1e-1, T=10)
    run(False, False, True, 40, 1e-2, tight=True)
    run(False, False, True, 40, 10.0)
    run(False, False, True, 2000, 1e-2)
