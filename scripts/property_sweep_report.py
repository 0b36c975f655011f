"""Diagnostic: the GPU property sweep of tests/test_property_random.py case by case -- dispatch path, worst relative error."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tests.test_property_random import _gpu_cases, random_problem, random_interior_start
from tests.util import handle_from_model, oracle_batch, rel_err
pkg = importlib.import_module("mpc-sensorlessao_amd")
for i, c in enumerate(_gpu_cases()):
    model, data = random_problem(c["seed"], c["n"], c["m"], c["T"], c["var_order"], c["dense_q"], c["dense_r"], c["xf"], c["lin"], batch=c["batch"])
    zi = random_interior_start(c["seed"], model, c["batch"]) if c["warm"] else None
    try:
        h = handle_from_model(pkg, model)
    except pkg.FastMPCError as e:
        print(i, "create failed", e.code); continue
    z, info = h.solve(data["x0"], data["x0_pre"], data["w"], z_init=zi, nu0=data["nu0"], n_newton=c["nw"], k=c["k"], return_info=True, check=False)
    zo, nuo, ito, sto, steps = oracle_batch(model, data, c["nw"], c["k"], z_init=zi)
    errs = [rel_err(z[p], zo[p]) for p in range(c["batch"])]
    print(i, {k: c[k] for k in ("n", "m", "T", "var_order", "dense_q", "dense_r", "xf", "lin", "nw", "k", "warm")}, "path", h.last_dispatch(),
          "worst %.2e" % max(errs), "status", info["status"].tolist(), sto.tolist(), "iters", info["iters"].tolist(), ito.tolist(), "FAIL" if max(errs) > 1e-9 else "")
    h.close()
