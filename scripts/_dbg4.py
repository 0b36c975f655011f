import importlib, os, sys
sys.path.insert(0, '.')
os.environ["FMPC_TILED"] = "1"
import numpy as np
pkg = importlib.import_module("mpc-sensorlessao_amd")
from tests.util import handle_from_model, oracle_batch, rel_err
nfail = {}
for T in (2, 10):
    model = pkg.synthetic.make_model(27, 144, T)
    data = pkg.synthetic.make_replay_batch(model, r=1, steps=40)
    zo, nuo, ito, sto, steps = oracle_batch(model, data, 1, 1e-2)
    for rep in range(6):
        for nw in ("2", "4"):
            os.environ["FMPC_TILED_NW"] = nw
            h = handle_from_model(pkg, model)
            z, info = h.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=1, k=1e-2, return_info=True, check=False)
            h.close()
            bad = [p for p in range(40) if rel_err(z[p], zo[p]) > 1e-9]
            nfail[(T, nw)] = nfail.get((T, nw), 0) + len(bad)
print("failed problems (of 240 each):", nfail)
