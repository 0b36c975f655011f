import importlib, os, sys
sys.path.insert(0, '.')
os.environ["FMPC_TILED"] = "1"
import numpy as np
pkg = importlib.import_module("mpc-sensorlessao_amd")
from tests.util import handle_from_model, oracle_batch, rel_err
T = 2
np.set_printoptions(linewidth=220, precision=2)
model = pkg.synthetic.make_model(27, 144, T)
data = pkg.synthetic.make_replay_batch(model, r=1, steps=40)
zo, nuo, ito, sto, steps = oracle_batch(model, data, 1, 1e-2)
for nw in sys.argv[1:]:
    os.environ["FMPC_TILED_NW"] = nw
    h = handle_from_model(pkg, model)
    z, info = h.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=1, k=1e-2, return_info=True, check=False)
    h.close()
    bad = [p for p in range(40) if rel_err(z[p], zo[p]) > 1e-9]
    print("NW", nw, "status", info["status"].tolist(), "bad", bad)
    for p in bad[:2]:
        dn = np.abs(info["nu"][p] - nuo[p]).reshape(T, 27)
        dz = np.abs(z[p] - zo[p]).reshape(T, 171)
        print(" problem", p, "step", info["step"][p], "nu err rows:"); print((dn > 1e-9).astype(int)); print(" z err > 1e-9 count per stage", (dz > 1e-9).sum(axis=1), "max", dz.max())
