import importlib, os, sys
sys.path.insert(0, '.')
os.environ["FMPC_TILED"] = "1"
import numpy as np
pkg = importlib.import_module("mpc-sensorlessao_amd")
from tests.util import handle_from_model, oracle_batch, rel_err
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = pkg.synthetic.make_model(27, 144, T)
data = pkg.synthetic.make_replay_batch(model, r=1, steps=40)
h = handle_from_model(pkg, model)
z, info = h.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=1, k=1e-2, return_info=True, check=False)
h.close()
zo, nuo, ito, sto, steps = oracle_batch(model, data, 1, 1e-2)
np.set_printoptions(linewidth=220, precision=2)
print("status", info["status"])
bad = [p for p in range(40) if rel_err(z[p], zo[p]) > 1e-9]
print("bad problems", bad)
for p in bad[:2]:
    dn = np.abs(info["nu"][p] - nuo[p]).reshape(T, 27)
    print("problem", p, "step", info["step"][p], "oracle step", steps[p])
    print((dn > 1e-9).astype(int))
