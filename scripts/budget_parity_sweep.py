"""One-off assurance run (GPU box): the bench workload at a Newton budget of 5, EVERY problem against the structured
oracle -- iteration counts, step lengths, z.  python scripts/budget_parity_sweep.py [batch]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mpc-sensorlessao_amd")
from tests.util import handle_from_model, oracle_batch, rel_err, canon_steps
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
md = pkg.synthetic.make_model(27, 144, 30)
data = pkg.synthetic.make_replay_batch(md, r=0, steps=B)
h = handle_from_model(pkg, md)
z, info = h.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=5, k=1e-2, return_info=True)
print("device: iters histogram", np.bincount(info["iters"]), "path", h.last_dispatch(), flush=True)
t0 = time.time()
zo, nuo, ito, sto, steps = oracle_batch(md, data, 5, 1e-2)
print("oracle: iters histogram", np.bincount(ito), "in %.0f s" % (time.time() - t0), flush=True)
bad = [p for p in range(B) if info["iters"][p] != ito[p] or info["status"][p] != sto[p]]
err = max(rel_err(z[p], zo[p]) for p in range(B))
st_ok = all(np.array_equal(canon_steps(info["step"][p][:ito[p]]), canon_steps(steps[p][:ito[p]])) for p in range(B))
print("iteration/status mismatches:", bad[:10], len(bad), "| max rel err z %.2e | step lengths equal: %s" % (err, st_ok))
