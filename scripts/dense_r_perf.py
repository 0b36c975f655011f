"""Device time of one Newton step with a dense R (tiled kernel, per-stage m x m Cholesky in LDS) at (27, 144, T)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import importlib
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
from tests.util import handle_from_model
dev = torch.device("cuda:0")
for T, B in ((10, 256), (30, 256)):
    md = pkg.synthetic.make_model(27, 144, T)
    G = np.random.default_rng(7).standard_normal((144, 144))
    md["R"] = G @ G.T / 144 + np.eye(144)
    h = handle_from_model(pkg, md)
    data = pkg.synthetic.make_replay_batch(md, r=1, steps=B)
    x0 = torch.tensor(data["x0"], device=dev); x0p = torch.tensor(data["x0_pre"], device=dev); nu0 = torch.tensor(data["nu0"], device=dev)
    z = torch.empty((B, T * 171), device=dev, dtype=torch.float64)
    for _ in range(2):
        h.solve_device(x0, x0p, None, nu0=nu0, n_newton=1, k=1e-2, z_out=z)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        h.solve_device(x0, x0p, None, nu0=nu0, n_newton=1, k=1e-2, z_out=z)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("dense R  T %d  batch %d: %.2f ms per Newton step  (%.0f problem-iterations/s)  path %d" % (T, B, ms, B / ms * 1e3, h.last_dispatch()[0]), flush=True)
    h.close()
