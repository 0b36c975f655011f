"""Device time of the cold-start solve without w per launch shape of the dense form (FMPC_INV_SHAPE0)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import importlib
import torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
from tests.util import handle_from_model
md = pkg.synthetic.make_model(27, 144, 30)
h = handle_from_model(pkg, md)
dev = torch.device("cuda:0")
out = []
for batch in (16, 512, 2000, 4096):
    data = pkg.synthetic.make_replay_batch(md, r=1, steps=batch)
    x0 = torch.tensor(data["x0"], device=dev); x0p = torch.tensor(data["x0_pre"], device=dev); nu0 = torch.tensor(data["nu0"], device=dev)
    z = torch.empty((batch, 5130), device=dev, dtype=torch.float64); u0 = torch.empty((batch, 144), device=dev, dtype=torch.float64)
    for _ in range(5):
        h.solve_device(x0, x0p, None, nu0=nu0, n_newton=1, k=1e-2, z_out=z, u0_out=u0)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        h.solve_device(x0, x0p, None, nu0=nu0, n_newton=1, k=1e-2, z_out=z, u0_out=u0)
    e1.record(); torch.cuda.synchronize()
    out.append("%d: %.1f" % (batch, e0.elapsed_time(e1) / 100 * 1e3))
print("shape", os.environ.get("FMPC_INV_SHAPE0", "0"), " ".join(out), flush=True)
