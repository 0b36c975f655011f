"""rocprofv3 target: cold-start solves in the dense form WITH w at one batch size.  python3 scripts/prof_dense_w.py <batch>"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
from tests.util import handle_from_model
md = pkg.synthetic.make_model(27, 144, 30)
dev = torch.device("cuda:0")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
h = handle_from_model(pkg, md)
h.set_dense_form(1, 1 << 30)
data = pkg.synthetic.make_replay_batch(md, r=1, steps=batch)
x0 = torch.tensor(data["x0"], device=dev); x0p = torch.tensor(data["x0_pre"], device=dev); nu0 = torch.tensor(data["nu0"], device=dev)
wt = torch.tensor(0.01 * np.random.default_rng(3).standard_normal((batch, 810)), device=dev)
z = torch.empty((batch, 5130), device=dev, dtype=torch.float64); u0 = torch.empty((batch, 144), device=dev, dtype=torch.float64)
for _ in range(4):
    h.solve_device(x0, x0p, wt, nu0=nu0, n_newton=1, k=1e-2, z_out=z, u0_out=u0)
torch.cuda.synchronize()
h.close()
