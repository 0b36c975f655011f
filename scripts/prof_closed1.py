"""rocprofv3 target: closed loop with ONE realisation, 40 steps (fmpc_loop_step_device per step)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
from tests.util import handle_from_model
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
md = pkg.synthetic.make_model(27, 144, 30)
steps = 40
a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(R)], axis=1)
h = handle_from_model(pkg, md)
loop = pkg.ClosedLoop(h, R, n_newton=1, k=1e-2)
at = torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda:0"))
for s in range(steps):
    loop.step(at[s])
torch.cuda.synchronize()
h.close()
