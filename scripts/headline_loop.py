"""The headline step (configs[1]: 2000 problems, cold start, one Newton step, w = NULL) N times -- a small target for
rocprofv3 --kernel-trace --stats.   python3 scripts/headline_loop.py [steps] [batch]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
model = pkg.synthetic.make_model(27, 144, 30)
h = pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"], model["u_max"], model["x_min"], model["x_max"], 30)
dev = torch.device("cuda:0")
sets = []
for i in range(4):
    d = pkg.synthetic.make_replay_batch(model, r=i, steps=B)
    sets.append((torch.from_numpy(d["x0"]).to(dev), torch.from_numpy(d["x0_pre"]).to(dev), torch.from_numpy(d["nu0"]).to(dev),
                 torch.empty((B, h.nz), dtype=torch.float64, device=dev), torch.empty((B, 144), dtype=torch.float64, device=dev),
                 torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)))
for it in range(steps + 10):
    if it == 10:
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    x0, x0p, nu0, z, u0, st, itr = sets[it % 4]
    h.solve_device(x0, x0p, None, None, nu0, 1, 1e-2, z_out=z, status=st, iters=itr, u0_out=u0)
e1.record(); torch.cuda.synchronize()
print("%.2f us per step (%d problems), dual form %d" % (e0.elapsed_time(e1) / steps * 1e3, B, h.last_dual_form()))
