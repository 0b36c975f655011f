"""Latency of one per-problem-factor Newton step when fewer problems than CUs are in flight (the continuation of a Newton
budget > 1): wave kernel against the tiled kernel, 190 problems from an explicit start."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpc-sensorlessao_amd")
dev = torch.device("cuda:0")
n, m, T, B = 27, 144, 30, int(sys.argv[1]) if len(sys.argv) > 1 else 190
model = pkg.synthetic.make_model(n, m, T)
data = pkg.synthetic.make_replay_batch(model, r=0, steps=B)
for tag, env in (("wave", {}), ("tiled NW=2", {"FMPC_TILED": "1", "FMPC_TILED_NW": "2"}), ("tiled NW=4", {"FMPC_TILED": "1", "FMPC_TILED_NW": "4"})):
    for k_ in ("FMPC_TILED", "FMPC_TILED_NW"):
        os.environ.pop(k_, None)
    os.environ.update(env)
    h = pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"], model["u_max"],
                          model["x_min"], model["x_max"], T)
    x0 = torch.from_numpy(data["x0"]).to(dev); x0p = torch.from_numpy(data["x0_pre"]).to(dev); nu0 = torch.from_numpy(data["nu0"]).to(dev)
    zi = torch.zeros((B, h.nz), dtype=torch.float64, device=dev)
    z = torch.empty((B, h.nz), dtype=torch.float64, device=dev)
    for nw in (1, 2):
        for _ in range(3):
            h.solve_device(x0, x0p, None, zi, nu0, nw, 1e-2, z_out=z)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            h.solve_device(x0, x0p, None, zi, nu0, nw, 1e-2, z_out=z)
        e1.record(); torch.cuda.synchronize()
        print("%-12s batch %d n_newton %d: %.3f ms per launch" % (tag, B, nw, e0.elapsed_time(e1) / 10), flush=True)
    h.close()
