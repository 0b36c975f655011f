"""Many lanes at a Newton budget of 5, for a kernel trace:  rocprofv3 --kernel-trace -- python3 scripts/lanes_trace.py [lanes] [steps]"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mpc-sensorlessao_amd")
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 12
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n, m, T, B = 27, 144, 30, 2000
md = pkg.synthetic.make_model(n, m, T)
data = pkg.synthetic.make_replay_batch(md, r=0, steps=B)
dev = torch.device("cuda", 0)
mk = lambda: pkg.FastMPCHandle(md["A1"], md["A2"], md["B"], md["Q"], md["R"], md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], T, device=0)
lanes = pkg.SolveLanes(mk, B, depth=depth, device=dev)
x0, x0p, nu0 = (torch.from_numpy(data[k]).to(dev) for k in ("x0", "x0_pre", "nu0"))
for _ in range(2 * depth):
    lanes.submit(x0, x0p, None, None, nu0, nw, 1e-2, after_current=False)
lanes.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    lanes.submit(x0, x0p, None, None, nu0, nw, 1e-2, after_current=False)
lanes.synchronize()
dt = time.perf_counter() - t0
print(f"{depth} lanes, budget {nw}: {dt / steps * 1e6:.1f} us per step, {B * steps / dt / 1e6:.2f} M steps/s")
