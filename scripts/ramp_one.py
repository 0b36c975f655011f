"""One configuration of the ramp-rate path for profiling:  python3 scripts/ramp_one.py [batch] [n_newton] [reps]"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mpc-sensorlessao_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n, m, T = 27, 144, 10
md = pkg.synthetic.make_model(n, m, T, var_order=1)
dev = torch.device("cuda", 0)
h = pkg.FastMPCHandle(md["A1"], None, md["B"], md["Q"], md["R"], md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], T, var_order=1, device=0)
h.set_ramp(-0.2121 * np.ones(m), 0.2121 * np.ones(m))
data = pkg.synthetic.make_replay_batch(md, r=1, steps=B)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
x0, nu0, up = t(data["x0"]), t(data["nu0"][:, :T * n]), t(0.05 * np.random.default_rng(0).standard_normal((B, m)))
z = torch.empty((B, h.nz), dtype=torch.float64, device=dev)
st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
for _ in range(reps):
    h.solve_device(x0, None, None, None, nu0, nw, 1e-2, z_out=z, status=st, iters=it, u_prev=up)
torch.cuda.synchronize()
print("done", int(it.sum()))
