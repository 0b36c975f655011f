"""Where the time of one literal per-timestep call goes (host pointers, one problem, cold start, budget 1):
fmpc_solve (host entry) vs the device entry + synchronisation vs bare copies.   python3 scripts/once_latency.py [reps]"""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
md = pkg.synthetic.make_model(27, 144, 30)
d = pkg.synthetic.make_replay_batch(md, r=0, steps=64)
h = pkg.FastMPCHandle(md["A1"], md["A2"], md["B"], md["Q"], md["R"], md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], 30)
dev = torch.device("cuda:0")
zo = np.empty((1, h.nz))


def med(f, n=reps):
    for _ in range(20):
        f(0)
    ts = []
    for i in range(n):
        t0 = time.perf_counter(); f(i % 60 + 1); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e6, ts[len(ts) // 10] * 1e6


host = lambda i: h.solve(d["x0"][i:i + 1], d["x0_pre"][i:i + 1], None, nu0=d["nu0"][i:i + 1], n_newton=1, k=1e-2, z_out=zo)
print("host entry, z out (h.solve, batch 1):            median %.1f us  (p10 %.1f)" % med(host))
u0o = np.empty((1, 144))
hostu = lambda i: h.solve_u0(d["x0"][i:i + 1], d["x0_pre"][i:i + 1], None, nu0=d["nu0"][i:i + 1], n_newton=1, k=1e-2, u0_out=u0o)
print("host entry, first moves only (h.solve_u0):        median %.1f us  (p10 %.1f)" % med(hostu))
x0, x0p, nu0 = (torch.from_numpy(d[k]).to(dev) for k in ("x0", "x0_pre", "nu0"))
z = torch.empty((1, h.nz), dtype=torch.float64, device=dev); st = torch.zeros(1, dtype=torch.int32, device=dev); it = torch.zeros(1, dtype=torch.int32, device=dev)


def devs(i):
    h.solve_device(x0[i:i + 1], x0p[i:i + 1], None, None, nu0[i:i + 1], 1, 1e-2, z_out=z, status=st, iters=it)
    torch.cuda.synchronize()
print("device entry + synchronize:                       median %.1f us  (p10 %.1f)" % med(devs))
pin_in = torch.empty(27 * 2 + 810, dtype=torch.float64).pin_memory(); pin_out = torch.empty(h.nz + 8, dtype=torch.float64).pin_memory()
din = torch.empty_like(pin_in, device=dev); dout = torch.empty(h.nz + 8, dtype=torch.float64, device=dev)


def copies(i):
    din.copy_(pin_in, non_blocking=True); pin_out.copy_(dout, non_blocking=True); torch.cuda.synchronize()
print("one H2D + one D2H of the same sizes + synchronize: median %.1f us  (p10 %.1f)" % med(copies))
