"""Diagnostic (timing build): per-wavefront time stamps of fmpc_cold_affine2 at the headline size.
   FMPC_LIB=mpc-sensorlessao_amd/lib/libfastmpc_timing.so python3 scripts/affine2_trace.py [batch]"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FMPC_LIB", os.path.join(ROOT, "mpc-sensorlessao_amd", "lib", "libfastmpc_timing.so"))
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
lib = pkg.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
model = pkg.synthetic.make_model(27, 144, 30)
data = pkg.synthetic.make_replay_batch(model, r=0, steps=B)
h = pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"], model["u_max"], model["x_min"], model["x_max"], 30)
dev = torch.device("cuda:0")
x0 = torch.from_numpy(data["x0"]).to(dev); x0p = torch.from_numpy(data["x0_pre"]).to(dev); nu0 = torch.from_numpy(data["nu0"]).to(dev)
for _ in range(5):
    z, st, it = h.solve_device(x0, x0p, None, None, nu0, 1, 1e-2)
torch.cuda.synchronize()
out = (C.c_ulonglong * (2048 * 8))()
lib.fmpc_debug_affine2_trace.restype = C.c_int
assert lib.fmpc_debug_affine2_trace(out) == 0
t = np.array(out[:], dtype=np.float64).reshape(2048, 8)
task = t[:2048]; task = task[task[:, 4] > 0]
t0 = t[t[:, 0] > 0][:, 0].min()
us = lambda a: (a - t0) * 0.01
q = lambda a: "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
print("task wavefronts: %d" % len(task))
for nm, k in (("start", 0), ("data staged (barrier passed)", 1), ("nu+ and x halves computed", 2), ("nu+ exchanged", 3), ("end", 4)):
    print("  %-30s %s" % (nm, q(us(task[:, k]))))
for nm, a, b in (("staging", 0, 1), ("nu+ and x halves (112 MFMA)", 1, 2), ("x stores + exchange", 2, 3), ("u rows (112 / 140 MFMA + stores)", 3, 4)):
    print("  duration %-28s %s" % (nm, q((task[:, b] - task[:, a]) * 0.01)))
f = t[:2048]; f = f[f[:, 5] > 0]
if len(f):
    print("decision wavefronts: end", q(us(f[:, 5])))
