"""Per-wavefront timeline of the dense-form product without w (timing build): when do the wavefronts start, have their
operands, finish?  FMPC_LIB=.../libfastmpc_timing.so python3 scripts/inv_trace.py [batch]"""
import os, sys
os.environ.setdefault("FMPC_LIB", os.path.abspath("mpc-sensorlessao_amd/lib/libfastmpc_timing.so"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import importlib, ctypes as C
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
from tests.util import handle_from_model
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
md = pkg.synthetic.make_model(27, 144, 30)
h = handle_from_model(pkg, md)
dev = torch.device("cuda:0")
data = pkg.synthetic.make_replay_batch(md, r=1, steps=batch)
x0 = torch.tensor(data["x0"], device=dev); x0p = torch.tensor(data["x0_pre"], device=dev); nu0 = torch.tensor(data["nu0"], device=dev)
z = torch.empty((batch, 5130), device=dev, dtype=torch.float64)
for _ in range(4):
    h.solve_device(x0, x0p, None, nu0=nu0, n_newton=1, k=1e-2, z_out=z)
torch.cuda.synchronize()
lib = pkg.load()
n = 8192
out = (C.c_ulonglong * (4 * n))()
lib.fmpc_debug_inv_trace(out, n)
a = np.array(out, dtype=np.int64).reshape(n, 4)
gate = a[7000:8000]; gate = gate[gate[:, 0] > 0]
a = a[:7000]
a = a[a[:, 0] > 0]
t0 = min(a[:, 0].min(), gate[:, 0].min())
gate = (gate - t0) * 0.01
print('gate workgroups', len(gate), 'start median %.2f, end median %.2f, max %.2f us' % (np.median(gate[:, 0]), np.median(gate[:, 3]), gate[:, 3].max()))
a = (a - t0) * 0.01          # us
live = a[a[:, 3] > a[:, 2]]
print("wavefronts traced", len(a))
for name, col in (("start", 0), ("operands in LDS", 1), ("products done", 2), ("end", 3)):
    v = a[:, col]
    print("  %-18s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))
d = a[:, 3] - a[:, 0]
print("  lifetime           median %.2f  max %.2f us;  load phase median %.2f, product phase median %.2f" % (np.median(d), d.max(), np.median(a[:, 1] - a[:, 0]), np.median(a[:, 2] - a[:, 1])))
