"""For rocprofv3: 200 closed-loop steps of ONE realisation, first moves only (the first-move form: two launches per step)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = 200
model = pkg.synthetic.make_model(27, 144, 30)
h = pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"], model["u_max"],
                      model["x_min"], model["x_max"], 30)
a = np.stack([pkg.synthetic.make_realisation(model, r=r, steps=steps)[1:steps + 1] for r in range(R)], axis=1)
at = torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda:0"))
loop = pkg.ClosedLoop(h, R, n_newton=1, k=1e-2, keep_z=False)
loop.run_recorded(at, want_x0=False)
torch.cuda.synchronize()
assert int(loop.status.abs().sum()) == 0
print("done", h.last_dispatch())

lib = pkg.load()
if hasattr(lib, "fmpc_debug_first_trace"):
    import ctypes as C
    tr = (C.c_ulonglong * 8)()
    lib.fmpc_debug_first_trace(tr)
    t = [(tr[i] - tr[0]) * 0.01 for i in range(6)]
    print("first-move kernel, wavefront 0 (us since kernel entry): loads landed %.2f, d ready %.2f, rows done %.2f, sums ready %.2f, end %.2f" % tuple(t[1:6]))
