"""Time fmpc_loop_inputs_device alone:  python scripts/loop_inputs_probe.py [batch]"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mpc-sensorlessao_amd")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
md = pkg.synthetic.make_model(27, 144, 30)
dev = torch.device("cuda", 0)
h = pkg.FastMPCHandle(md["A1"], md["A2"], md["B"], md["Q"], md["R"], md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], 30, device=0)
f = dict(dtype=torch.float64, device=dev)
a, xl, u1, u2 = torch.randn(R, 27, **f), torch.randn(R, 27, **f), torch.randn(R, 144, **f), torch.randn(R, 144, **f)
x0, x0p, w = torch.empty(R, 27, **f), torch.empty(R, 27, **f), torch.empty(R, 810, **f)
for _ in range(5):
    h.loop_inputs_device(a, xl, u1, u2, x0, x0p, w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    h.loop_inputs_device(a, xl, u1, u2, x0, x0p, w)
e1.record(); torch.cuda.synchronize()
print(f"batch {R}: {e0.elapsed_time(e1) * 10:.1f} us per call")
lib = pkg.load()
if hasattr(lib, "fmpc_debug_loop_inputs_timing"):
    import ctypes as C
    out = (C.c_ulonglong * 8)()
    torch.cuda.synchronize()
    lib.fmpc_debug_loop_inputs_timing(out)
    t = [out[i] * 0.01 for i in range(5)]
    print("workgroup (0,0), us: staging B', u %.1f | B u %.1f | x0, x0_pre %.1f | staging M %.1f | w %.1f" % (t[1] - t[0], t[2] - t[1], 0.0, t[3] - t[2], t[4] - t[3]))
