"""Time stamps of one launch of the first-move kernel (timing build): FMPC_LIB=.../libfastmpc_timing.so python3 scripts/first_move_trace.py [realisations]"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
lib = pkg.load()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
md = pkg.synthetic.make_model(27, 144, 30)
a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=40)[1:41] for r in range(R)], axis=1)
at = torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda:0"))
h = pkg.FastMPCHandle(md["A1"], md["A2"], md["B"], md["Q"], md["R"], md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], 30)
loop = pkg.ClosedLoop(h, R, n_newton=1, k=1e-2, keep_z=False)
for s in range(40):
    loop.step(at[s])
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
lib.fmpc_debug_first_trace.argtypes = [C.c_void_p]; lib.fmpc_debug_first_trace(out)
t = [(out[i] - out[0]) * 0.01 for i in range(6)]
print("first-move kernel, realisation 0, us after entry: inputs + rows requested and in LDS %.1f | B u %.1f | w rows / partial rows %.1f | forms + u0 written %.1f | decision %.1f"
      % (t[1], t[2], t[3], t[4], t[5]))
