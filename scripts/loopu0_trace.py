"""Timeline of one launch of the many-realisation first-move kernel (timing build: FMPC_LIB=.../libfastmpc_timing.so)."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
lib = pkg.load()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
md = pkg.synthetic.make_model(27, 144, 30)
steps = 20
a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(8)], axis=1)
at = torch.from_numpy(np.ascontiguousarray(np.tile(a, (1, (R + 7) // 8, 1))[:, :R])).to(torch.device("cuda:0"))
h = pkg.FastMPCHandle(md["A1"], md["A2"], md["B"], md["Q"], md["R"], md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], 30)
loop = pkg.ClosedLoop(h, R, n_newton=1, k=1e-2, keep_z=False)
for s in range(steps):
    loop.step(at[s])
torch.cuda.synchronize()
out = (C.c_ulonglong * 4096)()
lib.fmpc_debug_loopu0_trace.argtypes = [C.c_void_p]; lib.fmpc_debug_loopu0_trace(out)
t = np.array(out[:], dtype=np.int64).reshape(4, 128, 8)
nb = (R + 15) // 16
if h.last_dual_form() == 4:                                # the fused step (fmpc_loop_step27): roles w slice 0 / first moves (two halves) / forms
    t0 = t[:, :nb, 0].min()
    for y, name in ((0, "w slice 0"), (1, "first moves a"), (2, "first moves b"), (3, "forms")):
        k = (t[y, :nb] - t0) * 0.01
        print("%-14s start %.1f..%.1f | operands of B u there %.1f | B u reduced %.1f | %s end %.1f (medians; max end %.1f)"
              % (name, k[:, 0].min(), k[:, 0].max(), np.median(k[:, 1]), np.median(k[:, 2]), ("forms summed %.1f |" % np.median(k[:, 4])) if y == 3 else "", np.median(k[:, 3]), k[:, 3].max()))
    sys.exit(0)
t = t.reshape(-1)[:2048].reshape(2, 128, 8)
t0 = t[:, :nb, 0].min()
for y, name in ((0, "first moves"), (1, "forms")):
    k = (t[y, :nb] - t0) * 0.01
    print("%-12s start %.1f..%.1f | d there %.1f | images there %.1f | products+stores issued %.1f | barrier %.1f | end %.1f (medians; max end %.1f)"
          % (name, k[:, 0].min(), k[:, 0].max(), np.median(k[:, 1]), np.median(k[:, 2]), np.median(k[:, 3]), np.median(k[:, 4]) if y else 0, np.median(k[:, 5]) if y else np.median(k[:, 3]), k[:, 5 if y else 3].max()))
