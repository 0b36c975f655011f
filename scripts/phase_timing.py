"""Diagnostic: per-phase cycle totals of the wave kernel (needs `make -C mpc-sensorlessao_amd/csrc timing`).
Run with FMPC_LIB=mpc-sensorlessao_amd/lib/libfastmpc_timing.so."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, '.')
os.environ.setdefault("FMPC_LIB", os.path.abspath("mpc-sensorlessao_amd/lib/libfastmpc_timing.so"))
import numpy as np, torch
pkg = importlib.import_module('mpc-sensorlessao_amd')
lib = pkg.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 5
model = pkg.synthetic.make_model(27, 144, 30)
data = pkg.synthetic.make_replay_batch(model, r=0, steps=B)
h = pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"], model["u_max"], model["x_min"], model["x_max"], 30)
dev = torch.device('cuda:0')
x0 = torch.from_numpy(data["x0"]).to(dev); x0p = torch.from_numpy(data["x0_pre"]).to(dev); nu0 = torch.from_numpy(data["nu0"]).to(dev)
out = (C.c_ulonglong * 16)()
for rep in range(3):
    h.solve_device(x0, x0p, None, None, nu0, nw, 1e-2); torch.cuda.synchronize()
    lib.fmpc_debug_timing(out)
tot = sum(out[i] for i in range(4, 12))
nw_ = min(B, 2048)
print("per-wave-average cycles (batch %d, n_newton %d):" % (B, nw))
for nm, i in [("P0 init", 4), ("P1 residuals (C'nu, Cz)", 5), ("P2 rhs", 6), ("P3 factor+fwd | shared fwd sweep", 7), ("P4 backward | shared bwd sweep", 8), ("P5 dz+update", 9), ("between problems", 11)]:
    print("  %-26s %12.0f  %5.1f%%" % (nm, out[i] / nw_, 100.0 * out[i] / tot))
print("  total %.0f cycles/wave" % (tot / nw_))
ft = sum(out[i] for i in range(4))
if ft:       # per-stage sections of the per-problem factor phase (general path only)
    for nm, i in [("P3.images + B Rt^-1 B' + U'U (MFMA)", 0), ("P3.tiles -> LDS, row/col loads", 1), ("P3.fused potrf+trsm (VALU)", 2), ("P3.store + readback", 3)]:
        print("    %-38s %12.0f  %5.1f%% of P3" % (nm, out[i] / nw_, 100.0 * out[i] / ft))
