"""Diagnostic: per-phase cycle totals of the tiled kernel (workgroup 0, thread 0); needs `make -C mpc-sensorlessao_amd/csrc timing`.
  python scripts/tiled_phases.py [n] [T] [batch] [f32]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, '.')
os.environ.setdefault("FMPC_LIB", os.path.abspath("mpc-sensorlessao_amd/lib/libfastmpc_timing.so"))
os.environ["FMPC_TILED"] = "1"
import numpy as np, torch
pkg = importlib.import_module('mpc-sensorlessao_amd')
lib = pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 27
T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
f32 = len(sys.argv) > 4 and sys.argv[4] == "f32"
m = 144
model = pkg.synthetic.make_model(n, m, T)
data = pkg.synthetic.make_replay_batch(model, r=0, steps=B)
h = pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"], model["u_max"], model["x_min"], model["x_max"], T)
if f32 or n > 47:
    h.set_precision("f32")
dev = torch.device('cuda:0')
x0 = torch.from_numpy(data["x0"]).to(dev); x0p = torch.from_numpy(data["x0_pre"]).to(dev); nu0 = torch.from_numpy(data["nu0"]).to(dev)
out = (C.c_ulonglong * 16)()
for rep in range(2):
    lib.fmpc_debug_tiled_timing(out, 1)
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); h.solve_device(x0, x0p, None, None, nu0, 1, 1e-2); b.record(); torch.cuda.synchronize()
    lib.fmpc_debug_tiled_timing(out, 0)
names = {0: "P0 init", 1: "P1 residuals", 12: "P2 rhs", 13: "P2 S pre-pass", 2: "P2 slot clear", 6: "P3 factor + forward sweep (whole phase)", 7: "P4 backward",
         8: "P5 dz + update"}
sub = {3: "P3.A stage products", 11: "P3.B requests of the next stage's tiles", 9: "P3.B row products", 14: "P3.B before the potrf (wave 0 owns the tile)", 15: "P3.B potrf itself", 10: "P3.B diagonal tile: rest, stores", 4: "P3.B wait at the barrier",
       5: "P3.B scale + store (to barrier)"}
tot = sum(out[i] for i in names) or 1
print(f"tiled kernel n={n} T={T} batch={B} {'fp32' if (f32 or n > 47) else 'fp64'}: {a.elapsed_time(b):.3f} ms; cycles of workgroup 0, wave 0 over its problems:")
for i, nm in names.items():
    print("  %-48s %12d  %5.1f%%" % (nm, out[i], 100.0 * out[i] / tot))
    if i == 6:
        for j, sn in sub.items():
            print("      %-44s %12d  %5.1f%%" % (sn, out[j], 100.0 * out[j] / tot))
print("  total %d" % tot)
