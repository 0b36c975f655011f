"""Device time of the affine cold-start kernel at the headline batch (HIP events around solve_device), e.g. under FMPC_AFFINE_FLAGS.
    python3 scripts/affine_perf.py [batch] [reps]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
md = pkg.synthetic.make_model(27, 144, 30)
data = pkg.synthetic.make_replay_batch(md, r=0, steps=B)
dev = torch.device("cuda:0")
h = pkg.FastMPCHandle(md["A1"], md["A2"], md["B"], md["Q"], md["R"], md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], 30)
x0 = torch.from_numpy(data["x0"]).to(dev); x0p = torch.from_numpy(data["x0_pre"]).to(dev); nu0 = torch.from_numpy(data["nu0"]).to(dev)
zs = [torch.empty((B, h.nz), dtype=torch.float64, device=dev) for _ in range(4)]
u0 = torch.empty((B, h.m), dtype=torch.float64, device=dev)
st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
for want_z in (True, False):
    for _ in range(5):
        h.solve_device(x0, x0p, None, None, nu0, 1, 1e-2, z_out=zs[0] if want_z else None, status=st, iters=it, u0_out=u0, want_z=want_z)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        h.solve_device(x0, x0p, None, None, nu0, 1, 1e-2, z_out=zs[r % 4] if want_z else None, status=st, iters=it, u0_out=u0, want_z=want_z)
    e1.record(); torch.cuda.synchronize()
    print("flags %s batch %d %s: %.2f us per solve (form %d, handed %d)" % (os.environ.get("FMPC_AFFINE_FLAGS", "0"), B, "z + u0" if want_z else "u0 only",
          e0.elapsed_time(e1) / reps * 1e3, h.last_dual_form(), h.last_dispatch()[1]))
