"""The headline step with the z rows of consecutive problems `ldz` doubles apart (fmpc_set_z_ld through a view: a multiple of 16 makes
every 128-byte run of a tile one cache line and selects the kernel instance with non-temporal stores), five repetitions of `steps` steps.
    python3 scripts/headline_ldz.py [steps] [ldz]          (ldz 0: contiguous rows; FMPC_AFFINE_NO_NT=1: ordinary stores)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = 2000
model = pkg.synthetic.make_model(27, 144, 30)
h = pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"], model["u_max"], model["x_min"], model["x_max"], 30)
dev = torch.device("cuda:0")
ldz = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else h.nz
sets = []
for i in range(4):
    d = pkg.synthetic.make_replay_batch(model, r=i, steps=B)
    big = torch.zeros((B, ldz), dtype=torch.float64, device=dev)
    sets.append((torch.from_numpy(d["x0"]).to(dev), torch.from_numpy(d["x0_pre"]).to(dev), torch.from_numpy(d["nu0"]).to(dev),
                 big[:, :h.nz] if ldz != h.nz else big, torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)))
ts = []
for rep in range(5):
    for it in range(steps + 10):
        if it == 10:
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        x0, x0p, nu0, z, st, itr = sets[it % 4]
        h.solve_device(x0, x0p, None, None, nu0, 1, 1e-2, z_out=z, status=st, iters=itr)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / steps * 1e3)
print("ldz %d: %s us per step (median %.2f), dual form %d" % (ldz, " ".join("%.2f" % t for t in ts), float(np.median(ts)), h.last_dual_form()))
