import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("mpc-sensorlessao_amd")
md = pkg.synthetic.make_model(27, 144, 30)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=40)[1:41] for r in range(8)], axis=1)
a = np.ascontiguousarray(np.tile(a, (1, (R + 7) // 8, 1))[:, :R])
dev = torch.device("cuda", 0)
at = torch.from_numpy(a).to(dev)
h = pkg.FastMPCHandle(md["A1"], md["A2"], md["B"], md["Q"], md["R"], md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], 30, device=0)
loop = pkg.ClosedLoop(h, R, n_newton=1, k=1e-2)
for s in range(40):
    loop.step(at[s])
torch.cuda.synchronize()
print("done")
