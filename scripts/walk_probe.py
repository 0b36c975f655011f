"""Which steps of a closed loop are handed to the exact path, by bound width (exploration for the tests), and the walks of
fmpc_loop_run_device (FMPC_DEBUG_WALK=1)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
dev = torch.device("cuda:0")
sc = np.array([0.05, 0.3, 1.0, 2.0, 5.0])[None, :, None]
for T, ub, scale in ((10, 0.22, sc), (10, 0.24, sc), (10, 0.25, sc), (10, 0.26, sc), (10, 0.27, sc), (10, 0.28, sc)):
    md = pkg.synthetic.make_model(27, 144, T)
    md["u_min"] = -ub * np.ones(144); md["u_max"] = ub * np.ones(144)
    R, steps = 5, 8
    a = scale * np.stack([pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(R)], axis=1)
    at = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    h = pkg.FastMPCHandle(md["A1"], md["A2"], md["B"], md["Q"], md["R"], md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], T)
    lp = pkg.ClosedLoop(h, R, n_newton=1, k=1e-2, keep_z=False)
    handed = []
    for s in range(steps):
        lp.step(at[s]); torch.cuda.synchronize(); handed.append(h.last_dispatch()[1])
    print("T %d bounds +-%.2f, a x %s: handed per step %s" % (T, ub, np.ravel(scale), handed), flush=True)
    h.close()
