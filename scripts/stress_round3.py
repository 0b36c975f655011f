"""One-off stress of the round-3 kernels: the one-launch walk against one call per step (bitwise, 64 realisations x 1500 steps, and
at the bound width where single steps are handed over), the affine form against the three-kernel form over many seeds."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
dev = torch.device("cuda:0")
mk = lambda md: pkg.FastMPCHandle(md["A1"], md["A2"], md["B"], md["Q"], md["R"], md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], md["T"])
for ub, R, steps in ((None, 64, 1500), (0.24, 40, 400), (0.235, 64, 300)):
    md = pkg.synthetic.make_model(27, 144, 30 if ub is None else 10)
    if ub is not None:
        md["u_min"] = -ub * np.ones(144); md["u_max"] = ub * np.ones(144)
    a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(8)], axis=1)
    a = np.ascontiguousarray(np.tile(a, (1, R // 8, 1)) * np.linspace(0.05, 4.0, R)[None, :, None])
    at = torch.from_numpy(a).to(dev)
    h1, h2 = mk(md), mk(md)
    la = pkg.ClosedLoop(h1, R, n_newton=1, k=1e-2, keep_z=False); lb = pkg.ClosedLoop(h2, R, n_newton=1, k=1e-2, keep_z=False)
    Ua = []; handed = 0
    for s in range(steps):
        Ua.append(la.step(at[s]).clone())
        if s % 50 == 0:
            torch.cuda.synchronize(); handed += h1.last_dispatch()[1]
    Ua = torch.stack(Ua)
    Ub, _ = lb.run_recorded(at, want_x0=False)
    torch.cuda.synchronize()
    print("walk: bounds %s, %d realisations x %d steps: bitwise equal %s (handed over at the sampled steps: %d), max |u| %.3g"
          % (ub, R, steps, bool(torch.equal(Ua, Ub)), handed, float(Ub.abs().max())), flush=True)
    h1.close(); h2.close()
md = pkg.synthetic.make_model(27, 144, 30)
h = mk(md)
os.environ["FMPC_NO_AFFINE"] = "1"; h3 = mk(md); os.environ.pop("FMPC_NO_AFFINE")
worst = 0.0
for seed in range(12):
    B = [2000, 1, 63, 64, 65, 517, 1999, 130, 1024, 31, 4000, 777][seed]
    data = pkg.synthetic.make_replay_batch(md, r=seed, steps=B)
    t = {k: torch.from_numpy(v).to(dev) for k, v in data.items() if v is not None}
    za = torch.empty((B, h.nz), dtype=torch.float64, device=dev); zb = torch.empty_like(za)
    sa = torch.zeros(B, dtype=torch.int32, device=dev); sb = torch.zeros_like(sa)
    h.solve_device(t["x0"], t["x0_pre"], None, None, t["nu0"], 1, 1e-2, z_out=za, status=sa)
    h3.solve_device(t["x0"], t["x0_pre"], None, None, t["nu0"], 1, 1e-2, z_out=zb, status=sb)
    torch.cuda.synchronize()
    e = float(((za - zb).abs().max(dim=1).values / zb.abs().max(dim=1).values).max())
    worst = max(worst, e)
    assert h.last_dual_form() == 2 and h3.last_dual_form() == 1 and torch.equal(sa, sb), (seed, B)
print("affine vs three-kernel form over 12 batches (1 .. 4000 problems): worst relative difference %.2e" % worst)
