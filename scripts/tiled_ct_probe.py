"""The tiled-kernel instance <double,2,4,11> that is kept out of production: where does it differ?
    FMPC_TILED_CT_NW4=1 python3 scripts/tiled_ct_probe.py   (also run without the variable: the run-time-structure instance)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
dev = torch.device("cuda:0")
n, m, T, B = 27, 144, 30, 96
md = pkg.synthetic.make_model(n, m, T)
md["u_min"] = -0.6 * np.ones(m); md["u_max"] = 0.6 * np.ones(m)
def make(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    h = pkg.FastMPCHandle(md["A1"], md["A2"], md["B"], md["Q"], md["R"], md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], T)
    for k, v in old.items():
        if v is None: os.environ.pop(k)
        else: os.environ[k] = v
    return h
ht = make({})           # default handle: explicit-start batches <= 512 run the tiled kernel with 4 wavefronts per problem
hw = make({"FMPC_NO_SMALL_TILED": "1"})
for nn in (1, 2):
    for seed in range(6):
        rng = np.random.default_rng(1000 + seed)
        data = pkg.synthetic.make_replay_batch(md, r=50 + seed, steps=B)
        z0 = np.zeros((B, T, n + m)); z0[:, :, :m] = rng.uniform(-0.5, 0.5, (B, T, m)); z0[:, :, m:] = rng.standard_normal((B, T, n))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        x0, x0p, nu0, zi = t(data["x0"]), t(data["x0_pre"]), t(data["nu0"]), t(z0.reshape(B, -1))
        res = {}
        for name, h in (("tiled", ht), ("wave", hw)):
            nu = torch.empty((B, T * n), dtype=torch.float64, device=dev); stp = torch.empty((B, nn), dtype=torch.float64, device=dev)
            z, st, it = h.solve_device(x0, x0p, None, zi, nu0, nn, 1e-2, nu_out=nu, step=stp)
            torch.cuda.synchronize()
            res[name] = (z.cpu().numpy().reshape(B, T, n + m), nu.cpu().numpy().reshape(B, T, n), st.cpu().numpy(), it.cpu().numpy(), stp.cpu().numpy())
        a, b = res["tiled"], res["wave"]
        ez = np.abs(a[0] - b[0]).max(axis=2) / np.abs(b[0]).max()          # (B, T)
        en = np.abs(a[1] - b[1]).max(axis=2) / np.abs(b[1]).max()
        bad = np.argwhere(en > 1e-10)
        print("newton %d seed %d: path %s nw %d  max rel z %.2e nu %.2e  status equal %s  steps equal %s  bad (problem, stage) entries of nu: %d" %
              (nn, seed, ht.last_dispatch(), ht.last_tiled_wavefronts(), ez.max(), en.max(), np.array_equal(a[2], b[2]), np.allclose(a[4], b[4]), len(bad)))
        if len(bad):
            probs = sorted(set(int(p) for p, _ in bad))
            print("   problems:", probs[:20], "...", "stages of first bad problem:", [int(s) for p, s in bad if p == probs[0]])
            p = probs[0]
            s0 = min(int(s) for q, s in bad if q == p)
            print("   first bad problem %d stage %d: nu tiled %s\n                                 nu wave  %s" % (p, s0, a[1][p, s0, :6], b[1][p, s0, :6]))
            d = np.abs(a[1][p] - b[1][p])
            print("   per-stage max |diff| of nu:", " ".join("%.1e" % v for v in d.max(axis=1)))
            print("   per-entry (stage %d) |diff|:" % s0, " ".join("%.1e" % v for v in d[s0]))
