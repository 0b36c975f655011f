"""Device time of the per-problem-factor paths (HIP events on torch's stream): wave kernel vs tiled kernel (fp64, fp32) at
(27,144,30) x 2000 from a warm start (every problem factors its own Y), and the tiled fp32 kernel at configs[4].
  python scripts/tiled_perf.py [batch27] [batch65]"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mpc-sensorlessao_amd")
dev = torch.device("cuda:0")


def flops(n, m, T):
    return T * (2 * n * n * m + (19.0 / 3.0) * n ** 3 + 20 * n * n + 6 * (2 * n * n + n * m) + 40 * (n + m))


def bench(n, m, T, B, nw, tiled, prec, reps=5, label=""):
    if tiled:
        os.environ["FMPC_TILED"] = "1"
    else:
        os.environ.pop("FMPC_TILED", None)
    model = pkg.synthetic.make_model(n, m, T)
    data = pkg.synthetic.make_replay_batch(model, r=0, steps=B)
    h = pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"],
                          model["u_max"], model["x_min"], model["x_max"], T)
    if prec:
        h.set_precision(prec)
    x0 = torch.from_numpy(data["x0"]).to(dev); x0p = torch.from_numpy(data["x0_pre"]).to(dev)
    nu0 = torch.from_numpy(data["nu0"]).to(dev)
    zc = np.tile(np.concatenate([np.zeros(m), np.zeros(n)]), T)
    zi = torch.from_numpy(np.tile(zc, (B, 1))).to(dev)          # explicit start = per-problem factor path
    z = torch.empty((B, h.nz), dtype=torch.float64, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
    for _ in range(2):
        h.solve_device(x0, x0p, None, zi, nu0, nw, 1e-2, z_out=z, status=st, iters=it)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); h.solve_device(x0, x0p, None, zi, nu0, nw, 1e-2, z_out=z, status=st, iters=it); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = float(np.median(ts)); iters = float(it.sum().item())
    assert int((st < 0).sum()) == 0
    fl = flops(n, m, T) * iters
    peak = 157.3 if prec == "f32" else 78.6
    print(f"{label:28s} n={n} T={T} B={B} nw={nw}: {ms:8.3f} ms  iters/problem {iters / B:.2f}  "
          f"{fl / ms / 1e9:7.2f} TFLOP/s = {fl / ms / 1e9 / peak:.3f} of {peak} peak; path {h.last_dispatch()[0]}", flush=True)
    h.close()
    return ms


B27 = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B65 = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
for nw in (1, 3):
    bench(27, 144, 30, B27, nw, False, None, label="wave kernel fp64")
    bench(27, 144, 30, B27, nw, True, None, label="tiled fp64")
    bench(27, 144, 30, B27, nw, True, "f32", label="tiled fp32 factor")
bench(27, 144, 30, 512, 1, True, None, label="tiled fp64")
bench(27, 144, 30, 4096, 1, True, None, label="tiled fp64")
for nw in (1, 3):
    bench(65, 144, 60, B65, nw, False, None, label="configs[4] tiled fp32")
