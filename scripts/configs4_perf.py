"""configs[4] (n = 65, m = 144, T = 60, batch 1024, fp32 factor): device time of one Newton step per wavefront count.
    python3 scripts/configs4_perf.py            (FMPC_TILED_NW=4|8 is read when a handle is created)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
dev = torch.device("cuda:0")
n, m, T, B = 65, 144, 60, 1024
fl = T * (2 * n * n * m + (19.0 / 3.0) * n ** 3 + 20 * n * n + 6 * (2 * n * n + n * m) + 40 * (n + m))
model = pkg.synthetic.make_model(n, m, T)
data = pkg.synthetic.make_replay_batch(model, r=4, steps=B)
x0 = torch.from_numpy(data["x0"]).to(dev); x0p = torch.from_numpy(data["x0_pre"]).to(dev); nu0 = torch.from_numpy(data["nu0"]).to(dev)
res = {}
for nw in (sys.argv[1:] or ["8", "4"]):
    os.environ["FMPC_TILED_NW"] = nw
    h = pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"], model["u_max"],
                          model["x_min"], model["x_max"], T)
    z = torch.empty((B, h.nz), dtype=torch.float64, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
    for _ in range(2):
        h.solve_device(x0, x0p, None, None, nu0, 1, 1e-2, z_out=z, status=st, iters=it)
    torch.cuda.synchronize()
    os.environ.pop("FMPC_TILED_NW")                       # (read again when the kernel's images are built: at the first solve)
    ts = []
    for _ in range(5):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); h.solve_device(x0, x0p, None, None, nu0, 1, 1e-2, z_out=z, status=st, iters=it); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    assert int((st < 0).sum()) == 0
    res[nw] = z.cpu().numpy().copy()
    print("wavefronts per problem %s: %.3f ms per Newton step of %d problems = %.2f TFLOP/s = %.3f of the fp32 matrix peak (path %s)"
          % (nw, ms, B, fl * B / ms / 1e9, fl * B / ms / 1e9 / 157.3, h.last_dispatch()))
    h.close()
k = list(res)
if len(k) > 1:
    print("max rel diff between the two: %.2e" % (np.abs(res[k[0]] - res[k[1]]).max() / np.abs(res[k[0]]).max()))
