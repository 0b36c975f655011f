"""Per-problem-factor path (every problem factors its own Y: explicit start, SURVEY 8d "roofline is quoted on this one"):
time of one launch at a batch, agreement with the tiled kernel (an independent implementation in the same library), and --
with the timing build (FMPC_LIB=mpc-sensorlessao_amd/lib/libfastmpc_timing.so) -- the per-phase cycle totals.
    python3 scripts/general_perf.py [batch] [n_newton] [reps]
The start point is off-centre (u = 0.6 u_max on every third actuator) so that the barrier terms differ per problem."""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
lib = pkg.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n, m, T = 27, 144, 30
model = pkg.synthetic.make_model(n, m, T)
data = pkg.synthetic.make_replay_batch(model, r=0, steps=B)
dev = torch.device("cuda:0")


def handle(tiled):
    if tiled:
        os.environ["FMPC_TILED"] = "1"
    else:
        os.environ.pop("FMPC_TILED", None)
        os.environ["FMPC_NO_SMALL_TILED"] = "1"
    return pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"],
                             model["u_max"], model["x_min"], model["x_max"], T)


rng = np.random.default_rng(11)
zc = np.tile(np.concatenate([(model["u_min"] + model["u_max"]) / 2, (model["x_min"] + model["x_max"]) / 2]), T)
zi_h = np.tile(zc, (B, 1)).reshape(B, T, n + m)
zi_h[:, :, 0:m:3] += 0.6 * model["u_max"][0] * rng.random((B, T, (m + 2) // 3))
zi_h[:, :, m:] += rng.standard_normal((B, T, n))
zi_h = zi_h.reshape(B, -1)
x0 = torch.from_numpy(data["x0"]).to(dev); x0p = torch.from_numpy(data["x0_pre"]).to(dev); nu0 = torch.from_numpy(data["nu0"]).to(dev)
zi = torch.from_numpy(zi_h).to(dev)
res = {}
for tiled in ((0,) if os.environ.get('FMPC_PERF_WAVE_ONLY') == '1' else (0, 1)):
    h = handle(tiled)
    z = torch.empty((B, h.nz), dtype=torch.float64, device=dev)
    nu = torch.empty((B, h.nu_len), dtype=torch.float64, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
    stp = torch.empty((B, max(nw, 1)), dtype=torch.float64, device=dev)
    for _ in range(2):
        h.solve_device(x0, x0p, None, zi, nu0, nw, 1e-2, z_out=z, nu_out=nu, status=st, iters=it, step=stp)
    torch.cuda.synchronize()
    if hasattr(lib, "fmpc_debug_timing") and not tiled:
        lib.fmpc_debug_timing((C.c_ulonglong * 16)())
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        h.solve_device(x0, x0p, None, zi, nu0, nw, 1e-2, z_out=z, nu_out=nu, status=st, iters=it, step=stp)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 11643210.0 * B * float(it.sum()) / B
    print("%s kernel: batch %d n_newton %d  %.4f ms per launch  path %s  iters %d  status!=0: %d  -> %.2f TFLOP/s = %.3f of fp64 peak"
          % ("tiled" if tiled else "wave ", B, nw, ms, h.last_dispatch(), int(it.sum()), int((st != 0).sum()), flops / ms / 1e9, flops / ms / 1e9 / 78.6))
    res[tiled] = (z.cpu().numpy().copy(), nu.cpu().numpy().copy(), stp.cpu().numpy().copy(), it.cpu().numpy().copy())
    if hasattr(lib, "fmpc_debug_timing") and not tiled:
        out = (C.c_ulonglong * 16)()
        lib.fmpc_debug_timing(out)
        nwv = min(B, 2048) * reps
        tot = sum(out[i] for i in range(4, 12)) or 1
        print("  per-wave-average cycles per launch:")
        for nm, i in [("P0 init", 4), ("P1 residuals (C'nu, Cz)", 5), ("P2 rhs", 6), ("P3 factor+fwd", 7), ("P4 backward", 8), ("P5 dz+update", 9), ("end of problem", 10), ("between problems", 11)]:
            print("    %-26s %12.0f  %5.1f%%" % (nm, out[i] / nwv, 100.0 * out[i] / tot))
        print("    total %.0f cycles/wave" % (tot / nwv))
        print("    backward sweep alone (inside P4 / P5)  %12.0f" % (out[12] / nwv))
        print("    C'nu (r_d, Phi^-1 r_d) %12.0f   C z, rhs %12.0f" % (out[13] / nwv, out[14] / nwv))
        ft = sum(out[i] for i in range(4)) or 1
        for nm, i in [("P3.images + B Rt^-1 B' + U'U (MFMA)", 0), ("P3.tiles -> LDS, row/col loads", 1), ("P3.fused potrf+trsm (VALU)", 2), ("P3.store + readback", 3)]:
            print("      %-38s %12.0f  %5.1f%% of P3" % (nm, out[i] / nwv, 100.0 * out[i] / ft))
    h.close()
if 1 not in res:
    sys.exit(0)
zw, zt = res[0][0], res[1][0]
print("wave vs tiled: max rel diff z %.3e  nu %.3e  steps equal %s  iters equal %s"
      % (np.abs(zw - zt).max() / np.abs(zt).max(), np.abs(res[0][1] - res[1][1]).max() / np.abs(res[1][1]).max(),
         np.array_equal(res[0][2], res[1][2]), np.array_equal(res[0][3], res[1][3])))
print("t < 1 in %d of %d steps" % (int(((res[1][2] < 1.0) & (res[1][2] >= 0)).sum()), int((res[1][2] >= 0).sum())))
