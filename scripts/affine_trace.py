"""Timeline of one launch of the affine kernel (timing build: FMPC_LIB=.../libfastmpc_timing.so): per workgroup start / data staged /
first tile done / end, relative to the earliest start."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
lib = pkg.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
md = pkg.synthetic.make_model(27, 144, 30)
data = pkg.synthetic.make_replay_batch(md, r=0, steps=B)
dev = torch.device("cuda:0")
h = pkg.FastMPCHandle(md["A1"], md["A2"], md["B"], md["Q"], md["R"], md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], 30)
x0 = torch.from_numpy(data["x0"]).to(dev); x0p = torch.from_numpy(data["x0_pre"]).to(dev); nu0 = torch.from_numpy(data["nu0"]).to(dev)
z = torch.empty((B, h.nz), dtype=torch.float64, device=dev); u0 = torch.empty((B, h.m), dtype=torch.float64, device=dev)
st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
for want_z in (True, False):
    for _ in range(4):
        h.solve_device(x0, x0p, None, None, nu0, 1, 1e-2, z_out=z if want_z else None, status=st, iters=it, u0_out=u0, want_z=want_z)
        torch.cuda.synchronize()
    out = (C.c_ulonglong * 8192)()
    lib.fmpc_debug_affine_trace.argtypes = [C.c_void_p]; lib.fmpc_debug_affine_trace(out)
    t = np.array(out[:], dtype=np.int64).reshape(1024, 8)
    if hasattr(lib, "fmpc_debug_affine_trace_cycles"):
        outc = (C.c_ulonglong * 8192)()
        lib.fmpc_debug_affine_trace_cycles.argtypes = [C.c_void_p]; lib.fmpc_debug_affine_trace_cycles(outc)
        tc = np.array(outc[:], dtype=np.int64).reshape(1024, 8)
        u = t[:, 0] > 0
        def mhz(a, b):
            dw = (t[u, b] - t[u, a]) * 0.01; dc = (tc[u, b] - tc[u, a]).astype(np.float64)
            ok = dw > 0.3
            return float(np.median(dc[ok] / dw[ok])) if ok.any() else float("nan")
        print("   shader-clock counter per us of wall clock: start->staged %.0f, staged->first operand %.0f, first tile %.0f, first tile done->end %.0f"
              % (mhz(0, 1), mhz(1, 5), mhz(5, 6), mhz(6, 3)))
    nf = 0
    used = t[:, 0] > 0
    t0 = t[used, 0].min()
    rel = (t - t0) * 0.01                                   # us
    k = rel[used]
    md_ = lambda a: float(np.median(a))
    print("   medians: staged %.1f, D read %.1f, A there %.1f, first tile computed+stores issued %.1f, next A there %.1f, end %.1f" % (md_(k[:, 1]), md_(k[:, 4]), md_(k[:, 5]), md_(k[:, 6]), md_(k[:, 2]), md_(k[:, 3])))
    print("   %d product workgroups: start %.1f..%.1f (median %.1f), staged +%.1f (median), first tile +%.1f (median after staged), end %.1f..%.1f (median %.1f)"
          % (len(k), k[:, 0].min(), k[:, 0].max(), np.median(k[:, 0]), np.median(k[:, 1] - k[:, 0]), np.median(k[:, 2] - k[:, 1]), k[:, 3].min(), k[:, 3].max(), np.median(k[:, 3])))
    if want_z:
        wpg = 16
        byslot = [float(np.median(k[np.arange(len(k)) % wpg == sl, 3])) for sl in range(wpg)]
        print("   median end by workgroup slot in its group:", " ".join("%.1f" % v for v in byslot))
        bygrp = [float(np.median(k[(np.arange(len(k)) // wpg) == gq, 3])) for gq in range(len(k) // wpg)]
        print("   median end by group:", " ".join("%.1f" % v for v in bygrp))
        st0 = k[:, 0]
        print("   start by slot:", " ".join("%.1f" % float(np.median(st0[np.arange(len(k)) % wpg == sl])) for sl in range(wpg)))
