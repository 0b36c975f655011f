"""Diagnostic: per-phase cycle totals of the wave kernel (needs `make -C mpc-sensorlessao_amd/csrc timing`).
Run with FMPC_LIB=mpc-sensorlessao_amd/lib/libfastmpc_timing.so."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, '.')
os.environ.setdefault("FMPC_LIB", os.path.abspath("mpc-sensorlessao_amd/lib/libfastmpc_timing.so"))
import numpy as np, torch
pkg = importlib.import_module('mpc-sensorlessao_amd')
lib = pkg.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 5
model = pkg.synthetic.make_model(27, 144, 30)
data = pkg.synthetic.make_replay_batch(model, r=0, steps=B)
h = pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"], model["u_max"], model["x_min"], model["x_max"], 30)
dev = torch.device('cuda:0')
x0 = torch.from_numpy(data["x0"]).to(dev); x0p = torch.from_numpy(data["x0_pre"]).to(dev); nu0 = torch.from_numpy(data["nu0"]).to(dev)
out = (C.c_ulonglong * 16)()
for rep in range(3):
    h.solve_device(x0, x0p, None, None, nu0, nw, 1e-2); torch.cuda.synchronize()
    lib.fmpc_debug_timing(out)
    if rep < 2 and hasattr(lib, "fmpc_debug_panel_timing"):
        lib.fmpc_debug_panel_timing((C.c_ulonglong * 16)())
tot = sum(out[i] for i in range(4, 12)) or 1
nw_ = min(B, 2048)
print("per-wave-average cycles (batch %d, n_newton %d):" % (B, nw))
for nm, i in [("P0 init", 4), ("P1 residuals (C'nu, Cz)", 5), ("P2 rhs", 6), ("P3 factor+fwd | shared fwd sweep", 7), ("P4 backward | shared bwd sweep", 8), ("P5 dz+update", 9), ("between problems", 11)]:
    print("  %-26s %12.0f  %5.1f%%" % (nm, out[i] / nw_, 100.0 * out[i] / tot))
print("  total %.0f cycles/wave" % (tot / nw_))
ft = sum(out[i] for i in range(4))
if ft:       # per-stage sections of the per-problem factor phase (general path only)
    for nm, i in [("P3.images + B Rt^-1 B' + U'U (MFMA)", 0), ("P3.tiles -> LDS, row/col loads", 1), ("P3.fused potrf+trsm (VALU)", 2), ("P3.store + readback", 3)]:
        print("    %-38s %12.0f  %5.1f%% of P3" % (nm, out[i] / nw_, 100.0 * out[i] / ft))
if nw == 1 and hasattr(lib, "fmpc_debug_panel_timing"):
    lib.fmpc_debug_panel_timing(out)
    npan = (B + 15) // 16
    nwav = min(npan, 256) * 8
    tot = sum(out[i] for i in range(6))
    if tot:
        print("panel kernel (dual solve), per-wave-average cycles (%d panels):" % npan)
        for nm, i in [("S1 own work (before barrier)", 4), ("S1 barrier wait", 0), ("S2 forward sweep", 1), ("S3 Linv' y", 2), ("S4 backward sweep", 3), ("nu+ write-out", 5)]:
            print("  %-26s %12.0f  %5.1f%%" % (nm, out[i] / nwav, 100.0 * out[i] / tot))
        print("  total %.0f cycles/wave" % (tot / nwav))
        print("  S1 detail per wave: setup + lower bound of ||r_d|| %.0f, stages 0/1 %.0f, stages >= 2 %.0f" % tuple(out[8 + i] / nwav for i in range(3)))
if nw == 1 and hasattr(lib, "fmpc_debug_dz_trace"):
    import numpy as np
    nwav = min(8192, 8 * ((((B + 15) // 16 + 7) // 8 * 30 + 7) // 8) * 8)
    tr = np.zeros(4 * nwav, dtype=np.uint64)
    lib.fmpc_debug_dz_trace(tr.ctypes.data_as(C.c_void_p), nwav)
    tr = tr.reshape(-1, 4).astype(np.float64)
    tr = tr[tr[:, 0] > 0]
    t0 = tr[:, 0].min()
    tr = (tr - t0) * 0.01                      # us (100 MHz)
    q = lambda a: "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
    print("d_z trace of the last launch (%d waves), us since the first wave started:" % len(tr))
    print("  wave start          ", q(tr[:, 0]))
    print("  LDS image ready     ", q(tr[:, 1]))
    print("  loads issued        ", q(tr[:, 2]))
    print("  done                ", q(tr[:, 3]))
    print("  wave life time      ", q(tr[:, 3] - tr[:, 0]))
    print("  compute+store phase ", q(tr[:, 3] - tr[:, 2]))
    if os.environ.get("FMPC_TRACE_DETAIL"):
        raw = np.zeros(4 * nwav, dtype=np.uint64)
        lib.fmpc_debug_dz_trace(raw.ctypes.data_as(C.c_void_p), nwav)
        raw = raw.reshape(-1, 4).astype(np.float64)
        ok = raw[:, 0] > 0
        wid = np.arange(len(raw))[ok]; rr = (raw[ok] - t0) * 0.01
        blk = wid // 8
        for x in range(8):
            sel = (blk & 7) == x
            print("   XCD %d: loads done median %.1f p90 %.1f max %.1f | done median %.1f max %.1f" % (x, np.median(rr[sel, 2]), np.percentile(rr[sel, 2], 90), rr[sel, 2].max(), np.median(rr[sel, 3]), rr[sel, 3].max()))
        nwpw = 8
        for x in range(nwpw):
            sel = (wid % nwpw) == x
            print("   wave %2d of its workgroup: loads done median %.1f p90 %.1f | done median %.1f" % (x, np.median(rr[sel, 2]), np.percentile(rr[sel, 2], 90), np.median(rr[sel, 3])))
        late = rr[:, 2] > 9.0
        print("   late waves (loads done > 9 us): %d; their block ids (first 40):" % late.sum(), np.unique(blk[late])[:40])
        print("   their stage j = task %% 30 histogram:", np.bincount((((blk[late] >> 3) * 8 + (wid[late] & 7)) % 30), minlength=30))

