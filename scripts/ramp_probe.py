"""Timing of the ramp-rate path (fmpc_kernel_ramp.hip) at BASELINE configs[0]: VAR(1), n = 27, m = 144, T = 10.
Run on the GPU box:  python scripts/ramp_probe.py"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mpc-sensorlessao_amd")
n, m, T = 27, 144, int(sys.argv[1]) if len(sys.argv) > 1 else 10
md = pkg.synthetic.make_model(n, m, T, var_order=1)
dev = torch.device("cuda", 0)
h = pkg.FastMPCHandle(md["A1"], None, md["B"], md["Q"], md["R"], md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], T,
                      var_order=1, device=0)
du = 0.2121 * np.ones(m)
h.set_ramp(-du, du)
for B in (1, 64, 256, 512, 2000):
    data = pkg.synthetic.make_replay_batch(md, r=1, steps=B)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x0, nu0 = t(data["x0"]), t(data["nu0"][:, :T * n])
    up = t(0.05 * np.random.default_rng(0).standard_normal((B, m)))
    z = torch.empty((B, h.nz), dtype=torch.float64, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
    for nw in (1, 5):
        for _ in range(2):
            h.solve_device(x0, None, None, None, nu0, nw, 1e-2, z_out=z, status=st, iters=it, u_prev=up)
        torch.cuda.synchronize()
        K = 5
        t0 = time.perf_counter()
        for _ in range(K):
            h.solve_device(x0, None, None, None, nu0, nw, 1e-2, z_out=z, status=st, iters=it, u_prev=up)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        its = float(it.sum().item())
        print(f"T={T} batch {B:5d} n_newton {nw}: {dt * 1e3:8.3f} ms per solve, {B / dt:10.1f} MPC steps/s, "
              f"{its / B:.2f} Newton iterations per problem, {dt / max(its, 1) * B * 1e6 / B:8.1f} us per iteration (batch)",
              flush=True)

lib = pkg.load()
if hasattr(lib, "fmpc_debug_ramp_timing"):
    import ctypes as C
    out = (C.c_ulonglong * 16)()
    lib.fmpc_debug_ramp_timing(out)
    data = pkg.synthetic.make_replay_batch(md, r=1, steps=1)
    NR = 10
    for rep in range(3 + NR):                          # (three warm calls: the constants are in L2 in a running loop)
        if rep == 3:
            lib.fmpc_debug_ramp_timing(out)
        h.solve(data["x0"], None, None, nu0=data["nu0"][:, :T * n], n_newton=1, k=1e-2, u_prev=np.zeros((1, m)))
    lib.fmpc_debug_ramp_timing(out)
    for i in range(16):
        out[i] = out[i] // NR
    names = ["P1 residuals", "P2 tridiagonal LDL', inverse, rhs", "P3 Y assembly", "P4 Cholesky", "P4 substitutions", "P5 d_z, line search, update"]
    if h.last_dual_form() == 5:
        names = ["P0+P1 delta, rho, bhat, r_d, exit test", "P2 y_u0", "P3 tiles of [M | y_u0]", "P3 Cholesky + solve", "P4 pass through the constant operators", "P5 line search, outputs"]
    print("one problem, one Newton step, workgroup 0 (us):")
    for i, nm in enumerate(names):
        print("  %-36s %9.1f" % (nm, out[i] * 0.01))
    if h.last_dual_form() == 5:
        print("  of P4: B' reload + s + beta %.1f, nu+ = Yinv beta %.1f, kappa %.1f, d_u = phi + Gf kappa %.1f" % tuple(out[i] * 0.01 for i in (8, 9, 10, 11)))
        print("  of the Cholesky: pass A incl. the diagonal tiles %.1f, of which the 16-step factorisations of the diagonal tiles %.1f" % (out[6] * 0.01, out[7] * 0.01))
    else:
        print("  of P4: pass A + potrf %.1f, backward substitution %.1f" % (out[6] * 0.01, out[7] * 0.01))
