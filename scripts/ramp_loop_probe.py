"""configs[0] as the reference runs it: 200 sequential closed-loop steps of ONE realisation with the ramp-rate rows on (bench.py's
`config0_var1_ramp.closed_loop_200_sequential_steps` leg alone).  python3 scripts/ramp_loop_probe.py"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mpc-sensorlessao_amd")
n, m, T0 = 27, 144, 10
m0 = pkg.synthetic.make_model(n, m, T0, var_order=1)
dev = torch.device("cuda", 0)
h0 = pkg.FastMPCHandle(m0["A1"], None, m0["B"], m0["Q"], m0["R"], m0["Qf"], m0["u_min"], m0["u_max"], m0["x_min"], m0["x_max"], T0, var_order=1, device=0)
h0.set_ramp(-0.2121 * np.ones(m), 0.2121 * np.ones(m))
a0 = pkg.synthetic.make_realisation(m0, r=0, steps=201)[1:201]
ta0 = torch.from_numpy(np.ascontiguousarray(a0[:, None, :])).to(dev)
for rep in range(3):
    loop0 = pkg.ClosedLoop(h0, 1, n_newton=1, k=1e-2, ramp=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for s_ in range(200):
        loop0.step(ta0[s_])
    th = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    print("200 sequential steps: %.3f ms per loop step (host submission alone %.3f ms), form %d, status sum %d" % (dt / 200 * 1e3, th / 200 * 1e3, h0.last_dual_form(), int(loop0.status.abs().sum())))
lib = pkg.load()
if hasattr(lib, "fmpc_debug_ramp_timing"):
    import ctypes as C
    out = (C.c_ulonglong * 16)()
    lib.fmpc_debug_ramp_timing(out)
    loop0 = pkg.ClosedLoop(h0, 1, n_newton=1, k=1e-2, ramp=True)
    for s_ in range(40):
        loop0.step(ta0[s_])
    torch.cuda.synchronize(dev)
    lib.fmpc_debug_ramp_timing(out)
    for s_ in range(40, 200):
        loop0.step(ta0[s_])
    torch.cuda.synchronize(dev)
    lib.fmpc_debug_ramp_timing(out)
    names = ["P0+P1", "P2 y_u0", "P3 tiles", "P3 Cholesky + solve", "(P4 total: unused)", "P5 line search, outputs", "chol: pass A", "chol: potrf16 of diagonal tiles",
             "P4 s + beta", "P4 nu+ = Yinv beta", "P4 kappa", "P4 d_u = phi + Gf kappa"]
    print("steady state, per step (us):", ", ".join("%s %.1f" % (nm, out[i] * 0.01 / 160) for i, nm in enumerate(names)))
