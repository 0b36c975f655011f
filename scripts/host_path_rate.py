"""PCIe-inclusive rate of the host-pointer entry point fmpc_solve (numpy buffers in pageable host memory):
H2D of x0/x0_pre/nu0, the solve, D2H of z/nu/status/iters/step.  Reported in DESIGN.md; never bench `value`."""
import importlib, sys, time
sys.path.insert(0, '.')
import numpy as np
pkg = importlib.import_module('mpc-sensorlessao_amd')
model = pkg.synthetic.make_model(27, 144, 30)
data = pkg.synthetic.make_replay_batch(model, r=0, steps=2000)
h = pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"], model["u_max"], model["x_min"], model["x_max"], 30)
for _ in range(3):
    h.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=1, k=1e-2)
t0 = time.perf_counter(); K = 10
for _ in range(K):
    h.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=1, k=1e-2)
dt = (time.perf_counter() - t0) / K
print("host-pointer path fmpc_solve, batch 2000, n_newton 1: %.3f ms per call -> %.0f MPC steps/s (PCIe + pageable staging + ctypes included)" % (dt * 1e3, 2000 / dt))
