"""Closed-loop step time (one fmpc_loop_step_device per step, sequential): realisations x {keep z, first moves only}.
    python3 scripts/closed_loop_perf.py [steps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
model = pkg.synthetic.make_model(27, 144, 30)
mk = lambda: pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"], model["u_max"],
                               model["x_min"], model["x_max"], 30)
for R in (1, 64, 512, 2048):
    a = np.stack([pkg.synthetic.make_realisation(model, r=r, steps=steps)[1:steps + 1] for r in range(min(R, 8))], axis=1)
    a = np.ascontiguousarray(np.tile(a, (1, (R + a.shape[1] - 1) // a.shape[1], 1))[:, :R])
    at = torch.from_numpy(a).to(dev)
    for keep_z, env in ((True, None), (False, None), (False, "FMPC_NO_FIRST_MOVE")):
        if env:
            os.environ[env] = "1"
        h = mk()
        if env:
            os.environ.pop(env)
        for rep in range(2):
            loop = pkg.ClosedLoop(h, R, n_newton=1, k=1e-2, keep_z=keep_z)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(steps):
                loop.step(at[s])
            t_enq = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        assert int(loop.status.abs().sum()) == 0
        print("realisations %4d  %-34s %.2f us per loop step  (%.3g MPC steps/s; host enqueue %.2f us per step), handed over in the last step: %d"
              % (R, "keep z" if keep_z else ("first moves only" + (" (four-launch form)" if env else "")), dt / steps * 1e6, R * steps / dt, t_enq / steps * 1e6, h.last_dispatch()[1]))
        h.close()

    h = mk()
    for lsteps in (steps, 4 * steps):
        al = at.repeat((lsteps + steps - 1) // steps, 1, 1)[:lsteps].contiguous()
        for rep in range(2):
            loop = pkg.ClosedLoop(h, R, n_newton=1, k=1e-2, keep_z=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop.run_recorded(al, want_x0=False)
            t_enq = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("realisations %4d  %-34s %.2f us per loop step  (%.3g MPC steps/s; %d steps in %.3f ms)"
              % (R, "recorded stretch, one host call", dt / lsteps * 1e6, R * lsteps / dt, lsteps, dt * 1e3))
    h.close()
