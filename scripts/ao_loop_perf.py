"""Time per timestep of the loop with its estimator (AOLoop) at the reference's size (len 512): one C call for b_ref + fastMPC
(fmpc_ao_step_device) against loop inputs + solve as two.   python3 scripts/ao_loop_perf.py [realisations] [steps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
op = pkg.synthetic.estimator_optics(512)
md = pkg.synthetic.make_model(27, 144, 30)
h = pkg.FastMPCHandle(md["A1"], md["A2"], md["B"], md["Q"], md["R"], md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], 30)
est = pkg.PhaseDiversityEstimator(op["pupil"], op["W"], op["zd_list"], op["dx"], op["range_min"] + 1, op["range_max"] + 1, op["A_s"], op["b_s"])
ph = torch.from_numpy(0.2 * np.random.default_rng(0).standard_normal((R, 512, 512))).to(dev)
for one_call, cm in ((True, True), (True, False), (False, False)):
    for rep in range(2):
        loop = pkg.AOLoop(h, est, op["Z"][1:], R, n_newton=1, k=1e-2, one_call=one_call)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(steps):
            loop.step(ph, colmajor=cm)
        te = time.perf_counter() - t0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("realisations %d, %s%s: %.1f us per timestep (host enqueue %.1f us; dual form %d)" % (R, "one call" if one_call else "two calls", ", screens handed over column-major" if cm else "", dt / steps * 1e6, te / steps * 1e6, h.last_dual_form()))
