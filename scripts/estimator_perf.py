"""Device time of the estimator (README.md:456-480) at the reference's size (len 512, 31 x 31 window, three diversities).
    python3 scripts/estimator_perf.py [batch ...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
op = pkg.synthetic.estimator_optics(512)
est = pkg.PhaseDiversityEstimator(op["pupil"], op["W"], op["zd_list"], op["dx"], op["range_min"] + 1, op["range_max"] + 1, op["A_s"], op["b_s"])
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
flops = 3 * (6 * 512 * 512 * 32 + 8 * 32 * 512 * 32)          # executed: three real products per complex one in the first (2 flops each), four in the second, per diversity, padded window
for B in [int(a) for a in sys.argv[1:]] or [1, 8, 64, 256]:
    scr = torch.from_numpy(0.3 * rng.standard_normal((B, 512, 512))).to(dev)
    for _ in range(3):
        est.apply_device(scr, colmajor=True)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        est.apply_device(scr, colmajor=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("batch %4d: %.3f ms per call, %.1f us per screen, %.1f TFLOP/s executed (%.2f of fp64 peak), %.0f GB/s of screens"
          % (B, ms, ms / B * 1e3, flops * B / ms / 1e9, flops * B / ms / 1e9 / 78.6, B * 512 * 512 * 8 / ms / 1e6))
