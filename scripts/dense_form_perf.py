"""Device time of one cold-start solve (n_newton = 1), dense form of the dual solve against the sweeps of the panel
kernel, per batch size and with / without w.  Usage: python3 scripts/dense_form_perf.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import importlib
import numpy as np
import torch

pkg = importlib.import_module("mpc-sensorlessao_amd")
from tests.util import handle_from_model

md = pkg.synthetic.make_model(27, 144, 30)
h = handle_from_model(pkg, md)
dev = torch.device("cuda:0")
for batch in (1, 16, 64, 128, 512, 1024, 2000, 4096):
    data = pkg.synthetic.make_replay_batch(md, r=1, steps=batch)
    rng = np.random.default_rng(3)
    x0 = torch.tensor(data["x0"], device=dev); x0p = torch.tensor(data["x0_pre"], device=dev)
    nu0 = torch.tensor(data["nu0"], device=dev)
    wt = torch.tensor(0.01 * rng.standard_normal((batch, 30 * 27)), device=dev)
    z = torch.empty((batch, 30 * 171), device=dev, dtype=torch.float64)
    u0 = torch.empty((batch, 144), device=dev, dtype=torch.float64)
    for w in (None, wt):
        res = []
        for dense in (1, 0):
            h.set_dense_form(dense, 1 << 30)
            for _ in range(5):
                h.solve_device(x0, x0p, w, nu0=nu0, n_newton=1, k=1e-2, z_out=z, u0_out=u0)
            torch.cuda.synchronize()
            reps = 50
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                h.solve_device(x0, x0p, w, nu0=nu0, n_newton=1, k=1e-2, z_out=z, u0_out=u0)
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / reps * 1e3)
            assert h.last_dual_form() == dense
        print("batch %5d  w %-5s  dense %8.1f us   sweeps %8.1f us   ratio %.2f" % (batch, "yes" if w is not None else "NULL", res[0], res[1], res[1] / res[0]), flush=True)
