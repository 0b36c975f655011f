"""Exports tests/golden/*.npz (outputs of the dense restatement oracle/dense_ref.py -- PARITY UNPINNED, see tests/golden/
make_golden.py) as MATLAB v5 .mat files under matlab/fixtures/, for matlab/verify_against_reference.m: a maintainer WITH
MATLAB can then run the reference's own Fast_MPC2 on the same inputs and the same nu0 and close the parity pin that this
build image (no MATLAB, no Octave) cannot.  Run from the repo root:  python matlab/make_fixtures.py"""
import glob
import os

import numpy as np
from scipy.io import savemat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "matlab", "fixtures")

for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
    d = np.load(path)
    n, m, T, var_order, nw, has_xf = (int(v) for v in d["meta"])
    B = d["x0"].shape[0]
    col = lambda a: np.asarray(a, dtype=np.float64).reshape(-1, 1)
    out = {"n": float(n), "m": float(m), "T": float(T), "var_order": float(var_order), "nw": float(nw), "k": float(d["k"][0]),
           "num_problems": float(B)}
    for key in ("A1", "A2", "B", "Q", "R", "Qf"):
        out[key] = np.asarray(d["model_" + key], dtype=np.float64)
    for key in ("u_min", "u_max", "x_min", "x_max"):
        out[key] = col(d["model_" + key])
    out["xf"] = col(d["model_xf"]) if has_xf else np.zeros((0, 0))
    # per problem, one column each (MATLAB column vectors)
    out["x0"] = d["x0"].T.copy(); out["x0_pre"] = d["x0_pre"].T.copy(); out["nu0"] = d["nu0"].T.copy()
    out["w"] = (d["w"].T.copy() if "w" in d.files else np.zeros((T * n, B)))       # T*n entries as the reference indexes them (quirk D7)
    out["x_init"] = d["z_init"].T.copy() if "z_init" in d.files else np.zeros((0, 0))
    out["z_expected"] = d["z"].T.copy(); out["nu_expected"] = d["nu"].T.copy()
    out["iters_expected"] = d["iters"].astype(np.float64).reshape(1, -1)
    out["steps_expected"] = d["steps"].T.copy()                                    # accepted t per Newton step, -1 = unused
    name = os.path.splitext(os.path.basename(path))[0]
    savemat(os.path.join(OUT, name + ".mat"), out, format="5", do_compression=True, oned_as="column")
    print(name, "->", os.path.getsize(os.path.join(OUT, name + ".mat")), "bytes")
