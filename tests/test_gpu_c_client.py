"""The drop-in boundary from a compiled language: examples/c_client.c (plain C, gcc, links libfastmpc.so only) issues the
reference's per-timestep call -- Fast_MPC2(...) then mpc_fixed_log_newton(nw, k) (VAR_2/Fast_MPC2.m:28-29,124-130) -- as one
fmpc_solve_once with MATLAB's column-major arrays.  Its x_opt against the oracle (1e-9) and the Python binding (1e-11)."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from tests.util import handle_from_model, oracle_batch, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,m,T,nw", [(27, 144, 30, 1), (8, 5, 10, 4)])
def test_plain_c_client_of_the_c_abi(pkg, gpu, tmp_path, n, m, T, nw):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    lib_dir = os.path.join(ROOT, "mpc-sensorlessao_amd", "lib")
    exe = str(tmp_path / "c_client")
    subprocess.run(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_client.c"), "-o", exe,
                    "-L" + lib_dir, "-lfastmpc", "-Wl,-rpath," + lib_dir, "-lm"], check=True)
    if n == 27:
        md = pkg.synthetic.make_model(n, m, T)
        data = pkg.synthetic.make_replay_batch(md, r=2, steps=1)
        data["w"] = 0.01 * np.random.default_rng(0).standard_normal((1, T * n))
    else:
        md, data = pkg.synthetic.make_test_problem(n, m, T, seed=4, batch=1)
    k = 1e-2
    cm = lambda a: np.asfortranarray(np.asarray(a, dtype=np.float64)).ravel(order="F")     # MATLAB storage
    with open(tmp_path / "model.bin", "wb") as f:
        f.write(struct.pack("<4id", n, m, T, nw, k))
        for a in (md["Q"], md["R"], md["Qf"], md["x_min"], md["x_max"], md["u_min"], md["u_max"], data["x0"][0], data["x0_pre"][0],
                  md["A1"], md["A2"], md["B"], data["w"][0], data["nu0"][0]):
            f.write(cm(a).tobytes())
    env = dict(os.environ)
    subprocess.run([exe, str(tmp_path / "model.bin"), str(tmp_path / "out.bin")], check=True, env=env, timeout=300)
    raw = open(tmp_path / "out.bin", "rb").read()
    rc, iters = struct.unpack("<2i", raw[:8])
    z = np.frombuffer(raw[8:], dtype=np.float64)
    assert rc == 0 and z.shape == (T * (n + m),)
    zo, _, ito, sto, _ = oracle_batch(md, data, nw, k)
    assert iters == ito[0] and sto[0] == 0
    assert rel_err(z, zo[0]) <= 1e-9
    h = handle_from_model(pkg, md)
    zp = h.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=nw, k=k)
    h.close()
    assert rel_err(z, zp[0]) <= 1e-11
