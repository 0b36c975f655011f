"""The structured C baseline (oracle/banded_cpu.c, bench.py's `cpu_baseline_structured`) against the structured numpy
oracle it restates (oracle/banded_ref.py, itself pinned to the dense restatement of the reference): <= 1e-10."""
import numpy as np
import pytest

from oracle import banded_cpu
from tests.util import canon_steps, oracle_batch, rel_err


@pytest.mark.parametrize("n,m,T,xf,var,nw,k,umax", [(8, 5, 10, False, 2, 5, 0.01, 2.0), (8, 5, 10, True, 2, 5, 0.01, 2.0),
                                                      (8, 5, 10, False, 1, 3, 0.01, 2.0), (8, 5, 10, False, 2, 8, 10.0, 0.3),
                                                      (6, 9, 1, False, 2, 2, 0.01, 2.0), (6, 9, 2, True, 2, 0, 0.01, 2.0)])
def test_banded_cpu_matches_banded_ref_demo(pkg, n, m, T, xf, var, nw, k, umax):
    model, data = pkg.synthetic.make_test_problem(n, m, T, seed=n + T, umax=umax, xf=xf, var_order=var, batch=4)
    z, nu, it, st, step = banded_cpu.solve_batch(model, data, nw, k, threads=2)
    zo, nuo, ito, sto, steps = oracle_batch(model, data, nw, k)
    assert np.array_equal(it, ito) and np.array_equal(st, sto)
    for p in range(4):
        assert rel_err(z[p], zo[p]) <= 1e-10 and rel_err(nu[p], nuo[p]) <= 1e-9
        assert np.array_equal(canon_steps(step[p][:ito[p]]), canon_steps(steps[p]))


def test_banded_cpu_ao_config_and_threads(pkg):
    model = pkg.synthetic.make_model(27, 144, 30)
    data = pkg.synthetic.make_replay_batch(model, r=1, steps=6)
    z1, _, it1, st1, _ = banded_cpu.solve_batch(model, data, 5, 1e-2, threads=1)
    z4, _, it4, _, _ = banded_cpu.solve_batch(model, data, 5, 1e-2, threads=4)
    assert np.array_equal(z1, z4) and np.array_equal(it1, it4)          # a problem never depends on the thread count
    zo, _, ito, sto, _ = oracle_batch(model, data, 5, 1e-2)
    assert np.array_equal(it1, ito) and np.array_equal(st1, sto)
    assert max(rel_err(z1[p], zo[p]) for p in range(6)) <= 1e-10
    model["u_min"] = -0.05 * np.ones(144); model["u_max"] = 0.05 * np.ones(144)
    z, _, it, st, _ = banded_cpu.solve_batch(model, data, 6, 1e-2)
    zo, _, ito, sto, _ = oracle_batch(model, data, 6, 1e-2)
    assert np.array_equal(it, ito) and np.array_equal(st, sto)
    assert max(rel_err(z[p], zo[p]) for p in range(6)) <= 1e-10
