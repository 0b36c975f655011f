"""GPU tests of the dense form of the cold-start dual solve (fmpc_kernel_inv.hip): nu+ = nuc + J [x0; x0_pre; w] as one
product on the matrix cores instead of the two sweeps of fmpc_cold_panel through the block factor (both implement
inf_newton_solver.m:27-32 at the constant start of fast_mpc_init.m:19-20).  Checkers: the structured oracle
(oracle/banded_ref.py) and the sweep form of the same handle.  Tolerance: 1e-9 relative on z, 1e-7 on nu against the
oracle; 1e-11 between the two forms (same factor, different summation order)."""
import numpy as np
import pytest

from tests.util import canon_steps, handle_from_model, oracle_batch, rel_err

pytestmark = pytest.mark.gpu
TOL_Z, TOL_NU = 1e-9, 1e-7


def _case(pkg, T, batch, xf, use_w, use_nu, seed, var_order=2, m=144):
    md = pkg.synthetic.make_model(27, m, T, var_order=var_order)
    rng = np.random.default_rng(seed)
    if xf:
        md["xf"] = 0.01 * rng.standard_normal(27)
    data = pkg.synthetic.make_replay_batch(md, r=seed, steps=batch)
    data["w"] = 0.01 * rng.standard_normal((batch, T * 27)) if use_w else None
    data["nu0"] = rng.standard_normal((batch, (T + (1 if xf else 0)) * 27)) if use_nu else None
    if var_order == 1:
        data["x0_pre"] = None
    return md, data


@pytest.mark.parametrize("T,batch,xf,use_w,use_nu,var_order", [
    (30, 37, False, True, True, 2),      # few panels with w: 16 wavefronts split the k range
    (30, 16, True, True, False, 2),      # one panel, terminal row (nb = T + 1: more rows of nu than entries of w)
    (30, 100, False, True, True, 2),     # 7 panels: 4 wavefronts split the k range, odd number of panels
    (30, 512, False, True, False, 2),    # BASELINE configs[2]
    (30, 2000, False, False, True, 2),   # the headline workload: w = NULL, only [x0; x0_pre] enter
    (30, 33, True, False, False, 2),     # w = NULL with the terminal row
    (10, 40, False, True, True, 1),      # VAR(1): x0_pre does not enter
    (2, 17, False, True, True, 2),       # the horizon the README itself runs (README.md:338)
    (1, 5, True, True, True, 2),         # one stage + terminal row
    (7, 300, False, True, True, 2),      # T n not a multiple of 4 k-steps? (189 = 47.25 k-steps)
])
def test_dense_form_matches_oracle_and_sweeps(pkg, gpu, T, batch, xf, use_w, use_nu, var_order):
    md, data = _case(pkg, T, batch, xf, use_w, use_nu, seed=23, var_order=var_order)
    h = handle_from_model(pkg, md)
    args = (data["x0"], data["x0_pre"], data["w"])
    zd, idn = h.solve(*args, nu0=data["nu0"], n_newton=1, k=1e-2, return_info=True)
    assert h.last_dispatch()[0] == pkg.FMPC_PATH_PANEL and h.last_dual_form() >= 1
    h.set_dense_form(False)
    zs, isw = h.solve(*args, nu0=data["nu0"], n_newton=1, k=1e-2, return_info=True)
    assert h.last_dispatch()[0] == pkg.FMPC_PATH_PANEL and h.last_dual_form() == 0
    assert np.array_equal(idn["iters"], isw["iters"]) and np.array_equal(idn["status"], isw["status"])
    assert np.array_equal(idn["step"], isw["step"])
    nchk = min(batch, 48)
    sub = {k: (v[:nchk] if v is not None else None) for k, v in data.items()}
    zo, nuo, ito, sto, steps = oracle_batch(md, sub, 1, 1e-2)
    assert np.array_equal(idn["iters"][:nchk], ito) and np.array_equal(idn["status"][:nchk], sto)
    assert np.array_equal(canon_steps(idn["step"][:nchk, 0]), canon_steps([s[0] for s in steps]))
    for p in range(batch):
        assert rel_err(zd[p], zs[p]) <= 1e-11 and rel_err(idn["nu"][p], isw["nu"][p]) <= 1e-11
    for p in range(nchk):
        assert rel_err(zd[p], zo[p]) <= TOL_Z and rel_err(idn["nu"][p], nuo[p]) <= TOL_NU
    h.close()


def test_dense_form_is_independent_of_the_batch_it_runs_in(pkg, gpu):
    """A problem gives the same bits alone, in a ragged batch and at any position: the k range is split the same way for
    every launch shape of a variant, and partial tiles are added in a fixed order."""
    md, data = _case(pkg, 30, 64, False, True, True, seed=5)
    h = handle_from_model(pkg, md)
    z_all = h.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=1, k=1e-2)
    for lo, hi in [(0, 1), (17, 18), (3, 40), (48, 64)]:
        z = h.solve(data["x0"][lo:hi], data["x0_pre"][lo:hi], data["w"][lo:hi], nu0=data["nu0"][lo:hi], n_newton=1, k=1e-2)
        assert h.last_dual_form() >= 1
        assert np.array_equal(z, z_all[lo:hi])
    h.close()


def test_dense_form_bound_and_barrier_weight_changes(pkg, gpu):
    md, data = _case(pkg, 30, 40, False, True, True, seed=9)
    h = handle_from_model(pkg, md)
    h.set_dense_form(True, max_batch_with_w=32)
    h.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=1, k=1e-2)
    assert h.last_dual_form() == 0                                    # 40 problems with w: beyond the bound, the sweeps
    h.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=1, k=1e-2)
    assert h.last_dual_form() >= 1                                    # without w always
    h.set_dense_form(True, max_batch_with_w=1024)
    for k in (1e-2, 1.0, 1e-4, 1e-2):                                 # J is rebuilt per barrier weight
        z, info = h.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=1, k=k, return_info=True)
        assert h.last_dual_form() >= 1
        zo, nuo, ito, sto, _ = oracle_batch(md, data, 1, k)
        assert np.array_equal(info["iters"], ito) and np.array_equal(info["status"], sto)
        for p in range(40):
            assert rel_err(z[p], zo[p]) <= TOL_Z and rel_err(info["nu"][p], nuo[p]) <= TOL_NU
    h.close()


def test_dense_form_with_a_newton_budget(pkg, gpu):
    """Budget 5 with the exit test (test_fast_mpc.m:53,59): the dense form is the first step, the continuation is the
    exact path's."""
    md, data = _case(pkg, 30, 48, False, True, True, seed=31)
    h = handle_from_model(pkg, md)
    z, info = h.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=5, k=1e-2, return_info=True)
    assert h.last_dual_form() >= 1
    zo, nuo, ito, sto, _ = oracle_batch(md, data, 5, 1e-2)
    assert np.array_equal(info["iters"], ito) and np.array_equal(info["status"], sto)
    for p in range(48):
        assert rel_err(z[p], zo[p]) <= TOL_Z and rel_err(info["nu"][p], nuo[p]) <= TOL_NU
    h.close()


@pytest.mark.parametrize("T,batch,use_nu,var_order,m", [(30, 2000, True, 2, 144), (30, 37, False, 2, 144), (10, 40, True, 1, 144),
                                                       (2, 17, True, 2, 144), (3, 20, True, 2, 97), (1, 5, True, 2, 144)])
def test_dual_solve_fused_into_dz(pkg, gpu, T, batch, use_nu, var_order, m):
    """Experimental variant (FMPC_FUSE_DZ=1 at create time; measured slower): without w, without the terminal row
    and with a Newton budget of 1, d_z computes nu+ itself from [x0; x0_pre] (fmpc_cold_dz<.., true>: no nu+ round trip through
    HBM).  Against the default dense form (1e-11) and the oracle; nu_out comes from d_z in both."""
    import os
    md, data = _case(pkg, T, batch, False, False, use_nu, seed=41, var_order=var_order, m=m)
    os.environ["FMPC_NO_AFFINE"] = "1"         # both handles on the three-kernel form (the affine form would take these solves)
    try:
        hu = handle_from_model(pkg, md)
        os.environ["FMPC_FUSE_DZ"] = "1"
        try:
            hf = handle_from_model(pkg, md)
        finally:
            del os.environ["FMPC_FUSE_DZ"]
    finally:
        del os.environ["FMPC_NO_AFFINE"]
    zf, inf_ = hf.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=1, k=1e-2, return_info=True)
    zu, inu = hu.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=1, k=1e-2, return_info=True)
    assert hf.last_dual_form() == 1 and hu.last_dual_form() == 1
    assert np.array_equal(inf_["iters"], inu["iters"]) and np.array_equal(inf_["status"], inu["status"])
    assert np.array_equal(inf_["step"], inu["step"])
    for p in range(batch):
        assert rel_err(zf[p], zu[p]) <= 1e-11 and rel_err(inf_["nu"][p], inu["nu"][p]) <= 1e-11
    nchk = min(batch, 32)
    sub = {k: (v[:nchk] if v is not None else None) for k, v in data.items()}
    zo, nuo, ito, sto, steps = oracle_batch(md, sub, 1, 1e-2)
    assert np.array_equal(inf_["iters"][:nchk], ito) and np.array_equal(inf_["status"][:nchk], sto)
    for p in range(nchk):
        assert rel_err(zf[p], zo[p]) <= TOL_Z and rel_err(inf_["nu"][p], nuo[p]) <= TOL_NU
    hf.close(); hu.close()


@pytest.mark.parametrize("batch", [15, 17, 31, 33, 63, 65, 81, 257])
def test_dense_form_at_panel_and_shape_boundaries(pkg, gpu, batch):
    """Batch sizes either side of a panel (16 problems) and of the launch shapes of the product (4 / 16 / 64 panels), with w:
    every problem against the sweep form of the same handle (1e-11) and a sample against the oracle."""
    md, data = _case(pkg, 30, batch, False, True, True, seed=batch)
    h = handle_from_model(pkg, md)
    zd, idn = h.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=1, k=1e-2, return_info=True)
    assert h.last_dual_form() >= 1
    h.set_dense_form(False)
    zs, isw = h.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=1, k=1e-2, return_info=True)
    assert np.array_equal(idn["iters"], isw["iters"]) and np.array_equal(idn["step"], isw["step"])
    assert max(rel_err(zd[p], zs[p]) for p in range(batch)) <= 1e-11
    pick = sorted(set([0, batch // 2, batch - 1]))
    sub = {k: (v[pick] if v is not None else None) for k, v in data.items()}
    zo, nuo, ito, sto, _ = oracle_batch(md, sub, 1, 1e-2)
    for q, p in enumerate(pick):
        assert rel_err(zd[p], zo[q]) <= TOL_Z and rel_err(idn["nu"][p], nuo[q]) <= TOL_NU
    h.close()


def test_budget_continuation_on_the_tiled_kernel_with_handed_over_problems(pkg, gpu):
    """A tight box: the step-length decision hands many problems back, the others continue from the dense-form step; both
    kinds go through the compacted list to the tiled kernel (FtParams::list).  Against the oracle, and against the
    one-wavefront continuation (FMPC_NO_SMALL_TILED=1) to 1e-10."""
    import os
    md = pkg.synthetic.make_model(27, 144, 10)
    md["u_min"] = -0.05 * np.ones(144); md["u_max"] = 0.05 * np.ones(144)
    data = pkg.synthetic.make_replay_batch(md, r=8, steps=53)
    data["w"] = 0.01 * np.random.default_rng(8).standard_normal((53, 270))
    h = handle_from_model(pkg, md)
    z, info = h.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=6, k=1e-2, return_info=True)
    path, handed = h.last_dispatch()
    assert path == pkg.FMPC_PATH_PANEL and handed > 0 and h.last_dual_form() >= 1
    h.close()
    os.environ["FMPC_NO_SMALL_TILED"] = "1"
    try:
        hw = handle_from_model(pkg, md)
    finally:
        del os.environ["FMPC_NO_SMALL_TILED"]
    zw, iw = hw.solve(data["x0"], data["x0_pre"], data["w"], nu0=data["nu0"], n_newton=6, k=1e-2, return_info=True)
    hw.close()
    zo, nuo, ito, sto, steps = oracle_batch(md, data, 6, 1e-2)
    assert np.array_equal(info["iters"], ito) and np.array_equal(info["status"], sto)
    assert np.array_equal(iw["iters"], ito) and np.array_equal(iw["status"], sto)
    assert info["iters"].max() >= 3
    for p in range(53):
        assert rel_err(z[p], zo[p]) <= TOL_Z and rel_err(info["nu"][p], nuo[p]) <= TOL_NU
        assert rel_err(z[p], zw[p]) <= 1e-10
        assert np.array_equal(canon_steps(info["step"][p][:ito[p]]), canon_steps(steps[p]))
