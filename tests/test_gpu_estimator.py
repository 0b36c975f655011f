"""GPU tests of the phase-diversity estimator (fmpc_kernel_estimator.hip, fmpc_est_*; README.md:456-480): the PSF windows as
partial DFTs on the matrix cores + ad_est = G (Y_M - b_s), against the numpy restatement of the reference's FFT-based code
(oracle/estimator_ref.py) on synthetic optics (the reference's Zs.mat / SNR_10.mat are not shipped; its model_approx.mat is: tests/test_golden_model_approx.py).
Tolerances: 1e-10 relative on Y_M (fp64 both ways; a 512-point sum against an FFT), 1e-8 on ad_est."""
import numpy as np
import pytest

from oracle import estimator_ref as er
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _estimator(pkg, op):
    return pkg.PhaseDiversityEstimator(op["pupil"], op["W"], op["zd_list"], op["dx"], op["range_min"] + 1, op["range_max"] + 1,
                                       op["A_s"], op["b_s"], AU=op["AU"])


def _screens(op, batch, seed, strength=0.3, rough=0.05):
    """Residual screens: a few Zernike modes + pixel noise (the estimator must not rely on the screen being in the span of Z)."""
    rng = np.random.default_rng(seed)
    al = strength * rng.standard_normal((batch, op["nx"])) / np.sqrt(op["nx"])
    scr = np.tensordot(al, op["Z"][1:], axes=1) + rough * rng.standard_normal((batch, op["len"], op["len"]))
    return al, scr


@pytest.mark.parametrize("length,batch,with_noise", [(128, 5, True), (256, 3, False), (64, 2, True)])
def test_estimator_matches_the_fft_restatement(pkg, gpu, length, batch, with_noise):
    import torch
    op = pkg.synthetic.estimator_optics(length)
    est = _estimator(pkg, op)
    assert est.rank == op["nx"] and est.d == op["d"] and est.p == 3 * op["d"] ** 2
    _, scr = _screens(op, batch, seed=length)
    rng = np.random.default_rng(1)
    noise = 1e-3 * np.abs(op["b_s"]).max() * rng.standard_normal((batch, est.p)) if with_noise else None
    dev = torch.device("cuda:0")
    ad, Y = est.apply_device(torch.from_numpy(scr).to(dev), None if noise is None else torch.from_numpy(noise).to(dev), want_Y=True)
    torch.cuda.synchronize()
    ad, Y = ad.cpu().numpy(), Y.cpu().numpy()
    for b in range(batch):
        ad_o, Y_o = er.estimator_step(scr[b], er.pupil_mask(length, op["dx"]), op["W"], op["zd_list"], op["dx"], op["A_s"], op["b_s"],
                                      None if noise is None else noise[b], AU=op["AU"])
        assert rel_err(Y[b], Y_o) <= 1e-10, (b, rel_err(Y[b], Y_o))
        assert rel_err(ad[b], ad_o) <= 1e-8, (b, rel_err(ad[b], ad_o))
    # host-pointer entry: same numbers
    ad_h, Y_h = est.apply(scr, noise, want_Y=True)
    assert np.array_equal(ad_h, ad) and np.array_equal(Y_h, Y)
    est.close()


def test_estimator_at_the_reference_size_and_recovers_small_aberrations(pkg, gpu):
    """len = 512, window 31 x 31, three diversities (README.md:237, 378-380, 396): against the FFT restatement, and the linear
    model does what it is for -- a small aberration in the span of the modes comes back to first order."""
    import torch
    op = pkg.synthetic.estimator_optics(512)
    assert op["d"] == 31 and (op["range_min"], op["range_max"]) == er.window_range(512, op["dx"])
    est = _estimator(pkg, op)
    rng = np.random.default_rng(7)
    al = 0.02 * rng.standard_normal((4, op["nx"]))
    scr = np.tensordot(al, op["Z"][1:], axes=1)
    ad = est.apply_device(torch.from_numpy(scr).to(torch.device("cuda:0")))
    torch.cuda.synchronize()
    ad = ad.cpu().numpy()
    for b in range(2):
        ad_o, _ = er.estimator_step(scr[b], op["pupil"], op["W"], op["zd_list"], op["dx"], op["A_s"], op["b_s"], AU=op["AU"])
        assert rel_err(ad[b], ad_o) <= 1e-8
    assert max(rel_err(ad[b], al[b]) for b in range(4)) <= 0.1          # second-order terms of the image model
    # a result does not depend on its position in the batch (the launch shape -- 4 or 8 wavefronts per 16 rows -- depends on the
    # batch SIZE: another summation order, the same numbers to rounding)
    ad1 = est.apply_device(torch.from_numpy(scr[2:4]).to(torch.device("cuda:0"))).cpu().numpy()
    assert np.array_equal(ad1[0], ad[2]) and np.array_equal(ad1[1], ad[3])
    big = np.concatenate([scr, scr, scr, scr, scr])                                  # 20 screens: the 4-wavefront shape
    adb = est.apply_device(torch.from_numpy(big).to(torch.device("cuda:0"))).cpu().numpy()
    assert np.array_equal(adb[:4], adb[16:]) and max(rel_err(adb[b], ad[b]) for b in range(4)) <= 1e-11
    est.close()


def test_estimator_with_large_and_out_of_range_phases(pkg, gpu):
    """The PSF kernel reduces the phase by pi/2 itself (three-part constant, exact for |phase| < 1e6 rad): screens of thousands of
    radians -- many turns of the argument reduction -- against the FFT restatement; a pixel beyond the range gives NaN, not a
    wrong number."""
    op = pkg.synthetic.estimator_optics(64)
    est = _estimator(pkg, op)
    rng = np.random.default_rng(12)
    scr = _screens(op, 2, seed=3)[1] + 2.0e3 * rng.standard_normal((2, 64, 64)) * op["pupil"]
    ad, Y = est.apply(scr, want_Y=True)
    for b in range(2):
        ado, Yo = er.estimator_step(scr[b], op["pupil"], op["W"], op["zd_list"], op["dx"], op["A_s"], op["b_s"], AU=op["AU"])
        assert rel_err(Y[b], Yo) <= 1e-9 and rel_err(ad[b], ado) <= 1e-7
    bad = scr.copy(); bad[1, 32, 32] = 3.0e6
    adb = est.apply(bad)
    assert np.all(np.isfinite(adb[0])) and np.all(np.isnan(adb[1]))
    est.close()


def test_estimator_argument_checks(pkg, gpu):
    import ctypes as C
    lib = pkg.load()
    h = C.c_void_p()
    z = np.zeros(64 * 64 * 3)
    p_ = z.ctypes.data_as(C.c_void_p)
    assert lib.fmpc_est_create(C.byref(h), 100, 0, 31, 3, p_, p_, 1.0, p_, p_, 3 * 31 * 31, 27, 0) == pkg.FMPC_E_DIM      # len % 64
    assert lib.fmpc_est_create(C.byref(h), 64, 40, 31, 3, p_, p_, 1.0, p_, p_, 3 * 31 * 31, 27, 0) == pkg.FMPC_E_DIM       # window outside
    assert lib.fmpc_est_create(C.byref(h), 64, 0, 31, 3, p_, p_, 1.0, p_, p_, 100, 27, 0) == pkg.FMPC_E_DIM                # p != ndiv d^2
    assert lib.fmpc_est_create(C.byref(h), 64, 0, 31, 3, None, p_, 1.0, p_, p_, 3 * 31 * 31, 27, 0) == pkg.FMPC_E_NULL
    assert lib.fmpc_est_apply_device(None, 1, p_, None, p_, None, None) == pkg.FMPC_E_NULL


def test_phase_residual_kernel_against_numpy(pkg, gpu):
    """README.md:453, 590-601: out = phase + sum_j (B u)_j Z_j; u = NULL: out = phase.  11 screens (two groups of eight)."""
    import ctypes as C
    import torch
    from tests.util import handle_from_model
    md = pkg.synthetic.make_model(27, 144, 10)
    h = handle_from_model(pkg, md)
    rng = np.random.default_rng(2)
    batch, npx = 11, 64 * 64 + 37                                                     # not a multiple of the workgroup's 256 pixels
    phase = rng.standard_normal((batch, npx)); u = rng.standard_normal((batch, 144)); Z = rng.standard_normal((27, npx))
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    tp, tu, tz = t(phase), t(u), t(Z)
    out = torch.empty_like(tp)
    vp = lambda x: None if x is None else C.c_void_p(x.data_ptr())
    assert h._lib.fmpc_phase_residual_device(h._h, batch, npx, vp(tp), vp(tu), vp(tz), vp(out), None) == 0
    torch.cuda.synchronize()
    ref = phase + (u @ md["B"].T) @ Z
    assert rel_err(out.cpu().numpy(), ref) <= 1e-13
    assert h._lib.fmpc_phase_residual_device(h._h, batch, npx, vp(tp), None, None, vp(out), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), phase)
    h.close()


@pytest.mark.parametrize("one_call,R", [(True, 2), (False, 2), (True, 70)])
def test_simulation_loop_with_the_estimator_against_the_numpy_loop(pkg, gpu, one_call, R):
    """README.md:444-626 for two realisations over six steps at len = 128: residual screen, estimator, fastMPC, correction.
    Turbulence screens in and beyond the span of the modes; the deformable mirror's B scaled so that the loop's correction
    stays in the estimator's linear range.  Against oracle/ao_loop_ref.py: 1e-7 on the estimates and the first moves after
    six fed-back steps.  one_call: b_ref and the fastMPC step through fmpc_ao_step_device (first-move form: one launch for up to
    64 realisations, the product form beyond), otherwise loop inputs + solve as two calls; 70 realisations: three against the
    numpy loop."""
    import torch
    from oracle.ao_loop_ref import ao_loop
    from tests.util import handle_from_model
    length, steps = 128, 6
    op = pkg.synthetic.estimator_optics(length)
    md = pkg.synthetic.make_model(27, 144, 10)
    rng = np.random.default_rng(4)
    a = np.stack([0.03 * pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(R)], axis=1)   # (steps, R, n)
    phase = np.tensordot(a, op["Z"][1:], axes=1) + 0.01 * rng.standard_normal((steps, R, length, length))
    h = handle_from_model(pkg, md)
    est = _estimator(pkg, op)
    loop = pkg.AOLoop(h, est, op["Z"][1:], R, n_newton=1, k=1e-2, one_call=one_call)
    dev = torch.device("cuda:0")
    U, X = [], []
    for s_ in range(steps):
        u, x0 = loop.step(torch.from_numpy(np.ascontiguousarray(phase[s_])).to(dev))
        U.append(u.clone()); X.append(x0.clone())
    torch.cuda.synchronize()
    assert int(loop.status.abs().sum()) == 0
    if one_call:
        assert h.last_dual_form() == (1 if R <= 64 else 4) and h.last_dispatch()[0] == pkg.FMPC_PATH_PANEL
    U = torch.stack(U).cpu().numpy(); X = torch.stack(X).cpu().numpy()
    for r in (range(R) if R <= 4 else (0, R // 2, R - 1)):
        ref = ao_loop(md, op, phase[:, r], 1, 1e-2)
        assert rel_err(X[:, r], ref["ad_est"]) <= 1e-7, rel_err(X[:, r], ref["ad_est"])
        assert rel_err(U[:, r], ref["u0"]) <= 1e-7, rel_err(U[:, r], ref["u0"])
    # the loop does what it is for: the residual it estimates is smaller than the uncorrected aberration
    assert np.linalg.norm(X[-1]) < np.linalg.norm(a[-1])
    est.close(); h.close()


@pytest.mark.parametrize("R,n_newton,keep_z", [(5, 1, False), (90, 1, False), (12, 1, True), (7, 3, False)])
def test_ao_step_entry_against_loop_inputs_plus_solve(pkg, gpu, R, n_newton, keep_z):
    """fmpc_ao_step_device(x0, x0_pre, u1, u2) == fmpc_loop_inputs_device (for w = b_ref, with a = 0) + fmpc_solve_u0_device on
    (x0, x0_pre, w): w within 1e-12, first moves and z within 1e-11 (the first-move forms round differently), status and iteration
    counts identical; also with z requested and with a Newton budget (the general route of the entry)."""
    import ctypes as C
    import torch
    from tests.util import handle_from_model
    md = pkg.synthetic.make_model(27, 144, 30)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(21)
    f = lambda *sh: torch.from_numpy(rng.standard_normal(sh)).to(dev)
    x0, x0p, u1, u2 = 0.3 * f(R, 27), 0.3 * f(R, 27), 0.1 * f(R, 144), 0.1 * f(R, 144)
    h = handle_from_model(pkg, md); h2 = handle_from_model(pkg, md)
    vp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    w = torch.zeros((R, 810), dtype=torch.float64, device=dev); u0 = torch.zeros((R, 144), dtype=torch.float64, device=dev)
    z = torch.zeros((R, h.nz), dtype=torch.float64, device=dev) if keep_z else None
    st = torch.full((R,), -9, dtype=torch.int32, device=dev); it = torch.full((R,), -9, dtype=torch.int32, device=dev)
    x0c, x0pc = x0.clone(), x0p.clone()
    rc = h._lib.fmpc_ao_step_device(h._h, R, vp(x0), vp(x0p), vp(u1), vp(u2), vp(w), None, n_newton, 1e-2, vp(z), None, vp(st), vp(it), None, vp(u0),
                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(x0, x0c) and torch.equal(x0p, x0pc)                  # inputs are inputs
    if n_newton == 1 and not keep_z:
        assert h.last_dual_form() == (1 if R <= 64 else 4)
    # the two-call route on a second handle
    za = torch.zeros((R, 27), dtype=torch.float64, device=dev); sx = torch.zeros_like(za); sxp = torch.zeros_like(za)
    w2 = torch.zeros_like(w); u02 = torch.zeros_like(u0); z2 = torch.zeros((R, h.nz), dtype=torch.float64, device=dev)
    st2 = torch.zeros_like(st); it2 = torch.zeros_like(it)
    h2.loop_inputs_device(za, None, u1, u2, sx, sxp, w2)
    h2.solve_device(x0, x0p, w2, None, None, n_newton, 1e-2, z_out=z2, status=st2, iters=it2, u0_out=u02)
    torch.cuda.synchronize()
    assert rel_err(w.cpu().numpy(), w2.cpu().numpy()) <= 1e-12                            # (the first-move kernel sums b_ref in its own order)
    assert torch.equal(st, st2) and torch.equal(it, it2) and int(st.abs().sum()) == 0
    assert rel_err(u0.cpu().numpy(), u02.cpu().numpy()) <= 1e-11
    if keep_z:
        assert rel_err(z.cpu().numpy(), z2.cpu().numpy()) <= 1e-11
    # x0_pre = NULL is refused for a VAR(2) model
    assert h._lib.fmpc_ao_step_device(h._h, R, vp(x0), None, vp(u1), vp(u2), vp(w), None, 1, 1e-2, None, None, vp(st), vp(it), None, vp(u0), None) == pkg._lib.FMPC_E_NULL
    h.close(); h2.close()


@pytest.mark.parametrize("R", [3, 80])
def test_ao_step_entry_with_a_var1_model_and_no_x0_pre(pkg, gpu, R):
    """VAR(1) model (A2 = 0: the reference's VAR_1 variant without ramp rows): x0_pre = NULL is allowed and means zeros; against loop
    inputs + solve, both first-move forms."""
    import ctypes as C
    import torch
    from tests.util import handle_from_model
    md = pkg.synthetic.make_model(27, 144, 10, var_order=1)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    f = lambda *sh: torch.from_numpy(rng.standard_normal(sh)).to(dev)
    x0, u1, u2 = 0.3 * f(R, 27), 0.1 * f(R, 144), 0.1 * f(R, 144)
    h = handle_from_model(pkg, md); h2 = handle_from_model(pkg, md)
    vp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    w = torch.zeros((R, 270), dtype=torch.float64, device=dev); u0 = torch.zeros((R, 144), dtype=torch.float64, device=dev)
    st = torch.full((R,), -9, dtype=torch.int32, device=dev); it = torch.full((R,), -9, dtype=torch.int32, device=dev)
    rc = h._lib.fmpc_ao_step_device(h._h, R, vp(x0), None, vp(u1), vp(u2), vp(w), None, 1, 1e-2, None, None, vp(st), vp(it), None, vp(u0),
                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0 and h.last_dual_form() == (1 if R <= 64 else 4)
    za = torch.zeros((R, 27), dtype=torch.float64, device=dev); sx = torch.zeros_like(za); sxp = torch.zeros_like(za)
    w2 = torch.zeros_like(w); u02 = torch.zeros_like(u0); st2 = torch.zeros_like(st); it2 = torch.zeros_like(it)
    h2.loop_inputs_device(za, None, u1, u2, sx, sxp, w2)
    h2.solve_device(x0, None, w2, None, None, 1, 1e-2, status=st2, iters=it2, u0_out=u02, want_z=False)
    torch.cuda.synchronize()
    assert torch.equal(st, st2) and torch.equal(it, it2) and int(st.abs().sum()) == 0
    assert rel_err(w.cpu().numpy(), w2.cpu().numpy()) <= 1e-12 and rel_err(u0.cpu().numpy(), u02.cpu().numpy()) <= 1e-11
    h.close(); h2.close()
