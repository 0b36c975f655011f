"""GPU test of the coefficient-space closed loop (README.md:482-497, 589): the device loop
(`ClosedLoop`: fmpc_loop_inputs_device + solve + unpack per step, nothing leaves HBM) against the CPU oracle
`oracle/closed_loop_ref.py`.  The loop feeds its own first moves back through B, so errors accumulate over the
steps: tolerance 1e-8 relative on the trajectories after 6 steps (1e-9 per solve)."""
import numpy as np
import pytest

from oracle.closed_loop_ref import closed_loop, design_matrices
from tests.util import handle_from_model, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m,T,nw", [(27, 144, 30, 1), (27, 144, 10, 3), (8, 5, 6, 2), (27, 97, 6, 1)])   # last: m not a multiple of 4
def test_closed_loop_matches_oracle(pkg, gpu, n, m, T, nw):
    import torch
    md = pkg.synthetic.make_model(n, m, T)
    R, steps = 5, 6
    a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(R)], axis=1)   # (steps, R, n)
    h = handle_from_model(pkg, md)
    dev = torch.device("cuda:0")
    loop = pkg.ClosedLoop(h, R, n_newton=nw, k=1e-2)
    U0, X0 = loop.run(torch.from_numpy(np.ascontiguousarray(a)).to(dev))
    torch.cuda.synchronize()
    U0, X0 = U0.cpu().numpy(), X0.cpu().numpy()
    assert int(loop.status.abs().sum()) == 0
    for r in range(R):
        ref = closed_loop(md, a[:, r], nw, 1e-2)
        assert (ref["status"] == 0).all()
        assert rel_err(X0[:, r], ref["x0"]) <= 1e-8 and rel_err(U0[:, r], ref["u0"]) <= 1e-8
    # the inputs kernel alone: b_ref of the last step against M1, M2 applied on the host
    M1, M2 = design_matrices(md["A1"], md["A2"], T)
    w_dev = loop.w.cpu().numpy()
    for r in range(R):
        w_ref = -M1 @ (md["B"] @ U0[steps - 2, r]) - M2 @ (md["B"] @ U0[steps - 3, r])
        assert rel_err(w_dev[r], w_ref) <= 1e-12
    h.close()


@pytest.mark.parametrize("n,m,T,nw,var_order", [(8, 5, 6, 2, 1), (27, 144, 10, 1, 1)])
def test_closed_loop_with_ramp_rows(pkg, gpu, n, m, T, nw, var_order):
    """BASELINE configs[0] as a loop: VAR(1), ramp-rate rows against the previous first move (README.md:355-356, 589;
    VAR_1/fast_mpc_ineq_const.m:58-76).  Oracle: the dense restatement with the ramp rows, step by step."""
    import torch
    md = pkg.synthetic.make_model(n, m, T, var_order=var_order)
    R, steps = 2, 4
    a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(R)], axis=1)
    du = 0.2121 * np.ones(m)
    h = handle_from_model(pkg, md)
    h.set_ramp(-du, du)
    loop = pkg.ClosedLoop(h, R, n_newton=nw, k=1e-2, ramp=True)
    U0, X0 = loop.run(torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda:0")))
    torch.cuda.synchronize()
    U0, X0 = U0.cpu().numpy(), X0.cpu().numpy()
    assert int((loop.status < 0).sum()) == 0 and h.last_dispatch()[0] == pkg.FMPC_PATH_RAMP
    # (no feasibility assertion: like the reference, a fixed Newton budget without a slack-positivity test may leave
    #  the ramp rows violated; parity with the oracle is the bar)
    for r in range(R if n < 20 else 1):
        ref = closed_loop(md, a[:, r], nw, 1e-2, ramp=(-du, du))
        assert rel_err(X0[:, r], ref["x0"]) <= 1e-8 and rel_err(U0[:, r], ref["u0"]) <= 1e-8
    h.close()


def test_config0_200_sequential_steps_with_ramp_rows(pkg, gpu):
    """BASELINE configs[0] as the reference runs it: VAR(1), n = 27, m = 144, T = 10, ramp-rate rows on, 200 SEQUENTIAL
    timesteps of one realisation (every step's u_prev is the previous first move).  Oracle: the dense restatement step by
    step (about 0.5 s of host BLAS per step)."""
    import torch
    md = pkg.synthetic.make_model(27, 144, 10, var_order=1)
    steps = 200
    a = pkg.synthetic.make_realisation(md, r=0, steps=steps)[1:steps + 1][:, None, :]
    du = 0.2121 * np.ones(144)
    h = handle_from_model(pkg, md)
    h.set_ramp(-du, du)
    loop = pkg.ClosedLoop(h, 1, n_newton=1, k=1e-2, ramp=True)
    U0, X0 = loop.run(torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda:0")))
    torch.cuda.synchronize()
    U0, X0 = U0.cpu().numpy()[:, 0], X0.cpu().numpy()[:, 0]
    assert int((loop.status < 0).sum()) == 0 and h.last_dispatch()[0] == pkg.FMPC_PATH_RAMP
    ref = closed_loop(md, a[:, 0], 1, 1e-2, ramp=(-du, du))
    # per step (the loop feeds its own first moves back, so the comparison is over the whole trajectory).  Steps whose line
    # search collapses (quirk D2: the reference ends at t ~ 2^-50, the device at exactly 0 with FMPC_W_LINESEARCH) leave
    # u0 at the mid-box start on both sides up to ~1e-16: absolute floor 1e-9 rad
    errs = [np.abs(U0[s] - ref["u0"][s]).max() / max(np.abs(ref["u0"][s]).max(), 1e-2) for s in range(steps)]
    assert max(errs) <= 1e-7, (int(np.argmax(errs)), max(errs))
    assert rel_err(X0, ref["x0"]) <= 1e-8
    # round 5: every step is the cold-start step in its Woodbury form (fmpc_ramp_cold); with first moves only (keep_z=False:
    # z_out = NULL at fmpc_solve_ramp_u0_device) the step's own kernel writes u0 and nothing of z -- the same trajectory bit for bit
    assert h.last_dual_form() == 5
    loop2 = pkg.ClosedLoop(h, 1, n_newton=1, k=1e-2, ramp=True, keep_z=False)
    U2, X2 = loop2.run(torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda:0")))
    torch.cuda.synchronize()
    assert loop2.z is None and np.array_equal(U2.cpu().numpy()[:, 0], U0) and np.array_equal(X2.cpu().numpy()[:, 0], X0)
    h.close()


@pytest.mark.parametrize("n,m,T", [(27, 144, 30), (27, 97, 7), (8, 5, 6), (96, 20, 4), (60, 300, 12)])
def test_loop_inputs_kernel_against_numpy(pkg, gpu, n, m, T):
    """fmpc_loop_inputs_device alone (several 16-problem tiles, a ragged last one; matrix-core kernel for n = 27, the
    plain one otherwise, the any-size one for n > 64 or beyond the LDS): x0 = a + B u1, x0_pre = x0_last, w = -M1 B u1 - M2 B u2,
    and the NULL variants of the first steps."""
    import torch
    md = pkg.synthetic.make_model(n, m, T)
    R = 37
    rng = np.random.default_rng(4)
    a, xl, u1, u2 = rng.standard_normal((R, n)), rng.standard_normal((R, n)), rng.standard_normal((R, m)), rng.standard_normal((R, m))
    M1, M2 = design_matrices(md["A1"], md["A2"], T)
    h = handle_from_model(pkg, md)
    dev = torch.device("cuda:0")
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(dev)
    f = dict(dtype=torch.float64, device=dev)
    for use1, use2, usel in ((True, True, True), (True, False, True), (False, False, False)):
        x0, x0p, w = torch.full((R, n), np.nan, **f), torch.full((R, n), np.nan, **f), torch.full((R, T * n), np.nan, **f)
        h.loop_inputs_device(t(a), t(xl) if usel else None, t(u1) if use1 else None, t(u2) if use2 else None, x0, x0p, w)
        torch.cuda.synchronize()
        bu1 = (md["B"] @ u1.T).T if use1 else np.zeros((R, n))
        bu2 = (md["B"] @ u2.T).T if use2 else np.zeros((R, n))
        assert rel_err(x0.cpu().numpy(), a + bu1) <= 1e-13
        assert np.array_equal(x0p.cpu().numpy(), xl if usel else np.zeros((R, n)))
        w_ref = -(M1 @ bu1.T).T - (M2 @ bu2.T).T
        assert np.abs(w.cpu().numpy() - w_ref).max() <= 1e-12 * max(1.0, np.abs(w_ref).max())
    h.close()


@pytest.mark.parametrize("R,T", [(5, 30), (40, 30), (3, 10)])
def test_loop_step_equals_inputs_plus_solve(pkg, gpu, R, T):
    """fmpc_loop_step_device (one call; the dense form of the dual solve takes [B u1; B u2] in place of w) against the two
    separate calls with the sweeps of the panel kernel: same trajectories to 1e-10 over 6 fed-back steps, w bit for bit."""
    import torch
    md = pkg.synthetic.make_model(27, 144, T)
    steps = 6
    a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(R)], axis=1)
    dev = torch.device("cuda:0")
    at = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    nu0 = torch.from_numpy(np.random.default_rng(1).random((steps, R, T * 27))).to(dev)
    h1 = handle_from_model(pkg, md); h2 = handle_from_model(pkg, md)
    h2.set_dense_form(False)
    la = pkg.ClosedLoop(h1, R, n_newton=1, k=1e-2)
    lb = pkg.ClosedLoop(h2, R, n_newton=1, k=1e-2, fused=False)
    Ua, Xa = la.run(at, nu0)
    assert h1.last_dual_form() == 1 and h1.last_dispatch()[0] == pkg.FMPC_PATH_PANEL
    Ub, Xb = lb.run(at, nu0)
    assert h2.last_dual_form() == 0
    torch.cuda.synchronize()
    assert int(la.status.abs().sum()) == 0 and int(lb.status.abs().sum()) == 0
    assert rel_err(Ua.cpu().numpy(), Ub.cpu().numpy()) <= 1e-10 and rel_err(Xa.cpu().numpy(), Xb.cpu().numpy()) <= 1e-10
    assert rel_err(la.z.cpu().numpy(), lb.z.cpu().numpy()) <= 1e-10
    assert rel_err(la.w.cpu().numpy(), lb.w.cpu().numpy()) <= 1e-10
    h1.close(); h2.close()


def test_loop_step_with_a_newton_budget_and_no_nu0(pkg, gpu):
    """fmpc_loop_step_device with n_newton = 3 (dense-form first step, continuation on the tiled kernel) and nu0 = NULL,
    against the oracle loop."""
    import torch
    md = pkg.synthetic.make_model(27, 144, 10)
    md["u_min"] = -0.08 * np.ones(144); md["u_max"] = 0.08 * np.ones(144)
    R, steps = 4, 5
    a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(R)], axis=1)
    h = handle_from_model(pkg, md)
    loop = pkg.ClosedLoop(h, R, n_newton=3, k=1e-2)
    U0, X0 = loop.run(torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda:0")))
    torch.cuda.synchronize()
    assert h.last_dual_form() == 1 and int(loop.status.abs().sum()) == 0
    U0, X0 = U0.cpu().numpy(), X0.cpu().numpy()
    for r in range(R):
        ref = closed_loop(md, a[:, r], 3, 1e-2)
        assert rel_err(X0[:, r], ref["x0"]) <= 1e-8 and rel_err(U0[:, r], ref["u0"]) <= 1e-8
    h.close()


@pytest.mark.parametrize("R,T,first_move", [(1, 30, True), (40, 30, True), (37, 10, True), (5, 30, False), (100, 30, "product"),
                                            (65, 30, "product"), (530, 10, "product"), (100, 30, "product3"), (530, 10, "product3"),
                                            (100, 30, None)])
def test_closed_loop_first_moves_only(pkg, gpu, R, T, first_move, monkeypatch):
    """ClosedLoop(keep_z=False): z_out = NULL at the C ABI (fmpc_loop_step_device), only u[k] = U(1:nu) leaves the solve
    (README.md:589).  Up to 64 realisations the step is ONE launch in the first-move form (fmpc_kernel_first.hip: u0 = u0c + K0 d,
    the decision from two quadratic forms; same algebra, different rounding): trajectories within 1e-11 of the loop that keeps z.
    More than 64 realisations: the same form as one product over the batch (fmpc_kernel_loopu0.hip), same bound.
    With the forms switched off (FMPC_NO_FIRST_MOVE=1 / FMPC_NO_LOOP_U0=1; d_z without z stores): bit for bit."""
    import torch
    md = pkg.synthetic.make_model(27, 144, T)
    steps = 8
    a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(R)], axis=1)
    at = torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda:0"))
    nu0 = torch.from_numpy(np.random.default_rng(1).random((steps, R, T * 27))).to(at.device)
    h1 = handle_from_model(pkg, md)
    if first_move is False:
        monkeypatch.setenv("FMPC_NO_FIRST_MOVE", "1")
    if first_move is None:
        monkeypatch.setenv("FMPC_NO_LOOP_U0", "1")
    if first_move == "product3":
        monkeypatch.setenv("FMPC_NO_LOOP_FUSE", "1")
    h2 = handle_from_model(pkg, md)
    monkeypatch.delenv("FMPC_NO_FIRST_MOVE", raising=False)
    monkeypatch.delenv("FMPC_NO_LOOP_U0", raising=False)
    monkeypatch.delenv("FMPC_NO_LOOP_FUSE", raising=False)
    la = pkg.ClosedLoop(h1, R, n_newton=1, k=1e-2)
    lb = pkg.ClosedLoop(h2, R, n_newton=1, k=1e-2, keep_z=False)
    Ua, Xa = la.run(at, nu0)
    Ub, Xb = lb.run(at, nu0)
    torch.cuda.synchronize()
    assert lb.z is None and h2.last_dispatch() == (pkg.FMPC_PATH_PANEL, 0)
    assert int(la.status.abs().sum()) == 0 and int(lb.status.abs().sum()) == 0
    assert torch.equal(la.iters, lb.iters)
    # more than 64 realisations: the same form as ONE product per batch on the matrix cores (fmpc_kernel_loopu0.hip; dual form 3)
    # (ClosedLoop alternates two x0 buffers, so the loop inputs ride in the same launch: dual form 4; FMPC_NO_LOOP_FUSE=1: 3)
    assert h2.last_dual_form() == {"product": 4, "product3": 3}.get(first_move, 1 if first_move else h2.last_dual_form())
    if first_move:
        assert rel_err(Ub.cpu().numpy(), Ua.cpu().numpy()) <= 1e-11 and rel_err(Xb.cpu().numpy(), Xa.cpu().numpy()) <= 1e-11
        assert rel_err(lb.w.cpu().numpy(), la.w.cpu().numpy()) <= 1e-11
    else:
        assert torch.equal(Ua, Ub) and torch.equal(Xa, Xb)
    h1.close(); h2.close()


def test_first_move_form_hands_unclear_realisations_to_the_exact_path(pkg, gpu):
    """Tight bounds: the barrier is active at the start, the line search does not accept t = 1 with a wide margin, and the
    first-move kernel must flag those realisations; the exact path (flag mode of fmpc_newton_wave) redoes them.  Against the
    oracle loop (1e-8 over the fed-back steps) and against the four-launch path."""
    import torch
    md = pkg.synthetic.make_model(27, 144, 10)
    md["u_min"] = -0.05 * np.ones(144); md["u_max"] = 0.05 * np.ones(144)
    R, steps = 6, 5
    a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(R)], axis=1)
    at = torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda:0"))
    h1 = handle_from_model(pkg, md); h2 = handle_from_model(pkg, md)
    la = pkg.ClosedLoop(h1, R, n_newton=1, k=1e-2)
    lb = pkg.ClosedLoop(h2, R, n_newton=1, k=1e-2, keep_z=False)
    Ua, Xa = la.run(at)
    Ub, Xb = lb.run(at)
    torch.cuda.synchronize()
    assert h2.last_dispatch()[1] > 0, "no realisation was handed over: the case does not test the fallback"
    assert torch.equal(la.status, lb.status) and torch.equal(la.iters, lb.iters)
    assert rel_err(Ub.cpu().numpy(), Ua.cpu().numpy()) <= 1e-11 and rel_err(Xb.cpu().numpy(), Xa.cpu().numpy()) <= 1e-11
    U0, X0 = Ub.cpu().numpy(), Xb.cpu().numpy()
    for r in range(R):
        ref = closed_loop(md, a[:, r], 1, 1e-2)
        assert rel_err(X0[:, r], ref["x0"]) <= 1e-8 and rel_err(U0[:, r], ref["u0"]) <= 1e-8
    h1.close(); h2.close()


@pytest.mark.parametrize("ub,var_order,fuse", [(0.05, 2, True), (0.24, 2, True), (0.24, 1, True), (0.24, 2, False), (0.05, 1, False)])
def test_first_move_product_hands_unclear_realisations_to_the_exact_path(pkg, gpu, ub, var_order, fuse, monkeypatch):
    """The product form (more than 64 realisations) with bounds where the decision tips (0.24: flagged and accepted side by side)
    and where every realisation backtracks (0.05): same status / iterations / step lengths as the four-launch path, trajectories
    within 1e-11, three realisations against the oracle loop."""
    import torch
    md = pkg.synthetic.make_model(27, 144, 10, var_order=var_order)
    md["u_min"] = -ub * np.ones(144); md["u_max"] = ub * np.ones(144)
    R, steps = 75, 4
    a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(R)], axis=1)
    a = a * np.linspace(0.05, 5.0, R)[None, :, None]
    at = torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda:0"))
    h1 = handle_from_model(pkg, md)
    if not fuse:
        monkeypatch.setenv("FMPC_NO_LOOP_FUSE", "1")
    h2 = handle_from_model(pkg, md)
    monkeypatch.delenv("FMPC_NO_LOOP_FUSE", raising=False)
    la = pkg.ClosedLoop(h1, R, n_newton=1, k=1e-2)
    lb = pkg.ClosedLoop(h2, R, n_newton=1, k=1e-2, keep_z=False)
    Ua, Xa = la.run(at)
    Ub, Xb = lb.run(at)
    torch.cuda.synchronize()
    assert h2.last_dual_form() == (4 if fuse else 3)
    handed = h2.last_dispatch()[1]
    assert handed == R if ub == 0.05 else 0 <= handed <= R
    assert torch.equal(la.status, lb.status) and torch.equal(la.iters, lb.iters)
    assert rel_err(Ub.cpu().numpy(), Ua.cpu().numpy()) <= 1e-11 and rel_err(Xb.cpu().numpy(), Xa.cpu().numpy()) <= 1e-11
    U0, X0 = Ub.cpu().numpy(), Xb.cpu().numpy()
    for r in (0, R // 2, R - 1):
        ref = closed_loop(md, a[:, r], 1, 1e-2)
        assert rel_err(X0[:, r], ref["x0"]) <= 1e-8 and rel_err(U0[:, r], ref["u0"]) <= 1e-8
    h1.close(); h2.close()


@pytest.mark.parametrize("R", [1, 7, 130])
def test_recorded_stretch_in_one_call(pkg, gpu, R):
    """fmpc_loop_run_device: a recorded stretch of the loop in one host call == the same steps one call at a time, bit for bit
    (same kernels in the same order), also when the stretch continues an earlier one."""
    import torch
    md = pkg.synthetic.make_model(27, 144, 30)
    steps = 9
    a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(R)], axis=1)
    at = torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda:0"))
    nu0 = torch.from_numpy(np.random.default_rng(2).random((steps, R, 30 * 27))).to(at.device)
    h1 = handle_from_model(pkg, md); h2 = handle_from_model(pkg, md)
    la = pkg.ClosedLoop(h1, R, n_newton=1, k=1e-2, keep_z=False)
    lb = pkg.ClosedLoop(h2, R, n_newton=1, k=1e-2, keep_z=False)
    Ua, Xa = la.run(at, nu0)
    U1, X1 = lb.run_recorded(at[:4].contiguous(), nu0[:4].contiguous())
    U2, X2 = lb.run_recorded(at[4:].contiguous(), nu0[4:].contiguous())
    torch.cuda.synchronize()
    assert int(lb.status.abs().sum()) == 0
    if R <= 64:
        assert torch.equal(torch.cat([U1, U2]), Ua) and torch.equal(torch.cat([X1, X2]), Xa)
        # the state a following step starts from is the step-by-step run's
        assert torch.equal(lb.x0, la.x0) and torch.equal(lb.x0_pre, la.x0_pre) and torch.equal(lb.w, la.w)
        assert torch.equal(lb.iters, la.iters)
        ua = la.step(at[0]); ub = lb.step(at[0])
        torch.cuda.synchronize()
        assert torch.equal(ua, ub)
    else:
        # beyond 64 realisations one call per step is the four-launch form, the walk the first-move form: same algebra, other rounding
        close = lambda x, y: rel_err(x.cpu().numpy(), y.cpu().numpy()) <= 1e-11
        assert close(torch.cat([U1, U2]), Ua) and close(torch.cat([X1, X2]), Xa)
        assert close(lb.x0, la.x0) and close(lb.x0_pre, la.x0_pre) and close(lb.w, la.w) and torch.equal(lb.iters, la.iters)
    h1.close(); h2.close()


@pytest.mark.parametrize("max_restarts", [None, "1", "0"])
def test_recorded_stretch_with_steps_the_exact_path_redoes(pkg, gpu, monkeypatch, max_restarts):
    """Tight bounds (see test_first_move_form_hands_unclear_realisations_to_the_exact_path): the one-launch walk of
    fmpc_loop_run_device stops at the steps that are not clear-cut, the exact path redoes them and the walk goes on behind
    them -- same results as one call per step, bit for bit, and the oracle loop to 1e-8.  With a cap on the number of walks
    (FMPC_WALK_MAX_RESTARTS, read per call; default 8, or a tenth of the realisations stopping in one walk) the rest of the
    stretch is done stepwise, the realisations furthest behind first: same results again."""
    import torch
    if max_restarts is not None:
        monkeypatch.setenv("FMPC_WALK_MAX_RESTARTS", max_restarts)
    md = pkg.synthetic.make_model(27, 144, 10)
    # bounds at the width where the decision tips (+-0.22: every step of every realisation is handed over, +-0.25: none) and
    # realisations of different strength: single realisations stop at single steps
    md["u_min"] = -0.24 * np.ones(144); md["u_max"] = 0.24 * np.ones(144)
    R, steps = 5, 12
    a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(R)], axis=1)
    a *= np.array([0.05, 0.3, 1.0, 2.0, 5.0])[None, :, None]
    at = torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda:0"))
    h1 = handle_from_model(pkg, md); h2 = handle_from_model(pkg, md)
    la = pkg.ClosedLoop(h1, R, n_newton=1, k=1e-2, keep_z=False)
    lb = pkg.ClosedLoop(h2, R, n_newton=1, k=1e-2, keep_z=False)
    Ua, Xa, handed = [], [], []
    for s_ in range(steps):
        Ua.append(la.step(at[s_]).clone()); Xa.append(la.x0.clone())
        torch.cuda.synchronize()
        handed.append(h1.last_dispatch()[1])
    Ua, Xa = torch.stack(Ua), torch.stack(Xa)
    assert 0 < sum(1 for c_ in handed[:-1] if c_ > 0), "no step before the last was handed over: the case does not test the stops"
    assert any(c_ == 0 for c_ in handed), "every step stops some realisation: the walk never walks (%s)" % handed
    assert any(0 < c_ < R for c_ in handed[:-1]), "no step stops only part of the realisations: %s" % handed
    Ub, Xb = lb.run_recorded(at)
    torch.cuda.synchronize()
    assert torch.equal(Ub, Ua) and torch.equal(Xb, Xa)
    assert torch.equal(lb.x0, la.x0) and torch.equal(lb.x0_pre, la.x0_pre) and torch.equal(lb.w, la.w)
    assert torch.equal(lb.status, la.status) and torch.equal(lb.iters, la.iters)
    U0, X0 = Ub.cpu().numpy(), Xb.cpu().numpy()
    for r in range(R):
        ref = closed_loop(md, a[:, r], 1, 1e-2)
        assert rel_err(X0[:, r], ref["x0"]) <= 1e-8 and rel_err(U0[:, r], ref["u0"]) <= 1e-8
    h1.close(); h2.close()


@pytest.mark.parametrize("ub", [0.4, 2.0])
def test_first_move_decision_forms_against_the_oracle(pkg, gpu, ub):
    """The two quadratic forms the first-move kernel decides on -- ||e||^2 (k P'DP d_z of the full step) and ||r_p||^2 -- against
    the same sums of the oracle's Newton step, realisation by realisation over a few fed-back steps (diagnostic output of the
    kernel: upper bound of the first, lower bound of the second, both within the rounding guard)."""
    import ctypes as C
    import torch
    from oracle.banded_ref import BandedFastMPC
    md = pkg.synthetic.make_model(27, 144, 10)
    md["u_min"] = -ub * np.ones(144); md["u_max"] = ub * np.ones(144)
    R, steps = 4, 4
    a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(R)], axis=1)
    at = torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda:0"))
    h = handle_from_model(pkg, md)
    forms = torch.zeros((R, 3), dtype=torch.float64, device=at.device)
    fn = h._lib.fmpc_debug_first_move_forms
    fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
    assert fn(h._h, C.c_void_p(forms.data_ptr())) == 0
    lp = pkg.ClosedLoop(h, R, n_newton=1, k=1e-2, keep_z=False)
    orc = BandedFastMPC(md["A1"], md["A2"], md["B"], md["Q"], md["R"], md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], 10)
    worst_e, worst_p = 0.0, 0.0
    for s_ in range(steps):
        lp.step(at[s_])
        torch.cuda.synchronize()
        f = forms.cpu().numpy()
        x0, x0p, w = lp.x0.cpu().numpy(), lp.x0_pre.cpu().numpy(), lp.w.cpu().numpy()
        for r in range(R):
            info = {}
            orc.solve(x0[r], x0p[r], w[r], 1, 1e-2, info=info)
            eps2, rp2 = info["eps2"][0], info["rp2"][0]
            guard = 1e-9 * (eps2 + rp2 + 1.0)
            assert f[r, 0] >= eps2 - guard and f[r, 1] <= rp2 + guard, (s_, r, f[r], eps2, rp2)
            worst_e = max(worst_e, abs(f[r, 0] - eps2) / (eps2 + 1e-30)); worst_p = max(worst_p, abs(f[r, 1] - rp2) / (rp2 + 1e-30))
            assert f[r, 2] <= info["rho2"][0] * (1 + 1e-9)
    assert fn(h._h, None) == 0
    assert worst_e <= 1e-6 and worst_p <= 1e-6, (worst_e, worst_p)
    h.close()


def test_loop_step_in_place_and_with_separate_x0_buffers(pkg, gpu):
    """fmpc_loop_step_device with more than 64 realisations and first moves only: x0 updated in place (x0_last == x0) takes the
    three-launch form (dual form 3), separate buffers the one-launch form (4); x0, x0_pre, w and the first moves agree to 1e-12
    (w and x0 bit for bit: the same products in the same order)."""
    import ctypes as C
    import torch
    md = pkg.synthetic.make_model(27, 144, 30)
    R, dev = 150, torch.device("cuda:0")
    rng = np.random.default_rng(11)
    f = lambda *sh: torch.from_numpy(rng.standard_normal(sh)).to(dev)
    a, xl, u1, u2 = 0.3 * f(R, 27), 0.3 * f(R, 27), 0.1 * f(R, 144), 0.1 * f(R, 144)
    nu0 = torch.from_numpy(rng.random((R, 30 * 27))).to(dev)
    h = handle_from_model(pkg, md)
    vp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    out = {}
    for mode in ("in_place", "separate"):
        x0 = xl.clone() if mode == "in_place" else torch.zeros_like(xl)
        x0p = torch.zeros_like(xl); w = torch.zeros((R, 810), dtype=torch.float64, device=dev); u0 = torch.zeros((R, 144), dtype=torch.float64, device=dev)
        st = torch.full((R,), -9, dtype=torch.int32, device=dev); it = torch.full((R,), -9, dtype=torch.int32, device=dev)
        rc = h._lib.fmpc_loop_step_device(h._h, R, vp(a), vp(x0 if mode == "in_place" else xl), vp(u1), vp(u2), vp(x0), vp(x0p), vp(w), vp(nu0), 1, 1e-2,
                                          None, None, vp(st), vp(it), None, vp(u0), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        torch.cuda.synchronize()
        assert rc == 0 and h.last_dual_form() == (3 if mode == "in_place" else 4)
        out[mode] = [t.cpu().numpy() for t in (x0, x0p, w, u0, st, it)]
    A, B_ = out["in_place"], out["separate"]
    assert np.array_equal(A[0], B_[0]) and np.array_equal(A[1], B_[1]) and np.array_equal(A[1], xl.cpu().numpy())
    assert np.array_equal(A[4], B_[4]) and np.array_equal(A[5], B_[5]) and int(np.abs(A[4]).sum()) == 0
    assert rel_err(B_[2], A[2]) <= 1e-13 and rel_err(B_[3], A[3]) <= 1e-12
    h.close()


@pytest.mark.parametrize("m,R", [(100, 70), (37, 70), (100, 5)])
def test_first_move_forms_with_other_actuator_counts(pkg, gpu, m, R):
    """The first-move forms with m = 100 and 37 actuators (row tiles, k-steps and the wavefronts' shares of the actuators end in
    partial ones): first moves only against the loop that keeps z, 1e-11 over six fed-back steps."""
    import torch
    md = pkg.synthetic.make_model(27, m, 10)
    steps = 6
    a = np.stack([pkg.synthetic.make_realisation(md, r=r, steps=steps)[1:steps + 1] for r in range(R)], axis=1)
    at = torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda:0"))
    h1 = handle_from_model(pkg, md); h2 = handle_from_model(pkg, md)
    la = pkg.ClosedLoop(h1, R, n_newton=1, k=1e-2)
    lb = pkg.ClosedLoop(h2, R, n_newton=1, k=1e-2, keep_z=False)
    Ua, Xa = la.run(at); Ub, Xb = lb.run(at)
    torch.cuda.synchronize()
    assert h2.last_dual_form() == (4 if R > 64 else 1) and int(lb.status.abs().sum()) == 0
    assert rel_err(Ub.cpu().numpy(), Ua.cpu().numpy()) <= 1e-11 and rel_err(Xb.cpu().numpy(), Xa.cpu().numpy()) <= 1e-11
    assert rel_err(lb.w.cpu().numpy(), la.w.cpu().numpy()) <= 1e-11
    h1.close(); h2.close()
