"""GPU tests of the affine form of the cold-start step without w (fmpc_kernel_affine.hip): z+ = zc + Kz [x0 ; x0_pre] as ONE
product on the matrix cores + the step-length decision from two quadratic forms; problems whose decision is not clear-cut are
redone by the exact path.  Taken by solves from the cold start with w = NULL and n_newton = 1 (the reference's replay call,
README.md:548-556); nu+, when requested, is further rows of the same product.  Checkers: the structured oracle (1e-9 on z), and the three-kernel form of the same step
(dense dual solve + d_z + decision, FMPC_NO_AFFINE=1: 1e-11, identical status / iteration counts / step lengths)."""
import os

import numpy as np
import pytest

from tests.util import canon_steps, handle_from_model, oracle_batch, rel_err

pytestmark = pytest.mark.gpu


def _solve_dev(pkg, h, data, want_z=True, want_u0=True, want_nu=False):
    import torch
    dev = torch.device("cuda:0")
    t = {k: (None if v is None else torch.from_numpy(np.ascontiguousarray(v)).to(dev)) for k, v in data.items()}
    B = data["x0"].shape[0]
    z = torch.full((B, h.nz), float("nan"), dtype=torch.float64, device=dev) if want_z else None
    u0 = torch.full((B, h.m), float("nan"), dtype=torch.float64, device=dev) if want_u0 else None
    st = torch.full((B,), -99, dtype=torch.int32, device=dev); it = torch.full((B,), -99, dtype=torch.int32, device=dev)
    stp = torch.full((B, 1), float("nan"), dtype=torch.float64, device=dev)
    nu = torch.full((B, h.nu_len), float("nan"), dtype=torch.float64, device=dev) if want_nu else None
    h.solve_device(t["x0"], t["x0_pre"], None, None, t["nu0"], 1, 1e-2, z_out=z, nu_out=nu, status=st, iters=it, step=stp, u0_out=u0, want_z=want_z)
    torch.cuda.synchronize()
    if want_nu:
        return z.cpu().numpy(), nu.cpu().numpy()
    return (None if z is None else z.cpu().numpy(), None if u0 is None else u0.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy(), stp.cpu().numpy())


def _handles(pkg, md):
    h = handle_from_model(pkg, md)
    os.environ["FMPC_NO_AFFINE"] = "1"
    try:
        h3 = handle_from_model(pkg, md)
    finally:
        os.environ.pop("FMPC_NO_AFFINE")
    return h, h3


@pytest.mark.parametrize("T,batch,xf,use_nu,var_order", [
    (30, 1, False, True, 2),
    (30, 15, False, False, 2),
    (30, 17, True, True, 2),         # terminal row
    (30, 65, False, True, 2),        # one problem into the second group of 64
    (30, 300, True, False, 2),
    (10, 40, False, True, 1),        # VAR(1): x0_pre = NULL
    (2, 33, False, True, 2),         # the horizon the README itself runs (README.md:338): 342 rows of z
    (1, 5, True, True, 2),
    (7, 130, False, True, 2),
    (8, 70, False, False, 2),        # no free slot: the decision forms in workgroups of their own
    (9, 33, False, True, 2), (11, 17, False, False, 2), (2, 40, False, True, 2),
])
def test_affine_form_matches_oracle_and_the_three_kernel_form(pkg, gpu, T, batch, xf, use_nu, var_order):
    md = pkg.synthetic.make_model(27, 144, T, var_order=var_order)
    rng = np.random.default_rng(5)
    if xf:
        md["xf"] = 0.01 * rng.standard_normal(27)
    data = pkg.synthetic.make_replay_batch(md, r=5, steps=batch)
    data["w"] = None
    data["nu0"] = rng.standard_normal((batch, (T + (1 if xf else 0)) * 27)) if use_nu else None
    if var_order == 1:
        data["x0_pre"] = None
    h, h3 = _handles(pkg, md)
    za, ua, sa, ia, ta = _solve_dev(pkg, h, data)
    assert h.last_dispatch()[0] == pkg.FMPC_PATH_PANEL and h.last_dual_form() == 2
    z3, u3, s3, i3, t3 = _solve_dev(pkg, h3, data)
    assert h3.last_dual_form() == 1
    assert np.array_equal(sa, s3) and np.array_equal(ia, i3) and np.array_equal(ta, t3)
    assert np.all(np.isfinite(za)) and np.array_equal(ua, za[:, :144])
    assert max(rel_err(za[p], z3[p]) for p in range(batch)) <= 1e-11
    nchk = min(batch, 24)
    sub = {k: (v[:nchk] if v is not None else None) for k, v in data.items()}
    zo, _, ito, sto, steps = oracle_batch(md, sub, 1, 1e-2)
    assert np.array_equal(ia[:nchk], ito) and np.array_equal(sa[:nchk], sto)
    assert np.array_equal(canon_steps(ta[:nchk, 0]), canon_steps([s[0] for s in steps]))
    assert max(rel_err(za[p], zo[p]) for p in range(nchk)) <= 1e-9
    # with the multipliers requested: nu+ = nuc + J d as further rows of the same product; z does not change by a bit
    zn, nun = _solve_dev(pkg, h, data, want_nu=True)
    assert h.last_dual_form() == 2 and np.array_equal(zn, za) and np.all(np.isfinite(nun))
    z3n, nu3 = _solve_dev(pkg, h3, data, want_nu=True)
    assert max(rel_err(nun[p], nu3[p]) for p in range(batch)) <= 1e-11
    _, nuo, *_ = oracle_batch(md, sub, 1, 1e-2)
    assert max(rel_err(nun[p], nuo[p]) for p in range(nchk)) <= 1e-7
    # first moves only (z_out = NULL): the first m rows of the same product
    _, uo, so, io, _ = _solve_dev(pkg, h, data, want_z=False)
    assert h.last_dual_form() == 2
    assert np.array_equal(uo, ua) and np.array_equal(so, sa) and np.array_equal(io, ia)
    h.close(); h3.close()


def test_affine_form_hands_unclear_problems_to_the_exact_path(pkg, gpu):
    """Tight bounds: the barrier is active at the start, t = 1 is not accepted with a wide margin; the affine kernel flags those
    problems and the exact path redoes them (backtracking included).  Bounds at the width where the decision tips, so that
    flagged and accepted problems sit side by side."""
    for ub, want_all in ((0.05, True), (0.24, False)):
        md = pkg.synthetic.make_model(27, 144, 10)
        md["u_min"] = -ub * np.ones(144); md["u_max"] = ub * np.ones(144)
        batch = 70
        data = pkg.synthetic.make_replay_batch(md, r=3, steps=batch)
        data["x0"] = data["x0"] * np.linspace(0.05, 5.0, batch)[:, None]
        data["x0_pre"] = data["x0_pre"] * np.linspace(0.05, 5.0, batch)[:, None]
        data["w"] = None
        h, h3 = _handles(pkg, md)
        za, ua, sa, ia, ta = _solve_dev(pkg, h, data)
        handed = h.last_dispatch()[1]
        assert h.last_dual_form() == 2
        assert handed == batch if want_all else handed >= 0
        z3, u3, s3, i3, t3 = _solve_dev(pkg, h3, data)
        assert np.array_equal(sa, s3) and np.array_equal(ia, i3) and np.array_equal(ta, t3)
        assert max(rel_err(za[p], z3[p]) for p in range(batch)) <= 1e-11 and np.array_equal(ua, za[:, :144])
        pick = [0, 1, batch // 2, batch - 1]
        sub = {k: (v[pick] if v is not None else None) for k, v in data.items()}
        zo, _, ito, sto, steps = oracle_batch(md, sub, 1, 1e-2)
        assert np.array_equal(ia[pick], ito) and np.array_equal(sa[pick], sto)
        assert np.array_equal(canon_steps(ta[pick, 0]), canon_steps([s[0] for s in steps]))
        assert max(rel_err(za[p], zo[q]) for q, p in enumerate(pick)) <= 1e-9
        zn, nun = _solve_dev(pkg, h, data, want_nu=True)
        z3n, nu3 = _solve_dev(pkg, h3, data, want_nu=True)
        assert np.array_equal(zn, za) and max(rel_err(nun[p], nu3[p]) for p in range(batch)) <= 1e-11
        if want_all:
            assert np.any(ta[:, 0] < 1.0), "no backtracking in the tight-box case"
        _, uo, _, _, _ = _solve_dev(pkg, h, data, want_z=False)
        assert np.array_equal(uo, ua)
        h.close(); h3.close()


def test_affine_form_at_the_headline_size(pkg, gpu):
    """BASELINE configs[1]: 2000 problems of one replay batch, (27, 144, 30).  Against the three-kernel form on all problems,
    the oracle on three."""
    md = pkg.synthetic.make_model(27, 144, 30)
    data = pkg.synthetic.make_replay_batch(md, r=0, steps=2000)
    data["w"] = None
    h, h3 = _handles(pkg, md)
    za, ua, sa, ia, ta = _solve_dev(pkg, h, data)
    assert h.last_dual_form() == 2 and int(np.abs(sa).sum()) == 0 and np.all(ia == 1)
    z3, _, s3, i3, t3 = _solve_dev(pkg, h3, data)
    assert np.array_equal(sa, s3) and np.array_equal(ia, i3) and np.array_equal(ta, t3)
    assert max(rel_err(za[p], z3[p]) for p in range(2000)) <= 1e-11 and np.array_equal(ua, za[:, :144])
    pick = [0, 999, 1999]
    sub = {k: (v[pick] if v is not None else None) for k, v in data.items()}
    zo, *_ = oracle_batch(md, sub, 1, 1e-2)
    assert max(rel_err(za[p], zo[q]) for q, p in enumerate(pick)) <= 1e-9
    h.close(); h3.close()


@pytest.mark.parametrize("ldz_extra,batch,tight", [(6, 2000, None), (6, 37, None), (1, 70, None), (6, 70, 0.05), (70, 70, 0.05), (22, 200, 0.15), (1, 200, 0.1)])
def test_padded_z_rows(pkg, gpu, ldz_extra, batch, tight):
    """fmpc_set_z_ld: row p of z_out at z_out + p ldz.  N_z = 5130 at (27, 144, 30): ldz = 5136 / 5200 are multiples of 16 (128 bytes:
    the kernel instance with non-temporal stores), 5131 is not (ordinary stores).  Same numbers as with contiguous rows, bit for bit
    -- also for the problems the exact path redoes (tight bounds) --, nothing written between the rows; solves the affine form does
    not take refuse padded rows before anything is enqueued."""
    import torch
    dev = torch.device("cuda:0")
    md = pkg.synthetic.make_model(27, 144, 30)
    if tight:
        md["u_min"] = -tight * np.ones(144); md["u_max"] = tight * np.ones(144)
    data = pkg.synthetic.make_replay_batch(md, r=5, steps=batch)
    if tight:
        data["x0"] = data["x0"] * np.linspace(0.05, 5.0, batch)[:, None]
        data["x0_pre"] = data["x0_pre"] * np.linspace(0.05, 5.0, batch)[:, None]
    h = handle_from_model(pkg, md)
    t = {k: (None if v is None else torch.from_numpy(np.ascontiguousarray(v)).to(dev)) for k, v in data.items()}
    ldz = h.nz + ldz_extra
    zc = torch.full((batch, h.nz), float("nan"), dtype=torch.float64, device=dev)
    big = torch.full((batch, ldz), -7.0, dtype=torch.float64, device=dev)
    nuc = torch.empty((batch, h.nu_len), dtype=torch.float64, device=dev); nup = torch.empty_like(nuc)
    uc = torch.empty((batch, h.m), dtype=torch.float64, device=dev); up = torch.empty_like(uc)
    _, sc, ic = h.solve_device(t["x0"], t["x0_pre"], None, None, t["nu0"], 1, 1e-2, z_out=zc, nu_out=nuc, u0_out=uc)
    handed = h.last_dispatch()[1]
    assert h.last_dual_form() == 2
    if tight and tight <= 0.05:
        assert 0 < handed, "no problem was handed to the exact path: the case does not test it"
    _, sp, ip = h.solve_device(t["x0"], t["x0_pre"], None, None, t["nu0"], 1, 1e-2, z_out=big[:, :h.nz], nu_out=nup, u0_out=up)
    torch.cuda.synchronize()
    assert h.last_dual_form() == 2 and h.last_dispatch()[1] == handed
    assert torch.equal(big[:, :h.nz], zc) and torch.equal(nup, nuc) and torch.equal(up, uc) and torch.equal(sp, sc) and torch.equal(ip, ic)
    assert bool((big[:, h.nz:] == -7.0).all()), "something was written between the rows"
    # without nu, and z alone
    big.fill_(-7.0)
    h.solve_device(t["x0"], t["x0_pre"], None, None, t["nu0"], 1, 1e-2, z_out=big[:, :h.nz])
    torch.cuda.synchronize()
    assert torch.equal(big[:, :h.nz], zc) and bool((big[:, h.nz:] == -7.0).all())
    # the handle is back to contiguous rows after a call with a view
    z2 = torch.empty_like(zc)
    h.solve_device(t["x0"], t["x0_pre"], None, None, t["nu0"], 1, 1e-2, z_out=z2)
    torch.cuda.synchronize()
    assert torch.equal(z2, zc)
    # a stride that stays (set_z_ld), and the solves that do not take padded rows
    h.set_z_ld(ldz)
    big.fill_(-7.0)
    h.solve_device(t["x0"], t["x0_pre"], None, None, t["nu0"], 1, 1e-2, z_out=big[:, :h.nz])
    torch.cuda.synchronize()
    assert torch.equal(big[:, :h.nz], zc)
    # with a stride that stays, a CONTIGUOUS z_out (or none) must not be written at the handle's row distance --
    # the row distance is an argument of the call (fmpc_solve_u0_device_ld); a guard region behind the array stays untouched
    guard = torch.full((batch * h.nz + 64 * ldz,), -3.0, dtype=torch.float64, device=dev)
    zg = guard[:batch * h.nz].view(batch, h.nz)
    h.solve_device(t["x0"], t["x0_pre"], None, None, t["nu0"], 1, 1e-2, z_out=zg)
    z_auto, _, _ = h.solve_device(t["x0"], t["x0_pre"], None, None, t["nu0"], 1, 1e-2)
    torch.cuda.synchronize()
    assert torch.equal(zg, zc) and bool((guard[batch * h.nz:] == -3.0).all()), "rows written at the handle's stride"
    assert z_auto.stride(0) == ldz and torch.equal(z_auto, zc)            # (the array the method allocates takes the stride)
    # the raw C ABI: the handle's persistent stride serves fmpc_solve_device, an explicit ldz overrides it for one call
    import ctypes as C
    lib = pkg.load()
    vp = lambda x: C.c_void_p(x.data_ptr())
    big.fill_(-7.0); zg.fill_(-3.0)
    st_ = torch.empty(batch, dtype=torch.int32, device=dev); it_ = torch.empty_like(st_)
    assert lib.fmpc_solve_device(h._h, batch, vp(t["x0"]), vp(t["x0_pre"]), None, None, vp(t["nu0"]), 1, 1e-2, vp(big), None, vp(st_), vp(it_), None, None) == 0
    assert lib.fmpc_solve_u0_device_ld(h._h, batch, vp(t["x0"]), vp(t["x0_pre"]), None, None, vp(t["nu0"]), 1, 1e-2, vp(zg), None, vp(st_), vp(it_), None,
                                       None, 0, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(big[:, :h.nz], zc) and torch.equal(zg, zc) and bool((guard[batch * h.nz:] == -3.0).all())
    assert lib.fmpc_solve_u0_device_ld(h._h, batch, vp(t["x0"]), vp(t["x0_pre"]), None, None, vp(t["nu0"]), 1, 1e-2, vp(zg), None, vp(st_), vp(it_), None,
                                       None, h.nz - 1, None) == pkg._lib.FMPC_E_DIM
    w = torch.zeros((batch, h.T * h.n), dtype=torch.float64, device=dev)
    for kw in (dict(w=w), dict(n_newton=5), dict(z_init=zc.clone())):
        args = dict(w=None, z_init=None, n_newton=1); args.update(kw)
        with pytest.raises(pkg.FastMPCError) as ei:
            h.solve_device(t["x0"], t["x0_pre"], args["w"], args["z_init"], t["nu0"], args["n_newton"], 1e-2, z_out=big[:, :h.nz])
        assert ei.value.code == pkg._lib.FMPC_E_UNSUPPORTED
    # the host-pointer entry stages contiguous rows whatever the handle's stride says
    zh = h.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=1, k=1e-2)
    assert np.array_equal(zh, zc.cpu().numpy())
    h.set_z_ld(0)
    with pytest.raises(pkg.FastMPCError):
        h.set_z_ld(h.nz - 1)
    h.close()
