"""RecordedSolves: device solves captured once into a HIP graph and replayed (include/fastmpc.h, "Recording solves into a HIP
graph").  Same numbers as the eager calls, bit for bit; eager calls on other streams still work afterwards (the handle's
cross-stream event is not touched under capture); a replay is refused once device buffers have been reallocated."""
import numpy as np
import pytest

from tests.util import handle_from_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("batch,padded", [(2000, True), (70, False)])
def test_recorded_solves_replay_equals_eager(pkg, gpu, batch, padded):
    import torch
    dev = torch.device("cuda:0")
    md = pkg.synthetic.make_model(27, 144, 30)
    h = handle_from_model(pkg, md)
    nsets, reps = 3, 2
    ldz = (h.nz + 15) // 16 * 16 if padded else h.nz
    sets = []
    for i in range(nsets):
        d = pkg.synthetic.make_replay_batch(md, r=10 + i, steps=batch)
        big = torch.full((batch, ldz), -3.0, dtype=torch.float64, device=dev)
        sets.append(dict(x0=torch.from_numpy(d["x0"]).to(dev), x0p=torch.from_numpy(d["x0_pre"]).to(dev), nu0=torch.from_numpy(d["nu0"]).to(dev),
                         big=big, z=big[:, :h.nz] if padded else big, u0=torch.empty((batch, h.m), dtype=torch.float64, device=dev),
                         st=torch.zeros(batch, dtype=torch.int32, device=dev), it=torch.zeros(batch, dtype=torch.int32, device=dev)))

    def one(s):
        h.solve_device(s["x0"], s["x0p"], None, None, s["nu0"], 1, 1e-2, z_out=s["z"], status=s["st"], iters=s["it"], u0_out=s["u0"])

    def stretch():
        for _ in range(reps):
            for s in sets:
                one(s)

    stretch()
    torch.cuda.synchronize()
    ref = [(s["big"].clone(), s["u0"].clone(), s["st"].clone(), s["it"].clone()) for s in sets]
    rec = pkg.RecordedSolves(stretch)
    assert rec.valid()
    for _ in range(2):
        for s in sets:
            s["big"].fill_(-3.0); s["u0"].fill_(0.0); s["st"].fill_(-9); s["it"].fill_(-9)
        rec.replay()
        torch.cuda.synchronize()
        for s, (zb, u0, st, it) in zip(sets, ref):
            assert torch.equal(s["big"], zb) and torch.equal(s["u0"], u0) and torch.equal(s["st"], st) and torch.equal(s["it"], it)
    # eager calls afterwards, on another stream too
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        one(sets[0])
    one(sets[1])
    torch.cuda.synchronize()
    assert torch.equal(sets[0]["big"], ref[0][0]) and torch.equal(sets[1]["big"], ref[1][0])
    # a larger batch makes the handle allocate: the recording is stale from then on
    d2 = pkg.synthetic.make_replay_batch(md, r=99, steps=2 * batch + 100)
    h.solve_device(torch.from_numpy(d2["x0"]).to(dev), torch.from_numpy(d2["x0_pre"]).to(dev), None, None, torch.from_numpy(d2["nu0"]).to(dev), 1, 1e-2)
    torch.cuda.synchronize()
    assert not rec.valid()
    with pytest.raises(pkg.FastMPCError) as ei:
        rec.replay()
    assert ei.value.code == pkg._lib.FMPC_E_UNSUPPORTED
    h.close()


def test_recorded_solves_with_flagged_problems_odd_stretch(pkg, gpu):
    """Tight bounds: some problems of every step are redone by the exact path (flag mode).  The two counters of flagged problems
    take turns call by call on the host; a recorded stretch of an ODD number of steps, replayed back to back and mixed with eager
    calls, meets a counter that the step before it has not cleared -- the exact-path launch then scans the flags (a stale count can
    only be too high): same results as the eager calls every time."""
    import torch
    dev = torch.device("cuda:0")
    md = pkg.synthetic.make_model(27, 144, 30)
    md["u_min"] = -0.1 * np.ones(144); md["u_max"] = 0.1 * np.ones(144)
    h = handle_from_model(pkg, md)
    batch = 90
    sets = []
    for i in range(3):
        d = pkg.synthetic.make_replay_batch(md, r=20 + i, steps=batch)
        sc = np.linspace(0.05, 5.0, batch)[:, None] if i != 1 else np.full((batch, 1), 0.01)     # (set 1: nothing flagged)
        sets.append(dict(x0=torch.from_numpy(d["x0"] * sc).to(dev), x0p=torch.from_numpy(d["x0_pre"] * sc).to(dev), nu0=torch.from_numpy(d["nu0"]).to(dev),
                         z=torch.empty((batch, h.nz), dtype=torch.float64, device=dev), st=torch.zeros(batch, dtype=torch.int32, device=dev),
                         it=torch.zeros(batch, dtype=torch.int32, device=dev), stp=torch.zeros((batch, 1), dtype=torch.float64, device=dev)))

    def one(s):
        h.solve_device(s["x0"], s["x0p"], None, None, s["nu0"], 1, 1e-2, z_out=s["z"], status=s["st"], iters=s["it"], step=s["stp"])

    handed = []
    for s in sets:
        one(s); torch.cuda.synchronize(); handed.append(h.last_dispatch()[1])
    assert handed[0] > 0 and handed[2] > 0 and handed[1] == 0, handed
    ref = [(s["z"].clone(), s["st"].clone(), s["it"].clone(), s["stp"].clone()) for s in sets]
    rec = pkg.RecordedSolves(lambda: [one(s) for s in sets])                     # three steps: odd

    def check():
        torch.cuda.synchronize()
        for s, (z, st, it, stp) in zip(sets, ref):
            assert torch.equal(s["z"], z) and torch.equal(s["st"], st) and torch.equal(s["it"], it) and torch.equal(s["stp"], stp)

    def wipe():
        for s in sets:
            s["z"].fill_(0.0); s["st"].fill_(-9); s["it"].fill_(-9); s["stp"].fill_(0.0)

    for pattern in ("rr", "rer", "err", "reer"):
        for c_ in pattern:
            wipe()
            if c_ == "r":
                rec.replay()
            else:
                one(sets[0]); one(sets[1]); one(sets[2])
            check()
    h.close()


def test_recorded_explicit_start_budget_split_and_ramp_cold_form(pkg, gpu):
    """Recording the two-launch paths: an explicit-start batch of more than 1024 problems with a Newton budget > 1 --
    first step by the one-wavefront kernel, the listed continuations by the tiled kernel, whose launch is planned BEFORE the first
    launch and, under capture, sized without the host read-back of an earlier call's list length and without any allocation --
    and the ramp path (cold-start Woodbury form + continuation by the dense kernel).  Replays equal the eager calls bit for bit."""
    import torch
    dev = torch.device("cuda:0")
    md = pkg.synthetic.make_model(27, 144, 10)
    h = handle_from_model(pkg, md)
    B = 1100
    d = pkg.synthetic.make_replay_batch(md, r=21, steps=B)
    rng = np.random.default_rng(4)
    zi = np.tile(np.concatenate([np.zeros(144), np.zeros(27)]), (B, 10)) + 0.05 * rng.standard_normal((B, 1710))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x0, x0p, nu0, zi = t(d["x0"]), t(d["x0_pre"]), t(d["nu0"][:, :270]), t(zi)
    z = torch.empty((B, h.nz), dtype=torch.float64, device=dev); nu = torch.empty((B, 270), dtype=torch.float64, device=dev)
    st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
    stp = torch.empty((B, 3), dtype=torch.float64, device=dev)
    # a small call first: the continuation list length the handle remembers from it is shorter than what the recorded batch needs
    h.solve_device(x0[:1030], x0p[:1030], None, zi[:1030], nu0[:1030], 3, 1e-2)
    torch.cuda.synchronize()

    def call():
        h.solve_device(x0, x0p, None, zi, nu0, 3, 1e-2, z_out=z, nu_out=nu, status=st, iters=it, step=stp)
    call(); torch.cuda.synchronize()
    assert h.last_dispatch()[0] == pkg.FMPC_PATH_WAVE and int((it >= 2).sum()) > 0, "no problem went on: the case does not test the continuation"
    ref = (z.clone(), nu.clone(), st.clone(), it.clone(), stp.clone())
    rec = pkg.RecordedSolves(call)
    for _ in range(3):
        z.fill_(0.0); nu.fill_(0.0); st.fill_(-9); it.fill_(-9); stp.fill_(0.0)
        rec.replay(); torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip((z, nu, st, it, stp), ref))
    h.close()
    # ramp rows: VAR(1), cold start, budget 3 (cold form + two dense steps) and budget 1 with first moves only
    m1 = pkg.synthetic.make_model(27, 144, 10, var_order=1)
    h1 = handle_from_model(pkg, m1)
    h1.set_ramp(-0.2121 * np.ones(144), 0.2121 * np.ones(144))
    Bp = 40
    d1 = pkg.synthetic.make_replay_batch(m1, r=3, steps=Bp)
    x1, n1 = t(d1["x0"]), t(d1["nu0"][:, :270])
    up = t(0.05 * rng.standard_normal((Bp, 144)))
    z1 = torch.empty((Bp, h1.nz), dtype=torch.float64, device=dev); u1 = torch.empty((Bp, 144), dtype=torch.float64, device=dev); u1b = torch.empty_like(u1)
    s1 = torch.zeros(Bp, dtype=torch.int32, device=dev); i1 = torch.zeros(Bp, dtype=torch.int32, device=dev)

    def ramp_calls():
        h1.solve_device(x1, None, None, None, n1, 3, 1e-2, z_out=z1, status=s1, iters=i1, u_prev=up, u0_out=u1)
        h1.solve_device(x1, None, None, None, n1, 1, 1e-2, u_prev=up, u0_out=u1b, want_z=False)
    ramp_calls(); torch.cuda.synchronize()
    assert h1.last_dual_form() == 5
    ref1 = (z1.clone(), u1.clone(), u1b.clone(), s1.clone(), i1.clone())
    rec1 = pkg.RecordedSolves(ramp_calls)
    z1.fill_(0.0); u1.fill_(0.0); u1b.fill_(0.0); s1.fill_(-9); i1.fill_(-9)
    rec1.replay(); torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip((z1, u1, u1b, s1, i1), ref1))
    assert torch.equal(u1, z1[:, :144])
    h1.close()
