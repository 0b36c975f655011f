"""GPU test of `SolveLanes` (mpc-sensorlessao_amd/lanes.py): batches dealt to several solver lanes (one handle + HIP
stream each, solves overlapping on the device) give bit-identical results to one handle solving them in turn,
and match the oracle on a sample."""
import numpy as np
import pytest

from tests.util import handle_from_model, oracle_batch, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [2, 3])
def test_lanes_equal_sequential(pkg, gpu, depth):
    import torch
    md = pkg.synthetic.make_model(27, 144, 30)
    B, nbatches = 200, 7
    dev = torch.device("cuda:0")
    sets = [pkg.synthetic.make_replay_batch(md, r=r, steps=B) for r in range(nbatches)]
    dv = [{k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in d.items() if k in ("x0", "x0_pre", "nu0")} for d in sets]
    torch.cuda.synchronize()
    lanes = pkg.SolveLanes(lambda: handle_from_model(pkg, md), B, depth=depth, device=dev)
    got = []
    for i, d in enumerate(dv):
        lane = lanes.submit(d["x0"], d["x0_pre"], None, None, d["nu0"], 1, 1e-2)   # after the clones of this lane's last result
        lanes.wait(lane)                     # torch's current stream waits for this lane only; the clones below run after it
        got.append((lane.z.clone(), lane.u0.clone(), lane.status.clone(), lane.iters.clone()))
    torch.cuda.synchronize()
    # reference: one handle, one batch after the other on the default stream
    h = handle_from_model(pkg, md)
    for i, d in enumerate(dv):
        z, st, it = h.solve_device(d["x0"], d["x0_pre"], None, None, d["nu0"], 1, 1e-2)
        _, _, u0 = h.unpack_device(z)
        torch.cuda.synchronize()
        assert torch.equal(z, got[i][0]) and torch.equal(u0, got[i][1])
        assert torch.equal(st, got[i][2]) and torch.equal(it, got[i][3])
        assert int(st.abs().sum()) == 0
    assert h.last_dispatch()[0] == pkg.FMPC_PATH_PANEL
    sub = {k: sets[3][k][:4] for k in ("x0", "x0_pre", "nu0")}
    zo, _, _, sto, _ = oracle_batch(md, sub, 1, 1e-2)
    zg = got[3][0].cpu().numpy()
    for p in range(4):
        assert sto[p] == 0 and rel_err(zg[p], zo[p]) <= 1e-9
    h.close(); lanes.close()


def test_lanes_overlap_without_waiting(pkg, gpu):
    """Submitting without waiting in between: every lane's last result is still the right one."""
    import torch
    md = pkg.synthetic.make_model(27, 144, 10)
    B, depth = 64, 2
    dev = torch.device("cuda:0")
    sets = [pkg.synthetic.make_replay_batch(md, r=r, steps=B) for r in range(4)]
    dv = [{k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in d.items() if k in ("x0", "x0_pre", "nu0")} for d in sets]
    torch.cuda.synchronize()
    lanes = pkg.SolveLanes(lambda: handle_from_model(pkg, md), B, depth=depth, device=dev)
    used = [lanes.submit(d["x0"], d["x0_pre"], None, None, d["nu0"], 2, 1e-2, after_current=False) for d in dv]
    lanes.wait()
    torch.cuda.synchronize()
    assert used[0] is used[2] and used[1] is used[3] and used[0] is not used[1]
    h = handle_from_model(pkg, md)
    for i in (2, 3):                          # the last batch each lane saw
        z, st, it = h.solve_device(dv[i]["x0"], dv[i]["x0_pre"], None, None, dv[i]["nu0"], 2, 1e-2)
        torch.cuda.synchronize()
        assert torch.equal(z, used[i].z) and torch.equal(it, used[i].iters)
    h.close(); lanes.close()


def test_lanes_first_move_ring_and_bound_calls(pkg, gpu):
    """u0_slots > 1: the first moves of consecutive submits of a lane land in consecutive ring slots; repeated submits
    with the same buffers take the prebuilt-call path and must give the same results as the checked one."""
    import torch
    md = pkg.synthetic.make_model(27, 144, 10)
    B = 48
    dev = torch.device("cuda:0")
    sets = [pkg.synthetic.make_replay_batch(md, r=r, steps=B) for r in range(2)]
    dv = [{k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in d.items() if k in ("x0", "x0_pre", "nu0")} for d in sets]
    torch.cuda.synchronize()
    lanes = pkg.SolveLanes(lambda: handle_from_model(pkg, md), B, depth=1, device=dev, u0_slots=3)
    lane = lanes.lanes[0]
    seq = [0, 0, 0, 1, 1, 0, 0]                      # same buffers repeatedly (bound path), then a change, then back
    for i, q in enumerate(seq):
        d = dv[q]
        lanes.submit(d["x0"], d["x0_pre"], None, None, d["nu0"], 1, 1e-2, after_current=False)
        assert lane.slot == i % 3
    lanes.synchronize()
    h = handle_from_model(pkg, md)
    ref = []
    for q in range(2):
        z, _, _ = h.solve_device(dv[q]["x0"], dv[q]["x0_pre"], None, None, dv[q]["nu0"], 1, 1e-2)
        ref.append(h.unpack_device(z)[2].clone())
    torch.cuda.synchronize()
    # the ring after 7 submits: slot 0 <- submit 6 (set 0), slot 1 <- submit 4 (set 1), slot 2 <- submit 5 (set 0)
    assert torch.equal(lane.u0_ring[0], ref[0]) and torch.equal(lane.u0_ring[1], ref[1]) and torch.equal(lane.u0_ring[2], ref[0])
    assert torch.equal(lane.u0, lane.u0_ring[0]) and int(lane.status.abs().sum()) == 0
    h.close(); lanes.close()
