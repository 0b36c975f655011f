"""Multi-GPU path on CPU: two gloo ranks shard a replicated batch, solve their block (the local solver
is replaced by the structured oracle -- this test covers partitioning and the one gather, not the
kernels) and all-gather u0 / z; every rank must end with the single-process result, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, batch, out_dir):
    import importlib
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("mpc-sensorlessao_amd")
    from tests.util import oracle_batch
    md, data = pkg.synthetic.make_test_problem(8, 5, 6, seed=13, batch=batch)

    def solve_fn(x0, x0_pre, w, nu0, nw, k):
        d = dict(x0=x0.numpy(), x0_pre=x0_pre.numpy(), w=w.numpy(), nu0=nu0.numpy())
        return torch.from_numpy(oracle_batch(md, d, nw, k)[0])

    sh = pkg.ShardedFastMPC(solve_fn, nz=6 * 13, m=5, T=6, n=8)
    t = {k: torch.from_numpy(v) for k, v in data.items()}
    u0 = sh.solve_gather(t["x0"], t["x0_pre"], t["w"], t["nu0"], 3, 0.01, what="u0")
    zl, lo, hi = sh.solve_local(t["x0"], t["x0_pre"], t["w"], t["nu0"], 3, 0.01)
    z = sh.gather(zl, batch, "z")
    U = sh.gather(zl, batch, "U")
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), u0=u0.numpy(), z=z.numpy(), U=U.numpy(), lo=lo, hi=hi)
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [7, 2])
def test_two_rank_shard_and_gather(tmp_path, batch, pkg):
    from tests.util import oracle_batch
    port = _free_port()
    mp.spawn(_worker, args=(2, port, batch, str(tmp_path)), nprocs=2, join=True)
    md, data = pkg.synthetic.make_test_problem(8, 5, 6, seed=13, batch=batch)
    zref = oracle_batch(md, data, 3, 0.01)[0]
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    per = -(-batch // 2)
    assert (int(r0["lo"]), int(r0["hi"])) == (0, per) and (int(r1["lo"]), int(r1["hi"])) == (per, batch)
    for r in (r0, r1):
        assert np.array_equal(r["z"], zref)
        assert np.array_equal(r["u0"], zref[:, :5])
        assert np.array_equal(r["U"], zref.reshape(batch, 6, 13)[:, :, :5].reshape(batch, 30))


def test_world_size_one_needs_no_process_group(pkg):
    from tests.util import oracle_batch
    md, data = pkg.synthetic.make_test_problem(8, 5, 6, seed=13, batch=3)
    fn = lambda x0, x0p, w, nu0, nw, k: torch.from_numpy(
        oracle_batch(md, dict(x0=x0.numpy(), x0_pre=x0p.numpy(), w=w.numpy(), nu0=nu0.numpy()), nw, k)[0])
    sh = pkg.ShardedFastMPC(fn, nz=78, m=5, T=6, n=8)
    t = {k: torch.from_numpy(v) for k, v in data.items()}
    u0 = sh.solve_gather(t["x0"], t["x0_pre"], t["w"], t["nu0"], 3, 0.01)
    assert u0.shape == (3, 5)


def _worker8(rank, world, port, batch, out_dir):
    """configs[3]'s shape: 8 ranks, rank-LOCAL inputs (nothing replicated), first moves gathered.  The local solver is a cheap
    deterministic map (this test covers partitioning, empty shards and the gather; the kernels are covered on the GPU)."""
    import importlib
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("mpc-sensorlessao_amd")
    n, m, T = 8, 5, 6
    calls = []

    def u0_fn(x0, x0_pre, w, nu0, nw, k):
        calls.append(x0.shape[0])
        return (x0[:, :m] * 2.0 + x0_pre[:, :m]).contiguous()

    def z_fn(x0, x0_pre, w, nu0, nw, k):
        z = torch.zeros((x0.shape[0], T * (n + m)), dtype=torch.float64)
        z[:, :m] = x0[:, :m] * 2.0 + x0_pre[:, :m]
        return z

    sh = pkg.ShardedFastMPC(z_fn, nz=T * (n + m), m=m, T=T, n=n, solve_u0_fn=u0_fn)
    lo, hi = sh.block(batch)
    g = torch.arange(lo, hi, dtype=torch.float64)[:, None]            # rank-local generation from the global problem index
    x0 = g + torch.arange(n, dtype=torch.float64)[None, :]
    x0p = 0.5 * x0
    u0 = sh.solve_gather_local(batch, x0, x0p, None, None, 1, 0.01, what="u0")
    z = sh.solve_gather_local(batch, x0, x0p, None, None, 1, 0.01, what="z")
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), u0=u0.numpy(), z0=z[:, :m].numpy(), lo=lo, hi=hi, calls=np.array(calls))
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [4096, 10])
def test_eight_ranks_local_blocks_and_empty_shards(tmp_path, batch, pkg):
    world = 8
    port = _free_port()
    mp.spawn(_worker8, args=(world, port, batch, str(tmp_path)), nprocs=world, join=True)
    gidx = np.arange(batch, dtype=np.float64)[:, None] + np.arange(8, dtype=np.float64)[None, :]
    ref = 2.5 * gidx[:, :5]
    per = -(-batch // world)
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        lo, hi = min(r * per, batch), min(r * per + per, batch)
        assert (int(d["lo"]), int(d["hi"])) == (lo, hi)
        assert np.array_equal(d["u0"], ref) and np.array_equal(d["z0"], ref)
        # an empty shard never calls its solver, yet takes part in the gather
        assert list(d["calls"]) == ([hi - lo] if hi > lo else [])
    if batch == 10:
        assert per == 2 and min(5 * per, batch) == batch      # ranks 5, 6, 7 are empty
