"""The `nccl` (= RCCL) branches on ONE GPU (BASELINE configs[3]): a process group of one rank with backend nccl
runs exactly the code eight ranks run -- bench.py's double-buffered asynchronous all_gather_into_tensor of the first moves on
device buffers, and ShardedFastMPC.gather's device-tensor all-gather -- where the gloo rehearsals take a host detour.  Child
processes throughout (never an exec from a process that has touched the GPU).  No scaling number is measured here."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _env(**kw):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    return env


def _one_json_line(p):
    assert p.returncode == 0, p.stderr.decode()[-4000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_bench_async_all_gather_on_a_one_rank_rccl_group(gpu):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extra", "--no-cpu-baseline"],
                       env=_env(FMPC_BENCH_FORCE_DIST="1", FMPC_BENCH_REGIONS="3"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT)
    out = _one_json_line(p)
    assert out["rccl_ranks"] == 1 and out["backend"] == "nccl" and out["n_gpus"] == 1 and out["steps"] == 20
    g = out["gather"]
    # 5 warm-up + 5 single steps + 3 regions of 20 steps, 8 steps per collective, a partial group flushed where a region ends
    assert g["async_all_gather_into_tensor"] == g["collectives_issued"] >= 3 * 3 and g["steps_per_collective"] == 8
    assert g["gathered_equals_local_bitwise"] is True
    assert out["submission"] == "eager" and out["config"]["gather"].startswith("all-gather of the first moves")
    assert len(out["ms_per_step_by_rank"]) == 1 and out["value"] > 0


def test_sharded_gather_device_branch_on_a_one_rank_rccl_group(gpu):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_one_rank_child.py")], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900, cwd=ROOT)
    res = _one_json_line(p)
    assert res["backend"] == "nccl" and res["world"] == 1
    for batch, nw in ((512, 1), (75, 3)):
        c, q = res["b%d_nw%d_collective" % (batch, nw)], res["b%d_nw%d_plain" % (batch, nw)]
        assert c["collectives"] == 3 and q["collectives"] == 0            # u0, z, U each through ONE device all-gather / none at all
        for r in (c, q):
            assert r["u0_bitwise"] and r["z_bitwise"] and r["U_bitwise"] and r["lo_hi"] == [0, batch]
