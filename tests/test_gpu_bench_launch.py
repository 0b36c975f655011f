"""`python bench.py --gpus N` from a plain shell starts its N ranks itself (VERDICT r3 "missing" 2; BASELINE configs[3]).

On a one-GPU box the ranks share device 0 and gather over gloo (FMPC_BENCH_REHEARSE=1); what is checked is the launcher,
the one-JSON-line contract and the fields of the line -- not a scaling number.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_args, env_extra, timeout=900):
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-4000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # the contract: ONE JSON line on stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_gpus2_self_launch_rehearsal(gpu):
    out = _run(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-extra", "--no-cpu-baseline"],
               {"FMPC_BENCH_REHEARSE": "1", "FMPC_BENCH_REGIONS": "3"})
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 20
    assert len(out["ms_per_step_by_rank"]) == 2 and all(t > 0 for t in out["ms_per_step_by_rank"])
    assert out["scaling"] == "weak" and out["unit"] == "MPC steps/s"
    # whole-job value = the units all ranks processed / the slowest rank's time
    assert abs(out["value"] - 2 * out["config"]["batch_per_gpu"] / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["bytes_per_unit"] == 8 * (2 * 27 + 30 * (27 + 144))      # w = NULL: no T*n doubles of w


@pytest.mark.gpu
def test_bench_single_gpu_line_has_both_rooflines(gpu):
    out = _run(["--steps", "20", "--warmup", "5", "--no-extra", "--no-cpu-baseline"], {"FMPC_BENCH_REGIONS": "3"})
    r, rp = out["roofline"], out["roofline_per_problem_factor"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # the headline's roofline describes the timed region: bytes per unit x units / device time per step
    assert abs(r["achieved"] - r["bytes_per_unit"] * r["units_per_launch"] / (r["ms_per_step_device"] * 1e-3) / 1e9) <= 1e-9 * r["achieved"]
    assert r["ms_per_step_device"] <= out["ms_per_step"] * 1.05
    # the w-less headline moves 8 (2n + T(n+m)) = 41 472 B per problem, the measured traffic is reported with its
    # ratio to that model (never dropped), and the contiguous-rows figure (the reference's layout) sits beside the padded one
    assert r["bytes_per_unit"] == 8 * (2 * 27 + 30 * 171) == 41472
    assert r["traffic"] is not None and abs(r["traffic_over_compulsory"] - r["traffic"] / (41472 * 2000)) < 1e-12
    c = r["contiguous_rows"]
    assert abs(r["frac_contiguous_rows"] - c["frac"]) < 1e-15 and abs(c["achieved"] - 41472 * 2000 / (c["ms_per_step_device"] * 1e-3) / 1e9) <= 1e-9 * c["achieved"]
    assert out["z_layout"] == "padded_5136" and out["submission"] in ("hip_graph", "eager")
    assert abs(out["value_contiguous_rows"] - c["value"]) <= 1e-9 * c["value"] and c["value"] < 1.2 * out["value"]
    assert rp["hbm_streamed_factor_model"]["bytes_per_unit"] == 41472 + 8 * 2 * 30 * 3 * 27 * 27
    assert rp["bound"] == "mfma" and "n_newton_5" in rp and rp["n_newton_5"]["frac"] > 0
    assert abs(rp["achieved"] - rp["flops_per_unit"] * rp["units_per_launch"] / (rp["ms_per_step"] * 1e-3) / 1e12) <= 1e-9 * rp["achieved"]
