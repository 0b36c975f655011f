#!/usr/bin/env python3
"""bench.py -- MPC steps/s of the HIP fastMPC path on BASELINE.json's configs[1].

Workload (one "step" = one pass of the hot path over one batch): VAR(2), n = 27 Zernike modes,
m = 144 actuators, horizon T = 30, a replay batch of 2000 timesteps of one turbulence realisation
per GPU, fp64, cold start, k = 1e-2 (README.md:551), Newton-step budget 5 with the reference's
tolerance exit (test_fast_mpc.m:53,59) -- every problem is an independent
`Fast_MPC2(...).mpc_fixed_log_newton(5, 1e-2)` call of the reference.  Inputs (x0, x0_pre, nu0)
are resident in HBM before the timed region.  Synthetic data per SURVEY.md §8(d).

  python bench.py [--gpus N] [--steps K] [--warmup W]
For N > 1 launch with torch.distributed.run (one rank per GPU); ranks shard realisations
(weak scaling, no data-path collective) and all-gather the first moves u0 over RCCL each step.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, live HIP-event timing) and
`cpu_baseline` (dense restatement of the reference timed on this box's host cores, N = 1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_MODES, N_ACT, HORIZON, BATCH = 27, 144, 30, 2000
N_NEWTON, K_BAR = 5, 1e-2
FP64_PEAK_TFLOPS = 78.6      # MI355X datasheet fp64 vector = matrix peak (not in the microarch guide)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_flops(n, m, T):
    """per problem x Newton iteration (VAR(2), box on u, diagonal Q/R)."""
    return T * (2 * n * n * m + (19.0 / 3.0) * n ** 3 + 20 * n * n + 6 * (2 * n * n + n * m) + 40 * (n + m))


def algorithmic_bytes(n, m, T, streamed_factor):
    """compulsory x0,x0_pre,w in + z out; + factor written once and read once."""
    b = 8 * (2 * n + T * n + T * (n + m))
    if streamed_factor:
        b += 8 * 2 * T * 3 * n * n
    return b


def cpu_baseline(pkg, model, data):
    """The dense op-for-op restatement of the reference (oracle/dense_ref.py, literal dense D)
    on ONE problem of the same workload: about 10-30 s of host work."""
    from tests.util import dense_from_model
    try:
        from threadpoolctl import threadpool_info
        cores = max([d.get("num_threads", 1) for d in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    import numpy as np
    w0 = np.zeros(model["T"] * model["n"])      # the reference needs T*n entries (quirk D7)
    d = dense_from_model(model, data["x0"][0], data["x0_pre"][0], w0)
    info = {}
    t0 = time.perf_counter()
    d.mpc_fixed_log_newton(N_NEWTON, K_BAR, nu0=data["nu0"][0], info=info, literal_D=True)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "MPC steps/s", "cores": int(cores), "kind": "port",
            "sample": f"1 problem of the same workload (dense P'DP, dense chol, dense Schur as the "
                      f"reference; {info['iters']} Newton step(s) + exit test), {dt:.1f} s"}


def main():
    global N_NEWTON
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n-newton", type=int, default=N_NEWTON)
    args = ap.parse_args()
    N_NEWTON = args.n_newton

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run "
                             "--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs a HIP device: there is no CPU path"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    pkg = importlib.import_module("mpc-sensorlessao_amd")
    n, m, T, B = N_MODES, N_ACT, HORIZON, args.batch
    model = pkg.synthetic.make_model(n, m, T)
    data = pkg.synthetic.make_replay_batch(model, r=rank, steps=B)     # one realisation per rank
    h = pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"],
                          model["u_min"], model["u_max"], model["x_min"], model["x_max"], T,
                          device=local_rank)
    x0 = torch.from_numpy(data["x0"]).to(dev)
    x0p = torch.from_numpy(data["x0_pre"]).to(dev)
    nu0 = torch.from_numpy(data["nu0"]).to(dev)
    z = torch.empty((B, h.nz), dtype=torch.float64, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    it = torch.empty(B, dtype=torch.int32, device=dev)
    u0 = torch.empty((B, m), dtype=torch.float64, device=dev)
    u0_all = torch.empty((world * B, m), dtype=torch.float64, device=dev) if world > 1 else None

    def step():
        h.solve_device(x0, x0p, None, None, nu0, N_NEWTON, K_BAR, z_out=z, status=st, iters=it)
        h.unpack_device(z, None, None, u0)                   # first move u0 (README.md:589)
        if world > 1:
            dist.all_gather_into_tensor(u0_all, u0)          # the one collective: final gather

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    # per-launch duration of the dominant kernel: HIP events on the launch stream (torch's current)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        h.solve_device(x0, x0p, None, None, nu0, N_NEWTON, K_BAR, z_out=z, status=st, iters=it)
        ev[i][1].record()
        h.unpack_device(z, None, None, u0)
        if world > 1:
            dist.all_gather_into_tensor(u0_all, u0)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    iters_cpu = it.cpu().numpy()
    status_cpu = st.cpu().numpy()
    assert (status_cpu >= 0).all(), "solver reported errors"
    units = float(iters_cpu.sum())                  # problem x Newton-iteration units per launch

    if rank == 0:
        flops = algorithmic_flops(n, m, T) * units
        ach_tf = flops / (kern_ms * 1e-3) / 1e12
        bytes_sf = algorithmic_bytes(n, m, T, True) * units
        ach_gbs = bytes_sf / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "MPC steps/sec (n=27, VAR(2), T=30)",
            "value": world * B * args.steps / elapsed,
            "unit": "MPC steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: VAR(2), n=27, m=144, T=30, replay batch of "
                                   f"{B} timesteps of one realisation per GPU, cold start",
                       "batch_per_gpu": B, "n_newton": N_NEWTON, "k": K_BAR,
                       "newton_iters_per_problem": units / B,
                       "gather": "u0 all-gather (RCCL)" if world > 1 else "none (1 GPU)"},
            "roofline": {"bound": "mfma", "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach_tf / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": "fmpc_newton_generic", "kernel_ms": kern_ms,
                         "flops_per_unit": algorithmic_flops(n, m, T), "units_per_launch": units},
            "roofline_hbm_streamed_factor": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS,
                                             "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
                                             "bytes_per_unit": algorithmic_bytes(n, m, T, True)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, model, data)
        print(json.dumps(out), flush=True)
    h.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
