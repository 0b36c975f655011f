#!/usr/bin/env python3
"""bench.py -- MPC steps/s of the HIP fastMPC path on BASELINE.json's configs[1].

Workload (one "step" = one pass of the hot path over one batch): VAR(2), n = 27 Zernike modes,
m = 144 actuators, horizon T = 30, a replay batch of 2000 timesteps of one turbulence realisation
per GPU, fp64.  Every problem is one independent call of the reference exactly as its notebook
issues it (README.md:548-556):
    Fast_MPC2(Q,R,[],Qf,[],[],[],x_min,x_max,u_min,u_max,du_min,du_max,N,x0,x0_pre,u_prev,A1,A2,B,w,xf,[])
    .mpc_fixed_log_newton(n_fix = 1, k_fix = 1e-2)          % x_init = [] -> cold start
Inputs (x0, x0_pre, nu0) are resident in HBM before the timed region.  Synthetic data per
SURVEY.md §8(d).  From a cold start all problems share Phi, Y and its Cholesky factor in the first
Newton step; the library computes that factor once per (handle, k) -- SURVEY §7.2a regime (ii) -- so
the headline unit is priced with the survey's shared-factor figure (1.60 MFLOP), not 11.64 MFLOP.
The same JSON line also carries the general per-problem-factor path (`extra.general_path`, priced at
11.64 MFLOP per unit) and the Newton-budget-5 variant of test_fast_mpc.m (`extra.budget5`).

  python bench.py [--gpus N] [--steps K] [--warmup W]
For N > 1 launch with torch.distributed.run (one rank per GPU); ranks shard realisations
(weak scaling, no data-path collective) and all-gather the first moves u0 over RCCL each step.
"""
import argparse
import importlib
import json
import os
import sys
import time

# The HIP runtime multiplexes streams onto 4 hardware queues by default; the many-lane runs below (Newton budget 5) want one
# queue per lane.  Has to be in the environment before the runtime starts; no effect on the two-lane headline run.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_MODES, N_ACT, HORIZON, BATCH = 27, 144, 30, 2000
N_NEWTON, K_BAR = 1, 1e-2    # README.md:551-552
FP64_PEAK_TFLOPS = 78.6      # MI355X datasheet fp64 vector = matrix peak (the microarch guide has no fp64 row);
                             # scripts/mfma_f64_rate.hip measures 77.7 TFLOP/s
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec


def flops_per_problem_factor(n, m, T):
    """one problem x one Newton iteration with its own factorisation."""
    return T * (2 * n * n * m + (19.0 / 3.0) * n ** 3 + 20 * n * n + 6 * (2 * n * n + n * m) + 40 * (n + m))


def flops_shared_factor(n, m, T):
    """the same iteration when the factorisation is hoisted (regime (ii))."""
    return T * (20 * n * n + 6 * (2 * n * n + n * m) + 40 * (n + m))


def bytes_compulsory(n, m, T):
    return 8 * (2 * n + T * n + T * (n + m))


def bytes_streamed_factor(n, m, T):
    return bytes_compulsory(n, m, T) + 8 * 2 * T * 3 * n * n


def cpu_baseline(pkg, model, data, n_newton):
    """The dense op-for-op restatement of the reference (oracle/dense_ref.py, literal dense D)
    on ONE problem of the same workload: a few seconds of host BLAS work."""
    import numpy as np
    from tests.util import dense_from_model
    try:
        from threadpoolctl import threadpool_info
        cores = max([d.get("num_threads", 1) for d in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    w0 = np.zeros(model["T"] * model["n"])      # the reference needs T*n entries (quirk D7)
    d = dense_from_model(model, data["x0"][0], data["x0_pre"][0], w0)
    info = {}
    t0 = time.perf_counter()
    d.mpc_fixed_log_newton(n_newton, K_BAR, nu0=data["nu0"][0], info=info, literal_D=True)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "MPC steps/s", "cores": int(cores), "kind": "port",
            "sample": f"1 problem of the same workload (dense H, P, C, dense P'DP, dense chol, dense Schur as "
                      f"the reference; {info['iters']} Newton step(s)), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--n-newton", type=int, default=N_NEWTON)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the general-path / budget-5 variants")
    ap.add_argument("--in-flight", type=int, default=2, help="independent steps in flight per GPU (SolveLanes: one handle + HIP "
                                                             "stream each); 1 = strictly one step after the other")
    ap.add_argument("--gather-every", type=int, default=8, help="N > 1: a lane all-gathers the first moves of this many of its "
                                                                "steps in one RCCL call (every first move is gathered once)")
    ap.add_argument("--graph", action="store_true", help="replay a captured hipGraph of one step instead of launching from Python "
                                                          "(measured slower here: the step is GPU-bound, not launch-bound)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs a HIP device: there is no CPU path"
    # Rehearsal on a one-GPU box: FMPC_BENCH_REHEARSE=1 puts every rank on device 0 and gathers over gloo
    # (NCCL refuses two ranks on one device).  The real multi-GPU run uses RCCL ("nccl") below.
    rehearse = os.environ.get("FMPC_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # FMPC_BENCH_FORCE_DIST=1 (1-GPU boxes): initialise RCCL with a single rank and run the N > 1 code path (the gather
    # of the first moves, its overlap with the next step) on it
    dist_on = world > 1 or os.environ.get("FMPC_BENCH_FORCE_DIST", "0") == "1"
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    pkg = importlib.import_module("mpc-sensorlessao_amd")
    n, m, T, B = N_MODES, N_ACT, HORIZON, args.batch
    model = pkg.synthetic.make_model(n, m, T)
    data = pkg.synthetic.make_replay_batch(model, r=rank, steps=B)     # one realisation per rank

    def make_handle():
        return pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"],
                                 model["u_min"], model["u_max"], model["x_min"], model["x_max"], T,
                                 device=local_rank)

    x0 = torch.from_numpy(data["x0"]).to(dev)
    x0p = torch.from_numpy(data["x0_pre"]).to(dev)
    nu0 = torch.from_numpy(data["nu0"]).to(dev)
    overlap = {"gather": dist_on and not rehearse}

    def sync():
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def run(lanes, n_newton, steps, warmup, use_graph=False):
        """W warm-up steps, then K timed steps between barriers.  A step = solve + first-move unpack (+ the gather of
        the first moves for N > 1) of one batch, submitted to the next of `lanes` (SolveLanes): consecutive steps are
        independent batches and overlap when there is more than one lane.  Per-launch kernel time: HIP events
        recorded on the lane's stream around the solve, one step at a time (no overlap), outside the timed region."""
        G = lanes.lanes[0].u0_ring.shape[0]
        u0_all = {id(l): torch.empty((world * G * B, m), dtype=torch.float64, device=dev) for l in lanes.lanes} if dist_on else None
        pending = {id(l): None for l in lanes.lanes}
        ungathered = {id(l): 0 for l in lanes.lanes}

        def gather(lane, cnt):
            """all-gather the first `cnt` ring slots of the lane (the one collective of the job: RCCL), on its stream"""
            with torch.cuda.stream(lane.stream):
                src = lane.u0_ring[:cnt].reshape(cnt * B, m)
                dst = u0_all[id(lane)][:world * cnt * B]
                if rehearse:
                    parts = [torch.empty((cnt * B, m), dtype=torch.float64) for _ in range(world)]
                    dist.all_gather(parts, src.cpu())
                elif overlap["gather"]:
                    try:
                        pending[id(lane)] = dist.all_gather_into_tensor(dst, src, async_op=True)
                    except Exception:                            # no async support: gather in line
                        overlap["gather"] = False
                        dist.all_gather_into_tensor(dst, src)
                else:
                    dist.all_gather_into_tensor(dst, src)
            ungathered[id(lane)] = 0

        def step(ev=None):
            lane = lanes.lanes[0] if ev else lanes.next_lane()   # the evented steps run back to back on ONE lane
            if pending[id(lane)] is not None and (lane.slot + 1) % G == 0:
                with torch.cuda.stream(lane.stream):             # the ring is about to be rewritten: its gather must be done
                    pending[id(lane)].wait()
                pending[id(lane)] = None
            if ev:
                ev[0].record(lane.stream)
            lanes.submit(x0, x0p, None, None, nu0, n_newton, K_BAR, after_current=False,
                         lane=lane if ev else None)                      # solve + first move u0 (README.md:589)
            if ev:
                ev[1].record(lane.stream)
            if dist_on:
                ungathered[id(lane)] += 1
                if lane.slot == G - 1:
                    gather(lane, G)

        def drain():
            for lane in lanes.lanes:
                if dist_on and ungathered[id(lane)] > 0 and lane.slot != G - 1:
                    if pending[id(lane)] is not None:
                        with torch.cuda.stream(lane.stream):
                            pending[id(lane)].wait()
                        pending[id(lane)] = None
                    gather(lane, lane.slot + 1)                  # the last, partial group
                if pending[id(lane)] is not None:
                    with torch.cuda.stream(lane.stream):
                        pending[id(lane)].wait()
                    pending[id(lane)] = None
        for _ in range(warmup):
            step()
        drain()
        sync()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(3, min(steps, 20)))]
        for ev in evs:
            step(ev)
        drain()
        sync()
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        graph = None
        if use_graph and world == 1 and lanes.depth == 1:
            # capture the launches of one step once (hipGraph) and replay them
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=lanes.lanes[0].stream):
                step()
            graph.replay()
            sync()
        t0 = time.perf_counter()
        for i in range(steps):
            if graph is not None:
                graph.replay()
            else:
                step()
        host_s = time.perf_counter() - t0                       # host time to enqueue the K steps (diagnostic)
        drain()
        sync()
        elapsed = time.perf_counter() - t0
        run.host_enqueue_ms = host_s / steps * 1e3
        if dist_on:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        iters_cpu = lanes.lanes[0].iters.cpu().numpy()
        for lane in lanes.lanes:
            assert (lane.status.cpu().numpy() >= 0).all(), "solver reported errors"
        return elapsed, kern_ms, iters_cpu

    depth = max(1, args.in_flight)
    main_lanes = pkg.SolveLanes(make_handle, B, depth=depth, device=dev, u0_slots=max(1, args.gather_every) if dist_on else 1)
    h = main_lanes.lanes[0].handle
    u0 = main_lanes.lanes[0].u0
    elapsed, kern_ms, iters_cpu = run(main_lanes, args.n_newton, args.steps, args.warmup, use_graph=args.graph)
    host_ms_main = run.host_enqueue_ms
    path, handed = h.last_dispatch()
    shared = os.environ.get("FMPC_NO_SHARED", "0") != "1" and os.environ.get("FMPC_FORCE_GENERIC", "0") != "1"
    units_first = float((iters_cpu >= 1).sum())           # first Newton steps (shared factor when enabled)
    units_later = float(iters_cpu.sum()) - units_first    # later steps: per-problem factorisation
    f_first = flops_shared_factor(n, m, T) if shared else flops_per_problem_factor(n, m, T)
    b_first = bytes_compulsory(n, m, T) if shared else bytes_streamed_factor(n, m, T)

    extra = {}
    one_lane = pkg.SolveLanes(lambda: h, B, depth=1, device=dev) if depth > 1 else main_lanes     # shares lane 0's handle
    if rank == 0 and world == 1 and depth > 1:
        e1, k1, _ = run(one_lane, args.n_newton, args.steps, args.warmup)
        extra["one_step_at_a_time"] = {"what": "same workload with --in-flight 1: every step waits for the previous one "
                                               "(the latency-bound dual solve leaves half of the CUs idle)",
                                       "value": B * args.steps / e1, "unit": "MPC steps/s", "ms_per_step": e1 / args.steps * 1e3}
    if rank == 0 and world == 1 and not args.no_extra:
        ksteps = max(5, args.steps // 5)
        os.environ["FMPC_NO_SHARED"] = "1"
        try:
            lg = pkg.SolveLanes(make_handle, B, depth=1, device=dev)     # per-problem factorisation in every Newton step
        finally:
            del os.environ["FMPC_NO_SHARED"]
        e2, k2, i2 = run(lg, args.n_newton, ksteps, 2)
        fl = flops_per_problem_factor(n, m, T) * float(i2.sum())
        bs = bytes_streamed_factor(n, m, T) * float(i2.sum())
        extra["general_path"] = {
            "what": "same workload, every problem factors its own Y (no shared cold-start factor)",
            "value": B * ksteps / e2, "unit": "MPC steps/s", "kernel_ms": k2,
            "roofline": {"bound": "mfma", "achieved": fl / (k2 * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": fl / (k2 * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                         "flops_per_unit": flops_per_problem_factor(n, m, T)},
            "roofline_hbm_streamed_factor": {"bound": "hbm", "achieved": bs / (k2 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                             "unit": "GB/s", "frac": bs / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "bytes_per_unit": bytes_streamed_factor(n, m, T)}}
        lg.close()
        # closed loop (SURVEY 8d C2 "sequential closed loop (latency)" and C3 "512 realisations"): loop inputs +
        # solve + first-move unpack per step, all device-resident (mpc-sensorlessao_amd/closed_loop.py)
        cl = {}
        for R_, nsteps_ in ((1, 300), (512, 100)):
            a_np = np.stack([pkg.synthetic.make_realisation(model, r=r_, steps=nsteps_)[1:nsteps_ + 1] for r_ in range(min(R_, 8))], axis=1)
            a_np = np.ascontiguousarray(np.tile(a_np, (1, (R_ + a_np.shape[1] - 1) // a_np.shape[1], 1))[:, :R_])
            a_t = torch.from_numpy(a_np).to(dev)
            loop = pkg.ClosedLoop(h, R_, n_newton=args.n_newton, k=K_BAR)
            for s_ in range(5):
                loop.step(a_t[s_])
            torch.cuda.synchronize(dev)
            loop = pkg.ClosedLoop(h, R_, n_newton=args.n_newton, k=K_BAR)
            t0 = time.perf_counter()
            for s_ in range(nsteps_):
                loop.step(a_t[s_])
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            assert int(loop.status.abs().sum()) == 0
            cl["realisations_%d" % R_] = {"value": R_ * nsteps_ / dt, "unit": "MPC steps/s", "ms_per_loop_step": dt / nsteps_ * 1e3,
                                          "sequential_steps": nsteps_}
        extra["closed_loop"] = dict(what="coefficient-space closed loop (README.md:482-497,589; estimator out of scope): every step "
                                         "depends on the previous first move, so only realisations batch", **cl)
        # BASELINE configs[0] on the device: VAR(1), T = 10, ramp-rate rows on (VAR_1/fast_mpc_ineq_const.m:58-76;
        # README.md:355-356 du = +-0.2121): dense Schur complement per problem and Newton step (fmpc_kernel_ramp.hip)
        T0 = 10
        m0 = pkg.synthetic.make_model(n, m, T0, var_order=1)
        h0 = pkg.FastMPCHandle(m0["A1"], None, m0["B"], m0["Q"], m0["R"], m0["Qf"], m0["u_min"], m0["u_max"], m0["x_min"],
                               m0["x_max"], T0, var_order=1, device=local_rank)
        h0.set_ramp(-0.2121 * np.ones(m), 0.2121 * np.ones(m))
        fl0 = T0 * (T0 + 1) / 2 * 2.0 * n * n * m + (T0 * n) ** 3 / 3.0 + 2.0 * (T0 * n) ** 2 + T0 * (6.0 * (2 * n * n + n * m) + 60.0 * (n + m))
        rc0 = {"what": "configs[0] on the device: VAR(1), n=27, m=144, T=10, ramp-rate rows on; 200 timesteps of one realisation as a "
                       "replay batch, and one problem at a time (latency)",
               "flops_per_newton_iteration": fl0}
        for B0, tag in ((200, "replay_200"), (1, "single")):
            d0 = pkg.synthetic.make_replay_batch(m0, r=0, steps=B0)
            tx0 = torch.from_numpy(d0["x0"]).to(dev); tn0 = torch.from_numpy(np.ascontiguousarray(d0["nu0"][:, :T0 * n])).to(dev)
            tup = torch.from_numpy(0.05 * np.random.default_rng(7).standard_normal((B0, m))).to(dev)
            z0 = torch.empty((B0, h0.nz), dtype=torch.float64, device=dev)
            s0 = torch.empty(B0, dtype=torch.int32, device=dev); i0 = torch.empty(B0, dtype=torch.int32, device=dev)
            for nw0 in (1, 5):
                for _ in range(2):
                    h0.solve_device(tx0, None, None, None, tn0, nw0, K_BAR, z_out=z0, status=s0, iters=i0, u_prev=tup)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(10):
                    h0.solve_device(tx0, None, None, None, tn0, nw0, K_BAR, z_out=z0, status=s0, iters=i0, u_prev=tup)
                torch.cuda.synchronize(dev)
                dt = (time.perf_counter() - t0) / 10
                assert int((s0 < 0).sum()) == 0
                its = float(i0.sum().item())
                rc0["%s_budget%d" % (tag, nw0)] = {"value": B0 / dt, "unit": "MPC steps/s", "ms_per_solve": dt * 1e3,
                                                    "newton_iters_per_problem": its / B0,
                                                    "tflops": fl0 * its / dt / 1e12, "frac_of_fp64_peak": fl0 * its / dt / 1e12 / FP64_PEAK_TFLOPS}
        # ... and as the reference runs it: 200 SEQUENTIAL timesteps of one realisation, u_prev = the last first move
        a0 = pkg.synthetic.make_realisation(m0, r=0, steps=201)[1:201]
        ta0 = torch.from_numpy(np.ascontiguousarray(a0[:, None, :])).to(dev)
        for _ in range(2):
            loop0 = pkg.ClosedLoop(h0, 1, n_newton=1, k=K_BAR, ramp=True)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for s_ in range(200):
                loop0.step(ta0[s_])
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
        assert int((loop0.status < 0).sum()) == 0
        rc0["closed_loop_200_sequential_steps"] = {"value": 200 / dt, "unit": "MPC steps/s", "ms_per_loop_step": dt / 200 * 1e3, "n_newton": 1}
        h0.close()
        if not args.no_cpu_baseline:
            # the reference's dense algebra for this config on the host (oracle/dense_ref.py, checker code, timed only)
            from oracle.dense_ref import DenseFastMPC
            dref = DenseFastMPC.var1(m0["Q"], m0["R"], None, m0["Qf"], None, None, None, m0["x_min"], m0["x_max"], m0["u_min"],
                                     m0["u_max"], -0.2121 * np.ones(m), 0.2121 * np.ones(m), T0, a0[0], np.zeros(m), m0["A1"],
                                     m0["B"], np.zeros(T0 * n), None, None, ramp=True)
            t0 = time.perf_counter()
            dref.mpc_fixed_log_newton(1, K_BAR, nu0=np.zeros(T0 * n))
            rc0["cpu_port"] = {"value": 1.0 / (time.perf_counter() - t0), "unit": "MPC steps/s", "cores": os.cpu_count(),
                               "sample": "1 problem, 1 Newton step, dense H, P (4Tm x Tz), C as the reference"}
        extra["config0_var1_ramp"] = rc0
        e5, k5, i5 = run(one_lane, 5, ksteps, 2)
        extra["budget5"] = {"what": "Newton budget 5 with the reference's exit test (test_fast_mpc.m:53,59), one step at a time",
                            "value": B * ksteps / e5, "unit": "MPC steps/s", "kernel_ms": k5,
                            "newton_iters_per_problem": float(i5.sum()) / B}
        # the few problems that need more than one iteration are compacted onto a few workgroups, so independent steps
        # overlap almost completely: many lanes
        for d5 in (12,):
            l5 = pkg.SolveLanes(make_handle, B, depth=d5, device=dev)
            k5s = max(4 * d5, ksteps)
            e5l, _, i5l = run(l5, 5, k5s, d5)
            extra["budget5_in_flight_%d" % d5] = {"value": B * k5s / e5l, "unit": "MPC steps/s", "ms_per_step": e5l / k5s * 1e3,
                                                  "newton_iters_per_problem": float(i5l.sum()) / B}
            l5.close()

    if rank == 0:
        if path == pkg.FMPC_PATH_PANEL:
            kernel_name = "fmpc_cold_panel + fmpc_cold_dz + fmpc_newton_wave<27> (decision pass; %d problems redone exactly)" % handed
            nbk = T                                            # no terminal row in the bench model
            mfma = 2 * (2 * nbk - 3) * 14 + nbk * 14 + T * (7 * ((m + 15) // 16) + 28) + 63    # per 16-problem panel
            ex_fl = mfma * 2048.0 / 16.0
            executed = {"mfma_flops_per_unit": ex_fl, "tflops": ex_fl * units_first / (kern_ms * 1e-3) / 1e12,
                        "frac_of_peak": ex_fl * units_first / (kern_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}
        else:
            kernel_name = "fmpc_newton_wave<27>" if path != pkg.FMPC_PATH_GENERIC else "fmpc_newton_generic"
            executed = None
        flops = f_first * units_first + flops_per_problem_factor(n, m, T) * units_later
        byts = b_first * units_first + bytes_streamed_factor(n, m, T) * units_later
        ach_tf = flops / (kern_ms * 1e-3) / 1e12
        ach_gbs = byts / (kern_ms * 1e-3) / 1e9
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
                                   f"{B} timesteps of one realisation per GPU; each problem = the reference call "
                                   "Fast_MPC2(...,x_init=[]).mpc_fixed_log_newton(n_fix, k_fix) of README.md:548-556",
                       "batch_per_gpu": B, "n_newton": args.n_newton, "k": K_BAR,
                       "newton_iters_per_problem": float(iters_cpu.sum()) / B,
                       "cold_start_factor": "shared: one factorisation per (handle, k), SURVEY regime (ii)"
                                            if shared else "per problem",
                       "in_flight": depth,
                       "in_flight_note": "consecutive steps are independent batches (other realisations / horizon windows) and are "
                                         "dealt round-robin to `in_flight` solver lanes, one handle + HIP stream each "
                                         "(mpc-sensorlessao_amd/lanes.py); extra.one_step_at_a_time has the strictly sequential figure",
                       "gather": ("u0 all-gather (RCCL) of every %d steps of a lane, overlapped with the following solves" % max(1, args.gather_every)
                                  if overlap["gather"] else "u0 all-gather")
                                 if dist_on else "none (1 GPU)",
                       "host_enqueue_ms_per_step": host_ms_main,
                       "launch": "hipGraph replay of one step (solve + first-move unpack)" if (world == 1 and args.graph)
                                 else "one Python call per step"},
            "roofline": {"bound": "mfma", "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tf / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": kernel_name, "kernel_ms": kern_ms,
                         "flops_per_unit": f_first, "units_per_launch": units_first + units_later,
                         "executed": executed,
                         "note": "achieved = SURVEY 8d's algorithmic figure for the shared-factor regime (1.60 MFLOP and 48 KB per "
                                 "unit) x units / device time of one solve (all its kernels, HIP events on the launch stream); "
                                 "`executed` is what the panel kernels actually issue on the matrix cores (padded 16x16x4 "
                                 "tiles), less than the algorithmic figure because the constant primal start lets the step "
                                 "skip the nu-dependent residual products; the per-problem factor path is priced at 11.64 MFLOP "
                                 "under extra.general_path"},
            "roofline_hbm": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach_gbs / HBM_PEAK_GBS, "bytes_per_unit": b_first},
        }
        if extra:
            out["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, model, data, args.n_newton)
        print(json.dumps(out), flush=True)
    main_lanes.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
