#!/usr/bin/env python3
"""bench.py -- MPC steps/s of the HIP fastMPC path on BASELINE.json's configs[1], plus the other configs as `extra` legs.

Headline workload (one "step" = one pass of the hot path over one batch): VAR(2), n = 27 Zernike modes, m = 144
actuators, horizon T = 30, a replay batch of 2000 timesteps of one turbulence realisation per GPU, fp64, ONE step at a
time.  Every problem is one independent call of the reference exactly as its notebook issues it (README.md:548-556):
    Fast_MPC2(Q,R,[],Qf,[],[],[],x_min,x_max,u_min,u_max,du_min,du_max,N,x0,x0_pre,u_prev,A1,A2,B,w,xf,[])
    .mpc_fixed_log_newton(n_fix = 1, k_fix = 1e-2)          % x_init = [] -> cold start
Inputs (x0, x0_pre, nu0) are resident in HBM before the timed region; synthetic data per SURVEY.md §8(d).

The JSON line carries
  roofline             the headline's OWN kernels (what the timed region ran): bound "hbm", SURVEY §8(d)'s compulsory 47 952 B
                       per problem x problems of a step / device time of a step (HIP events around the timed regions);
                       `executed` = matrix instructions issued against the fp64 peak
  roofline_per_problem_factor
                       the PER-PROBLEM-FACTOR kernel in a leg of its own (every problem factors its own Y: explicit start),
                       priced at SURVEY §8(d)'s 11.64 MFLOP per problem-iteration, with its own ms_per_step; `n_newton_5` =
                       the Newton budget of test_fast_mpc.m:53,59, the setting §8(d) quotes the roofline on
  cpu_baseline         the dense op-for-op restatement of the reference on the host (all cores); `cpu_baseline_1thread`
                       and `cpu_baseline_structured` (oracle/banded_cpu.c, the same algorithm as the GPU) beside it
  extra                configs[2] (512 realisations), configs[4] (n = 65, T = 60, fp32 factor), two steps in flight,
                       Newton budget 5, the closed loops, configs[0] (VAR(1) + ramp rows), the literal fmpc_solve_once call

  python bench.py [--gpus N] [--steps K] [--warmup W]
For N > 1 either launch with torch.distributed.run (one rank per GPU) or run `python bench.py --gpus N` from a plain shell:
without WORLD_SIZE in the environment it starts the N ranks itself as child processes (self_launch).  Weak scaling, every rank solves its own 2000-timestep
replay batch per step and the first moves are all-gathered over RCCL; `extra.configs3_sharded` is BASELINE configs[3]
literally (4096 realisations sharded over the ranks, one all-gather of u0).
"""
import argparse
import importlib
import json
import os
import sys
import time

# set before any HIP / torch initialisation: dmabuf IPC for RCCL, one hardware queue per lane for the many-lane leg
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
if os.environ.get("FMPC_BENCH_MANY_LANES", "1") == "1":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")     # only the `budget5_in_flight_12` leg needs more than 4 queues

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_MODES, N_ACT, HORIZON, BATCH = 27, 144, 30, 2000
N_NEWTON, K_BAR = 1, 1e-2    # README.md:551-552
FP64_PEAK_TFLOPS = 78.6      # MI355X datasheet fp64 vector = matrix peak (scripts/probes/mfma_f64_rate.hip measures 77.7)
FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 matrix = vector peak
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s
MIN_LEG_MS = 50.0            # every timed GPU leg lasts at least this long
HEAD_SETS = 4                # buffer sets the headline rotates through: 4 x 82 MB of z > the 256 MB Infinity Cache
HEAD_REGIONS = int(os.environ.get("FMPC_BENCH_REGIONS", "15"))   # timed regions of EXACTLY --steps steps each; the median is reported


def flops_per_problem_factor(n, m, T):
    """one problem x one Newton iteration with its own factorisation."""
    return T * (2 * n * n * m + (19.0 / 3.0) * n ** 3 + 20 * n * n + 6 * (2 * n * n + n * m) + 40 * (n + m))


def flops_shared_factor(n, m, T):
    """the same iteration when the factorisation is hoisted (regime (ii))."""
    return T * (20 * n * n + 6 * (2 * n * n + n * m) + 40 * (n + m))


def bytes_compulsory(n, m, T, word=8, with_w=True):
    """x0, x0_pre and the disturbance w in, z out.  A call with w = NULL (the replay call every leg of this
    script makes: the affine form's precondition, fastmpc.h) moves no T*n doubles of w: with_w=False drops that term."""
    return word * (2 * n + (T * n if with_w else 0) + T * (n + m))


def bytes_streamed_factor(n, m, T, word=8, with_w=True):
    return bytes_compulsory(n, m, T, 8, with_w) + word * 2 * T * 3 * n * n


def executed_mfma_flops_panel(m, T, dense_form=False, n=27):
    """What the kernels of the panel path issue per PROBLEM: MFMAs of 2048 flop on 16-problem panels.
    Sweep form: edges of the two sweeps + S3 + d_z + S1.  Dense form without w (fmpc_cold_inv_rg): 14 k-steps per
    16-row tile of nu+, the gate's product E [x0; x0_pre], d_z."""
    dz = T * (7 * ((m + 15) // 16) + 28)
    if dense_form:
        mfma = ((n * T + 15) // 16) * 14 + 4 * 14 + dz
    else:
        mfma = 2 * (2 * T - 3) * 14 + T * 14 + dz + 63
    return mfma * 2048.0 / 16.0


def executed_mfma_flops_tiled(n, m, T):
    """What the tiled kernel issues per problem-iteration: 16x16x4 MFMAs (2048 flop) of the S pre-pass, the stage
    products, the rank-1 factorisation of the diagonal tiles, the scaling of the block rows and the residual GEMMs."""
    NB, mb, TA = n // 16 + 1, (m + 15) // 16, (T + 15) // 16
    NS, NQ = NB * (NB + 1) // 2, NB * NB
    per_stage = NS * 4 * mb + (NS * 8 * NB + NQ * 4 * NB)            # B W B' ; Ua'Ua + Uc'Uc ; Ua'Ub
    per_stage += sum(4 * kb * (NB - kb + 2 * NB) for kb in range(NB))  # products with the rows of the stage already done
    per_stage += NB * 32 + 4 * sum(NB - 1 - kb + 2 * NB for kb in range(NB))   # rank-1 updates on [S | I], scaling
    resid = TA * (2 * (4 * mb * NB + 8 * NB * NB) + 2 * (4 * NB * mb + 8 * NB * NB))  # P1 + P2 + P5 products
    return (T * per_stage + resid) * 2048.0


def cpu_cores():
    """Host threads this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box shows 256 CPUs
    in the mask but grants a share of them: round 2's 256 BLAS / OpenMP threads ran on that share)."""
    try:
        n_ = len(os.sched_getaffinity(0))
    except Exception:
        n_ = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n_ = min(n_, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q_ = int(txt[0]); per_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q_ > 0:
                    n_ = min(n_, max(1, int(q_ / per_ + 0.5)))
            break
        except Exception:
            continue
    return n_


def self_launch(n_ranks, real_out):
    """Start `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a child process
    and return its exit code.  The child ranks write their one JSON line (rank 0) to this process's real stdout."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as s_:                       # a free port on the loop-back interface
            s_.bind(("127.0.0.1", 0))
            port = str(s_.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    env.setdefault("OMP_NUM_THREADS", "1")
    real_out.flush()
    return subprocess.run(cmd, env=env, stdout=real_out.fileno(), stderr=2).returncode


def main():
    # Libraries (RCCL prints a version banner) write to stdout; the contract is ONE JSON line there.  Everything written to
    # fd 1 from here on goes to stderr, the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    real_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    try:
        _main(real_out)
    finally:
        real_out.flush()


def _main(real_out):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--n-newton", type=int, default=N_NEWTON)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="only the headline and the two roofline blocks")
    ap.add_argument("--in-flight", type=int, default=1, help="independent steps in flight per GPU for the HEADLINE (SolveLanes); the "
                                                             "configs[1] figure is 1, two in flight is reported as extra.two_in_flight")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` from a plain shell: start the N ranks as CHILD processes (one per GPU) before anything here
        # has touched the GPU, hand them the real stdout for rank 0's JSON line and leave with their exit code.  Never an exec.
        raise SystemExit(self_launch(args.gpus, real_out))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert args.gpus == world or os.environ.get("FMPC_BENCH_FORCE_DIST", "0") == "1", \
        "--gpus %d but WORLD_SIZE is %d" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a HIP device: there is no CPU path"
    # Rehearsal on a one-GPU box: FMPC_BENCH_REHEARSE=1 puts every rank on device 0 and gathers over gloo
    rehearse = os.environ.get("FMPC_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or os.environ.get("FMPC_BENCH_FORCE_DIST", "0") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if os.environ.get("FMPC_BENCH_FORCE_DIST", "0") != "1":
            assert dist.get_world_size() == args.gpus, "--gpus %d but the process group has %d ranks" % (args.gpus, dist.get_world_size())

    pkg = importlib.import_module("mpc-sensorlessao_amd")
    n, m, T, B = N_MODES, N_ACT, HORIZON, args.batch
    model = pkg.synthetic.make_model(n, m, T)
    data = pkg.synthetic.make_replay_batch(model, r=rank, steps=B)     # one realisation per rank

    def make_handle(md=model, env=None, prec=None):
        old = {}
        for k_, v_ in (env or {}).items():
            old[k_] = os.environ.get(k_); os.environ[k_] = v_
        try:
            h_ = pkg.FastMPCHandle(md["A1"], md["A2"] if md.get("var_order", 2) == 2 else None, md["B"], md["Q"], md["R"],
                                   md["Qf"], md["u_min"], md["u_max"], md["x_min"], md["x_max"], md["T"],
                                   var_order=md.get("var_order", 2), device=local_rank)
        finally:
            for k_, v_ in old.items():
                if v_ is None:
                    os.environ.pop(k_, None)
                else:
                    os.environ[k_] = v_
        if prec:
            h_.set_precision(prec)
        return h_

    def to_dev(a):
        return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    x0, x0p, nu0 = to_dev(data["x0"]), to_dev(data["x0_pre"]), to_dev(data["nu0"])

    def sync():
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed(step_fn, steps, warmup, min_ms=MIN_LEG_MS, after=None, exact=False, regions=1, all_regions=None, dev_regions=None,
              rank_times=None, region_fn=None):
        """W warm-up steps, then `steps` timed steps between barriers -- exactly `steps` for the headline (exact=True),
        otherwise as many more as it takes for the timed region to last min_ms.  regions > 1 (headline): that many timed
        regions of exactly `steps` steps back to back, each bracketed by barrier + synchronize; the MEDIAN region is
        returned (a single 20-step region lasts 1 ms and spread by 15 % from run to run in round 2), all of them in
        `all_regions`.  region_fn (headline, one GPU): the `steps` steps of a region as ONE host call -- a HIP graph of exactly
        that many solves recorded beforehand (RecordedSolves) -- instead of `steps` calls of step_fn.
        Returns (elapsed_s, steps_done, kernel_ms): kernel_ms = median device time of ONE step run alone, from HIP
        events on the stream the step is enqueued on, measured outside the timed region."""
        for _ in range(warmup):
            step_fn()
        if region_fn is not None:
            region_fn()                                                 # (the recorded form is warmed up too: one untimed replay)
        if after:
            after()
        sync()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in ev:
            torch.cuda.synchronize(dev)
            a.record(); step_fn(); b.record()
        if after:
            after()
        sync()
        kern_ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        if world == 1 and not exact:
            steps = max(steps, int(np.ceil(min_ms / max(kern_ms, 1e-3))))
        times, rev = [], []
        for _ in range(max(1, regions)):
            ra, rb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ra.record()                                                 # HIP events on the stream the steps are enqueued on
            if region_fn is not None:
                region_fn()                                             # (exactly `steps` steps, recorded once: one host call)
            else:
                for _ in range(steps):
                    step_fn()
            rb.record()
            if after:
                after()
            sync()
            times.append(time.perf_counter() - t0)
            rev.append((ra, rb))
        if dev_regions is not None:
            dev_regions.extend(a.elapsed_time(b) * 1e-3 for a, b in rev)     # seconds of device time per region
        if dist_on:
            if rank_times is not None:                                  # every rank's own median region, gathered
                mine = torch.tensor([float(np.median(times))], dtype=torch.float64, device="cpu" if rehearse else dev)
                allr = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(allr, mine)
                rank_times.extend(float(v_.item()) for v_ in allr)
            tmax = torch.tensor(times, dtype=torch.float64, device="cpu" if rehearse else dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)                 # per region: the slowest rank
            times = [float(v_) for v_ in tmax.tolist()]
        if all_regions is not None:
            all_regions.extend(times)
        elapsed = float(np.median(times))
        return elapsed, steps, kern_ms

    class Replay:
        """One handle + output buffers for a replay batch; step() = solve + first moves (README.md:589) on torch's stream."""
        def __init__(self, h, tx0, tx0p, tnu0, n_newton, z_init=None, want_z=True, pad_z=False):
            self.h, self.a = h, (tx0, tx0p, tnu0, z_init)
            Bn = tx0.shape[0]
            self.nw = n_newton
            # pad_z: the z rows of consecutive problems a multiple of 128 bytes apart (a view of a wider array, fmpc_set_z_ld): the
            # cold-start step then writes whole cache lines with non-temporal stores
            self.ldz = (h.nz + 15) // 16 * 16 if (pad_z and want_z) else h.nz
            self.zbuf = torch.empty((Bn, self.ldz), dtype=torch.float64, device=dev) if want_z else None
            self.z = (self.zbuf[:, :h.nz] if self.ldz != h.nz else self.zbuf) if want_z else None
            self.u0 = torch.empty((Bn, h.m), dtype=torch.float64, device=dev)
            self.st = torch.zeros(Bn, dtype=torch.int32, device=dev)
            self.it = torch.zeros(Bn, dtype=torch.int32, device=dev)

        def step(self, u0_out=None):
            tx0, tx0p, tnu0, zi = self.a
            self.h.solve_device(tx0, tx0p, None, zi, tnu0, self.nw, K_BAR, z_out=self.z, status=self.st, iters=self.it,
                                u0_out=self.u0 if u0_out is None else u0_out, want_z=self.z is not None)

        def check(self):
            assert int((self.st < 0).sum().item()) == 0, "solver reported errors"
            return float(self.it.sum().item())

    # ------------------------------------------------------------------ headline: configs[1], one step at a time
    depth = max(1, args.in_flight)
    h = make_handle()

    class Ring:
        """HEAD_SETS replay batches (different stretches of the realisation, own inputs AND outputs) solved in turn by one
        handle: consecutive steps neither re-read the same inputs nor overwrite the same 82 MB of z, so the output stream of a
        step really goes to HBM (4 x 82 MB exceed the 256 MB Infinity Cache; round 2 wrote one buffer over and over)."""
        def __init__(self, h_, n_newton, want_z=True, pad_z=False):
            self.sets = []
            for i_ in range(HEAD_SETS):
                d_ = data if i_ == 0 else pkg.synthetic.make_replay_batch(model, r=rank + 1000 * i_, steps=B)
                self.sets.append(Replay(h_, to_dev(d_["x0"]), to_dev(d_["x0_pre"]), to_dev(d_["nu0"]), n_newton, want_z=want_z, pad_z=pad_z))
            self.i = 0

        def step(self, u0_out=None):
            self.sets[self.i].step(u0_out)
            self.i = (self.i + 1) % len(self.sets)

        def check(self):
            return sum(s_.check() for s_ in self.sets) / len(self.sets)

    PAD_Z = args.n_newton == 1 and os.environ.get("FMPC_BENCH_CONTIGUOUS_Z", "0") != "1"   # (padded rows: the cold-start step with a budget of 1 only)
    head = Ring(h, args.n_newton, pad_z=PAD_Z)
    LDZ = head.sets[0].ldz
    # Multi-GPU: the first moves of every step are gathered (the one collective of the job, RCCL all-gather).  A collective
    # costs ~30 us of host time to submit, more than half a step: the first moves of GROUP consecutive steps share one buffer
    # and one all-gather ("fewer, larger collectives"), two buffers in turn so that a gather in flight never holds up a solve;
    # the last, partial group is gathered when the timed region ends (head_after runs inside it).
    GROUP = int(os.environ.get("FMPC_BENCH_GATHER_GROUP", "8"))
    gather_state = {"pending": [None, None], "bufs": None, "slot": 0, "fill": 0, "issued": 0, "async": 0}
    if dist_on:
        gather_state["bufs"] = [torch.empty((GROUP, B, m), dtype=torch.float64, device=dev) for _ in range(2)]
        gather_state["all"] = [torch.empty((world, GROUP, B, m), dtype=torch.float64, device=dev) for _ in range(2)]

    def gather_flush():
        s_ = gather_state["slot"]
        if gather_state["fill"] == 0:
            return
        if rehearse:
            parts = [torch.empty((GROUP, B, m), dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, gather_state["bufs"][s_].cpu())
        else:
            gather_state["pending"][s_] = dist.all_gather_into_tensor(gather_state["all"][s_], gather_state["bufs"][s_], async_op=True)
            gather_state["async"] += 1
        gather_state["issued"] += 1
        gather_state["slot"] = 1 - s_
        gather_state["fill"] = 0

    def head_step():
        if not dist_on:
            head.step()
        else:
            s_, f_ = gather_state["slot"], gather_state["fill"]
            if f_ == 0 and gather_state["pending"][s_] is not None:        # the buffer's previous gather must have read it
                gather_state["pending"][s_].wait(); gather_state["pending"][s_] = None
            head.step(u0_out=gather_state["bufs"][s_][f_])                  # the solve writes its first moves straight into the buffer
            gather_state["fill"] = f_ + 1
            if f_ + 1 == GROUP:
                gather_flush()

    def head_after():
        gather_flush()
        for s_ in (0, 1):
            if gather_state["pending"][s_] is not None:
                gather_state["pending"][s_].wait(); gather_state["pending"][s_] = None

    head_regions, head_dev_regions, head_rank_times = [], [], []
    # One GPU: the K steps of a timed region are recorded once into a HIP graph (the inputs of a replay batch are known in advance)
    # and a region is one replay -- inside a graph the launches follow each other more closely than the host can submit them.
    # With ranks the first moves are gathered every GROUP steps by the host (RCCL): submitted step by step as before.
    GRAPH = depth == 1 and not dist_on and os.environ.get("FMPC_BENCH_EAGER", "0") != "1"
    head_rec = None
    if GRAPH:
        try:
            head_rec = pkg.RecordedSolves(lambda: [head.step() for _ in range(args.steps)])
        except Exception as ex_:                                           # (a runtime without stream capture: eager submission)
            print("[bench] HIP graph capture of the headline failed (%s): eager submission" % repr(ex_)[:200], file=sys.stderr)
            head_rec, GRAPH = None, False
    if depth == 1:
        elapsed, steps_done, kern_ms = timed(head_step, args.steps, args.warmup, after=head_after, exact=True,
                                             regions=HEAD_REGIONS, all_regions=head_regions, dev_regions=head_dev_regions,
                                             rank_times=head_rank_times, region_fn=head_rec.replay if head_rec is not None else None)
    else:
        lanes = pkg.SolveLanes(make_handle, B, depth=depth, device=dev)
        lane_step = lambda: lanes.submit(x0, x0p, None, None, nu0, args.n_newton, K_BAR, after_current=False)
        elapsed, steps_done, _ = timed(lane_step, args.steps, args.warmup, after=lanes.synchronize, exact=True)
        _, _, kern_ms = timed(head_step, 5, 2)
        lanes.close()
    iters_head = head.check()
    gather_ok = None
    if dist_on and not rehearse:
        # what the all-gathers delivered: this rank's slice of every gathered buffer is, bit for bit, the buffer the solves wrote
        # (a full group fills a buffer completely; both buffers have been gathered at least once after the warm-up + regions)
        torch.cuda.synchronize(dev)
        gather_ok = bool(gather_state["issued"] >= 2 and all(torch.equal(gather_state["all"][s_][rank], gather_state["bufs"][s_]) for s_ in (0, 1)))
        assert gather_ok, "the gathered first moves differ from the local ones"
    # The same leg with the z rows of consecutive problems CONTIGUOUS (N_z doubles apart: the layout a caller of the reference's
    # N_z x batch column-major array has), same submission, same regions: reported inside `roofline` beside the padded figure.
    contig = None
    if depth == 1 and PAD_Z and not dist_on:
        hc = Ring(h, args.n_newton)
        rec_c = None
        if GRAPH:
            try:
                rec_c = pkg.RecordedSolves(lambda: [hc.step() for _ in range(args.steps)])
            except Exception as ex_:
                print("[bench] HIP graph capture of the contiguous-rows leg failed (%s): eager submission" % repr(ex_)[:200], file=sys.stderr)
        cdev = []
        eC, sC, kC = timed(hc.step, args.steps, args.warmup, exact=True, regions=HEAD_REGIONS, dev_regions=cdev,
                           region_fn=rec_c.replay if rec_c is not None else None)
        hc.check()
        contig = {"elapsed": eC, "steps": sC, "kernel_ms": kC, "ms_dev": float(np.median(cdev)) / sC * 1e3,
                  "submission": "hip_graph" if rec_c is not None else "eager"}
        del hc, rec_c
    path, handed = h.last_dispatch()
    dual_form = h.last_dual_form() if path == pkg._lib.FMPC_PATH_PANEL else 0
    dense = bool(dual_form)
    affine = dual_form == 2          # the whole step as one product (fmpc_kernel_affine.hip)
    shared = path in (pkg._lib.FMPC_PATH_PANEL, pkg._lib.FMPC_PATH_SHARED)

    extra = {}
    roof_pp = None
    if rank == 0 and world == 1:
        # ------------------------------------------------------------------ per-problem-factor regime (the roofline block)
        zc = np.tile(np.concatenate([(model["u_min"] + model["u_max"]) / 2, (model["x_min"] + model["x_max"]) / 2]), T)
        zi = to_dev(np.tile(zc, (B, 1)))          # the cold-start values passed explicitly: no shared factor, every problem factors
        fl_unit = flops_per_problem_factor(n, m, T)
        cand = {}
        for name, env, prec in (("fmpc_newton_wave<27>", {}, None),
                                ("fmpc_newton_tiled<double,2,2,11>", {"FMPC_TILED": "1"}, None)):
            hg = make_handle(env=env, prec=prec)
            rp = Replay(hg, x0, x0p, nu0, 1, z_init=zi)                  # one Newton step: every problem factors exactly once
            e_, s_, k_ = timed(rp.step, max(5, args.steps // 20), 2)
            its = rp.check()
            rp5 = Replay(hg, x0, x0p, nu0, 5, z_init=zi)                 # budget 5 with the exit test (test_fast_mpc.m:53,59)
            e5_, s5_, k5_ = timed(rp5.step, 5, 2)
            its5 = rp5.check()
            cand[name] = dict(elapsed=e_, steps=s_, kernel_ms=k_, iters=its, path=hg.last_dispatch()[0],
                              budget5=dict(kernel_ms=k5_, iters=its5, MPC_steps_per_s=B * s5_ / e5_,
                                           tflops=fl_unit * its5 / (k5_ * 1e-3) / 1e12))
            if not env:
                # eight such batches in ONE launch: every wavefront works through eight problems, the waves of a SIMD drift apart and
                # the memory-latency-bound residual phases of one overlap the matrix-core-bound factorisation of the other
                rep8 = lambda t_: t_.repeat(8, 1).contiguous()
                rp8 = Replay(hg, rep8(x0), rep8(x0p), rep8(nu0), 1, z_init=rep8(zi))
                e8_, s8_, k8_ = timed(rp8.step, 5, 2)
                its8 = rp8.check()
                cand[name]["batch_x8"] = dict(problems_per_launch=8 * B, kernel_ms=k8_, iters=its8, MPC_steps_per_s=8 * B * s8_ / e8_,
                                              tflops=fl_unit * its8 / (k8_ * 1e-3) / 1e12,
                                              frac=fl_unit * its8 / (k8_ * 1e-3) / 1e12 / FP64_PEAK_TFLOPS)
                del rp8
            hg.close()
        best = min(cand, key=lambda c_: cand[c_]["kernel_ms"])
        cb = cand[best]
        ach = fl_unit * cb["iters"] / (cb["kernel_ms"] * 1e-3) / 1e12
        by = bytes_streamed_factor(n, m, T, with_w=False) * cb["iters"] / (cb["kernel_ms"] * 1e-3) / 1e9
        ex_unit = executed_mfma_flops_tiled(n, m, T) if "tiled" in best else 6100 * 2048.0     # wave kernel: DESIGN.md §3
        roof_pp = {"bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
                   "traffic": None, "kernel": best, "kernel_ms": cb["kernel_ms"], "ms_per_step": cb["kernel_ms"],
                   "n_newton_5": {"what": "the same batch with the Newton budget of the reference's test and its exit test (test_fast_mpc.m:53,59, "
                                          "inf_newton_solver.m:19-22): the setting SURVEY 8d quotes the roofline on",
                                  "kernel_ms": cb["budget5"]["kernel_ms"], "units_per_launch": cb["budget5"]["iters"],
                                  "newton_iters_per_problem": cb["budget5"]["iters"] / B,
                                  "achieved": cb["budget5"]["tflops"], "frac": cb["budget5"]["tflops"] / FP64_PEAK_TFLOPS},
                   "flops_per_unit": fl_unit, "units_per_launch": cb["iters"], "newton_iters_per_problem": cb["iters"] / B,
                   "workload": "configs[1] batch from an explicit start z_init (no shared factor), ONE Newton step: every problem "
                               "factors its own Y exactly once per launch (the per-problem-factor regime of SURVEY 8d); the Newton-"
                               "budget-5 variant with the reference's exit test (test_fast_mpc.m:53,59; 9 % of the problems take a second "
                               "step and set the launch time) is under `candidates.*.budget5`",
                   "executed": {"mfma_flops_per_unit": ex_unit, "tflops": ex_unit * cb["iters"] / (cb["kernel_ms"] * 1e-3) / 1e12,
                                "frac_of_peak": ex_unit * cb["iters"] / (cb["kernel_ms"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
                   "hbm_streamed_factor_model": {"achieved": by, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / HBM_PEAK_GBS,
                                                 "bytes_per_unit": bytes_streamed_factor(n, m, T, with_w=False),
                                                 "bytes_per_launch": bytes_streamed_factor(n, m, T, with_w=False) * cb["iters"]},
                   "candidates": {c_: {"kernel_ms": v_["kernel_ms"], "MPC_steps_per_s": B * v_["steps"] / v_["elapsed"],
                                       "tflops": fl_unit * v_["iters"] / (v_["kernel_ms"] * 1e-3) / 1e12,
                                       "frac": fl_unit * v_["iters"] / (v_["kernel_ms"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                       "budget5": v_["budget5"], **({"batch_x8": v_["batch_x8"]} if "batch_x8" in v_ else {})}
                                  for c_, v_ in cand.items()},
                   "note": "achieved = SURVEY 8d's 11.64 MFLOP per problem-iteration x problem-iterations of one launch / its device "
                           "time (HIP events on the launch stream); the faster of the two per-problem-factor kernels is reported"}
        tpp = os.path.join(ROOT, "profiles", "traffic_general_latest.json")
        if os.path.exists(tpp) and B == BATCH:
            try:
                tj = json.load(open(tpp))
                # (the profile names the kernel with its template arguments, e.g. fmpc_newton_wave<27, false>: match on the stem)
                stem = best.split("<")[0]
                cand_t = next((v_.get("hbm_bytes_per_launch") for k_, v_ in tj.get("kernels", {}).items() if k_.split("<")[0] == stem), None)
                if cand_t is not None:
                    roof_pp["traffic"] = cand_t
                    roof_pp["traffic_over_streamed_factor_model"] = cand_t / (bytes_streamed_factor(n, m, T, with_w=False) * B)
                    roof_pp["traffic_source"] = "profiles/traffic_general_latest.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
            except Exception:
                pass

    if rank == 0 and world == 1 and not args.no_extra:
        ksteps = max(5, args.steps // 10)
        eL, sL, _ = timed(head_step, args.steps, 2)
        extra["headline_min_50ms"] = {"what": "the headline leg again over a timed region of at least %.0f ms" % MIN_LEG_MS,
                                      "value": B * sL / eL, "unit": "MPC steps/s", "steps": sL, "ms_per_step": eL / sL * 1e3}
        if GRAPH:
            eE, sE, _ = timed(head_step, args.steps, args.warmup, exact=True, regions=HEAD_REGIONS)
            extra["headline_eager_submission"] = {"what": "the headline leg with the %d steps of a region submitted one host call at a time (no HIP graph)" % sE,
                                                  "value": B * sE / eE, "unit": "MPC steps/s", "ms_per_step": eE / sE * 1e3}
        if PAD_Z:
            hc = Ring(h, args.n_newton)
            eC, sC, kC = timed(hc.step, args.steps, args.warmup)
            hc.check()
            extra["headline_contiguous_z"] = {"what": "the headline leg with the z rows of consecutive problems contiguous (N_z = %d doubles apart: a row starts "
                                                      "%d bytes off a cache line, the stores go through the L2), one host call per step over a region of at least "
                                                      "%.0f ms; the same layout in regions of exactly --steps steps and the headline's submission is "
                                                      "`roofline.contiguous_rows`" % (h.nz, (h.nz * 8) % 128, MIN_LEG_MS),
                                              "value": B * sC / eC, "unit": "MPC steps/s", "ms_per_step": eC / sC * 1e3, "kernel_ms": kC}
            del hc
        # ------------------------------------------------------------------ output options: first moves only (README.md:589)
        hu = Ring(h, args.n_newton, want_z=False)
        eU, sU, kU = timed(hu.step, args.steps, args.warmup)
        hu.check()
        extra["headline_u0_only"] = {"what": "same workload, z_out = NULL: only the first moves u0 = U(1:nu) leave the solve (what the reference's loop "
                                             "applies, README.md:589); fmpc_cold_dz writes 1.1 KB instead of 41 KB per problem",
                                     "value": B * sU / eU, "unit": "MPC steps/s", "ms_per_step": eU / sU * 1e3, "kernel_ms": kU}
        del hu
        # ------------------------------------------------------------------ two steps in flight (independent batches)
        lanes = pkg.SolveLanes(make_handle, B, depth=2, device=dev)
        e2, s2, _ = timed(lambda: lanes.submit(x0, x0p, None, None, nu0, args.n_newton, K_BAR, after_current=False),
                          args.steps, args.warmup, after=lanes.synchronize)
        extra["two_in_flight"] = {"what": "same workload, consecutive steps are INDEPENDENT batches dealt to two solver lanes (handle + "
                                          "stream each, mpc-sensorlessao_amd/lanes.py): 4000 problems resident, not the configs[1] figure",
                                  "value": B * s2 / e2, "unit": "MPC steps/s", "ms_per_step": e2 / s2 * 1e3}
        lanes.close()
        # ------------------------------------------------------------------ configs[2]: 512 realisations x 1 step
        d512 = pkg.synthetic.make_replay_batch(model, r=3, steps=512)
        r512 = Replay(h, to_dev(d512["x0"]), to_dev(d512["x0_pre"]), to_dev(d512["nu0"]), args.n_newton)
        e_, s_, k_ = timed(r512.step, args.steps, args.warmup)
        r512.check()
        extra["configs2_batch512"] = {"what": "configs[2]: 512 independent problems per step (the per-rank shape of configs[3] too), cold start",
                                      "value": 512 * s_ / e_, "unit": "MPC steps/s", "ms_per_step": e_ / s_ * 1e3, "kernel_ms": k_}
        r512b = Replay(h, to_dev(d512["x0"]), to_dev(d512["x0_pre"]), to_dev(d512["nu0"]), 5)     # every config also with n_newton = 5
        e_, s_, k_ = timed(r512b.step, 5, 2)
        extra["configs2_batch512"]["budget5"] = {"value": 512 * s_ / e_, "unit": "MPC steps/s", "kernel_ms": k_,
                                                 "newton_iters_per_problem": r512b.check() / 512}
        # ------------------------------------------------------------------ configs[4]: n = 65, T = 60, fp32 factor, batch 1024
        n4, T4, B4 = 65, 60, 1024
        m4 = pkg.synthetic.make_model(n4, m, T4)
        d4 = pkg.synthetic.make_replay_batch(m4, r=4, steps=B4)
        h4 = make_handle(md=m4, prec="f32")                      # (the fp32 factor of BASELINE configs[4]; the default at n = 65 is fp64)
        r4 = Replay(h4, to_dev(d4["x0"]), to_dev(d4["x0_pre"]), to_dev(d4["nu0"]), 1)
        e_, s_, k_ = timed(r4.step, 5, 2)
        it4 = r4.check()
        fl4 = flops_per_problem_factor(n4, m, T4)
        ex4 = executed_mfma_flops_tiled(n4, m, T4)
        extra["configs4_n65_T60_fp32"] = {
            "what": "configs[4]: VAR(2), n=65 (radial order 10), m=144, T=60, batch 1024, fp32 factor + fp64 residuals "
                    "(fmpc_newton_tiled<float,5,4,1>: two workgroups of 4 wavefronts per CU), n_newton = 1",
            "value": B4 * s_ / e_, "unit": "MPC steps/s", "ms_per_step": e_ / s_ * 1e3, "kernel_ms": k_, "dtype": "f32 factor / f64 residuals",
            "path": h4.last_dispatch()[0],
            "roofline": {"bound": "mfma", "achieved": fl4 * it4 / (k_ * 1e-3) / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": fl4 * it4 / (k_ * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, "flops_per_unit": fl4,
                         "executed": {"mfma_flops_per_unit": ex4, "frac_of_peak": ex4 * it4 / (k_ * 1e-3) / 1e12 / FP32_PEAK_TFLOPS},
                         "hbm_streamed_factor_model": {"bytes_per_unit": bytes_streamed_factor(n4, m, T4, 4, with_w=False),
                                                       "frac": bytes_streamed_factor(n4, m, T4, 4, with_w=False) * it4 / (k_ * 1e-3) / 1e9 / HBM_PEAK_GBS}}}
        r4b = Replay(h4, to_dev(d4["x0"]), to_dev(d4["x0_pre"]), to_dev(d4["nu0"]), 5)             # the same with the Newton budget of the reference's test
        e_, s_, k_ = timed(r4b.step, 3, 1)
        it4b = r4b.check()
        extra["configs4_n65_T60_fp32"]["budget5"] = {
            "value": B4 * s_ / e_, "unit": "MPC steps/s", "kernel_ms": k_, "newton_iters_per_problem": it4b / B4,
            "note": "exit test of inf_newton_solver.m:19-22 (absolute 1e-6): an fp32 factor step leaves ||r|| ~ 1e-3 where the exact step "
                    "leaves 1e-9, so the fp32 path uses more of the budget than the fp64 oracle (DESIGN.md, tiled kernel)"}
        # the same problems in fp64 (the default arithmetic; set back with fmpc_set_precision): the eight-wavefront fp64 instance of the same kernel
        # (fmpc_newton_tiled<double,5,8>, round 5) -- the fp64 answer the fp32 step is measured against, on the matrix cores
        h4.set_precision("f64")
        r4g = Replay(h4, to_dev(d4["x0"]), to_dev(d4["x0_pre"]), to_dev(d4["nu0"]), 1)
        e_, s_, k_ = timed(r4g.step, 3, 1)
        it4g = r4g.check()
        p4g = h4.last_dispatch()[0]
        z64 = r4g.z.clone()
        h4.set_precision("f32")
        r4g.step(); torch.cuda.synchronize(dev)
        extra["configs4_fp64"] = {
            "what": "configs[4]'s model and batch with everything in fp64 (the library's default arithmetic: fmpc_newton_tiled<double,5,8>, one "
                    "workgroup of 8 wavefronts per CU)",
            "value": B4 * s_ / e_, "unit": "MPC steps/s", "kernel_ms": k_, "path": p4g, "dtype": "f64",
            "roofline": {"bound": "mfma", "achieved": fl4 * it4g / (k_ * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": fl4 * it4g / (k_ * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
            "fp32_factor_step_vs_this": float((r4g.z - z64).norm() / z64.norm())}
        h4.close()
        # ------------------------------------------------------------------ a size no matrix-core kernel takes (n > 79): the generic kernel's
        # workspace instance -- correctness at any size, no speed claim
        mb_ = pkg.synthetic.make_model(96, m, 30)
        Bb = 128
        db = pkg.synthetic.make_replay_batch(mb_, r=5, steps=Bb)
        hb = make_handle(md=mb_)
        rb_ = Replay(hb, to_dev(db["x0"]), to_dev(db["x0_pre"]), to_dev(db["nu0"]), 1)
        e_, s_, k_ = timed(rb_.step, 2, 1, min_ms=0.0)
        rb_.check()
        extra["any_size_fallback_n96"] = {"what": "VAR(2), n = 96, m = %d, T = 30, %d problems, one Newton step: fmpc_newton_generic<true> (tiles in the HBM "
                                                  "workspace), the path of every n > 79" % (m, Bb),
                                          "value": Bb * s_ / e_, "unit": "MPC steps/s", "kernel_ms": k_, "path": hb.last_dispatch()[0]}
        zb64 = rb_.z.clone()
        hb.set_precision("f32")                        # 79 < n <= 111: the fp32 factor on request (fmpc_newton_tiled<float,6,8>)
        e_, s_, k_ = timed(rb_.step, 3, 1)
        rb_.check()
        extra["any_size_fallback_n96"]["fp32_factor_on_request"] = {
            "value": Bb * s_ / e_, "unit": "MPC steps/s", "kernel_ms": k_, "path": hb.last_dispatch()[0],
            "step_vs_fp64": float((rb_.z - zb64).norm() / zb64.norm())}
        hb.close()
        # ------------------------------------------------------------------ tiled kernel at (27,144,30) in both arithmetic types
        for tag, prec in (("tiled_fp32_budget1", "f32"),):
            ht = make_handle(env={"FMPC_TILED": "1"}, prec=prec)
            rt_ = Replay(ht, x0, x0p, nu0, 1, z_init=zi)
            e_, s_, k_ = timed(rt_.step, 5, 2)
            rt_.check()
            extra[tag] = {"value": B * s_ / e_, "unit": "MPC steps/s", "kernel_ms": k_,
                          "tflops": fl_unit * B / (k_ * 1e-3) / 1e12}
            ht.close()
        # ------------------------------------------------------------------ Newton budget 5 from the cold start
        r5 = Replay(h, x0, x0p, nu0, 5)
        e5, s5, k5 = timed(r5.step, ksteps, 2)
        i5 = r5.check()
        extra["budget5"] = {"what": "Newton budget 5 with the reference's exit test (test_fast_mpc.m:53,59) from the cold start, one step at a time",
                            "value": B * s5 / e5, "unit": "MPC steps/s", "kernel_ms": k5, "newton_iters_per_problem": i5 / B}
        if os.environ.get("FMPC_BENCH_MANY_LANES", "1") == "1":
            def make_lane_handle():                  # many lanes: the one-wavefront continuation shares the chip better
                hl = make_handle()
                hl.set_small_batch_kernel(False)
                return hl
            l5 = pkg.SolveLanes(make_lane_handle, B, depth=12, device=dev)
            e5l, s5l, _ = timed(lambda: l5.submit(x0, x0p, None, None, nu0, 5, K_BAR, after_current=False), max(48, ksteps), 12,
                                after=l5.synchronize)
            extra["budget5_in_flight_12"] = {"value": B * s5l / e5l, "unit": "MPC steps/s", "ms_per_step": e5l / s5l * 1e3,
                                             "note": "12 independent steps in flight (GPU_MAX_HW_QUEUES=12 set by this script)"}
            l5.close()
        # ------------------------------------------------------------------ closed loops
        cl = {}
        for R_, nsteps_ in ((1, 300), (512, 100)):
            a_np = np.stack([pkg.synthetic.make_realisation(model, r=r_, steps=nsteps_)[1:nsteps_ + 1] for r_ in range(min(R_, 8))], axis=1)
            a_np = np.ascontiguousarray(np.tile(a_np, (1, (R_ + a_np.shape[1] - 1) // a_np.shape[1], 1))[:, :R_])
            a_t = torch.from_numpy(a_np).to(dev)
            for keep_z in (True, False):
                for rep in range(2):
                    loop = pkg.ClosedLoop(h, R_, n_newton=args.n_newton, k=K_BAR, keep_z=keep_z)
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for s_ in range(nsteps_):
                        loop.step(a_t[s_])
                    torch.cuda.synchronize(dev)
                    dt = time.perf_counter() - t0
                assert int(loop.status.abs().sum()) == 0
                cl["realisations_%d%s" % (R_, "" if keep_z else "_u0_only")] = {
                    "value": R_ * nsteps_ / dt, "unit": "MPC steps/s", "ms_per_loop_step": dt / nsteps_ * 1e3, "sequential_steps": nsteps_}
        for R_ in (1, 16, 64, 512):                       # a recorded stretch in ONE host call (fmpc_loop_run_device): all steps but the last
            nst = 1000                                        # in ONE launch (the walk of fmpc_kernel_first.hip), the last by the one-step call
            a_np = np.stack([pkg.synthetic.make_realisation(model, r=r_, steps=nst)[1:nst + 1] for r_ in range(R_)], axis=1)
            a_t = torch.from_numpy(np.ascontiguousarray(a_np)).to(dev)
            for rep in range(2):
                loop = pkg.ClosedLoop(h, R_, n_newton=args.n_newton, k=K_BAR, keep_z=False)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                loop.run_recorded(a_t, want_x0=False)
                torch.cuda.synchronize(dev)
                dt = time.perf_counter() - t0
            assert int(loop.status.abs().sum()) == 0
            cl["realisations_%d_u0_only_recorded" % R_] = {"value": R_ * nst / dt, "unit": "MPC steps/s", "ms_per_loop_step": dt / nst * 1e3,
                                                           "sequential_steps": nst, "host_calls": 1}
        extra["closed_loop"] = dict(what="coefficient-space closed loop (README.md:482-497,589; estimator out of scope): every step "
                                         "depends on the previous first move, so only realisations batch; one fmpc_loop_step_device call per step. *_u0_only: z_out = NULL "
                                         "(README.md:589 applies U(1:nu) only); up to 64 realisations that is the first-move form, ONE launch + an exact-path "
                                         "launch that returns at once (fmpc_kernel_first.hip); more realisations: the same form as a product on the matrix "
                                         "cores between the loop-input kernel and the exact-path launch (fmpc_kernel_loopu0.hip); *_recorded: the whole stretch in one host call (fmpc_loop_run_device: the reference's "
                                         "simulation knows its turbulence coefficients in advance, README.md:51-93), i.e. one launch in which the workgroup of a "
                                         "realisation walks through the steps with its rows of the first-move form in registers, stopping where a step "
                                         "is not clear-cut (the exact path redoes that step)", **cl)
        # ------------------------------------------------------------------ estimator (README.md:456-480) and the loop with it
        try:
            op_ = pkg.synthetic.estimator_optics(512)
            est_ = pkg.PhaseDiversityEstimator(op_["pupil"], op_["W"], op_["zd_list"], op_["dx"], op_["range_min"] + 1, op_["range_max"] + 1,
                                               op_["A_s"], op_["b_s"])
            fl_est = 3 * (6 * 512 * 512 * 32 + 8 * 32 * 512 * 32)        # executed: 6 real flops per complex multiply-add in the first product (three real
                                                                         # products, Gauss), 8 in the second, 32 of 31 columns
            es = {"what": "phase-diversity estimator at the reference's size (len 512, 31 x 31 window, three diversities; synthetic optics: "
                          "Zs.mat is not shipped; the reference's model_approx.mat pins the linear half in tests/test_golden_model_approx.py): PSF windows as partial DFTs on the fp64 matrix cores + ad_est = G (Y_M - b_s)",
                  "executed_flops_per_screen": fl_est}
            rng_ = np.random.default_rng(5)
            for Be in (1, 256):
                scr_ = to_dev(0.3 * rng_.standard_normal((Be, 512, 512)))
                ee, se, _ = timed(lambda: est_.apply_device(scr_, colmajor=True), 10, 3)
                es["screens_%d" % Be] = {"value": Be * se / ee, "unit": "screens/s", "us_per_call": ee / se * 1e6,
                                         "executed_tflops": fl_est * Be * se / ee / 1e12, "frac_of_fp64_peak": fl_est * Be * se / ee / 1e12 / FP64_PEAK_TFLOPS}
                del scr_
            # the reference's loop, one realisation: residual screen + estimator + loop inputs + fastMPC per timestep
            a_np = pkg.synthetic.make_realisation(model, r=3, steps=40)[1:41]
            ph_ = to_dev(np.tensordot(0.03 * a_np, op_["Z"][1:], axes=1)[:, None])            # (steps, 1, 512, 512)
            ao_ = pkg.AOLoop(h, est_, op_["Z"][1:], 1, n_newton=args.n_newton, k=K_BAR)
            for s_ in range(5):
                ao_.step(ph_[s_])
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for s_ in range(5, 40):
                ao_.step(ph_[s_])
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            assert int(ao_.status.abs().sum()) == 0
            es["loop_with_estimator_1_realisation"] = {"what": "README.md:444-626 per timestep on the device: residual screen (57 MB of mode maps), three PSF "
                                                                "windows, ad_est, b_ref, fastMPC step, first moves", "ms_per_loop_step": dt / 35 * 1e3,
                                                       "value": 35 / dt, "unit": "loop steps/s"}
            phc_ = ph_.transpose(-1, -2).contiguous()                                       # the order MATLAB keeps phase_valid(:,:,k) in
            for s_ in range(5):                                                             # (warm-up of this form too, as above)
                ao_.step(phc_[s_], colmajor=True)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for s_ in range(5, 40):
                ao_.step(phc_[s_], colmajor=True)
            torch.cuda.synchronize(dev)
            dtc = time.perf_counter() - t0
            es["loop_with_estimator_1_realisation"]["ms_per_loop_step_column_major_screens"] = dtc / 35 * 1e3
            del phc_
            extra["estimator"] = es
            est_.close()
            del ph_, ao_
        except Exception as ex_:                                                               # never lose the bench line over an extra
            extra["estimator"] = {"error": repr(ex_)}
        # ------------------------------------------------------------------ configs[0] on the device: VAR(1), T = 10, ramp rows
        T0 = 10
        m0 = pkg.synthetic.make_model(n, m, T0, var_order=1)
        h0 = make_handle(md=m0)
        h0.set_ramp(-0.2121 * np.ones(m), 0.2121 * np.ones(m))
        fl0 = T0 * (T0 + 1) / 2 * 2.0 * n * n * m + (T0 * n) ** 3 / 3.0 + 2.0 * (T0 * n) ** 2 + T0 * (6.0 * (2 * n * n + n * m) + 60.0 * (n + m))
        # the cold-start step in its Woodbury form (fmpc_ramp_cold): one m x m factorisation + the passes through the constant operators
        flc = m ** 3 / 3.0 + 2.0 * m * m + 2.0 * (T0 * n) ** 2 + 2.0 * m * (m + T0 * n) + 6.0 * T0 * m * n + 2.0 * T0 * T0 * m + 60.0 * T0 * (n + m)
        rc0 = {"what": "configs[0] on the device: VAR(1), n=27, m=144, T=10, ramp-rate rows on; 200 timesteps of one realisation as a "
                       "replay batch, and as the reference runs them (200 sequential steps).  From the cold start (the reference loop's call) the "
                       "first Newton step is taken in its Woodbury form: a constant KKT matrix + a diagonal term on u_0, one m x m factorisation "
                       "per problem (fmpc_ramp_cold, round 5); later steps of a budget factor the dense (T n)^2 Schur complement",
               "flops_per_newton_iteration": fl0, "flops_first_step_cold_form": flc,
               "note": "frac_of_fp64_peak prices every Newton iteration at the DENSE form's algorithmic flops (what the reference's structure costs once "
                       "the ramp rows make Y dense); frac_executed prices the first step at what the cold form executes"}
        d0 = pkg.synthetic.make_replay_batch(m0, r=0, steps=200)
        tx0 = to_dev(d0["x0"]); tn0 = to_dev(d0["nu0"][:, :T0 * n])
        tup = to_dev(0.05 * np.random.default_rng(7).standard_normal((200, m)))
        z0 = torch.empty((200, h0.nz), dtype=torch.float64, device=dev)
        s0 = torch.empty(200, dtype=torch.int32, device=dev); i0 = torch.empty(200, dtype=torch.int32, device=dev)
        for nw0 in (1, 5):
            fn0 = lambda: h0.solve_device(tx0, None, None, None, tn0, nw0, K_BAR, z_out=z0, status=s0, iters=i0, u_prev=tup)
            e_, s_, k_ = timed(fn0, 5, 2)
            assert int((s0 < 0).sum()) == 0
            its = float(i0.sum().item())
            cold_first = h0.last_dual_form() == 5
            ex_fl = (flc * 200 + fl0 * (its - 200)) if cold_first else fl0 * its
            rc0["replay_200_budget%d" % nw0] = {"value": 200 * s_ / e_, "unit": "MPC steps/s", "ms_per_solve": e_ / s_ * 1e3, "kernel_ms": k_,
                                                 "newton_iters_per_problem": its / 200, "first_step_cold_form": cold_first,
                                                 "frac_of_fp64_peak": fl0 * its / (k_ * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                                 "frac_executed": ex_fl / (k_ * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}
        a0 = pkg.synthetic.make_realisation(m0, r=0, steps=201)[1:201]
        ta0 = torch.from_numpy(np.ascontiguousarray(a0[:, None, :])).to(dev)
        for _ in range(2):
            loop0 = pkg.ClosedLoop(h0, 1, n_newton=1, k=K_BAR, ramp=True, keep_z=False)     # (README.md:589: the loop applies U(1:nu) only)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for s_ in range(200):
                loop0.step(ta0[s_])
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
        assert int((loop0.status < 0).sum()) == 0
        rc0["closed_loop_200_sequential_steps"] = {"value": 200 / dt, "unit": "MPC steps/s", "ms_per_loop_step": dt / 200 * 1e3, "n_newton": 1}
        # the literal per-timestep call of configs[0]: the reference's 21-argument VAR_1 constructor set through fmpc_solve_once (host pointers)
        try:
            import ctypes as C0
            lib0 = pkg.load()
            lib0.fmpc_solve_once_cache_clear()
            F0 = lambda M_: np.asfortranarray(M_).ravel(order="K").copy()
            P0 = lambda a_: None if a_ is None else np.ascontiguousarray(a_, dtype=np.float64).ctypes.data_as(C0.c_void_p)
            k0 = [F0(m0["Q"]), F0(m0["R"]), F0(m0["Qf"]), F0(m0["A1"]), F0(m0["B"]), -0.2121 * np.ones(m), 0.2121 * np.ones(m), np.zeros(T0 * n)]
            up0 = 0.05 * np.random.default_rng(9).standard_normal((64, m))
            zo0 = np.empty(T0 * (n + m)); it0_ = C0.c_int()

            def once0(i_):
                return lib0.fmpc_solve_once(n, m, T0, 1, P0(k0[0]), P0(k0[1]), None, P0(k0[2]), None, None, None, P0(m0["x_min"]), P0(m0["x_max"]),
                                            P0(m0["u_min"]), P0(m0["u_max"]), P0(k0[5]), P0(k0[6]), P0(d0["x0"][i_]), None, P0(up0[i_]), P0(k0[3]), None,
                                            P0(k0[4]), P0(k0[7]), None, None, P0(d0["nu0"][i_, :T0 * n]), 1, K_BAR, local_rank, P0(zo0), C0.byref(it0_))
            t0 = time.perf_counter(); assert once0(0) == 0; tf0 = time.perf_counter() - t0
            ts0 = []
            for i_ in range(1, 41):
                t0 = time.perf_counter(); assert once0(i_) == 0; ts0.append(time.perf_counter() - t0)
            rc0["solve_once_literal_call"] = {"what": "fmpc_solve_once with the VAR_1 constructor's argument set per timestep (du_min, du_max, u_prev: ramp rows on), host "
                                                      "pointers, PCIe inclusive: first call (create + upload + the constants of the cold-start form) vs the following calls",
                                              "first_call_ms": tf0 * 1e3, "next_calls_ms_median": float(np.median(ts0)) * 1e3, "MPC_steps_per_s": 1.0 / float(np.median(ts0))}
            lib0.fmpc_solve_once_cache_clear()
        except Exception as ex_:
            rc0["solve_once_literal_call"] = {"error": repr(ex_)}
        h0.close()
        extra["config0_var1_ramp"] = rc0
        # ------------------------------------------------------------------ the literal drop-in call (host pointers, one problem)
        import ctypes as C
        lib = pkg.load()
        lib.fmpc_solve_once_cache_clear()
        F = lambda M_: np.asfortranarray(M_).ravel(order="K").copy()
        P_ = lambda a_: None if a_ is None else np.ascontiguousarray(a_, dtype=np.float64).ctypes.data_as(C.c_void_p)
        keep = [F(model["Q"]), F(model["R"]), F(model["Qf"]), F(model["A1"]), F(model["A2"]), F(model["B"]), np.zeros(m)]
        zo = np.empty(T * (n + m)); it_ = C.c_int()

        def once(k_idx):
            return lib.fmpc_solve_once(n, m, T, 2, P_(keep[0]), P_(keep[1]), None, P_(keep[2]), None, None, None, P_(model["x_min"]),
                                       P_(model["x_max"]), P_(model["u_min"]), P_(model["u_max"]), None, None, P_(data["x0"][k_idx]),
                                       P_(data["x0_pre"][k_idx]), P_(keep[6]), P_(keep[3]), P_(keep[4]), P_(keep[5]), None, None, None,
                                       P_(data["nu0"][k_idx]), 1, K_BAR, local_rank, P_(zo), C.byref(it_))
        t0 = time.perf_counter(); assert once(0) == 0; t_first = time.perf_counter() - t0
        ts = []
        for k_idx in range(1, 41):
            t0 = time.perf_counter(); assert once(k_idx) == 0; ts.append(time.perf_counter() - t0)
        extra["solve_once_literal_call"] = {"what": "fmpc_solve_once with the reference's full argument set per timestep (README.md:548-556), host "
                                                    "pointers, PCIe inclusive: first call (create + upload + factor) vs the following calls (cached handle)",
                                            "first_call_ms": t_first * 1e3, "next_calls_ms_median": float(np.median(ts)) * 1e3,
                                            "MPC_steps_per_s": 1.0 / float(np.median(ts))}
        lib.fmpc_solve_once_cache_clear()
        # ------------------------------------------------------------------ end to end through the host-pointer entry
        th = []
        for _ in range(4):
            t0 = time.perf_counter()
            zh = h.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=args.n_newton, k=K_BAR)
            th.append(time.perf_counter() - t0)
        zkeep = np.empty_like(zh)
        tk = []
        for _ in range(4):
            t0 = time.perf_counter()
            h.solve(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=args.n_newton, k=K_BAR, z_out=zkeep)
            tk.append(time.perf_counter() - t0)
        assert np.array_equal(zkeep, zh)
        extra["host_pointer_entry_output_reused"] = {"what": "the same call writing into ONE output array kept by the caller (z_out=): no first touch of fresh pages, no nu",
                                                     "ms_per_solve_median": float(np.median(tk[1:])) * 1e3, "MPC_steps_per_s": B / float(np.median(tk[1:]))}
        u0keep = np.empty((B, m)); tu, tu0 = [], []
        for _ in range(6):
            t0 = time.perf_counter()
            h.solve_u0(data["x0"], data["x0_pre"], None, nu0=data["nu0"], n_newton=args.n_newton, k=K_BAR, u0_out=u0keep)
            tu.append(time.perf_counter() - t0)
        assert np.array_equal(u0keep, zh[:, :m])
        for _ in range(6):
            t0 = time.perf_counter()
            h.solve_u0(data["x0"], data["x0_pre"], None, nu0=None, n_newton=args.n_newton, k=K_BAR, u0_out=u0keep)
            tu0.append(time.perf_counter() - t0)
        extra["host_pointer_entry_u0"] = {"what": "fmpc_solve_u0 with HOST pointers on the headline batch (README.md:589: the caller applies U(1:nu) only): H2D of x0, x0_pre "
                                                  "(0.86 MB) and nu0 (13 MB), the solve with z_out = NULL, D2H of the first moves (%.1f MB instead of %.0f MB of z); "
                                                  "`nu0_null`: the same with nu0 = NULL (zeros: the caller that does not mirror MATLAB's rand stream)" % (u0keep.nbytes / 1e6, zh.nbytes / 1e6),
                                          "ms_per_solve_median": float(np.median(tu[1:])) * 1e3, "MPC_steps_per_s": B / float(np.median(tu[1:])),
                                          "nu0_null": {"ms_per_solve_median": float(np.median(tu0[1:])) * 1e3, "MPC_steps_per_s": B / float(np.median(tu0[1:]))}}
        extra["host_pointer_entry"] = {"what": "fmpc_solve with HOST pointers on the headline batch: H2D of x0, x0_pre, nu0, the solve, D2H of z (%.0f MB), "
                                               "pageable host memory, a FRESH output array per call (its first touch is most of the time: the same copy into a buffer "
                                               "that has been written before runs at 55 GB/s), ctypes; never `value`" % (zh.nbytes / 1e6),
                                       "ms_per_solve_median": float(np.median(th[1:])) * 1e3, "MPC_steps_per_s": B / float(np.median(th[1:]))}

    if dist_on and not args.no_extra:
        # ------------------------------------------------------------------ configs[3] literally: 4096 realisations sharded over the ranks
        BG = 4096
        sh = pkg.ShardedFastMPC.from_handle(h)
        lo, hi = sh.block(BG)                                                  # this rank's contiguous block of realisations
        # every rank generates ONLY its own block (realisation g = one problem, seeds 1000 + g / 5000 + g as in SURVEY 8d);
        # nothing of the global batch is replicated
        blk = [pkg.synthetic.make_replay_batch(model, r=100 + g_, steps=1) for g_ in range(lo, hi)]
        cat = lambda k_: to_dev(np.concatenate([b_[k_] for b_ in blk], axis=0)) if blk else torch.empty((0, n if k_ != "nu0" else T * n), dtype=torch.float64, device=dev)
        lx0, lx0p, lnu = cat("x0"), cat("x0_pre"), cat("nu0")
        fn3 = lambda: sh.solve_gather_local(BG, lx0, lx0p, None, lnu, args.n_newton, K_BAR, what="u0")
        e3, s3, k3 = timed(fn3, max(20, args.steps // 4), 5)
        if rank == 0:
            extra["configs3_sharded"] = {"what": "configs[3]: 4096 realisations sharded over the ranks (contiguous blocks generated rank-locally, no "
                                                 "data-path collective) + ONE all-gather of the first moves per step (ShardedFastMPC.solve_gather_local)",
                                         "value": BG * s3 / e3, "unit": "MPC steps/s", "ms_per_step": e3 / s3 * 1e3, "ranks": world,
                                         "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "device_ordinal": local_rank,
                                         "problems_per_rank": -(-BG // world), "this_rank_block": [lo, hi]}

    if rank == 0:
        units = iters_head
        ex_fl = executed_mfma_flops_panel(m, T, dense) if path == pkg._lib.FMPC_PATH_PANEL else None
        if affine:      # 14 k-steps per 16 x 16 tile of z (T (n + m) rows), + the two decision forms (4 row tiles each); per problem = / 16
            ex_fl = (((T * (n + m) + 15) // 16) * 14 + 2 * 4 * 14) * 2048.0 / 16.0
        # measured HBM traffic of one solve: from the committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes of THIS workload
        # (scripts/prof_round2.sh writes profiles/traffic_latest.json); not measured inside this run, and only
        # reported when it is consistent with the compulsory bytes
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and B == BATCH and path == pkg._lib.FMPC_PATH_PANEL:
            try:
                tj = json.load(open(tpath))
                cand_t = tj.get("hbm_bytes_per_launch")
                if cand_t is not None:                                   # reported as measured, whatever its ratio to the model
                    traffic, traffic_src = cand_t, "profiles/traffic_latest.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, " + tj.get("tag", "") + ")"
            except Exception:
                traffic = None
        # the headline's own roofline: SURVEY 8d's compulsory bytes per unit x the units of one step / the device time of one
        # step, from HIP events around every timed region on the launch stream (median region / its steps)
        ms_step_wall = elapsed / steps_done * 1e3
        ms_step_dev = (float(np.median(head_dev_regions)) / steps_done * 1e3) if head_dev_regions else ms_step_wall
        b_unit = bytes_compulsory(n, m, T, with_w=False)                 # every headline problem is solved with w = NULL: no T*n doubles of w move
        head_kernel = (("fmpc_cold_affine<true> (the whole step as one product z+ = zc + Kz [x0; x0_pre] on the matrix cores, decision forms included) "
                        "+ fmpc_newton_wave<27> (flag mode: %d problems redone exactly)" % handed) if affine else
                       ((("fmpc_cold_inv_rg<2,4,1,false> (dense form of the dual solve: nu+ = nuc + J [x0; x0_pre], w = NULL)" if dense else "fmpc_cold_panel")
                         + " + fmpc_cold_dz + fmpc_newton_wave<27> (decision pass; %d problems redone exactly)" % handed)
                        if path == pkg._lib.FMPC_PATH_PANEL else "path %d" % path))
        ach_gbs = b_unit * units / (ms_step_dev * 1e-3) / 1e9
        roof_head = {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_over_compulsory": None if traffic is None else traffic / (b_unit * B),
                     "bytes_per_unit": b_unit, "units_per_launch": units, "ms_per_step_device": ms_step_dev, "ms_per_step_wall": ms_step_wall,
                     "frac_on_wall_clock": b_unit * units / (ms_step_wall * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "kernel": head_kernel, "kernel_ms_one_step_alone": kern_ms,
                     "executed": None if ex_fl is None else {"mfma_flops_per_unit": ex_fl, "tflops": ex_fl * units / (ms_step_dev * 1e-3) / 1e12,
                                                              "peak": FP64_PEAK_TFLOPS,
                                                              "frac_of_peak": ex_fl * units / (ms_step_dev * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
                     "note": "what the timed region ran: achieved = 8 (2n + T(n+m)) = %d B per problem (SURVEY 8d's compulsory bytes 8 (2n + Tn + T(n+m)) "
                             "WITHOUT the T n doubles of w: the replay call passes w = NULL, so only x0, x0_pre come in and z goes out) "
                             "x problems of a step / device time of a step (HIP events on the launch stream around each timed region of exactly "
                             "--steps steps; median region). With w = NULL and one Newton step from the cold start the step is affine in [x0; x0_pre] "
                             "(SURVEY regime (ii), the factor hoisted per (handle, k)), so the kernel is a product whose output is the traffic; "
                             "`executed` = matrix instructions actually issued on padded tiles against the fp64 peak" % b_unit}
        if contig is not None:
            ach_c = b_unit * units / (contig["ms_dev"] * 1e-3) / 1e9
            roof_head["contiguous_rows"] = {"what": "the same leg with the z rows N_z = %d doubles apart (the reference's N_z x batch column-major layout: a row "
                                                    "starts %d bytes off a cache line, ordinary stores through the L2), same regions of exactly %d steps"
                                                    % (T * (n + m), (T * (n + m) * 8) % 128, contig["steps"]),
                                            "value": B * contig["steps"] / contig["elapsed"], "unit": "MPC steps/s", "submission": contig["submission"],
                                            "ms_per_step_wall": contig["elapsed"] / contig["steps"] * 1e3, "ms_per_step_device": contig["ms_dev"],
                                            "achieved": ach_c, "frac": ach_c / HBM_PEAK_GBS}
            roof_head["frac_contiguous_rows"] = ach_c / HBM_PEAK_GBS
        out = {
            "metric": "MPC steps/sec (n=27, VAR(2), T=30)",
            "value": world * B * steps_done / elapsed,
            "unit": "MPC steps/s",
            "n_gpus": world, "steps": steps_done, "warmup": args.warmup,
            "ms_per_step": elapsed / steps_done * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            # the two conditions `value` is measured under, as fields a reader can compare between rounds: how the steps
            # of a timed region are submitted and how the rows of z lie in HBM; the contiguous-rows figure is value_contiguous_rows
            "submission": "hip_graph" if GRAPH else "eager", "z_layout": ("padded_%d" % LDZ) if PAD_Z else "contiguous",
            "value_contiguous_rows": None if contig is None else world * B * contig["steps"] / contig["elapsed"],
            "config": {"workload": "configs[1]: VAR(2), n=27, m=144, T=30, replay batch of "
                                   f"{B} timesteps of one realisation per GPU, one step at a time; each problem = the reference call "
                                   "Fast_MPC2(...,x_init=[]).mpc_fixed_log_newton(n_fix, k_fix) of README.md:548-556",
                       "batch_per_gpu": B, "n_newton": args.n_newton, "k": K_BAR,
                       "newton_iters_per_problem": iters_head / B, "in_flight": depth,
                       "cold_start_factor": "shared: one factorisation per (handle, k), SURVEY regime (ii)" if shared else "per problem",
                       "cold_start_dual_solve": ("affine form of the whole step: z+ = zc + Kz [x0; x0_pre] (w = NULL, one Newton step), Kz built once per (handle, k) "
                                                 "on the host from the shared factor; one matrix product per batch" if shared and affine else
                                                 "dense form: nu+ = nuc + J [x0; x0_pre] (w = NULL), J built once per (handle, k) from the shared factor"
                                                 if shared and dense else "two sweeps through the shared block factor (panels of 16 problems)") if shared else None,
                       "gather": ("all-gather of the first moves u0 (RCCL), one collective per %d steps (and at the end of the timed region), two buffers in turn" % GROUP) if dist_on else "none (1 GPU)",
                       "steps_requested": args.steps,
                       "timing": "median of %d timed regions of exactly %d steps each (barrier + synchronize on both sides of every region); "
                                 "regions (ms): min %.3f, median %.3f, max %.3f" % (len(head_regions) or 1, steps_done,
                                                                                   1e3 * min(head_regions or [elapsed]), 1e3 * elapsed, 1e3 * max(head_regions or [elapsed])),
                       "buffers": "%d input/output sets solved in turn (%.0f MB of z in rotation: larger than the 256 MB Infinity Cache)" % (HEAD_SETS, HEAD_SETS * B * T * (n + m) * 8 / 1e6),
                       "submission": ("the %d steps of a timed region are recorded once into a HIP graph (RecordedSolves: torch.cuda.CUDAGraph around the "
                                      "C-ABI calls, capture-safe once the workspaces exist) and a region is one replay; `extra.headline_eager_submission` is "
                                      "the same leg with one host call per step" % steps_done) if GRAPH else "one host call per step",
                       "z_layout": ("rows of consecutive problems %d doubles apart (N_z = %d padded to a multiple of 128 bytes, fmpc_set_z_ld: whole cache lines, "
                                    "non-temporal stores); `extra.headline_contiguous_z` is the same leg with contiguous rows" % (LDZ, T * (n + m))) if PAD_Z
                                   else "contiguous rows (N_z = %d doubles)" % (T * (n + m)),
                       "ranks": world, "rccl_ranks": dist.get_world_size() if dist_on else 1, "device_ordinal": local_rank},
            "roofline": roof_head,
        }
        if roof_pp is not None:
            out["roofline_per_problem_factor"] = roof_pp
        if dist_on:
            out["rccl_ranks"] = dist.get_world_size()
            out["backend"] = dist.get_backend()
            out["ms_per_step_by_rank"] = [t_ / steps_done * 1e3 for t_ in head_rank_times]
            out["gather"] = {"collectives_issued": gather_state["issued"], "async_all_gather_into_tensor": gather_state["async"],
                             "steps_per_collective": GROUP, "gathered_equals_local_bitwise": gather_ok}
        if extra:
            out["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out.update(cpu_baselines(pkg, model, data, args.n_newton))
        real_out.write(json.dumps(out) + "\n"); real_out.flush()
    h.close()
    if dist_on:
        dist.destroy_process_group()


def cpu_baselines(pkg, model, data, n_newton):
    """Host-side baselines on a bounded sample, AFTER every GPU leg (host BLAS threads disturb launch timing):
    the dense op-for-op restatement of the reference (oracle/dense_ref.py, literal dense D; `kind: port`) on ONE problem
    with all cores (3 samples) and with 1 thread (1 sample), and the structured C solver (oracle/banded_cpu.c)."""
    import numpy as np
    from tests.util import dense_from_model
    cores = cpu_cores()
    w0 = np.zeros(model["T"] * model["n"])      # the reference needs T*n entries (quirk D7)

    def dense_once(k_idx):
        d = dense_from_model(model, data["x0"][k_idx], data["x0_pre"][k_idx], w0)
        info = {}
        t0 = time.perf_counter()
        d.mpc_fixed_log_newton(n_newton, K_BAR, nu0=data["nu0"][k_idx], info=info, literal_D=True)
        return time.perf_counter() - t0, info["iters"]
    out = {}
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    # thread counts: everything the affinity mask / cgroup quota shows, and 16 (the CPU share of a one-GPU box may be smaller
    # than what the mask shows: more threads than granted cores only add contention); the better one is the baseline
    cand_thr = sorted({cores, min(cores, 16)})
    res = {}
    for thr in cand_thr:
        if threadpool_limits:
            with threadpool_limits(limits=thr):
                res[thr] = [dense_once(k_)[0] for k_ in range(3 if thr == cand_thr[-1] or len(cand_thr) == 1 else 2)]
        else:
            res[thr] = [dense_once(k_)[0] for k_ in range(3)]
    best_thr = min(res, key=lambda t_: float(np.median(res[t_])))
    ts = res[best_thr]
    out["cpu_baseline"] = {"value": 1.0 / float(np.median(ts)), "unit": "MPC steps/s", "cores": best_thr, "kind": "port",
                           "sample": "%d problems of the same workload, one at a time (dense H, P, C, dense P'DP, dense chol, dense Schur as the "
                                     "reference; %d Newton step), BLAS on %d threads, median %.2f s (min %.2f, max %.2f); thread counts tried: %s"
                                     % (len(ts), n_newton, best_thr, float(np.median(ts)), min(ts), max(ts),
                                        ", ".join("%d -> %.2f s" % (t_, float(np.median(v_))) for t_, v_ in sorted(res.items())))}
    if threadpool_limits:
        with threadpool_limits(limits=1):
            t1, _ = dense_once(3)
        out["cpu_baseline_1thread"] = {"value": 1.0 / t1, "unit": "MPC steps/s", "cores": 1, "kind": "port",
                                       "sample": "1 problem, BLAS limited to 1 thread (the reference's tic/toc is a single MATLAB thread), %.1f s" % t1}
    try:
        from oracle import banded_cpu
        take = lambda cnt: {k_: (None if v_ is None else np.ascontiguousarray(np.tile(v_, ((cnt + len(v_) - 1) // len(v_), 1))[:cnt]))
                            for k_, v_ in data.items()}
        # one thread: 32 problems, repeated until 1 s has passed
        d1 = take(32)
        o1 = banded_cpu.solve_batch(model, d1, n_newton, K_BAR, threads=1)
        t0 = time.perf_counter(); r1 = 0
        while time.perf_counter() - t0 < 1.0:
            banded_cpu.solve_batch(model, d1, n_newton, K_BAR, threads=1, out=o1); r1 += 1
        t1 = (time.perf_counter() - t0) / (32 * r1)
        # many threads: 32 problems PER THREAD in one call (OpenMP over the batch, per-thread workspace, outputs first touched by
        # the warm-up call and then reused), repeated until 1 s has passed, for several thread counts up to what the affinity
        # mask shows; the best is reported with ITS thread count.  (Round 2 measured one problem per thread per call on 256
        # threads, i.e. mostly thread start-up and first touch; and a one-GPU box grants fewer cores than its mask shows.)
        best = None; tried = []
        for nthr in sorted({t_ for t_ in (8, 16, 32, 64, cores) if t_ <= cores}):
            dn = take(32 * nthr)
            on = banded_cpu.solve_batch(model, dn, n_newton, K_BAR, threads=nthr)
            t0 = time.perf_counter(); rn = 0
            while time.perf_counter() - t0 < 1.0:
                banded_cpu.solve_batch(model, dn, n_newton, K_BAR, threads=nthr, out=on); rn += 1
            tn = (time.perf_counter() - t0) / (32 * nthr * rn)
            tried.append((nthr, 1.0 / tn))
            if best is None or tn < best[1]:
                best = (nthr, tn, rn)
        nthr, tn, rn = best
        out["cpu_baseline_structured"] = {"value": 1.0 / tn, "unit": "MPC steps/s", "cores": nthr, "kind": "port",
                                          "one_thread_value": 1.0 / t1, "speedup_over_one_thread": t1 / tn,
                                          "sample": "oracle/banded_cpu.c (block-penta-diagonal Newton step, the GPU's algorithm; OpenMP over the "
                                                    "batch): %d problems per call (32 per thread) x %d calls on %d threads; 32 problems x %d calls on 1 "
                                                    "thread; steps/s by thread count: %s (the speed-up stops where the box's CPU share ends)"
                                                    % (32 * nthr, rn, nthr, r1, ", ".join("%d: %.0f" % t_ for t_ in tried))}
    except Exception as e:     # the baseline library is optional
        out["cpu_baseline_structured"] = {"error": repr(e)}
    # The headline's OWN algorithm on the host cores (VERDICT r4 "missing" 4): from the cold start with w = NULL and one Newton step
    # the step is affine in d = [x0; x0_pre], z+ = zc + Kz d -- the GPU multiplies by a prebuilt 5130 x 56 matrix and factors nothing,
    # so the like-for-like CPU figure is ONE dgemm Z (B x N_z) = [D | 1] (B x 55) . [Kz | zc]' (55 x N_z) over the whole batch.
    # Kz, zc come from the checker (oracle/banded_cpu.c on 2n + 1 perturbed copies of problem 0: exact differences of an affine map)
    # and the product is checked against the checker's own solves before it is timed.
    try:
        from oracle import banded_cpu
        n_, T_ = model["n"], model["T"]
        nd = 2 * n_
        eps_ = 1e-2
        base = np.concatenate([data["x0"][0], data["x0_pre"][0]])
        Dp = np.tile(base, (nd + 1, 1)); Dp[1:] += eps_ * np.eye(nd)
        probe = {"x0": np.ascontiguousarray(Dp[:, :n_]), "x0_pre": np.ascontiguousarray(Dp[:, n_:]), "w": None, "nu0": np.tile(data["nu0"][0], (nd + 1, 1))}
        zp, _, _, stp, tp = banded_cpu.solve_batch(model, probe, 1, K_BAR)
        assert int(np.abs(stp).sum()) == 0 and bool((tp[:, 0] == 1.0).all()), "probe problems must take the full step"
        Kz = (zp[1:] - zp[0]).T / eps_                                  # N_z x 2n
        zc_ = zp[0] - Kz @ base
        K1 = np.ascontiguousarray(np.concatenate([Kz, zc_[:, None]], axis=1).T)      # (2n + 1) x N_z
        Bn = data["x0"].shape[0]
        D1 = np.ascontiguousarray(np.concatenate([data["x0"], data["x0_pre"], np.ones((Bn, 1))], axis=1))
        Z = np.empty((Bn, K1.shape[1]))
        np.matmul(D1, K1, out=Z)
        chk = {k_: (None if v_ is None else np.ascontiguousarray(v_[100:108])) for k_, v_ in data.items()}
        zo_, _, _, sto_, to_ = banded_cpu.solve_batch(model, chk, 1, K_BAR)
        ok_ = [p_ for p_ in range(8) if to_[p_, 0] == 1.0 and sto_[p_] == 0]
        err_ = max(float(np.linalg.norm(Z[100 + p_] - zo_[p_]) / np.linalg.norm(zo_[p_])) for p_ in ok_)
        assert err_ <= 1e-8, err_
        resA = {}
        for thr in sorted({1, min(cores, 8), min(cores, 16), cores}):
            def run_():
                np.matmul(D1, K1, out=Z)                                 # (warm: Z has been written before)
                t0 = time.perf_counter(); r_ = 0
                while time.perf_counter() - t0 < 1.5:
                    np.matmul(D1, K1, out=Z); r_ += 1
                return (time.perf_counter() - t0) / r_, r_
            if threadpool_limits:
                with threadpool_limits(limits=thr):
                    resA[thr] = run_()
            else:
                resA[cores] = run_(); break
        bt = min(resA, key=lambda t_: resA[t_][0])
        out["cpu_baseline_affine"] = {"value": Bn / resA[bt][0], "unit": "MPC steps/s", "cores": bt, "kind": "port",
                                      "ms_per_step": resA[bt][0] * 1e3, "gflops": 2.0 * Bn * K1.shape[0] * K1.shape[1] / resA[bt][0] / 1e9,
                                      "sample": "the headline's own algorithm on the host: one dgemm [D | 1] (%d x %d) . [Kz | zc]' (%d x %d) per step of %d problems "
                                                "(numpy / OpenBLAS, output array reused), %d repetitions on %d threads; steps/s by thread count: %s; the product equals "
                                                "the structured checker's solves to %.1e on 8 problems (Kz from exact differences of the checker on %d probes)"
                                                % (Bn, K1.shape[0], K1.shape[0], K1.shape[1], Bn, resA[bt][1], bt,
                                                   ", ".join("%d: %.0f" % (t_, Bn / v_[0]) for t_, v_ in sorted(resA.items())), err_, nd + 1)}
    except Exception as e:
        out["cpu_baseline_affine"] = {"error": repr(e)}
    return out


if __name__ == "__main__":
    main()
