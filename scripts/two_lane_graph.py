"""The headline step recorded into ONE HIP graph on one stream (the bench's submission) and on TWO streams with one handle each
(consecutive steps on alternating streams: no edge between step k and step k + 1, so the prologue of one runs under the store
tail of the other).   python3 scripts/two_lane_graph.py [steps] [batch] [lanes] [replays]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
LANES = int(sys.argv[3]) if len(sys.argv) > 3 else 2
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
model = pkg.synthetic.make_model(27, 144, 30)
dev = torch.device("cuda:0")
mk = lambda: pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"], model["u_max"],
                               model["x_min"], model["x_max"], 30)
hs = [mk() for _ in range(LANES)]
nz = hs[0].nz
ldz = (nz + 15) // 16 * 16
sets = []
for i in range(4):
    d = pkg.synthetic.make_replay_batch(model, r=i, steps=B)
    big = torch.empty((B, ldz), dtype=torch.float64, device=dev)
    sets.append((torch.from_numpy(d["x0"]).to(dev), torch.from_numpy(d["x0_pre"]).to(dev), torch.from_numpy(d["nu0"]).to(dev),
                 big[:, :nz], torch.empty((B, 144), dtype=torch.float64, device=dev),
                 torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev), big))


def one(h, i):
    x0, x0p, nu0, z, u0, st, itr, _ = sets[i % 4]
    h.solve_device(x0, x0p, None, None, nu0, 1, 1e-2, z_out=z, status=st, iters=itr, u0_out=u0)


def single():
    for i in range(steps):
        one(hs[0], i)


side = [torch.cuda.Stream() for _ in range(LANES - 1)]


def lanes():
    main = torch.cuda.current_stream()
    for s in side:
        s.wait_stream(main)
    for i in range(steps):
        l = i % LANES
        if l == 0:
            one(hs[0], i)
        else:
            with torch.cuda.stream(side[l - 1]):
                one(hs[l], i)
    for s in side:
        main.wait_stream(s)


def time_rec(rec):
    for _ in range(3):
        rec.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); rec.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


single(); torch.cuda.synchronize()
ref = [s[7].clone() for s in sets]
rec1 = pkg.RecordedSolves(single)
m1 = time_rec(rec1)
print("one stream : median %.2f us per step, best %.2f" % m1, flush=True)
if LANES == 1:
    sys.exit(0)
for s in sets:
    s[7].fill_(0.0)
rec2 = pkg.RecordedSolves(lanes)
m2 = time_rec(rec2)
torch.cuda.synchronize()
ok = all(torch.equal(s[7][:, :nz], r[:, :nz]) for s, r in zip(sets, ref))
print("%d streams : median %.2f us per step, best %.2f   outputs bitwise equal: %s" % (LANES, m2[0], m2[1], ok), flush=True)
m1b = time_rec(rec1)
print("one stream again: median %.2f, best %.2f" % m1b)
