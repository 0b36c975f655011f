"""Where the time of a 20-step timed region of the headline goes (the driver runs bench.py --steps 20): device time per step between
events recorded behind chosen steps -- the first step after a synchronisation carries the launch latency from idle (~20 us), and the
steps speed up over the region (clocks after the idle gap): 36-37 us per step against 34 in a 400-step region.
    python3 scripts/region_trace.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("mpc-sensorlessao_amd")
B = 2000
model = pkg.synthetic.make_model(27, 144, 30)
h = pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"], model["u_max"], model["x_min"], model["x_max"], 30)
dev = torch.device("cuda:0")
ldz = (h.nz + 15) // 16 * 16
sets = []
for i in range(4):
    d = pkg.synthetic.make_replay_batch(model, r=i, steps=B)
    big = torch.zeros((B, ldz), dtype=torch.float64, device=dev)
    sets.append((torch.from_numpy(d["x0"]).to(dev), torch.from_numpy(d["x0_pre"]).to(dev), torch.from_numpy(d["nu0"]).to(dev),
                 big[:, :h.nz], torch.empty((B, 144), dtype=torch.float64, device=dev),
                 torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)))
def step(it):
    x0, x0p, nu0, z, u0, st, itr = sets[it % 4]
    h.solve_device(x0, x0p, None, None, nu0, 1, 1e-2, z_out=z, status=st, iters=itr, u0_out=u0)
for it in range(60): step(it)
torch.cuda.synchronize()
K = 20
for marks in ([1, 3, 6, 12, 20], [2, 5, 10, 20], [20]):
    res = []
    for rep in range(15):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(marks) + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record(); j = 1
        for it in range(K):
            step(it)
            if it + 1 == marks[j - 1]:
                ev[j].record(); j += 1
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e6
        res.append([wall] + [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(len(marks))])
    med = np.median(np.array(res), axis=0)
    prev = 0; parts = []
    for m_, t_ in zip(marks, med[1:]):
        parts.append("steps %d-%d: %.1f us/step" % (prev + 1, m_, t_ / (m_ - prev))); prev = m_
    print("wall %.0f us | %s" % (med[0], "; ".join(parts)))
