"""Throughput with several independent solves in flight (one handle + one stream each) vs. one stream.
Run on the GPU box:  python scripts/inflight_probe.py [batch]"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mpc-sensorlessao_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n, m, T = 27, 144, 30
model = pkg.synthetic.make_model(n, m, T)
data = pkg.synthetic.make_replay_batch(model, r=0, steps=B)
dev = torch.device("cuda", 0)


def mk():
    h = pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"],
                          model["u_max"], model["x_min"], model["x_max"], T, device=0)
    return dict(h=h, x0=torch.from_numpy(data["x0"]).to(dev), x0p=torch.from_numpy(data["x0_pre"]).to(dev),
                nu0=torch.from_numpy(data["nu0"]).to(dev), z=torch.empty((B, h.nz), dtype=torch.float64, device=dev),
                st=torch.empty(B, dtype=torch.int32, device=dev), it=torch.empty(B, dtype=torch.int32, device=dev),
                u0=torch.empty((B, m), dtype=torch.float64, device=dev), s=torch.cuda.Stream(dev))


for nfl in (1, 2, 3, 4):
    slots = [mk() for _ in range(nfl)]
    def step(i):
        c = slots[i % nfl]
        with torch.cuda.stream(c["s"]):
            c["h"].solve_device(c["x0"], c["x0p"], None, None, c["nu0"], 1, 0.01, z_out=c["z"], status=c["st"], iters=c["it"])
            c["h"].unpack_device(c["z"], None, None, c["u0"])
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    K = 400
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"in flight {nfl}: {dt / K * 1e6:.1f} us per step, {B * K / dt / 1e6:.2f} M steps/s (host enqueue {th / K * 1e6:.1f} us per step)", flush=True)
    if nfl == 2:
        # one hipGraph holding one step of each lane (fork/join from a capture stream)
        g = torch.cuda.CUDAGraph()
        cs = torch.cuda.Stream(dev)
        with torch.cuda.graph(g, stream=cs):
            for c in slots:
                c["s"].wait_stream(cs)
            for i in range(nfl):
                step(i)
            for c in slots:
                cs.wait_stream(c["s"])
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K // nfl):
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"  graph of {nfl} lanes: {dt / K * 1e6:.1f} us per step, {B * K / dt / 1e6:.2f} M steps/s", flush=True)
    for c in slots:
        c["h"].close()
