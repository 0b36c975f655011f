"""Lanes on disjoint CU sets (hipExtStreamCreateWithCUMask): does the dual solve of one lane overlap the write-out of
the other?  Run on the GPU box:  python scripts/cumask_probe.py"""
import ctypes as C
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mpc-sensorlessao_amd")
hip = C.CDLL("libamdhip64.so")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n, m, T = 27, 144, 30
model = pkg.synthetic.make_model(n, m, T)
data = pkg.synthetic.make_replay_batch(model, r=0, steps=B)
dev = torch.device("cuda", 0)
torch.cuda.init(); torch.zeros(1, device=dev)


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def mk(stream):
    h = pkg.FastMPCHandle(model["A1"], model["A2"], model["B"], model["Q"], model["R"], model["Qf"], model["u_min"],
                          model["u_max"], model["x_min"], model["x_max"], T, device=0)
    return dict(h=h, x0=torch.from_numpy(data["x0"]).to(dev), x0p=torch.from_numpy(data["x0_pre"]).to(dev),
                nu0=torch.from_numpy(data["nu0"]).to(dev), z=torch.empty((B, h.nz), dtype=torch.float64, device=dev),
                st=torch.empty(B, dtype=torch.int32, device=dev), it=torch.empty(B, dtype=torch.int32, device=dev),
                u0=torch.empty((B, m), dtype=torch.float64, device=dev), s=stream)


configs = {
    "2 lanes, unmasked": [torch.cuda.Stream(dev), torch.cuda.Stream(dev)],
    "2 lanes, CUs [0,128) / [128,256)": [masked_stream(range(0, 128)), masked_stream(range(128, 256))],
    "2 lanes, even / odd CU bits": [masked_stream(range(0, 256, 2)), masked_stream(range(1, 256, 2))],
    "1 lane, CUs [0,128)": [masked_stream(range(0, 128))],
    "1 lane, unmasked": [torch.cuda.Stream(dev)],
}
for name, streams in configs.items():
    slots = [mk(s) for s in streams]
    nfl = len(slots)
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
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {dt / K * 1e6:.1f} us per step, {B * K / dt / 1e6:.2f} M steps/s", flush=True)
    for c in slots:
        c["h"].close()
