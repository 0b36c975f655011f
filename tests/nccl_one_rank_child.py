"""Child process of tests/test_gpu_nccl_one_rank.py (never collected by pytest): ONE rank, backend nccl (= RCCL) on device 0.
ShardedFastMPC with always_collective=True runs the device-tensor all_gather_into_tensor branch of sharded.py -- the branch
eight ranks run (BASELINE configs[3]) -- and the results are compared with the plain single-process solve.  Prints one JSON line."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    pkg = importlib.import_module("mpc-sensorlessao_amd")
    from tests.util import handle_from_model
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    model = pkg.synthetic.make_model(27, 144, 30)
    h = handle_from_model(pkg, model)
    for batch, nw in ((512, 1), (75, 3)):                 # 512 = the per-rank shape of configs[3]
        data = pkg.synthetic.make_replay_batch(model, r=11, steps=batch)
        t = {k: (None if v is None else torch.from_numpy(v).to(dev)) for k, v in data.items()}
        z_ref, st, it = h.solve_device(t["x0"], t["x0_pre"], None, None, t["nu0"], nw, 1e-2)
        z_ref = z_ref.clone()
        for collective in (True, False):
            sh = pkg.ShardedFastMPC.from_handle(h, always_collective=collective)
            assert sh.block(batch) == (0, batch)
            calls = {"n": 0}
            orig = dist.all_gather_into_tensor

            def counted(out, inp, *a, **kw):
                calls["n"] += 1
                assert out.is_cuda and inp.is_cuda, "the nccl branch gathers in HBM"
                return orig(out, inp, *a, **kw)
            dist.all_gather_into_tensor = counted
            try:
                u0 = sh.solve_gather_local(batch, t["x0"], t["x0_pre"], None, t["nu0"], nw, 1e-2, what="u0")
                zl, lo, hi = sh.solve_local(t["x0"], t["x0_pre"], None, t["nu0"], nw, 1e-2)
                z = sh.gather(zl, batch, "z")
                U = sh.gather(zl, batch, "U")
            finally:
                dist.all_gather_into_tensor = orig
            torch.cuda.synchronize()
            key = "b%d_nw%d_%s" % (batch, nw, "collective" if collective else "plain")
            res[key] = {"collectives": calls["n"], "u0_bitwise": bool(torch.equal(u0, z_ref[:, :144])), "z_bitwise": bool(torch.equal(z, z_ref)),
                        "U_bitwise": bool(torch.equal(U, z_ref.view(batch, 30, 171)[:, :, :144].reshape(batch, -1))), "lo_hi": [lo, hi]}
    h.close()
    dist.destroy_process_group()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
