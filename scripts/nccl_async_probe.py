"""Probe (GPU box): the async all-gather pattern of bench.py with a 1-rank RCCL group."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
u = [torch.randn(2000, 144, dtype=torch.float64, device=dev) for _ in range(2)]
out = [torch.empty(2000, 144, dtype=torch.float64, device=dev) for _ in range(2)]
pend = [None, None]
for i in range(6):
    b = i & 1
    if pend[b] is not None:
        pend[b].wait(); pend[b] = None
    u[b].mul_(1.0001)
    pend[b] = dist.all_gather_into_tensor(out[b], u[b], async_op=True)
for b in range(2):
    if pend[b] is not None:
        pend[b].wait()
torch.cuda.synchronize()
print("ok", bool(torch.equal(out[0], u[0])), bool(torch.equal(out[1], u[1])))
dist.destroy_process_group()
