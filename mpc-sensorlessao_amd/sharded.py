"""Multi-GPU driver: independent problems are sharded across ranks (one process per GPU), each
rank solves its contiguous block locally with no data-path collective, and ONE gather at the end
collects the outputs (RCCL over xGMI when the process group is `nccl`).  SURVEY.md §8(e).

The reference has no distributed code at all; this is the batch partitioning the north-star asks
for, placed above the drop-in boundary (`mpc_fixed_log_newton`)."""
from __future__ import annotations


def shard_range(batch, world_size, rank):
    """Contiguous block of ceil(batch/G) problems per rank (the last blocks may be short/empty)."""
    per = -(-batch // world_size)
    lo = min(rank * per, batch)
    return lo, min(lo + per, batch)


class ShardedFastMPC:
    """solve_fn(x0, x0_pre, w, nu0, n_newton, k) -> z (local_batch, N_z) tensor on the rank's
    device.  On a GPU rank this is `FastMPCHandle.solve_device`; CPU gloo tests inject a checker.
    """

    def __init__(self, solve_fn, nz, m, T, n, group=None, solve_u0_fn=None, always_collective=False):
        # always_collective: issue the all-gather also in a group of ONE rank (a 1-rank RCCL group on a one-GPU box runs the very
        # branch 8 ranks run: tests/test_gpu_nccl_one_rank.py); by default a single rank returns its rows without a collective
        self.always_collective = bool(always_collective)
        self.solve_fn = solve_fn
        self.solve_u0_fn = solve_u0_fn     # optional: (x0, x0_pre, w, nu0, n_newton, k) -> first moves (local_batch, m) only
        self.nz, self.m, self.T, self.n = nz, m, T, n
        self.group = group

    @classmethod
    def from_handle(cls, handle, group=None, always_collective=False):
        """The production wiring: every rank solves its block with its own `FastMPCHandle.solve_device` (HIP kernels on
        the rank's GPU, asynchronous on torch's current stream)."""
        def solve_fn(x0, x0_pre, w, nu0, n_newton, k):
            return handle.solve_device(x0, x0_pre, w, None, nu0, n_newton, k)[0]

        def solve_u0_fn(x0, x0_pre, w, nu0, n_newton, k):          # z_out = NULL at the C ABI: nothing but u0 leaves the solve
            u0 = x0.new_empty((x0.shape[0], handle.m))
            handle.solve_device(x0, x0_pre, w, None, nu0, n_newton, k, u0_out=u0, want_z=False)
            return u0
        return cls(solve_fn, handle.nz, handle.m, handle.T, handle.n, group, solve_u0_fn, always_collective)

    def _dist(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist, dist.get_world_size(self.group), dist.get_rank(self.group)
        return None, 1, 0

    def block(self, batch):
        """(lo, hi) of this rank's contiguous block of a global batch."""
        _, ws, rank = self._dist()
        return shard_range(batch, ws, rank)

    def solve_local(self, x0, x0_pre, w, nu0, n_newton, k):
        """Solve this rank's shard of a replicated global batch.  Returns (z_local, lo, hi)."""
        _, ws, rank = self._dist()
        lo, hi = shard_range(x0.shape[0], ws, rank)
        sl = lambda t: None if t is None else t[lo:hi].contiguous()
        if hi == lo:
            return x0.new_zeros((0, self.nz)), lo, hi
        return self.solve_fn(sl(x0), sl(x0_pre), sl(w), sl(nu0), n_newton, k), lo, hi

    def gather(self, local, batch, what="z"):
        """One all-gather of the per-rank outputs (rows of `local`) into the global batch order.
        what: "z" (N_z), "u0" (first move, README.md:589), "U" (T*m)."""
        import torch
        dist, ws, rank = self._dist()
        if what == "rows":
            pass                                     # `local` already holds what is to be gathered (e.g. first moves)
        elif what == "u0":
            local = local[:, :self.m].contiguous()
        elif what == "U":
            local = local.reshape(local.shape[0], self.T, self.n + self.m)[:, :, :self.m] \
                         .reshape(local.shape[0], self.T * self.m).contiguous()
        if ws == 1 and not (self.always_collective and dist is not None):
            return local
        per = -(-batch // ws)
        cols = local.shape[1]
        pad = local.new_zeros((per, cols))
        pad[:local.shape[0]] = local
        if local.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal on one GPU / CPU clusters: gloo gathers host tensors (RCCL, the production backend, gathers in HBM)
            host = pad.cpu()
            out = host.new_empty((ws * per, cols))
            dist.all_gather_into_tensor(out, host, group=self.group)
            return out[:batch].to(local.device)
        out = local.new_empty((ws * per, cols))
        dist.all_gather_into_tensor(out, pad, group=self.group)
        return out[:batch]

    def solve_gather(self, x0, x0_pre, w, nu0, n_newton, k, what="u0"):
        z, lo, hi = self.solve_local(x0, x0_pre, w, nu0, n_newton, k)
        return self.gather(z, x0.shape[0], what)

    def solve_gather_local(self, batch, x0, x0_pre, w, nu0, n_newton, k, what="u0"):
        """Like solve_gather, but the inputs are THIS RANK'S BLOCK only (rows lo..hi of a global batch of `batch` problems,
        see `block`): nothing is replicated.  An empty block (more ranks than problems) takes part in the gather with no rows."""
        lo, hi = self.block(batch)
        assert x0.shape[0] == hi - lo, "solve_gather_local: pass exactly this rank's block"
        if what == "u0" and self.solve_u0_fn is not None:
            local = self.solve_u0_fn(x0, x0_pre, w, nu0, n_newton, k) if hi > lo else x0.new_zeros((0, self.m))
            return self.gather(local, batch, "rows")
        z = self.solve_fn(x0, x0_pre, w, nu0, n_newton, k) if hi > lo else x0.new_zeros((0, self.nz))
        return self.gather(z, batch, what)
