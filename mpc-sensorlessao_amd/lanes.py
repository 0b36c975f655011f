"""Several independent solves in flight on one GPU.

The dual-solve kernel of the cold-start path (fmpc_cold_panel) is latency-bound and occupies one CU per
16-problem panel: a replay batch of 2000 timesteps fills 125 of the 256 CUs for most of a solve.  Independent
batches (other realisations, other horizon windows: README.md:444-622 runs one realisation; SURVEY.md §8(e) batches
them) therefore overlap well: each LANE owns a `FastMPCHandle` (its own device workspaces), a HIP stream and its
output buffers, and `SolveLanes.submit` deals consecutive batches round-robin to the lanes.  Two lanes lift the
2000-problem step from 27 to 40 M steps/s on one MI355X; more lanes add nothing.

Nothing here synchronises with the host: `submit` only enqueues, `wait` makes torch's current stream wait for a
lane (or all), `synchronize` blocks the host.
"""
from __future__ import annotations


class Lane:
    def __init__(self, handle, batch, device, u0_slots=1):
        import torch
        f64 = dict(dtype=torch.float64, device=device)
        self.handle = handle
        self.stream = torch.cuda.Stream(device)
        self.z = torch.empty((batch, handle.nz), **f64)
        self.status = torch.zeros(batch, dtype=torch.int32, device=device)
        self.iters = torch.zeros(batch, dtype=torch.int32, device=device)
        # first moves: a ring of `u0_slots` buffers, one per submit in turn (so that several steps' first moves can be
        # collected, e.g. gathered across GPUs, in one go); `u0` is the slot of the last submit
        self.u0_ring = torch.empty((u0_slots, batch, handle.m), **f64)
        self.slot = u0_slots - 1
        self.u0 = self.u0_ring[self.slot]
        self.done = torch.cuda.Event()
        self._bound_key = None                      # argument set of the last submit and its prebuilt C calls
        self._bound = None


class SolveLanes:
    def __init__(self, make_handle, batch, depth=2, device=None, u0_slots=1):
        """make_handle: () -> FastMPCHandle (called `depth` times: every lane needs its own workspaces)."""
        import torch
        if depth < 1:
            raise ValueError("depth >= 1")
        handles = [make_handle() for _ in range(depth)]
        dev = torch.device("cuda", handles[0].device) if device is None else device
        self.device = dev
        self.batch = int(batch)
        self.lanes = [Lane(h, self.batch, dev, u0_slots) for h in handles]
        self.submitted = 0

    @property
    def depth(self):
        return len(self.lanes)

    def next_lane(self):
        """The lane the next `submit` will use."""
        return self.lanes[self.submitted % len(self.lanes)]

    def submit(self, x0, x0_pre=None, w=None, z_init=None, nu0=None, n_newton=1, k=1e-2, after_current=True, lane=None):
        """Enqueue one batch on the next lane: solve + first moves (fmpc_solve_u0_device) into the lane's buffers.  Returns the lane.
        The inputs must stay untouched until the lane is waited for.  after_current: the lane's stream first waits
        for what torch's current stream has enqueued so far (the producer of the inputs); pass False when the inputs
        are already complete, so that lanes never serialise through the caller's stream.  lane: use this lane
        instead of the next one in turn."""
        import torch
        if lane is None:
            lane = self.lanes[self.submitted % len(self.lanes)]
            self.submitted += 1
        if after_current:
            lane.stream.wait_stream(torch.cuda.current_stream(self.device))
        # Steady state (the same device buffers as in this lane's last submit): the C call is replayed from
        # prebuilt ctypes arguments on the lane's stream -- no tensor checks, no stream context (host time per
        # submit 41 -> ~15 us, which matters once a step takes 50 us on the device).
        key = (x0.data_ptr(), 0 if x0_pre is None else x0_pre.data_ptr(), 0 if w is None else w.data_ptr(),
               0 if z_init is None else z_init.data_ptr(), 0 if nu0 is None else nu0.data_ptr(), int(n_newton), float(k),
               x0.shape[0])
        lane.slot = (lane.slot + 1) % lane.u0_ring.shape[0]
        lane.u0 = lane.u0_ring[lane.slot]
        if key != lane._bound_key:
            with torch.cuda.stream(lane.stream):    # first time: the checked path
                lane.handle.solve_device(x0, x0_pre, w, z_init, nu0, n_newton, k, z_out=lane.z, status=lane.status,
                                         iters=lane.iters, u0_out=lane.u0)
            import ctypes as C
            h = lane.handle
            vp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
            st = C.c_void_p(lane.stream.cuda_stream)
            args = [(h._h, int(x0.shape[0]), vp(x0), vp(x0_pre), vp(w), vp(z_init), vp(nu0), int(n_newton), float(k),
                     vp(lane.z), None, vp(lane.status), vp(lane.iters), None, vp(lane.u0_ring[q]), st)
                    for q in range(lane.u0_ring.shape[0])]           # one per ring slot
            lane._bound = (h._lib.fmpc_solve_u0_device, args, (x0, x0_pre, w, z_init, nu0))   # (keeps the inputs alive)
            lane._bound_key = key
        else:
            f1, a1, _ = lane._bound
            rc = f1(*a1[lane.slot])
            if rc != 0:
                from ._lib import FastMPCError
                raise FastMPCError(rc, "SolveLanes.submit")
        lane.done.record(lane.stream)
        return lane

    def wait(self, lane=None):
        """torch's current stream waits for the lane's last submit (all lanes if None)."""
        import torch
        cur = torch.cuda.current_stream(self.device)
        for ln in (self.lanes if lane is None else [lane]):
            cur.wait_event(ln.done)

    def synchronize(self):
        for ln in self.lanes:
            ln.stream.synchronize()

    def close(self):
        for ln in self.lanes:
            ln.handle.close()
        self.lanes = []
