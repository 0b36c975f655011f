"""A stretch of device solves recorded once into a HIP graph and replayed with one host call (include/fastmpc.h, "Recording
solves into a HIP graph").  The reference replays a realisation step by step through Fast_MPC2(...).mpc_fixed_log_newton
(README.md:548-556); when the inputs of the steps are known in advance -- a replay batch -- the host does not have to submit
them one by one: inside a graph the launches follow each other more closely than the host can submit them."""
from . import _lib
from ._lib import FastMPCError


class RecordedSolves:
    """record_fn(): any number of solve_device / solve-like calls on torch tensors that all exist already (status and iters
    included: nothing may be allocated while the calls are recorded).  It is run once eagerly (the handle builds its constants
    and workspaces), then once under capture.  replay() runs the recorded launches on torch's current stream; it refuses when
    any handle of the process has allocated or released device memory since the recording (the graph holds addresses)."""

    def __init__(self, record_fn):
        import gc
        import torch
        self._lib = _lib.load()
        self._record_fn = record_fn           # (keeps the tensors the closure refers to alive: the graph holds their addresses)
        record_fn()
        torch.cuda.synchronize()
        self._stream = torch.cuda.Stream()
        self.graph = torch.cuda.CUDAGraph()
        # No garbage collection while the stream is capturing: a collection that happens to run inside the capture may destroy
        # objects whose release is not allowed then -- another graph with its memory pool (hipFree), a stream -- and the
        # runtime aborts the process (seen in the test suite: the previous test's recording was collected here).
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.stream(self._stream):
                with torch.cuda.graph(self.graph, stream=self._stream):
                    record_fn()
        finally:
            if gc_was_on:
                gc.enable()
        torch.cuda.synchronize()
        self._gen = int(self._lib.fmpc_alloc_generation())

    def valid(self):
        return int(self._lib.fmpc_alloc_generation()) == self._gen

    def replay(self):
        if not self.valid():
            raise FastMPCError(_lib.FMPC_E_UNSUPPORTED, "RecordedSolves.replay: device buffers were (re)allocated since the recording -- record again")
        self.graph.replay()
