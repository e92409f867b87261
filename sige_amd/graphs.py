"""hipGraph capture for a host that re-captures often (a new edit = a new mask = new grids: one graph per mask).

`torch.cuda.graph(...)` gives every capture a NEW private memory pool and runs `gc.collect()` + `empty_cache()` on entry: every
tensor the forward allocates is hipMalloc'ed again (~100 allocations), and the blocks of the graph just destroyed are handed
back to the driver first.  Measured on MI355X for the DDPM-256 sparse forward (tools/probe/recapture_probe.py): 7.1 ms per
capture; with ONE capture stream and ONE pool for every capture 4.7 ms, and capturing straight away -- the capture IS the
first forward under the new mask -- 1.6 (set_masks) + 5.5 + 2.2 (first replay) = 9.4 ms from a new mask to the first output WITH
the steady-state graph in hand (bench.py: dynamic.mask_change_capture_first), against 7.5 ms to an eager first output and 17.6 ms
to the graph.  (Steady state of a session: an edit larger than any before grows the pool once, by hipMalloc inside the capture.)

Not part of the reference (it has no graphs); pure host code on top of torch's CUDAGraph."""
from typing import Callable, Optional, Tuple

import torch


class GraphPool:
    """One capture stream and one graph memory pool for every capture on a device.

    A pool lives as long as a graph uses it: a one-kernel graph captured at construction and kept for the life of the object
    keeps the pool -- and the blocks of destroyed graphs cached inside it -- alive.  Destroy the previous graph of a forward
    BEFORE capturing its successor (its blocks are what the new capture reuses); graphs of one pool must not be replayed
    concurrently (they may share memory)."""

    def __init__(self, device: Optional[torch.device] = None, capture_error_mode: str = "thread_local"):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.mode = capture_error_mode
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.Stream()
            self.pool = torch.cuda.graph_pool_handle()
            self._keeper = torch.cuda.CUDAGraph()
            self._dummy = torch.zeros(8, device=self.device)
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self._keeper.capture_begin(pool=self.pool, capture_error_mode=self.mode)
                try:
                    self._dummy.add_(1.0)
                finally:
                    self._keeper.capture_end()
            torch.cuda.current_stream().wait_stream(self.stream)

    def capture(self, fn: Callable[[], torch.Tensor]) -> Tuple[torch.cuda.CUDAGraph, torch.Tensor]:
        """Capture one call of `fn` (no warm-up call: whatever `fn` sets up on its first run -- tile tables, packed weights
        -- has to be capturable, which the sparse forward's set-up is).  Returns (graph, fn's output = the graph's output
        buffer); nothing has run yet: `graph.replay()` produces the first values."""
        with torch.cuda.device(self.device):
            g = torch.cuda.CUDAGraph()
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                g.capture_begin(pool=self.pool, capture_error_mode=self.mode)
                try:
                    out = fn()
                finally:
                    g.capture_end()
            torch.cuda.current_stream().wait_stream(self.stream)
        return g, out
