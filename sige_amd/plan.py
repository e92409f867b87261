"""Launch plans: a sparse forward that survives a mask change (host side of csrc/plan.hpp / include/sige_hip.h
`sige_hip_plan_*`).

The reference sizes every launch from `activeIndices.size(0)` at call time (sige/cuda/gather_kernel.cu:78-84,111;
sige/utils.py:30; sige/nn/gather.py:101-107) and two of its three applications run ONE sparse forward per mask
(gaugan/runner.py:150-195, diffusion_demo/runner.py:134-164).  On MI355X a sparse forward is ~100 launches of a few
microseconds each: issued from Python they cost 4 ms of host time, replayed from a hipGraph 1.4 ms -- but a graph bakes the
grids and tile counts of ONE mask in.  A `LaunchPlan` records the library calls once and replays them from C:

    plan = LaunchPlan(model)
    out = plan.record(mask, build_masks, forward)     # one-off: warm-up, then the two recordings
    ...
    plan.bind_mask(new_mask)                          # a new edit: mask -> pyramid -> index lists -> tables, refresh of the
    out = plan.run()                                  #   persistent outputs; then the forward -- no Python per launch
    plan.capture(); plan.replay()                     # optional: a hipGraph of the SAME calls for the steady state

`build_masks(mask)` is the user's mask recipe (e.g. `downsample_mask(dilate_mask(mask, 5), 8)`), `forward()` runs the model in
sparse mode on STATIC input tensors and returns its output.  Both must reach the GPU only through libsige_hip.so (bool GPU
masks and the fused channels-last path do): a torch kernel in between is invisible to the plan -- `record` checks the result
of a replay against the eager forward to find out.  What the plan owns: a persistent copy of the mask, persistent index
lists / tile tables / scatter maps (sized for every candidate tile), the memory pool the recorded forward allocated from.
"""
import ctypes
from typing import Callable, Dict, Optional

import torch

from . import hip
from .graphs import GraphPool
from .nn.base import SIGEModel
from .nn.gather import Gather

MASKS, FORWARD = 0, 1


class _Recorder:
    """What sige_amd.hip consults while a plan records on this thread (hip.plan_recorder())."""

    def __init__(self, plan: "LaunchPlan", section: int):
        self.plan = plan
        self.handle = plan.handle
        self.section = section
        self.keep = plan._keep
        self.idx_info = plan._idx_info  # data_ptr of an index list -> (slot, capacity in tiles, base buffer)

    def new_slots(self, n: int) -> int:
        first = hip.lib().sige_hip_plan_new_slots(self.handle, n)
        if first < 0:
            raise RuntimeError("sige_hip_plan_new_slots failed")
        return first

    def bind_index_list(self, buf: torch.Tensor, slot: int, n: int):
        hip._check(hip.lib().sige_hip_plan_bind_ptr(self.handle, buf.data_ptr(), slot), "plan_bind_ptr")
        hip._check(hip.lib().sige_hip_plan_set_slot(self.handle, slot, n), "plan_set_slot")
        self.idx_info[buf.data_ptr()] = (slot, buf.shape[0], buf)

    def bind_constant(self, idx: torch.Tensor):
        """`idx` is an index list whose tile count does not depend on the mask (hip.all_tiles: a dense layer)."""
        if idx.numel():
            hip._check(hip.lib().sige_hip_plan_bind_const(self.handle, idx.data_ptr()), "plan_bind_const")
            self.keep.append(idx)

    def bind_alias(self, table: torch.Tensor, idx: torch.Tensor):
        info = self.idx_info.get(hip.base_ptr(idx))
        if info is not None:
            hip._check(hip.lib().sige_hip_plan_bind_ptr(self.handle, table.data_ptr(), info[0]), "plan_bind_ptr")


class LaunchPlan:
    def __init__(self, model: SIGEModel, device: Optional[torch.device] = None):
        self.model = model
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(self.device):
            self.handle = hip.lib().sige_hip_plan_create()
        if not self.handle:
            raise RuntimeError("sige_hip_plan_create failed")
        self._keep = []
        self._idx_info: Dict[int, tuple] = {}
        self.mask: Optional[torch.Tensor] = None
        self.out: Optional[torch.Tensor] = None
        self.pool: Optional[GraphPool] = None
        self._record_graph = None
        self.graph = None
        self.counts = []
        self._recorded_counts = None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                hip.lib().sige_hip_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ------------------------------------------------------------------ recording --
    class _Recording:
        def __init__(self, plan, section, append=False):
            self.plan, self.section, self.append = plan, section, append

        def __enter__(self):
            if hip.plan_recorder() is not None:
                raise RuntimeError("a launch plan is already recording on this thread")
            hip._check(hip.lib().sige_hip_plan_begin(self.plan.handle, self.section, int(self.append)), "plan_begin")
            hip._plan_tls.rec = _Recorder(self.plan, self.section)
            return self

        def __exit__(self, *exc):
            hip._plan_tls.rec = None
            hip._check(hip.lib().sige_hip_plan_end(self.plan.handle), "plan_end")
            return False

    def _scatter_modules(self):
        return [m for m in self.model.modules() if hasattr(m, "refresh_outputs") and hasattr(m, "plan_tables")]

    def record(self, mask: torch.Tensor, build_masks: Callable[[torch.Tensor], dict], forward: Callable[[], torch.Tensor],
               warmup: int = 2, check: bool = True) -> torch.Tensor:
        """Record both sections under `mask` (bool [H,W] on the plan's device) and return the output of a first replay.
        `warmup` eager forwards come first: they pack weights, register the activated twins and create the persistent
        outputs, so that the recorded call sequence is the steady-state one.  `check`: compare the replayed output with the
        last eager one (raises if they differ by more than fp32 noise: something reached the GPU outside the library)."""
        if mask.dim() != 2 or mask.dtype != torch.bool or not mask.is_cuda:
            raise ValueError("LaunchPlan.record: `mask` must be a 2-D bool tensor on the GPU")
        self.build_masks, self.forward = build_masks, forward
        with torch.cuda.device(self.device):
            self.mask = mask.clone()
            self.model.set_mode("sparse")
            self.model.set_masks(build_masks(self.mask))
            eager = None
            for _ in range(max(1, warmup)):
                eager = forward()
            eager = eager.clone()
            # section 0: mask -> pyramid -> index lists (persistent) -> scatter maps, tile tables, refreshed persistent outputs
            with LaunchPlan._Recording(self, MASKS):
                self.model.set_masks(build_masks(self.mask))
                for m in self._scatter_modules():
                    m.plan_tables()
                    m.refresh_outputs(new_mask=True)
            torch.cuda.synchronize(self.device)
            # section 1: the forward, recorded under a hipGraph capture -- every tensor it allocates comes from the capture's
            # private pool, which stays alive with `_record_graph`: the pointers the plan holds stay valid
            self.pool = GraphPool(self.device)
            with LaunchPlan._Recording(self, FORWARD):
                self._record_graph, self.out = self.pool.capture(forward)
            if hip.lib().sige_hip_plan_calls(self.handle, FORWARD) <= 0:
                raise RuntimeError("LaunchPlan.record: the forward made no library call")
            self._read_counts()
            self._recorded_counts = list(self.counts)
            if not all(self.counts):
                raise RuntimeError("LaunchPlan.record: record under a mask with at least one active tile at every resolution (an empty "
                                   "index list hands the entry points a null pointer, which no later mask can be looked up under)")
            out = self.run()
            if check:
                torch.cuda.synchronize(self.device)
                err = float((out - eager).abs().max())
                ref = float(eager.abs().max())
                if not err <= 1e-4 * (1.0 + ref):
                    raise RuntimeError("LaunchPlan.record: the replayed forward differs from the eager one (max |diff| %.3g): part "
                                       "of the forward does not go through libsige_hip.so and cannot be planned" % err)
        return out

    @property
    def shape_bound(self) -> bool:
        """True if a recorded call takes tile counts the plan cannot follow (NCHW / two-kernel forms): only the recorded mask's
        counts replay."""
        return bool(hip.lib().sige_hip_plan_shape_bound(self.handle))

    @property
    def unbound_counts(self) -> int:
        """How many recorded count arguments hang on a pointer the plan knows nothing about (an index list that did not come from
        the recorded mask pipeline and is not a registered constant).  Any such argument makes the plan shape bound."""
        return int(hip.lib().sige_hip_plan_unbound(self.handle))

    def calls(self, section: int) -> int:
        return int(hip.lib().sige_hip_plan_calls(self.handle, section))

    # -------------------------------------------------------------------- replay --
    def _read_counts(self):
        n = hip.lib().sige_hip_plan_get_slots(self.handle, None, 0)
        arr = (ctypes.c_int32 * max(n, 1))()
        hip.lib().sige_hip_plan_get_slots(self.handle, ctypes.cast(arr, ctypes.c_void_p), n)
        self.counts = [int(arr[i]) for i in range(n)]

    def _run(self, section: int):
        dev = self.device.index
        hip._tls.pending_device = dev  # (device guard of the _Guarded call, as hip._stream() would set it)
        stream = hip._raw_stream(dev) if hip._raw_stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        status = hip.lib().sige_hip_plan_run(self.handle, section, stream)
        if status == hip.UNSUPPORTED:
            raise RuntimeError("LaunchPlan: a recorded call has no kernel for the new mask's tile counts -- record again under this mask")
        hip._check(status, "plan_run")

    def bind_mask(self, mask: torch.Tensor, adopt: bool = True):
        """A new edit.  `mask`: bool [H,W] like the recorded one.  Replays section 0 (one stream synchronisation inside: the
        tile counts come back to the host) -- afterwards `run()` / `capture()` work under the new mask.  `adopt`: also point the
        model's Gather modules at the new index lists, so that module-level (eager) forwards agree with the plan."""
        if self.mask is None:
            raise RuntimeError("LaunchPlan.bind_mask before record")
        if mask.shape != self.mask.shape or mask.dtype != torch.bool:
            raise ValueError("bind_mask: a bool mask shaped like the recorded one")
        bound = self.shape_bound
        prev = self.mask.clone() if bound else None
        self.mask.copy_(mask, non_blocking=True)
        self._run(MASKS)
        self._read_counts()
        if bound and self.counts != self._recorded_counts:
            # a shape-bound plan replays the recorded counts only: put the previous mask's lists, tables and persistent outputs
            # back (the FORWARD section would otherwise run the recorded counts over the new mask's lists) and refuse
            self.mask.copy_(prev, non_blocking=True)
            self._run(MASKS)
            self._read_counts()
            raise RuntimeError("LaunchPlan: this plan recorded calls whose sizes cannot follow a new mask (NCHW or unfused tile "
                               "kernels, or %d count argument(s) on index lists the plan does not own); it only replays masks "
                               "with the recorded tile counts -- the previous mask is still bound" % self.unbound_counts)
        self.graph = None  # (a graph of the previous mask's counts)
        if adopt:
            self._adopt()

    def _adopt(self):
        views = {}
        for ptr, (slot, cap, buf) in self._idx_info.items():
            views[ptr] = buf[:self.counts[slot]]
        for m in self.model.modules():
            if isinstance(m, Gather) and m.active_indices is not None:
                v = views.get(hip.base_ptr(m.active_indices))
                if v is not None:
                    m.active_indices = v

    def run(self) -> torch.Tensor:
        """The sparse forward under the bound mask, issued from C on the current stream.  Returns the output tensor (the same
        tensor every time: copy it before the next run if it must survive)."""
        self._run(FORWARD)
        return self.out

    def capture(self):
        """A hipGraph of section 1 under the bound mask (for many forwards under one mask: replay() then costs one launch)."""
        if self.pool is None:
            raise RuntimeError("LaunchPlan.capture before record")
        self.graph = None
        self.graph, _ = self.pool.capture(lambda: self.run())
        return self.graph

    def replay(self) -> torch.Tensor:
        if self.graph is None:
            self.capture()
        self.graph.replay()
        return self.out
