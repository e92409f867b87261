"""Scatter / ScatterWithBlockResidual: conv-output tiles back into the cached
full activation.

API parity with sige/nn/scatter.py:9-136 (same attributes: `gather`,
`original_outputs`, `original_residuals`, `output_res`; same full / sparse /
profile behaviour; `sparse_update` refreshes the cache in place).

On the GPU the sparse mode uses the fused single-pass kernels of
libsige_hip.so (tile table + one streaming pass) -- results are bit-identical
to the reference's clone-then-overwrite.
"""
from typing import Optional

import torch

from . import deferred
from .base import SIGEModule, SIGEModuleWrapper
from .gather import Gather


def _fused_ok(x: torch.Tensor) -> bool:
    return x.is_cuda


def hip_is_cl(t: torch.Tensor) -> bool:
    from .. import hip

    return hip.is_cl(t)


def _cl_ok(cached: torch.Tensor, *others) -> bool:
    """Channels-last kernels: a channels-last cache with C % 4 == 0 and full-size operands."""
    if not cached.is_cuda or cached.dtype not in (torch.float32, torch.float16):  # (fp16: SIGEModel.set_cache_dtype)
        return False
    from .. import hip

    if not (hip.is_cl(cached) and cached.shape[1] % 4 == 0):
        return False
    return all(o is None or tuple(o.shape) == tuple(cached.shape) for o in others)


def _fill(buf: torch.Tensor, cached: torch.Tensor, build=None) -> torch.Tensor:
    """buf <- cached (build None) or SiLU(scale * cached + shift) (build = (scale [C], shift [C])).  On the GPU, channels-last:
    ONE library launch (sige_hip_copy_f32 / sige_hip_affine_act_nhwc_f32), which a launch plan records -- a torch copy would
    be invisible to it."""
    if cached.is_cuda and cached.dtype in (torch.float32, torch.float16) and buf.stride() == cached.stride():
        from .. import hip

        if (build is None and (cached.is_contiguous() or cached.is_contiguous(memory_format=torch.channels_last))
                and (cached.dtype == torch.float32 or cached.numel() % 4 == 0)):
            return hip.copy_dense_(buf, cached)  # (an fp16-stored cache is widened on the way)
        if build is not None and hip.is_cl(cached):
            sc, sh = build
            done = hip.affine_act_cl(cached, sc.reshape(1, -1, 1, 1), sh.reshape(1, -1, 1, 1), "swish", out=buf)
            if done is not None:
                return buf
    if build is None:
        buf.copy_(cached)
    else:
        sc, sh = build
        buf.copy_(torch.nn.functional.silu(cached.float() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)))
    return buf


class _OutputBuffers:
    """Persistent outputs of a Scatter module for the in-place mode (SIGEModel.set_scatter_inplace).

    The reference returns `y.clone()` with the tiles written over it (sige/cpu/scatter.cpp:83):
    two full-tensor passes per call although only the tiles change.  With 288 GB of HBM the
    module keeps, per cache id, one buffer that already equals the cached tensor outside the
    current mask's tiles; a forward then writes the covered pixels only.  The buffer is rebuilt
    (one copy) whenever the cache or the mask changes.  The caller must not modify the returned
    tensor in place, and it is only valid until the module's next forward.

    "The cache changed" is an explicit per-cache-id generation counter, bumped by whoever writes the
    cache (a full-mode forward, `sparse_update`, `parallel.pack_caches` / `broadcast_cache`) --
    not the tensor's address or version: the caching allocator recycles addresses, a fresh tensor
    starts at version 0 and a collective may write a buffer without touching its version."""

    def __init__(self):
        self.bufs = {}
        self.gen = {}

    def get(self, cache_id, cached: torch.Tensor, stamp, build=None):
        """`build`: what a fresh buffer holds -- None: a copy of the cache; (scale [C], shift [C]): SiLU(scale * cache + shift),
        an activated twin."""
        entry = self.bufs.get(cache_id)
        key = (stamp, self.gen.get(cache_id, 0), tuple(cached.shape), cached.stride())
        if entry is None or entry[0] != key:
            if entry is not None and entry[0][1:] == key[1:] and entry[1].device == cached.device:
                # only the MASK changed: restore the buffer in place -- same address, so a launch plan or a captured hipGraph
                # that points at it stays valid, and nothing is allocated.  CONSEQUENCE: a hipGraph captured under an
                # EARLIER mask shares this buffer and does not contain the restore -- after set_masks() graphs of earlier masks
                # are invalid (alternating replays of two masks' graphs would leave the other mask's tiles in the buffer):
                # re-capture per mask (GraphPool / LaunchPlan.capture do), and clone() a model output that must outlive set_masks
                buf = _fill(entry[1], cached, build)
            else:
                # (persistent outputs and twins are fp32 whatever the cache is stored as: they hold this edit's RESULTS)
                buf = _fill(torch.empty_like(cached, dtype=torch.float32, memory_format=torch.preserve_format), cached, build)
            entry = (key, buf, build)
            self.bufs[cache_id] = entry
        return entry[1]

    def invalidate(self, cache_id=None):
        """The cached tensor of `cache_id` (None: of every id) was replaced or rewritten."""
        for k in ([cache_id] if cache_id is not None else set(self.bufs) | set(self.gen)):
            self.gen[k] = self.gen.get(k, 0) + 1
            self.bufs.pop(k, None)

    def refresh(self, caches: dict, stamp=None):
        """The cached tensors were rewritten IN PLACE (same tensors, new values -- e.g. a collective into the packed
        cache), or (`stamp`) the mask changed: rebuild the existing buffers from them, addresses kept -- a captured hipGraph
        or a launch plan may hold them.  `stamp`: the buffers then count as built for that mask."""
        for cid, (key, buf, build) in list(self.bufs.items()):
            cached = caches.get(cid)
            if cached is not None and tuple(cached.shape) == tuple(buf.shape) and cached.stride() == buf.stride():
                _fill(buf, cached, build)
                if stamp is not None:
                    self.bufs[cid] = ((stamp,) + tuple(key[1:]), buf, build)
            else:
                self.bufs.pop(cid)

    def clear(self):
        self.bufs = {}


def twin_key_cache_id(key):
    """Consumers key their registration (consumer id, input part, affine generation, cache id): the registered (scale,
    shift) is the consumer's cached affine of THAT cache id (one cache per denoising step, sige/nn/scatter.py:59-60).  Keys of
    any other shape apply to every cache id."""
    return key[3] if isinstance(key, tuple) and len(key) >= 4 else None


def twins_for(regs: dict, cache_id) -> dict:
    """The registrations a launch under `cache_id` serves."""
    return {k: v for k, v in regs.items() if twin_key_cache_id(k) in (None, cache_id)}


class _TwinBuffers:
    """Activated twins of a Scatter module's persistent output (not in the reference).

    A CONSUMER whose conv1 applies a cached GroupNorm affine + SiLU to this module's output can register (scale, shift)
    under a key; the fused conv -> scatter launch then also writes SiLU(scale * value + shift) for every pixel it writes
    into a second persistent buffer that equals SiLU(scale * cache + shift) elsewhere -- the consumer stages raw values
    (the activation is computed once per element by the producer instead of once per output-channel block).  At most two
    registrations (a skip tensor has two consumers); further ones are refused and those consumers keep activating."""

    MAX = 2

    def __init__(self):
        self.regs = {}   # key -> (scale [C], shift [C])
        self.bufs = {}   # key -> _OutputBuffers

    def register(self, key, scale: torch.Tensor, shift: torch.Tensor) -> bool:
        if key not in self.regs and len(twins_for(self.regs, twin_key_cache_id(key))) >= self.MAX:
            return False
        self.regs[key] = (scale, shift)
        self.bufs.pop(key, None)
        return True

    def unregister(self, key):
        self.regs.pop(key, None)
        self.bufs.pop(key, None)

    def launch_args(self, cache_id, cached: torch.Tensor, stamp):
        """[(key, buffer, scale, shift)] for the fused launch; buffers are (re)built from the cache when stale."""
        out = []
        for key, (sc, sh) in twins_for(self.regs, cache_id).items():
            buf = self.bufs.setdefault(key, _OutputBuffers()).get(cache_id, cached, stamp, build=(sc, sh))
            out.append((key, buf, sc, sh))
        return out

    def invalidate(self, cache_id=None):
        for b in self.bufs.values():
            b.invalidate(cache_id)

    def refresh(self, caches: dict, stamp=None):
        for b in self.bufs.values():
            b.refresh(caches, stamp)

    def clear(self):
        self.bufs = {}


# Tests only (tests/test_host_logic.py): where a launch cannot write twins itself (no GPU: the CPU path goes through the
# plain Scatter kernels), compute them from the finished output with torch -- the registration / invalidation logic of the
# twins can then be exercised end to end on the oracle backend.  Never set in the product.
EMULATE_TWINS = False


def emulated_twins(out: torch.Tensor, regs: dict) -> dict:
    return {k: torch.nn.functional.silu(out * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) for k, (sc, sh) in list(regs.items())[:2]}


def tag_twins(out: torch.Tensor, twins: dict, producer=None) -> torch.Tensor:
    """Attach {consumer key: activated twin} (possibly empty: persistent outputs keep their Python object, a stale
    entry must not survive a launch that did not write it) and the producing module to an output tensor."""
    out._sige_twins = twins
    if producer is not None:
        out._sige_producer = producer
    return out


class Scatter(SIGEModule):
    def __init__(self, gather: Gather):
        super(Scatter, self).__init__()
        self.gather = SIGEModuleWrapper(gather)

        self.load_runtime("scatter")
        self.output_res = None
        self.original_outputs = {}
        self.inplace = False
        self._out_bufs = _OutputBuffers()
        self.twins = _TwinBuffers()

    def clear_cache(self):
        self.original_outputs = {}
        self._out_bufs.clear()
        self.twins.clear()

    def refresh_outputs(self, new_mask: bool = False):
        """(sige_amd.parallel) the cache tensors were rewritten in place -- or (`new_mask`, sige_amd.plan) the mask changed:
        persistent outputs follow, addresses kept."""
        stamp = self.gather.module.timestamp if new_mask else None
        self._out_bufs.refresh(self.original_outputs, stamp)
        self.twins.refresh(self.original_outputs, stamp)

    def plan_tables(self):
        """(sige_amd.plan) build, now, the per-mask lookup tables this module's sparse forward would build lazily."""
        g: Gather = self.gather.module
        for cached in self.original_outputs.values():
            if cached.is_cuda and g.active_indices is not None:
                g.tile_table(cached.shape[2:], cached.device)

    def forward_fused(self, conv, tiles: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(not in the reference) `self(conv(tiles), residual)` in ONE launch when the in-place mode is on and `tiles`
        are pending (deferred) channels-last tiles: the conv's epilogue writes into this module's persistent output.
        Falls back to the two-module form otherwise; values are identical."""
        if self.mode == "sparse" and self.inplace and not self.sparse_update:
            g: Gather = self.gather.module
            cached = self.original_outputs[self.cache_id]
            if _cl_ok(cached, residual) and isinstance(tiles, deferred.DeferredTiles) and tiles.spec is not None:
                out = self._out_bufs.get(self.cache_id, cached, g.timestamp)
                tw = self.twins.launch_args(self.cache_id, cached, g.timestamp)
                done = _fused_conv_into(conv, tiles, out, g, residual=residual, twins=[(b, sc, sh) for _, b, sc, sh in tw])
                if done is not None:
                    return tag_twins(done, {k: b for k, b, _, _ in tw})
        return self.forward(conv(tiles), residual)

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        self.check_dtype(x, residual)
        self.check_dim(x, residual)
        if self.mode == "sparse":
            x = deferred.resolve(x)
            g: Gather = self.gather.module
            cached = self.original_outputs[self.cache_id]
            if _cl_ok(cached, residual):
                from .. import hip

                out = self._out_bufs.get(self.cache_id, cached, g.timestamp) if (self.inplace and not self.sparse_update) else None
                output = hip.scatter_cl(x, cached, g.offset, g.model_stride, g.indices_on(x.device),
                                        g.tile_table(cached.shape[2:], x.device), residual, out=out)
            elif _fused_ok(x):
                from .. import hip

                output = hip.scatter_fused(
                    x.contiguous(), deferred.from_cache(cached), g.tile_table(cached.shape[2:], x.device), g.active_indices.size(0),
                    None if residual is None else residual.contiguous())
            else:
                fn = self.native(self.runtime, x)
                output = fn(x.contiguous(), deferred.from_cache(cached).contiguous(), g.offset[0], g.offset[1],
                            g.model_stride[0], g.model_stride[1], g.indices_on(x.device),
                            None if residual is None else residual.contiguous())
            if self.sparse_update:
                cached.copy_(output)
                self._out_bufs.invalidate(self.cache_id)
                self.twins.invalidate(self.cache_id)
            # (this launch wrote no twin: consumers activate for themselves)
            return tag_twins(output, emulated_twins(output, twins_for(self.twins.regs, self.cache_id)) if EMULATE_TWINS else {})
        if self.mode == "full":
            output = x if residual is None else x + residual
            self.output_res = output.shape[2:]
            self.original_outputs[self.cache_id] = deferred.to_cache(output, self.cache_dtype)
            self._out_bufs.invalidate(self.cache_id)
            self.twins.invalidate(self.cache_id)
            return output
        if self.mode == "profile":
            c = x.shape[1]
            output = torch.full((self.original_outputs[self.cache_id].size(0), c, *self.output_res),
                                fill_value=x[0, 0, 0, 0], dtype=x.dtype, device=x.device)
            if residual is not None:
                output = output + residual
            return output
        raise NotImplementedError("Unknown mode: [%s]!!!" % self.mode)


def _fused_conv_into(conv, tiles, out, g: Gather, residual=None, x1=None, table1=None, twins=None):
    """Run `conv` on the pending (deferred) tiles with its output written straight into `out` at the tile positions of
    gather `g`.  Returns `out`, or None when the combination has no fused kernel."""
    from .. import hip

    spec = tiles.spec if isinstance(tiles, deferred.DeferredTiles) else None
    if spec is None or not spec.get("cl", False) or conv.groups != 1 or tuple(conv.dilation) != (1, 1):
        return None
    if conv.out_channels % 4 or tuple(conv.stride) != tuple(g.model_stride):
        return None
    packed = conv._packed_weights(tiles, True)
    if packed is None:
        return None
    if spec["kind"] == "gather":
        if x1 is not None:
            return None
        return hip.gather_conv_cl(spec["x"], spec.get("x2"), spec["block"], spec["idx"], spec["scale"], spec["shift"], spec["act"],
                                  packed, conv.bias, conv.out_channels, conv.kernel_size, conv.stride,
                                  full=dict(offset=g.offset, out_res=tuple(out.shape[2:]), residual=residual),
                                  upsample2x=spec.get("up", False), out=out, twins=twins)
    if tuple(conv.kernel_size) != (3, 3) or tuple(conv.stride) != (1, 1):
        return None
    return hip.scatter_gather_conv_scatter_cl(spec["x"], spec["y"], spec["block"], spec["idx"], spec["map"], spec["scale"],
                                              spec["shift"], spec["act"], packed, conv.bias, conv.out_channels,
                                              conv.kernel_size, g.offset, out, residual=residual, x1=x1, table1=table1,
                                              twins=twins)


class ScatterWithBlockResidual(SIGEModule):
    """scatter(main tiles, residual = cached shortcut) + `x1 - y1` correction on the
    shortcut branch's own tiles (sige/nn/scatter.py:66-136)."""

    def __init__(self, main_gather: Gather, shortcut_gather: Gather):
        super(ScatterWithBlockResidual, self).__init__()
        self.main_gather = SIGEModuleWrapper(main_gather)
        self.shortcut_gather = SIGEModuleWrapper(shortcut_gather)

        self.load_runtime("scatter_with_block_residual")
        self.scatter_runtime = None
        self.output_res = None
        self.original_outputs = {}
        self.original_residuals = {}
        self.inplace = False
        self._out_bufs = _OutputBuffers()
        self.twins = _TwinBuffers()

    def clear_cache(self):
        self.original_outputs = {}
        self.original_residuals = {}
        self._out_bufs.clear()
        self.twins.clear()

    def refresh_outputs(self, new_mask: bool = False):
        """(sige_amd.parallel) the cache tensors were rewritten in place -- or (`new_mask`, sige_amd.plan) the mask changed:
        persistent outputs follow, addresses kept."""
        stamp = (self.main_gather.module.timestamp, self.shortcut_gather.module.timestamp) if new_mask else None
        self._out_bufs.refresh(self.original_outputs, stamp)
        self.twins.refresh(self.original_outputs, stamp)

    def plan_tables(self):
        """(sige_amd.plan) build, now, the per-mask lookup tables this module's sparse forward would build lazily."""
        for cached in self.original_outputs.values():
            if cached.is_cuda:
                for g in (self.main_gather.module, self.shortcut_gather.module):
                    if g.active_indices is not None:
                        g.tile_table(cached.shape[2:], cached.device)

    def forward_fused(self, conv, tiles: torch.Tensor, residual: torch.Tensor) -> torch.Tensor:
        """(not in the reference) `self(conv(tiles), residual)` in ONE launch (see Scatter.forward_fused): `residual` are
        the shortcut conv's tiles; the block-residual correction runs in the conv's epilogue."""
        if self.mode == "sparse" and self.inplace and not self.sparse_update:
            mg: Gather = self.main_gather.module
            sg: Gather = self.shortcut_gather.module
            y0 = self.original_outputs[self.cache_id]
            y1 = self.original_residuals[self.cache_id]
            if (_cl_ok(y0, y1) and hip_is_cl(y1) and isinstance(tiles, deferred.DeferredTiles) and tiles.spec is not None
                    and sg.active_indices.size(0) <= mg.active_indices.size(0)):
                out = self._out_bufs.get(self.cache_id, y0, (mg.timestamp, sg.timestamp))
                x1 = deferred.resolve(residual)
                tw = self.twins.launch_args(self.cache_id, y0, (mg.timestamp, sg.timestamp))
                done = _fused_conv_into(conv, tiles, out, mg, residual=y1, x1=x1, table1=sg.tile_table(y0.shape[2:], y0.device),
                                        twins=[(b, sc, sh) for _, b, sc, sh in tw])
                if done is not None:
                    return tag_twins(done, {k: b for k, b, _, _ in tw})
        return self.forward(conv(tiles), residual)

    def forward(self, x: torch.Tensor, residual: torch.Tensor, x_is_sum: bool = False) -> torch.Tensor:
        """`x_is_sum` (full mode only, not in the reference): `x` already is main + residual."""
        self.check_dtype(x, residual)
        self.check_dim(x, residual)
        assert not x_is_sum or self.mode == "full"
        if self.mode == "sparse":
            x, residual = deferred.resolve(x), deferred.resolve(residual)
            mg: Gather = self.main_gather.module
            sg: Gather = self.shortcut_gather.module
            y0 = self.original_outputs[self.cache_id]
            y1 = self.original_residuals[self.cache_id]
            res = y0.shape[2:]
            if _cl_ok(y0, y1) and hip_is_cl(y1):
                from .. import hip

                out = self._out_bufs.get(self.cache_id, y0, (mg.timestamp, sg.timestamp)) \
                    if (self.inplace and not self.sparse_update) else None
                output = hip.scatter_with_block_residual_cl(
                    x, y0, residual, y1, mg.offset, mg.model_stride, mg.indices_on(x.device), mg.tile_table(res, x.device),
                    sg.indices_on(x.device), sg.tile_table(res, x.device), out=out)
            elif _fused_ok(x):
                from .. import hip

                output = hip.scatter_with_block_residual_fused(
                    x.contiguous(), deferred.from_cache(y0), residual.contiguous(), deferred.from_cache(y1),
                    mg.tile_table(res, x.device), mg.active_indices.size(0),
                    sg.tile_table(res, x.device), sg.active_indices.size(0))
            else:
                fn = self.native(self.runtime, x)
                output = fn(x.contiguous(), deferred.from_cache(y0).contiguous(), residual.contiguous(), deferred.from_cache(y1).contiguous(),
                            mg.offset[0], mg.offset[1], mg.model_stride[0], mg.model_stride[1],
                            mg.indices_on(x.device), sg.indices_on(x.device))
            if self.sparse_update:
                if self.scatter_runtime is None:
                    self.scatter_runtime = self.load_runtime("scatter", {})
                y0.copy_(output)
                self._out_bufs.invalidate(self.cache_id)
                self.twins.invalidate(self.cache_id)
                if _fused_ok(x):
                    from .. import hip

                    y1.copy_(hip.scatter_fused(residual.contiguous(), deferred.from_cache(y1), sg.tile_table(res, x.device),
                                               sg.active_indices.size(0), None))
                else:
                    fn = self.native(self.scatter_runtime, x)
                    y1.copy_(fn(residual.contiguous(), deferred.from_cache(y1).contiguous(), sg.offset[0], sg.offset[1],
                                sg.model_stride[0], sg.model_stride[1], sg.indices_on(x.device), None))
            return tag_twins(output, emulated_twins(output, twins_for(self.twins.regs, self.cache_id)) if EMULATE_TWINS else {})  # (no twin written)
        if self.mode == "full":
            # (x_is_sum: the caller's conv already added the residual in its epilogue -- one launch less, one pass less)
            output = x if x_is_sum else x + residual
            self.output_res = output.shape[2:]
            self.original_outputs[self.cache_id] = deferred.to_cache(output, self.cache_dtype)
            cl = self.original_outputs[self.cache_id].is_contiguous(memory_format=torch.channels_last)
            self.original_residuals[self.cache_id] = deferred.to_cache(
                residual.contiguous(memory_format=torch.channels_last) if cl and not residual.is_contiguous() else residual,
                self.cache_dtype)
            self._out_bufs.invalidate(self.cache_id)
            self.twins.invalidate(self.cache_id)
            return output
        if self.mode == "profile":
            c = x.shape[1]
            return torch.full((self.original_outputs[self.cache_id].size(0), c, *self.output_res),
                              fill_value=x[0, 0, 0, 0] + residual[0, 0, 0, 0], dtype=x.dtype, device=x.device)
        raise NotImplementedError("Unknown mode: [%s]!!!" % self.mode)
