"""ScatterGather: "scatter conv-1's output tiles, gather conv-2's input tiles"
without materialising the full tensor.

API parity with sige/nn/scatter_gather.py:10-117 (attributes `gather`,
`scatter_map`, `original_outputs`, `output_res`; the per-mask scatter map is
memoised in the SIGEModel.set_masks cache under the same key).
"""
from typing import Dict, Optional

import torch

from . import deferred
from .base import SIGEModule, SIGEModuleWrapper
from .gather import Gather
from .utils import activation


class ScatterGather(SIGEModule):
    def __init__(self, gather: Gather, activation_name: str = "identity", activation_first: bool = False):
        super(ScatterGather, self).__init__()
        self.gather = SIGEModuleWrapper(gather)
        self.activation_name = activation_name
        self.activation_first = activation_first

        self.load_runtime("scatter_gather")
        self.scatter_runtime = self.load_runtime("scatter", {})
        self.get_scatter_map_runtime = self.load_runtime("get_scatter_map", {})

        self.scatter_map = None
        self.output_res = None
        self.original_outputs = {}
        self.activated_outputs = {}  # see cache_activated()
        self._maps: Dict = {}

    def clear_cache(self):
        self.original_outputs = {}
        self.activated_outputs = {}  # see cache_activated()

    def _build_map(self, idx: torch.Tensor) -> torch.Tensor:
        g: Gather = self.gather.module
        h, w = g.mask.shape
        fn = self.native(self.get_scatter_map_runtime, idx)
        return fn(h, w, g.block_size[0], g.block_size[1], g.kernel_size[0], g.kernel_size[1],
                  g.offset[0], g.offset[1], g.model_stride[0], g.model_stride[1], idx)

    def _map_on(self, device: torch.device) -> torch.Tensor:
        m = self.scatter_map
        if m is None or m.device != device:
            m = self._maps.get(device)
            if m is None:
                if self.scatter_map is not None:
                    m = self.scatter_map.to(device)
                else:  # mask lives on a device with no backend (CPU mask, GPU activations)
                    m = self._build_map(self.gather.module.indices_on(device))
                    self.scatter_map = m
                self._maps[device] = m
        return m if m.is_contiguous() else m.contiguous()

    def cache_activated(self, scale: torch.Tensor, shift: torch.Tensor):
        """(not in the reference) After a full-mode forward: also keep act(scale * cached + shift).  A sparse
        forward may then be handed tiles that the producing conv already activated (SIGEConv2d `out_affine`)
        with `preactivated=True`: both sources of the scatter-gather are final values and the conv stages them
        raw -- the affine + SiLU is computed once per element instead of once per output-channel block."""
        from .utils import activation as act_fn

        y = self.original_outputs[self.cache_id]
        old = self.activated_outputs.get(self.cache_id)
        if y.is_cuda and y.dtype in (torch.float32, torch.float16):
            from .. import hip

            if hip.is_cl(y) and scale.dim() == 4 and shift.dim() == 4 and self.activation_name in hip.ACT:
                # one streaming pass, straight into the existing copy when there is one (same address: a captured hipGraph
                # that reads it stays valid).  An fp16-stored cache gets an fp16 activated copy: no fp32 working copy of it.
                reuse = old if (old is not None and old.shape == y.shape and old.stride() == y.stride() and old.device == y.device
                                and old.dtype == y.dtype) else None
                done = hip.affine_act_cl(y, scale, shift, self.activation_name, out=reuse)
                if done is not None:
                    self.activated_outputs[self.cache_id] = done
                    return
        new = deferred.to_cache(act_fn(deferred.from_cache(y) * scale + shift, self.activation_name), self.cache_dtype)
        if old is not None and old.shape == new.shape and old.stride() == new.stride() and old.device == new.device and old.dtype == new.dtype:
            old.copy_(new)  # same address: a captured hipGraph that reads the activated copy stays valid
        else:
            self.activated_outputs[self.cache_id] = new

    def forward(
        self, x: torch.Tensor, scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
        preactivated: bool = False
    ) -> torch.Tensor:
        self.check_dtype(x, scale, shift)
        self.check_dim(x, scale, shift)
        g: Gather = self.gather.module
        if self.mode == "sparse" and preactivated:
            assert scale is None and shift is None and not self.sparse_update
            cached = self.activated_outputs[self.cache_id]
            x = deferred.resolve(x)
            cl = deferred.channels_last_ok(cached, cache=True)
            x = x.contiguous(memory_format=torch.channels_last) if cl else x.contiguous()
            idx, smap = g.indices_on(x.device), self._map_on(x.device)
            bh, bw = g.block_size

            def run_pre():
                if cl:
                    from .. import hip

                    return hip.scatter_gather_cl(x, cached, bh, bw, idx, smap, None, None, "identity")
                return self.native(self.runtime, x)(x, deferred.from_cache(cached).contiguous(), bh, bw, idx, smap, None, None, "identity", False)

            if deferred.defer_ok(x, None, None, False, False, "identity"):
                return deferred.DeferredTiles(
                    (cached.shape[0] * idx.shape[0], x.shape[1], bh, bw), x.dtype, x.device, run_pre,
                    dict(kind="scatter_gather", x=x, y=cached, block=(bh, bw), idx=idx, map=smap, scale=None, shift=None,
                         act="identity", cl=cl))
            return run_pre()
        if self.mode == "sparse":
            cached = self.original_outputs[self.cache_id]
            fn = self.native(self.runtime, x)
            x = deferred.resolve(x)
            # the cache decides the layout: channels-last cache -> channels-last tiles
            cl = deferred.channels_last_ok(cached, scale, shift, self.activation_first, cache=True)
            x = x.contiguous(memory_format=torch.channels_last) if cl else x.contiguous()
            idx, smap = g.indices_on(x.device), self._map_on(x.device)
            scale = None if scale is None else scale.contiguous()
            shift = None if shift is None else shift.contiguous()
            bh, bw = g.block_size
            act, first = self.activation_name, self.activation_first

            def run():
                if cl:
                    from .. import hip

                    return hip.scatter_gather_cl(x, cached, bh, bw, idx, smap, scale, shift, act)
                return fn(x, deferred.from_cache(cached).contiguous(), bh, bw, idx, smap, scale, shift, act, first)

            if deferred.defer_ok(x, scale, shift, first, self.sparse_update, act):
                return deferred.DeferredTiles(
                    (cached.shape[0] * idx.shape[0], x.shape[1], bh, bw), x.dtype, x.device, run,
                    dict(kind="scatter_gather", x=x, y=cached, block=(bh, bw), idx=idx, map=smap, scale=scale,
                         shift=shift, act=act, cl=cl))
            output = run()
            if self.sparse_update:
                if x.is_cuda:
                    from .. import hip

                    cached.copy_(hip.scatter_fused(x.contiguous(), deferred.from_cache(cached), g.tile_table(cached.shape[2:], x.device),
                                                   g.active_indices.size(0), None))
                else:
                    sfn = self.native(self.scatter_runtime, x)
                    cached.copy_(sfn(x.contiguous(), deferred.from_cache(cached).contiguous(), g.offset[0], g.offset[1],
                                     g.model_stride[0], g.model_stride[1], g.indices_on(x.device), None))
            return output
        if self.mode == "full":
            self.output_res = x.shape[2:]
            self.original_outputs[self.cache_id] = deferred.to_cache(x, self.cache_dtype)
            return x
        if self.mode == "profile":
            c = x.shape[1]
            output = torch.full(
                (self.original_outputs[self.cache_id].size(0) * g.active_indices.size(0), c, *g.block_size),
                fill_value=x[0, 0, 0, 0], dtype=x.dtype, device=x.device)
            if scale is not None:
                output = output * scale[0, 0, 0, 0]
            if shift is not None:
                output = output + shift[0, 0, 0, 0]
            return activation(output, self.activation_name)
        raise NotImplementedError("Unknown mode: [%s]!!!" % self.mode)

    def set_mask(self, masks: Dict, cache: Dict, timestamp: int):
        if self.timestamp == timestamp:
            return
        super(ScatterGather, self).set_mask(masks, cache, timestamp)
        g: Gather = self.gather.module
        g.set_mask(masks, cache, timestamp)
        h, w = g.mask.shape
        key = ("scatter_map", h, w, *g.block_size, *g.kernel_size, *g.offset, *g.model_stride)
        if key not in cache:
            idx = g.active_indices
            # built on the mask's device when it has a backend, else lazily on the
            # activations' device at the first sparse forward
            cache[key] = self._build_map(idx) if self.get_scatter_map_runtime[idx.device.type] is not None else None
        self.scatter_map = cache[key]
        self._maps = {}
