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


class Scatter(SIGEModule):
    def __init__(self, gather: Gather):
        super(Scatter, self).__init__()
        self.gather = SIGEModuleWrapper(gather)

        self.load_runtime("scatter")
        self.output_res = None
        self.original_outputs = {}

    def clear_cache(self):
        self.original_outputs = {}

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        self.check_dtype(x, residual)
        self.check_dim(x, residual)
        if self.mode == "sparse":
            x = deferred.resolve(x)
            g: Gather = self.gather.module
            cached = self.original_outputs[self.cache_id]
            if _fused_ok(x):
                from .. import hip

                output = hip.scatter_fused(
                    x.contiguous(), cached, g.tile_table(cached.shape[2:], x.device), g.active_indices.size(0),
                    None if residual is None else residual.contiguous())
            else:
                fn = self.native(self.runtime, x)
                output = fn(x.contiguous(), cached.contiguous(), g.offset[0], g.offset[1],
                            g.model_stride[0], g.model_stride[1], g.indices_on(x.device),
                            None if residual is None else residual.contiguous())
            if self.sparse_update:
                cached.copy_(output)
            return output
        if self.mode == "full":
            output = x if residual is None else x + residual
            self.output_res = output.shape[2:]
            self.original_outputs[self.cache_id] = output.contiguous()
            return output
        if self.mode == "profile":
            c = x.shape[1]
            output = torch.full((self.original_outputs[self.cache_id].size(0), c, *self.output_res),
                                fill_value=x[0, 0, 0, 0], dtype=x.dtype, device=x.device)
            if residual is not None:
                output = output + residual
            return output
        raise NotImplementedError("Unknown mode: [%s]!!!" % self.mode)


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

    def clear_cache(self):
        self.original_outputs = {}
        self.original_residuals = {}

    def forward(self, x: torch.Tensor, residual: torch.Tensor) -> torch.Tensor:
        self.check_dtype(x, residual)
        self.check_dim(x, residual)
        if self.mode == "sparse":
            x, residual = deferred.resolve(x), deferred.resolve(residual)
            mg: Gather = self.main_gather.module
            sg: Gather = self.shortcut_gather.module
            y0 = self.original_outputs[self.cache_id]
            y1 = self.original_residuals[self.cache_id]
            res = y0.shape[2:]
            if _fused_ok(x):
                from .. import hip

                output = hip.scatter_with_block_residual_fused(
                    x.contiguous(), y0, residual.contiguous(), y1,
                    mg.tile_table(res, x.device), mg.active_indices.size(0),
                    sg.tile_table(res, x.device), sg.active_indices.size(0))
            else:
                fn = self.native(self.runtime, x)
                output = fn(x.contiguous(), y0.contiguous(), residual.contiguous(), y1.contiguous(),
                            mg.offset[0], mg.offset[1], mg.model_stride[0], mg.model_stride[1],
                            mg.indices_on(x.device), sg.indices_on(x.device))
            if self.sparse_update:
                if self.scatter_runtime is None:
                    self.scatter_runtime = self.load_runtime("scatter", {})
                y0.copy_(output)
                if _fused_ok(x):
                    from .. import hip

                    y1.copy_(hip.scatter_fused(residual.contiguous(), y1, sg.tile_table(res, x.device),
                                               sg.active_indices.size(0), None))
                else:
                    fn = self.native(self.scatter_runtime, x)
                    y1.copy_(fn(residual.contiguous(), y1.contiguous(), sg.offset[0], sg.offset[1],
                                sg.model_stride[0], sg.model_stride[1], sg.indices_on(x.device), None))
            return output
        if self.mode == "full":
            output = x + residual
            self.output_res = output.shape[2:]
            self.original_outputs[self.cache_id] = output.contiguous()
            self.original_residuals[self.cache_id] = residual.contiguous()
            return output
        if self.mode == "profile":
            c = x.shape[1]
            return torch.full((self.original_outputs[self.cache_id].size(0), c, *self.output_res),
                              fill_value=x[0, 0, 0, 0] + residual[0, 0, 0, 0], dtype=x.dtype, device=x.device)
        raise NotImplementedError("Unknown mode: [%s]!!!" % self.mode)
