"""Gather: active input tiles (with halo) of a conv out of the full activation.

API parity with sige/nn/gather.py:12-108.  Geometry (effective block size,
block stride, offset) is derived exactly as the reference's constructor does;
`set_mask` memoises the index list per (resolution, geometry) in the cache
shared by SIGEModel.set_masks.  Extra here: per-mask lookup tables for the
fused single-pass scatter kernels (tile tables), built lazily on the GPU.
"""
import warnings
from typing import Dict, Optional, Tuple, Union

import torch
from torch import nn

from ..utils import reduce_mask
from . import deferred
from .base import SIGEModule
from .utils import activation


class Gather(SIGEModule):
    def __init__(
        self,
        conv: nn.Conv2d,
        block_size: Union[int, Tuple[int, int]],
        offset: Optional[Union[int, Tuple[int, int]]] = None,
        activation_name: str = "identity",
        activation_first: bool = False,
        verbose: bool = False,
    ):
        super(Gather, self).__init__()
        if isinstance(block_size, int):
            block_size = (block_size, block_size)
        k, s = conv.kernel_size, conv.stride
        # out tile = how many conv outputs fit in the requested block; the block is
        # then shrunk to exactly cover them (6 -> 5 for a stride-2 3x3 conv)
        out_tile = tuple(max(block_size[i] - k[i], 0) // s[i] + 1 for i in (0, 1))
        fitted = tuple((out_tile[i] - 1) * s[i] + k[i] for i in (0, 1))
        if fitted != tuple(block_size):
            warnings.warn("Change the block size from (%d, %d) to (%d, %d)" % (*block_size, *fitted))

        self.model_stride = conv.stride
        self.kernel_size = conv.kernel_size
        self.block_size = fitted
        self.block_stride = (out_tile[0] * s[0], out_tile[1] * s[1])
        self.out_tile = out_tile
        if offset is None:
            self.offset = conv.padding
        else:
            self.offset = (offset, offset) if isinstance(offset, int) else offset
        self.activation_name = activation_name
        self.activation_first = activation_first
        self.verbose = verbose

        self.load_runtime("gather")

        self.input_res: Optional[Tuple[int, int]] = None
        self.active_indices: Optional[torch.Tensor] = None
        self._tables: Dict = {}

    def note_full_input(self, res):
        """What a full-mode forward records (sige/nn/gather.py:51-57: the input resolution) -- for a full pass whose input tensor
        never exists as one tensor (a torch.cat / F.interpolate fused into the consuming conv)."""
        self.input_res = torch.Size(tuple(res))

    def forward(
        self, x: torch.Tensor, scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
        upsample2x: bool = False, preactivated: bool = False
    ) -> torch.Tensor:
        """`preactivated` (sparse mode, not in the reference): `x` already IS activation(scale * input + shift) -- an
        activated twin its producer wrote (sige_amd.hip: twins) -- so the tiles are gathered raw whatever this module's
        activation is.  Zero fill stays zero (gather.cpp:27-30: padding is never transformed; SiLU(0) = 0 anyway).
        `upsample2x` (sparse mode, not in the reference): `x` is the HALF-resolution tensor and the tiles are
        taken from its x2 nearest-neighbour upsampling -- which the fused gather -> conv kernel never materialises
        (the `F.interpolate` in front of the U-Net's upsampling convs).  Only valid where `fuses_upsample(x)` holds;
        anywhere else the MODEL upsamples, as the reference's does (sige_fused_unet.py:222-227), and calls the
        gather without the flag."""
        self.check_dtype(x, scale, shift)
        self.check_dim(x, scale, shift)
        if upsample2x:
            x = deferred.resolve(x)
            if not self.fuses_upsample(x, scale, shift):
                raise ValueError("Gather(upsample2x=True) needs the fused channels-last gather -> conv path "
                                 "(sparse mode, channels-last fp32 GPU tensor, per-channel affine); upsample in the model instead")
        if self.mode == "sparse":
            act_name = self.activation_name
            if preactivated:
                if scale is not None or shift is not None:
                    raise ValueError("Gather(preactivated=True) takes no scale / shift: the input already carries them")
                act_name = "identity"
            x2 = None
            if isinstance(x, deferred.LazyCat) and x.spec is not None:
                # a pending cat: keep the two tensors apart if the fused conv can read them in place
                a, b = x.parts
                from .. import hip

                if (hip.cat_fusable(a.shape[0], a.shape[1], self.kernel_size) and a.shape[1] % 4 == 0 and b.shape[1] % 4 == 0
                        and deferred.defer_ok(a, scale, shift, self.activation_first, self.sparse_update, act_name)
                        and hip.is_cl(a) and hip.is_cl(b)):
                    x, x2 = a, b
            fn = self.native(self.runtime, x)
            x = deferred.resolve(x)
            # channels-last activations stay channels-last (16-byte coalesced tile gathers)
            cl = x2 is not None or deferred.channels_last_ok(x, scale, shift, self.activation_first)
            if not cl:
                x = x.contiguous()
            idx = self.indices_on(x.device)
            scale = None if scale is None else scale.contiguous()
            shift = None if shift is None else shift.contiguous()
            bh, bw = self.block_size
            act, first = act_name, self.activation_first

            def run():
                xx = x if x2 is None else torch.cat([x, x2], dim=1)
                if upsample2x:
                    xx = torch.nn.functional.interpolate(xx, scale_factor=2.0, mode="nearest")
                if cl:
                    from .. import hip

                    return hip.gather_cl(xx, bh, bw, idx, scale, shift, act)
                return fn(xx, bh, bw, idx, scale, shift, act, first)

            if deferred.defer_ok(x, scale, shift, first, self.sparse_update, act):
                # not computed yet: a SIGEConv2d consumer fuses it into its prologue,
                # any other consumer materialises it through the gather kernel
                channels = x.shape[1] + (0 if x2 is None else x2.shape[1])
                return deferred.DeferredTiles(
                    (x.shape[0] * idx.shape[0], channels, bh, bw), x.dtype, x.device, run,
                    dict(kind="gather", x=x, x2=x2, block=(bh, bw), idx=idx, scale=scale, shift=shift, act=act, cl=cl,
                         up=upsample2x))
            return run()
        if self.mode == "full":
            self.note_full_input(x.shape[2:])
            assert scale is None
            assert shift is None
            return x
        if self.mode == "profile":
            # dummy of the right shape that still depends on the inputs, so a MACs
            # tracer follows the graph (sige/nn/gather.py:59-70)
            b, c = x.shape[:2]
            output = torch.full((b * self.active_indices.size(0), c, *self.block_size), fill_value=x[0, 0, 0, 0],
                                dtype=x.dtype, device=x.device)
            if scale is not None:
                output = output * scale[0, 0, 0, 0]
            if shift is not None:
                output = output + shift[0, 0, 0, 0]
            return activation(output, self.activation_name)
        raise NotImplementedError("Unknown mode: [%s]!!!" % self.mode)

    # ------------------------------------------------------------------ masks --
    def set_mask(self, masks: Dict, cache: Dict, timestamp: int):
        if self.timestamp == timestamp:
            return
        super(Gather, self).set_mask(masks, cache, timestamp)
        assert self.input_res is not None
        res = tuple(self.input_res)
        self.mask = masks[res]
        key = self.index_key(res)
        if key not in cache:
            cache[key] = reduce_mask(self.mask, self.block_size, self.block_stride, self.offset, verbose=self.verbose)
        self.active_indices = cache[key]
        # (tile tables also depend on the conv's stride and output tile, which the index key does not contain)
        self._tables = cache.setdefault(("tile_tables", *key[1:], *self.model_stride, *self.out_tile), {})

    def fuses_upsample(self, x: torch.Tensor, scale=None, shift=None) -> bool:
        """Can `forward(x, upsample2x=True)` be used for this input (see there)?"""
        return (self.mode == "sparse" and not isinstance(x, deferred.LazyCat)
                and deferred.channels_last_ok(deferred.resolve(x), scale, shift, self.activation_first)
                and deferred.defer_ok(deferred.resolve(x), scale, shift, self.activation_first, self.sparse_update,
                                      self.activation_name))

    def index_key(self, res) -> tuple:
        """Key of this gather's index list in the cache shared by SIGEModel.set_masks."""
        return ("active_indices", *res, *self.block_size, *self.block_stride, *self.offset)

    def indices_on(self, device: torch.device) -> torch.Tensor:
        """active_indices on `device` (the reference requires mask and activations
        on the same device; a CPU mask is migrated once per set_mask here)."""
        idx = self.active_indices
        if idx.device != device:
            idx = self._tables.get(("idx", device))
            if idx is None:
                idx = self.active_indices.to(device)
                self._tables[("idx", device)] = idx
        return idx if idx.is_contiguous() else idx.contiguous()

    def tile_table(self, out_res: Tuple[int, int], device: torch.device) -> torch.Tensor:
        """[ceil(H/o), ceil(W/o)] int32 lookup "which active tile covers this
        output cell" for the fused scatter kernels; memoised per mask."""
        key = ("table", tuple(out_res), device)
        table = self._tables.get(key)
        if table is None:
            from .. import hip

            table = hip.tile_table(self.indices_on(device), self.offset, self.model_stride, self.out_tile, out_res)
            self._tables[key] = table
        return table
