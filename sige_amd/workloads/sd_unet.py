"""Stable-Diffusion v1 U-Net workload (BASELINE.json configs[3]) on sige_amd.nn.

The network of stable-diffusion/ldm/modules/diffusionmodules/sige_openaimodel.py (`SIGEUNetModel`): tiled residual blocks
(GroupNorm affine + SiLU cached per SAMPLE: classifier-free guidance runs batch 2, so every cached (scale, shift) is
[B,C,1,1]), tiled stride-2 / upsampling convs, sparse-query spatial transformers (sd_transformer.py), a dense middle block.
Parameter names follow the reference (`input_blocks.4.0.in_layers.2`, `output_blocks.5.1.transformer_blocks.0.attn1.to_q`,
`middle_block.0.out_layers.3`, `time_embed.0`, `out.2`), so its state dict loads here (tests/test_reference_models.py).

Sparse mode: the timestep embedding is folded into the cached second shift (sige_openaimodel.py:165-176), so a sparse
forward takes no embedding; a residual block is gather(+affine+SiLU) -> conv -> scatter_gather(+affine+SiLU) -> conv ->
scatter with (block) residual, i.e. three fused launches on channels-last GPU tensors; the skip concatenation of the up path
is a deferred cat read through two base pointers.
"""
import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import nn
from torch.nn import functional as F

from ..nn import Gather, Scatter, ScatterGather, ScatterWithBlockResidual, SIGEConv2d, SIGEModel, SIGEModule, paired_convs
from ..nn.deferred import lazy_cat
from .sd_transformer import SpatialTransformer, group_norm_affine


# the 1x1 skip_connection of a channel-changing residual block rides in conv1's launch (sige_amd.hip.conv_pair).  A paired
# conv1 stays on csrc/conv_mfma.hpp; unpaired it may be routed to the tile conv v3 (tools/sd_route_bench.py measures both)
PAIRED_SHORTCUT = True


@dataclass
class SDConfig:
    in_channels: int = 4
    model_channels: int = 320
    out_channels: int = 4
    num_res_blocks: int = 2
    attention_resolutions: Tuple[int, ...] = (4, 2, 1)
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_heads: int = 8
    context_dim: int = 768
    transformer_depth: int = 1
    main_block_size: int = 6
    shortcut_block_size: int = 4


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    return torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1) if dim % 2 else emb


class GroupNorm32(nn.GroupNorm):
    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


class ResBlock(SIGEModule):
    """`tiled=False`: the plain residual block of the middle of the network (statistics recomputed on whatever comes in)."""

    takes_emb = True

    def __init__(self, cfg: SDConfig, cin: int, cout: int, emb_ch: int, tiled: bool = True):
        super().__init__()
        self.cin, self.cout, self.tiled = cin, cout, tiled
        Conv = SIGEConv2d if tiled else nn.Conv2d
        self.in_layers = nn.Sequential(GroupNorm32(32, cin), nn.SiLU(), Conv(cin, cout, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_ch, cout))
        self.out_layers = nn.Sequential(GroupNorm32(32, cout), nn.SiLU(), nn.Dropout(0.0), Conv(cout, cout, 3, padding=1))
        self.skip_connection = nn.Identity() if cin == cout else Conv(cin, cout, 1)
        if tiled:
            self.main_gather = Gather(self.in_layers[2], cfg.main_block_size, activation_name="swish")
            self.scatter_gather = ScatterGather(self.main_gather, activation_name="swish")
            if cin != cout:
                self.shortcut_gather = Gather(self.skip_connection, cfg.shortcut_block_size)
                self.scatter = ScatterWithBlockResidual(self.main_gather, self.shortcut_gather)
            else:
                self.scatter = Scatter(self.main_gather)
        self.affine = None  # (scale1, shift1, scale2, shift2), each [B,C,1,1]

    def forward(self, x, emb):
        if not self.tiled:
            h = self.in_layers(x)
            h = h + self.emb_layers(emb)[:, :, None, None]
            return self.skip_connection(x) + self.out_layers(h)
        if self.mode == "full":
            skip = x if self.cin == self.cout else self.skip_connection(self.shortcut_gather(x))
            h, s1, t1 = group_norm_affine(self.main_gather(x), self.in_layers[0])
            h = self.scatter_gather(self.in_layers[2](F.silu(h)))
            e = self.emb_layers(emb)[:, :, None, None]
            h, s2, t2 = group_norm_affine(h + e, self.out_layers[0])
            self.affine = tuple(v.contiguous() for v in (s1, t1, s2, s2 * e + t2))  # the embedding add folded into the shift
            return self.scatter(self.out_layers[3](F.silu(h)), skip)
        if self.mode in ("sparse", "profile"):
            s1, t1, s2, t2 = self.affine
            with paired_convs(x, enabled=PAIRED_SHORTCUT and self.cin != self.cout and self.mode == "sparse"):  # the 1x1 rides in conv1's launch
                skip = x if self.cin == self.cout else self.skip_connection(self.shortcut_gather(x))
                h = self.in_layers[2](self.main_gather(x, s1, t1))
            tiles = self.scatter_gather(h, s2, t2)
            if self.mode == "sparse":
                return self.scatter.forward_fused(self.out_layers[3], tiles, skip)
            return self.scatter(self.out_layers[3](tiles), skip)
        raise NotImplementedError("Unknown mode [%s]!!!" % self.mode)


class Downsample(SIGEModule):
    def __init__(self, cfg: SDConfig, ch: int):
        super().__init__()
        self.op = SIGEConv2d(ch, ch, 3, stride=2, padding=1)
        self.gather = Gather(self.op, cfg.main_block_size)
        self.scatter = Scatter(self.gather)

    def forward(self, x):
        t = self.gather(x)
        return self.scatter.forward_fused(self.op, t) if self.mode == "sparse" else self.scatter(self.op(t))


class Upsample(SIGEModule):
    def __init__(self, cfg: SDConfig, ch: int):
        super().__init__()
        self.conv = SIGEConv2d(ch, ch, 3, padding=1)
        self.gather = Gather(self.conv, cfg.main_block_size)
        self.scatter = Scatter(self.gather)

    def forward(self, x):
        if self.mode == "sparse" and self.gather.fuses_upsample(x):
            return self.scatter.forward_fused(self.conv, self.gather(x, upsample2x=True))
        t = self.gather(F.interpolate(x, scale_factor=2, mode="nearest"))
        return self.scatter.forward_fused(self.conv, t) if self.mode == "sparse" else self.scatter(self.conv(t))


class SDUNet(SIGEModel):
    def __init__(self, cfg: SDConfig = SDConfig()):
        super().__init__()
        self.cfg = cfg
        mc = cfg.model_channels
        emb_ch = 4 * mc
        self.time_embed = nn.Sequential(nn.Linear(mc, emb_ch), nn.SiLU(), nn.Linear(emb_ch, emb_ch))

        def transformer(ch, block_size: Optional[int] = 4):
            return SpatialTransformer(ch, cfg.num_heads, ch // cfg.num_heads, depth=cfg.transformer_depth,
                                      context_dim=cfg.context_dim, block_size=block_size)

        self.input_blocks = nn.ModuleList([nn.ModuleList([nn.Conv2d(cfg.in_channels, mc, 3, padding=1)])])
        chans, ch, ds = [mc], mc, 1
        for level, mult in enumerate(cfg.channel_mult):
            for _ in range(cfg.num_res_blocks):
                layers = [ResBlock(cfg, ch, mult * mc, emb_ch)]
                ch = mult * mc
                if ds in cfg.attention_resolutions:
                    layers.append(transformer(ch))
                self.input_blocks.append(nn.ModuleList(layers))
                chans.append(ch)
            if level != len(cfg.channel_mult) - 1:
                self.input_blocks.append(nn.ModuleList([Downsample(cfg, ch)]))
                chans.append(ch)
                ds *= 2
        self.middle_block = nn.ModuleList([ResBlock(cfg, ch, ch, emb_ch, tiled=False), transformer(ch, None),
                                           ResBlock(cfg, ch, ch, emb_ch, tiled=False)])
        self.output_blocks = nn.ModuleList()
        for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
            for i in range(cfg.num_res_blocks + 1):
                layers = [ResBlock(cfg, ch + chans.pop(), mc * mult, emb_ch)]
                ch = mc * mult
                if ds in cfg.attention_resolutions:
                    layers.append(transformer(ch))
                if level and i == cfg.num_res_blocks:
                    layers.append(Upsample(cfg, ch))
                    ds //= 2
                self.output_blocks.append(nn.ModuleList(layers))
        self.out = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(), nn.Conv2d(mc, cfg.out_channels, 3, padding=1))

    @staticmethod
    def _run(layers, h, emb, context):
        for layer in layers:
            if getattr(layer, "takes_emb", False):
                h = layer(h, emb)
            elif isinstance(layer, SpatialTransformer):
                h = layer(h, context)
            else:
                h = layer(h)
        return h

    def forward(self, x, timesteps=None, context=None):
        emb = self.time_embed(timestep_embedding(timesteps, self.cfg.model_channels))
        hs = []
        h = x
        for blk in self.input_blocks:
            h = self._run(blk, h, emb, context)
            hs.append(h)
        h = self._run(self.middle_block, h, emb, context)
        for blk in self.output_blocks:
            skip = hs.pop()
            first = blk[0]
            if self.mode == "sparse" and first.cin != first.cout:
                h = lazy_cat(h, skip)  # consumed by the block's two Gathers only
            else:
                h = torch.cat([h, skip], dim=1)
            h = self._run(blk, h, emb, context)
        return self.out(h)
