"""GauGAN SPADE generator workload (BASELINE.json configs[2]) built on sige_amd.nn.

The network of gaugan/models/spade_generators/sige_fused_spade_generator.py (the reference's `SIGEFusedSPADEGenerator`,
norm_G = "spadesyncbatch3x3"): a 36-channel label map drives seven SPADE residual blocks; the `num_sparse_layers`
highest-resolution blocks run tiled.  Parameter / buffer names follow the reference (`head_0.conv_0`,
`up_2.norm_s.mlp_gamma_beta`, `up_1.mlp_shared.0`, `*.param_free_norm.running_mean`, `fc`, `conv_img`), so its state dict
loads here; tests/test_reference_models.py checks in the build container that both give the same full and sparse outputs.

What SIGE caches for this model: every SPADE layer normalises with the BatchNorm's RUNNING statistics, i.e. a fixed
per-channel affine (scale = 1/sqrt(var + eps), shift = -mean * scale) that the gathers apply on the fly; the label-map
branch (mlp_shared -> mlp_gamma_beta) is recomputed on the edited tiles only and re-tiled through ScatterGather.

Sparse mode has two forms with identical arithmetic:
  * the module chain the reference runs: Gather / ScatterGather produce normalised tiles, a second ScatterGather re-tiles
    gamma|beta, then split, `n * (1 + gamma) + beta`, leaky ReLU as torch ops over [N,2C,6,6] (sige_normalization.py:62-88);
  * fused (channels-last GPU tensors): ONE `spade_modulate` pass (sige_amd/csrc/nhwc_ops.hip) writes the conv's input
    tile slab directly -- both re-tilings, the modulation and the activation; SURVEY.md 8(f) row 2.
"""
from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn
from torch.nn import functional as F

from ..nn import Gather, Scatter, ScatterGather, ScatterWithBlockResidual, SIGEConv2d, SIGEModel, SIGEModule
from ..nn import deferred


@dataclass
class SPADEConfig:
    ngf: int = 64
    semantic_nc: int = 36
    num_upsampling_layers: str = "more"      # normal | more | most
    main_block_size: Optional[int] = 6
    shortcut_block_size: Optional[int] = 4
    num_sparse_layers: int = 5
    crop_size: int = 512
    aspect_ratio: float = 2.0
    leaky_slope: float = 0.2
    bn_eps: float = 1e-5
    fused: bool = True                       # use the one-pass SPADE modulation kernel where it applies


def _cl_gpu(t: torch.Tensor) -> bool:
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.shape[1] % 4 == 0):
        return False
    from .. import hip

    return hip.is_cl(t)


def _refuse_in_stacked_mode(x: torch.Tensor, what: str) -> None:
    """A torch fallback is not seam-aware: on the tall [1,C,E*h,w] tensor of a stacked forward (sige_amd/stacked.py) it would read
    the neighbouring edit's rows as halo.  The standard configuration never gets here; anything that does fails loudly."""
    if x.is_cuda:
        from .. import hip

        if hip.get_edit_batch() > 1:
            raise RuntimeError("SpadeGenerator, stacked edits: %s has no library (seam-aware) form for this shape; refusing the "
                               "torch fallback, which would mix neighbouring edits" % what)


def _dense_conv(conv: nn.Conv2d, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A dense (non-tiled) conv of a sparse-mode forward (+ residual): on channels-last GPU tensors one launch of the MFMA tile
    kernel with every tile active (sige_amd.nn.dense.fused_conv2d), else the plain conv -- the blocks below `num_sparse_layers`
    are recomputed densely in sparse mode exactly as in the reference (sige_fused_spade_generator.py:133-173)."""
    from ..nn.dense import fusable, fused_conv2d

    if _cl_gpu(x) and fusable(conv) and conv.out_channels % 4 == 0:
        return fused_conv2d(conv, x, residual=residual)
    _refuse_in_stacked_mode(x, "a dense conv")
    out = conv(x)
    return out if residual is None else residual + out


_RESIZE_MEMO = None  # {(data_ptr, shape, size): tensor} while a generator forward runs (set and cleared by SPADEGenerator.forward)


def _resize(x: torch.Tensor, size) -> torch.Tensor:
    """F.interpolate(x, size=size, mode="nearest"); on channels-last GPU tensors one library launch (csrc/spade_ops.hip), so that
    a launch plan sees it."""
    size = (int(size[0]), int(size[1]))
    if tuple(x.shape[2:]) == size:
        return x  # (nearest resize to the same size is the identity: skip the copy)
    if _cl_gpu(x):
        from .. import hip

        # (within ONE generator forward the label map is resized to the same size by every block of a resolution -- fc and head_0,
        #  G_middle_0 and G_middle_1: the second asks for the tensor the first made)
        key = (x.data_ptr(), tuple(x.shape), size)
        if _RESIZE_MEMO is not None and key in _RESIZE_MEMO:
            return _RESIZE_MEMO[key]
        out = hip.resize_nearest_cl(x, size)
        if out is not None:
            if _RESIZE_MEMO is not None:
                _RESIZE_MEMO[key] = out
            return out
    _refuse_in_stacked_mode(x, "the nearest resize")
    return F.interpolate(x, size=size, mode="nearest")


class SpadeNorm(SIGEModule):
    """One SPADE layer: param-free BatchNorm (running statistics) modulated by gamma|beta = conv3x3(label features).
    `role`: "main" (tiles of a 3x3 conv: gamma|beta re-tiled by ScatterGather) or "shortcut" (tiles of the 1x1 shortcut
    conv: gamma|beta scattered to the full tensor, then gathered with the shortcut conv's own tile geometry)."""

    def __init__(self, cfg: SPADEConfig, channels: int, seg_gather: Optional[Gather], shortcut_conv: Optional[nn.Conv2d] = None):
        super().__init__()
        self.channels = channels
        self.role = "shortcut" if shortcut_conv is not None else "main"
        self.tiled = seg_gather is not None
        self.param_free_norm = nn.BatchNorm2d(channels, affine=False, eps=cfg.bn_eps)
        self.mlp_gamma_beta = (SIGEConv2d if self.tiled else nn.Conv2d)(cfg.ngf * 2, 2 * channels, 3, padding=1)
        if self.tiled and self.role == "shortcut":
            self.scatter = Scatter(seg_gather)
            self.gather = Gather(shortcut_conv, cfg.shortcut_block_size)
        elif self.tiled:
            self.scatter_gather = ScatterGather(seg_gather)
        self.scale = self.shift = None

    def affine4(self):
        return self.scale.view(1, -1, 1, 1), self.shift.view(1, -1, 1, 1)

    def _retile(self, gb: torch.Tensor) -> torch.Tensor:
        if not self.tiled:
            return gb
        if self.role == "shortcut":
            return self.gather(self.scatter(gb))
        return self.scatter_gather(gb)

    def forward(self, x: torch.Tensor, actv: torch.Tensor) -> torch.Tensor:
        """`x`: the full tensor (full mode; normalised here and its statistics remembered) or already-normalised tiles /
        tensor (sparse mode)."""
        if self.mode == "full":
            bn = self.param_free_norm
            n = bn(x)
            std = torch.sqrt(bn.running_var + bn.eps)
            self.scale = 1 / std
            self.shift = -(bn.running_mean / std)
        elif self.mode in ("sparse", "profile"):
            n = x
        else:
            raise NotImplementedError("Unknown mode [%s]!!!" % self.mode)
        gb = _dense_conv(self.mlp_gamma_beta, actv) if (not self.tiled and self.mode == "sparse") else self.mlp_gamma_beta(actv)
        gamma, beta = torch.split(self._retile(gb), self.channels, dim=1)
        return n * (1 + gamma) + beta

    # -- fused sparse forms ---------------------------------------------------------
    def modulated_dense(self, x: torch.Tensor, actv: torch.Tensor, slope: Optional[float]) -> Optional[torch.Tensor]:
        """A NON-tiled layer in sparse mode on channels-last GPU tensors: leaky((scale * x + shift) * (1 + gamma) + beta) in one
        pass over the full tensor, the param-free norm as its cached affine (what the tiled layers do); None if it does not
        apply.  Replaces BatchNorm + split + three elementwise kernels + leaky ReLU of the module chain."""
        if self.tiled or self.mode != "sparse" or self.scale is None or not (_cl_gpu(x) and _cl_gpu(actv)):
            return None
        from .. import hip

        gb = _dense_conv(self.mlp_gamma_beta, actv)
        if not _cl_gpu(gb):
            return None
        sc, sh = self.affine4()
        return hip.spade_modulate_dense_cl(x, sc.contiguous(), sh.contiguous(), gb, slope)

    def modulated_tiles(self, source, actv_tiles: torch.Tensor, slope: Optional[float]) -> Optional[torch.Tensor]:
        """leaky(norm(x) * (1 + gamma) + beta) on this layer's tiles in one pass, or None if the fused kernel does not apply.
        source = ("gather", gather_module, x_full) | ("scatter_gather", sg_module, conv_tiles)."""
        if not (self.tiled and self.mode == "sparse" and not self.sparse_update):
            return None
        from .. import hip

        kind, mod, x = source
        x = deferred.resolve(x)
        if self.role == "shortcut":
            gb_mod, out_gather = self.scatter, self.gather
        else:
            gb_mod, out_gather = self.scatter_gather, self.scatter_gather.gather.module
        gb_full = gb_mod.original_outputs[gb_mod.cache_id]
        seg_sg = self._map_owner()
        if seg_sg is None:
            return None
        if kind == "gather":
            x_full, x_tiles, map_x = x, None, None
            tgt = mod
        else:
            x_full, x_tiles = mod.original_outputs[mod.cache_id], x
            map_x = mod._map_on(x.device)
            tgt = mod.gather.module
        if not (_cl_gpu(x_full) and _cl_gpu(gb_full)) or tuple(tgt.block_size) != tuple(out_gather.block_size):
            return None
        gb = self.mlp_gamma_beta(actv_tiles)  # [N, 2C, 4, 4] on the label branch's tile list
        gb = deferred.resolve(gb)
        if not hip.is_cl(gb):
            gb = gb.contiguous(memory_format=torch.channels_last)
        sc, sh = self.affine4()
        return hip.spade_modulate_cl(x_full, x_tiles, map_x, sc.contiguous(), sh.contiguous(), gb, gb_full,
                                     seg_sg._map_on(x_full.device), out_gather.indices_on(x_full.device), out_gather.block_size, slope)

    def _map_owner(self) -> Optional[ScatterGather]:
        """The ScatterGather whose scatter map describes where the label branch's conv tiles lie (set by the block)."""
        return getattr(self, "_seg_map_owner", None)


class SpadeResBlock(SIGEModule):
    def __init__(self, cfg: SPADEConfig, fin: int, fout: int, tiled: bool):
        super().__init__()
        self.cfg = cfg
        self.fin, self.fout = fin, fout
        self.nhidden = cfg.ngf * 2
        self.learned_shortcut = fin != fout
        fmid = min(fin, fout)
        self.tiled = tiled and cfg.main_block_size is not None
        self.tiled_shortcut = self.learned_shortcut and self.tiled and cfg.shortcut_block_size is not None
        Conv = SIGEConv2d if self.tiled else nn.Conv2d
        parts = 3 if self.learned_shortcut else 2
        self.mlp_shared = nn.Sequential(Conv(cfg.semantic_nc, self.nhidden * parts, 3, padding=1), nn.ReLU())
        self.conv_0 = Conv(fin, fmid, 3, padding=1)
        self.conv_1 = Conv(fmid, fout, 3, padding=1)
        seg_gather = None
        if self.tiled:
            self.seg_gather = Gather(self.mlp_shared[0], cfg.main_block_size)
            self.seg_scatter_gather = ScatterGather(self.seg_gather)
            self.main_gather = Gather(self.conv_0, cfg.main_block_size)
            self.main_scatter_gather = ScatterGather(self.main_gather)
            seg_gather = self.seg_gather
        if self.learned_shortcut:
            self.conv_s = (SIGEConv2d if self.tiled_shortcut else nn.Conv2d)(fin, fout, 1, bias=False)
            if self.tiled_shortcut:
                self.shortcut_gather = Gather(self.conv_s, cfg.shortcut_block_size)
                self.scatter = ScatterWithBlockResidual(self.main_gather, self.shortcut_gather)
        if self.tiled and not self.tiled_shortcut:
            self.scatter = Scatter(self.main_gather)
        self.norm_0 = SpadeNorm(cfg, fin, seg_gather)
        self.norm_1 = SpadeNorm(cfg, fmid, seg_gather)
        if self.learned_shortcut:
            self.norm_s = SpadeNorm(cfg, fin, seg_gather, shortcut_conv=self.conv_s)
        if self.tiled:
            for n in (self.norm_0, self.norm_1, getattr(self, "norm_s", None)):
                if n is not None:
                    object.__setattr__(n, "_seg_map_owner", self.seg_scatter_gather)  # (not a child module: no double registration)

    def _lrelu(self, x):
        return F.leaky_relu(x, self.cfg.leaky_slope)

    def _label_features(self, seg, res):
        """ReLU(conv3x3(label map at this resolution)), split into one part per SPADE layer of the block."""
        seg = _resize(seg, res)
        fused = self.cfg.fused and self.mode == "sparse" and _cl_gpu(seg)
        if fused:
            parts = self._label_features_fused(seg)
            if parts is not None:
                return parts
        if self.tiled:
            seg = self.seg_gather(seg)
        if not self.tiled and self.mode == "sparse":
            a = F.relu(_dense_conv(self.mlp_shared[0], seg))
        else:
            a = self.mlp_shared(seg)
        if self.tiled:
            a = deferred.resolve(self.seg_scatter_gather(a))  # (a split is not a conv: the tiles are needed as a tensor)
        parts = torch.split(a, self.nhidden, dim=1)
        if self.tiled and self.mode == "sparse" and _cl_gpu(a):
            # channel slices of channels-last tiles are strided: give every consumer conv its own dense channels-last slab
            parts = tuple(p.contiguous(memory_format=torch.channels_last) for p in parts)
        return parts

    def _label_features_fused(self, seg):
        """The label branch of a sparse forward without a torch kernel: conv (tiles or dense), then ONE pass that re-tiles
        (ScatterGather), applies the ReLU and writes one dense slab per SPADE layer of the block (csrc/spade_ops.hip) -- in place
        of ReLU, ScatterGather, torch.split and one copy per part."""
        from .. import hip

        n_parts = 3 if self.learned_shortcut else 2
        if not self.tiled:
            a = _dense_conv(self.mlp_shared[0], seg)
            return hip.act_split_cl(a, n_parts, "relu") if _cl_gpu(a) else None
        sg = self.seg_scatter_gather
        if sg.sparse_update or sg.mode != "sparse":
            return None
        cached = sg.original_outputs[sg.cache_id]
        if not (_cl_gpu(cached) and cached.shape[1] % (4 * n_parts) == 0):
            return None
        t = deferred.resolve(self.mlp_shared[0](self.seg_gather(seg)))
        if not hip.is_cl(t):
            t = t.contiguous(memory_format=torch.channels_last)
        g = sg.gather.module
        return hip.scatter_gather_split_cl(t, cached, g.block_size[0], g.block_size[1], g.indices_on(t.device), sg._map_on(t.device),
                                           n_parts, "relu")

    def forward(self, x, seg):
        if self.mode == "full":
            return self._full(x, seg)
        if self.mode in ("sparse", "profile"):
            return self._sparse(x, seg)
        raise NotImplementedError("Unknown mode [%s]!!!" % self.mode)

    def _full(self, x, seg):
        a = self._label_features(seg, x.shape[2:])
        if self.learned_shortcut:
            xs = self.shortcut_gather(x) if self.tiled_shortcut else x
            xs = self.conv_s(self.norm_s(xs, a[2]))
        else:
            xs = x
        dx = self.main_gather(x) if self.tiled else x
        dx = self.conv_0(self._lrelu(self.norm_0(dx, a[0])))
        if self.tiled:
            dx = self.main_scatter_gather(dx)
        dx = self.conv_1(self._lrelu(self.norm_1(dx, a[1])))
        return self.scatter(dx, xs) if self.tiled else xs + dx

    def _sparse(self, x, seg):
        a = self._label_features(seg, x.shape[2:])
        fused = self.cfg.fused and self.tiled and self.mode == "sparse"
        slope = self.cfg.leaky_slope
        # shortcut branch
        if self.learned_shortcut:
            t = self.norm_s.modulated_tiles(("gather", self.shortcut_gather, x), a[2], None) if (fused and self.tiled_shortcut) else None
            if t is None and self.cfg.fused and not self.tiled and self.mode == "sparse":
                t = self.norm_s.modulated_dense(x, a[2], None)
            if t is None:
                xs = self.shortcut_gather(x, *self.norm_s.affine4()) if self.tiled_shortcut else self.norm_s.param_free_norm(x)
                t = self.norm_s(xs, a[2])
            xs = self.conv_s(t)
        else:
            xs = x
        # main branch
        dense_fused = self.cfg.fused and not self.tiled and self.mode == "sparse"
        t = self.norm_0.modulated_tiles(("gather", self.main_gather, x), a[0], slope) if fused else (
            self.norm_0.modulated_dense(x, a[0], slope) if dense_fused else None)
        if t is None:
            dx = self.main_gather(x, *self.norm_0.affine4()) if self.tiled else self.norm_0.param_free_norm(x)
            t = self._lrelu(self.norm_0(dx, a[0]))
        dx = self.conv_0(t) if self.tiled else _dense_conv(self.conv_0, t)
        t = self.norm_1.modulated_tiles(("scatter_gather", self.main_scatter_gather, dx), a[1], slope) if fused else (
            self.norm_1.modulated_dense(dx, a[1], slope) if dense_fused else None)
        if t is None:
            dx = self.main_scatter_gather(dx, *self.norm_1.affine4()) if self.tiled else self.norm_1.param_free_norm(dx)
            t = self._lrelu(self.norm_1(dx, a[1]))
        if self.tiled:
            return self.scatter(self.conv_1(t), xs)
        return _dense_conv(self.conv_1, t, residual=xs)  # (xs + dx inside the conv's epilogue)


class SpadeGenerator(SIGEModel):
    def __init__(self, cfg: SPADEConfig = SPADEConfig()):
        super().__init__()
        self.cfg = cfg
        nf = cfg.ngf
        ups = {"normal": 5, "more": 6, "most": 7}[cfg.num_upsampling_layers]
        self.sw = cfg.crop_size // (2 ** ups)
        self.sh = round(self.sw / cfg.aspect_ratio)
        most = cfg.num_upsampling_layers == "most"
        k = cfg.num_sparse_layers
        self.fc = nn.Conv2d(cfg.semantic_nc, 16 * nf, 3, padding=1)
        # (name, fin, fout, tiled) from the lowest resolution up; the last `num_sparse_layers` blocks are tiled
        self.head_0 = SpadeResBlock(cfg, 16 * nf, 16 * nf, k >= 7 + most)
        self.G_middle_0 = SpadeResBlock(cfg, 16 * nf, 16 * nf, k >= 6 + most)
        self.G_middle_1 = SpadeResBlock(cfg, 16 * nf, 16 * nf, k >= 5 + most)
        self.up_0 = SpadeResBlock(cfg, 16 * nf, 8 * nf, k >= 4 + most)
        self.up_1 = SpadeResBlock(cfg, 8 * nf, 4 * nf, k >= 3 + most)
        self.up_2 = SpadeResBlock(cfg, 4 * nf, 2 * nf, k >= 2 + most)
        self.up_3 = SpadeResBlock(cfg, 2 * nf, nf, k >= 1 + most)
        final = nf
        if most:
            self.up_4 = SpadeResBlock(cfg, nf, nf // 2, k >= 1)
            final = nf // 2
        self.conv_img = nn.Conv2d(final, 3, 3, padding=1)
        self.edit_batch = 1  # (sige_amd.stacked.edit_batch: E edits of one original stacked along H in every sparse-mode tensor)

    def _up(self, x):
        if self.cfg.fused and self.mode == "sparse" and _cl_gpu(x):
            return _resize(x, (2 * x.shape[2], 2 * x.shape[3]))
        return F.interpolate(x, scale_factor=2.0, mode="nearest")

    def _tail(self, x):
        cfg = self.cfg
        if self.edit_batch > 1:
            from ..stacked import untall

            x = untall(x, self.edit_batch)  # (the image conv is a whole-image op: batch E on the same bytes)
        if cfg.fused and self.mode == "sparse" and _cl_gpu(x):
            from .. import hip
            from ..nn.dense import _plain_weight

            # (a channels-last model holds the weight permuted: the dense copy is made once per weight version, not per forward)
            out = hip.conv3x3_small_cout_act_cl(x, _plain_weight(self.conv_img), self.conv_img.bias, "leaky", cfg.leaky_slope, "tanh")
            if out is not None:
                return out
        return torch.tanh(self.conv_img(F.leaky_relu(x, cfg.leaky_slope)))

    def forward(self, seg: torch.Tensor) -> torch.Tensor:
        global _RESIZE_MEMO
        _RESIZE_MEMO = {}
        try:
            return self._forward(seg)
        finally:
            _RESIZE_MEMO = None

    def _forward(self, seg: torch.Tensor) -> torch.Tensor:
        cfg = self.cfg
        E = self.edit_batch
        if E > 1:
            # stacked edits (sige_amd/stacked.py): seg is [E,36,H,W], one edited label map per edit; from here to the image conv
            # every tensor is the tall image [1,C,E*h,w] -- the same bytes.  Every op of the all-library sparse forward is either
            # per pixel (nearest resize by an integer factor, ReLU + split, the dense SPADE modulation), seam-aware (the tile and
            # dense-layer convs, the SPADE modulation of tiles, scatter_gather_split) or tile-local (convs over tile slabs)
            if not (cfg.fused and self.mode == "sparse" and _cl_gpu(seg) and seg.shape[0] == E):
                raise RuntimeError("stacked edits: the fused sparse forward on channels-last GPU label maps, one per edit")
            from ..stacked import tall

            seg = tall(seg)
        if cfg.fused and self.mode == "sparse" and _cl_gpu(seg):
            x = _dense_conv(self.fc, _resize(seg, (E * self.sh, self.sw)))
        else:
            x = self.fc(F.interpolate(seg, size=(self.sh, self.sw)))
        x = self._up(self.head_0(x, seg))
        x = self.G_middle_0(x, seg)
        if cfg.num_upsampling_layers in ("more", "most"):
            x = self._up(x)
        x = self._up(self.G_middle_1(x, seg))
        x = self._up(self.up_0(x, seg))
        x = self._up(self.up_1(x, seg))
        x = self._up(self.up_2(x, seg))
        x = self.up_3(x, seg)
        if cfg.num_upsampling_layers == "most":
            x = self.up_4(self._up(x), seg)
        return self._tail(x)
