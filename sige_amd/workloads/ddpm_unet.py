"""DDPM 256x256 U-Net workload (LSUN-Church configuration) built on sige_amd.nn.

This is the benchmark harness for BASELINE.json configs[1]/[4]: the network the
reference's headline numbers are quoted on (diffusion/configs/
church_ddpm256-sige.yml: ch 128, mult 1,1,2,2,4,4, 2 res-blocks per level,
attention at 16x16, SIGE tiles 6/4, sparse at resolutions >= 64).  It follows
the public DDPM checkpoint layout (parameter names `down.{l}.block.{i}.conv1`,
`mid.block_1`, `up.{l}.upsample.conv`, `temb.dense.{k}`, ...), so a state dict of
the reference's SIGEFusedUNet (diffusion/models/ddpm_arch/sige_fused_unet.py:
251-434) loads into it -- tests/test_reference_models.py checks that both
produce the same full and sparse outputs.

Sparse-mode contract (what SIGE caches):
  * GroupNorm statistics are NOT recomputed: the full pass stores per-channel
    (scale, shift) = (gamma/sigma, beta - mu*gamma/sigma) of the original image,
    with the timestep-embedding add folded into the second shift; the sparse
    pass feeds them to Gather / ScatterGather as the fused affine + SiLU.
  * resolutions below `sparse_threshold` run dense convs on the cached affine.
"""
import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import nn
from torch.nn import functional as F

from ..nn import Gather, Scatter, ScatterGather, ScatterWithBlockResidual, SIGEConv2d, SIGEModel, SIGEModule, paired_convs
from ..nn.deferred import lazy_cat
from ..nn.dense import fast_full_pass, full_conv2d, fused_conv2d, group_norm_affine, input_conv2d


@dataclass
class DDPMConfig:
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 1, 2, 2, 4, 4)
    num_res_blocks: int = 2
    attn_resolutions: Tuple[int, ...] = (16,)
    in_ch: int = 3
    out_ch: int = 3
    resolution: int = 256
    main_block: Optional[int] = 6       # tile edge for 3x3 convs
    shortcut_block: Optional[int] = 4   # tile edge for 1x1 convs
    sparse_threshold: int = 64
    groups: int = 32
    eps: float = 1e-6
    # The reference's SIGEFusedAttnBlock stores its cached (scale, shift) un-keyed
    # (sige_fused_unet.py:170) and then indexes them with cache_id, so its sparse
    # pass normalises every channel with channel 0's statistics.  True reproduces
    # that (bit-for-bit model-level parity tests); False is the intended math.
    reference_attn_quirk: bool = False
    # conv1's epilogue applies the cached affine-2 + SiLU (and the ScatterGather cache keeps an activated copy),
    # so conv2 stages raw values: the activation is computed once per element, not once per output-channel block
    preactivate: bool = True
    # launch the shortcut 1x1 of a ResBlock inside the kernel of its conv1 (sige_amd.hip.conv_pair: horizontal fusion)
    pair_shortcut: bool = True
    # conv1 inputs activated by their PRODUCERS: every module whose output feeds a ResBlock's conv1 also writes
    # SiLU(scale1 * out + shift1) (an "activated twin", sige_amd.nn.scatter._TwinBuffers / dense.fused_conv2d(twins=...)), so
    # conv1 stages raw values -- the cached affine + SiLU is computed once per element instead of once per
    # output-channel block (8-16 per layer).  Consumers register on their first sparse forward and activate for
    # themselves until the twin exists.  (WIP: written without GPU time left in round 2 -- validate first:
    # tests/test_gpu_round2.py::test_ddpm_forward_twins_vs_no_twins, then bench.py; measured upper bound -76 us / forward.)
    conv1_twins: bool = True
    # run the shortcut branch of a ResBlock (1x1 conv on the block input) on a second stream (it is independent of
    # conv1 -> conv2).  Measured on MI355X / ROCm 7.2: the cross-stream graph edges cost more than the overlap gains
    # (2.05 ms vs 1.82 ms per forward), so it is off by default.
    overlap_shortcut: bool = False


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32, device=t.device) * (-math.log(10000) / (half - 1)))
    arg = t.float()[:, None] * freq[None, :]
    emb = torch.cat([arg.sin(), arg.cos()], dim=1)
    return F.pad(emb, (0, 1)) if dim % 2 else emb


def norm_affine(x: torch.Tensor, norm: nn.GroupNorm, fast: bool = False, cbias: Optional[torch.Tensor] = None,
                x2: Optional[torch.Tensor] = None):
    """Per-channel (scale, shift) with GroupNorm(x) == x * scale + shift, batch 1.  `fast` (the full pass on the fp16 matrix
    cores): the library's split reduction (fp64 combine) instead of torch's var_mean + five elementwise kernels.
    `cbias` [C]: the statistics are those of x + cbias (the timestep embedding); with `fast` the returned affine is for x itself
    (GroupNorm(x + cbias) == x * scale + shift), otherwise for x + cbias as before -- the caller folds the bias."""
    if fast and x.is_cuda:
        assert x.shape[0] == 1, "SIGE caches one original image"
        # (`x2`: the norm of torch.cat([x, x2], 1); tensors that carry their producer's per-channel statistics are not read)
        sc, sh = group_norm_affine(x, norm, cbias, x2=x2, make_stats=True)
        return sc.reshape(-1), sh.reshape(-1)
    if x2 is not None:
        x = torch.cat([x, x2], dim=1)
    n, c, h, w = x.shape
    assert n == 1, "SIGE caches one original image"
    if cbias is not None:
        x = x + cbias.reshape(1, -1, 1, 1)
    g = norm.num_groups
    var, mean = torch.var_mean(x.reshape(g, -1), dim=1, unbiased=False)
    inv = torch.rsqrt(var + norm.eps).repeat_interleave(c // g)
    mu = mean.repeat_interleave(c // g)
    scale = inv * norm.weight
    shift = norm.bias - mu * scale
    return scale, shift


def _as4(v: torch.Tensor) -> torch.Tensor:
    return v.reshape(1, -1, 1, 1)


class _TwinProducer:
    """Mixin of the modules whose output is written by a full-tensor conv epilogue and can therefore carry activated twins
    for the conv1 of a consumer (cfg.conv1_twins).  `_twin_scatter()`: the Scatter module that owns the persistent in-place
    output (SIGE layers: persistent twins live next to it), or None (dense layers: a fresh twin per forward)."""

    def _twin_scatter(self):
        return None

    def _twins_ok(self) -> bool:
        return True

    def register_twin(self, key, scale: torch.Tensor, shift: torch.Tensor) -> bool:
        if not self._twins_ok():
            return False
        sct = self._twin_scatter()
        if sct is not None:
            return sct.twins.register(key, scale, shift)
        from ..nn.scatter import twin_key_cache_id, twins_for

        regs = self.__dict__.setdefault("twin_regs", {})
        if key not in regs and len(twins_for(regs, twin_key_cache_id(key))) >= 2:
            return False
        regs[key] = (scale, shift)
        return True

    def unregister_twin(self, key):
        sct = self._twin_scatter()
        if sct is not None:
            sct.twins.unregister(key)
        self.__dict__.setdefault("twin_regs", {}).pop(key, None)

    def _my_twins(self) -> dict:
        """(dense producers) the registrations this forward serves: those of the current cache id."""
        from ..nn.scatter import twins_for

        return twins_for(self.__dict__.get("twin_regs") or {}, getattr(self, "cache_id", 0))

    def _produced(self, out: torch.Tensor) -> torch.Tensor:
        """Mark `out` as this module's output of the current sparse forward (consumers find their producer through it)."""
        out._sige_producer = self
        if not hasattr(out, "_sige_twins"):
            out._sige_twins = {}
        return out


class ResBlock(SIGEModule, _TwinProducer):
    def __init__(self, cfg: DDPMConfig, cin: int, cout: int, sparse: bool):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.sparse_main = sparse and cfg.main_block is not None
        Conv = SIGEConv2d if self.sparse_main else nn.Conv2d
        self.norm1 = nn.GroupNorm(cfg.groups, cin, eps=cfg.eps)
        self.conv1 = Conv(cin, cout, 3, 1, 1)
        self.norm2 = nn.GroupNorm(cfg.groups, cout, eps=cfg.eps)
        self.conv2 = Conv(cout, cout, 3, 1, 1)
        self.sparse_shortcut = False
        if self.sparse_main:
            self.main_gather = Gather(self.conv1, cfg.main_block, activation_name="swish")
            self.scatter_gather = ScatterGather(self.main_gather, activation_name="swish")
        if cin != cout:
            self.sparse_shortcut = self.sparse_main and cfg.shortcut_block is not None
            self.nin_shortcut = (SIGEConv2d if self.sparse_shortcut else nn.Conv2d)(cin, cout, 1, 1, 0)
            if self.sparse_shortcut:
                self.shortcut_gather = Gather(self.nin_shortcut, cfg.shortcut_block)
                self.scatter = ScatterWithBlockResidual(self.main_gather, self.shortcut_gather)
        if self.sparse_main and not self.sparse_shortcut:
            self.scatter = Scatter(self.main_gather)
        self.affine = {}  # cache_id -> (scale1, shift1, scale2, shift2) as [1,C,1,1]
        self.plain = False
        self.preactivate = cfg.preactivate
        self.overlap = cfg.overlap_shortcut
        self.pair = cfg.pair_shortcut
        self.use_twins = cfg.conv1_twins
        self.twin_regs = {}     # as a producer (dense blocks): consumer key -> (scale, shift)
        self._twin_links = {}   # as a consumer: key -> the producer it registered with
        self._aff_gen = 0       # bumped when the cached affine is recomputed: old twins are for the old affine
        self._side = None

    def clear_cache(self):
        self.affine = {}
        self._drop_twin_links()

    # ---- activated twins (cfg.conv1_twins) ----
    def _twin_scatter(self):
        return self.scatter if self.sparse_main else None

    def _drop_twin_links(self):
        for key, (prod, _) in self._twin_links.items():
            prod.unregister_twin(key)
        self._twin_links = {}
        self._aff_gen += 1

    def _twin_inputs(self, parts, s1, t1):
        """The activated twins of this block's conv1 inputs, one per part of the (possibly concatenated) input -- or None
        (then conv1 activates in its staging path, as without twins).  A missing twin is requested from the module that
        produced the part; it exists from the next forward on."""
        from ..nn import scatter as _scatter

        if not (self.use_twins and self.preactivate and self.mode == "sparse" and (parts[0].is_cuda or _scatter.EMULATE_TWINS)):
            return None
        if s1.shape[0] != 1:
            return None  # (per-sample affine: the epilogue's twin vectors are per channel)
        # A registration holds VIEWS of the affine tensors it was made with.  If the cached affine has been re-pointed since
        # (sige_amd.parallel.pack_caches moves every cache tensor into one flat buffer; anything that assigns `affine[cid]`
        # without going through _full), the producers would keep activating with the old tensors: drop every link and start
        # over with the current ones (sparse -> pack_caches -> broadcast gave stale twins, max error 0.18).
        anchor = (s1.data_ptr(), t1.data_ptr())
        if any(a != anchor for k, (_, a) in self._twin_links.items() if k[3] == self.cache_id):
            self._drop_twin_links()
        twins, off, complete = [], 0, True
        for i, p in enumerate(parts):
            c = p.shape[1]
            key = (id(self), i, self._aff_gen, self.cache_id)  # (the registered affine is this cache id's)
            t = getattr(p, "_sige_twins", {}).get(key)
            if t is None:
                complete = False
                prod = getattr(p, "_sige_producer", None)
                if prod is not None and key not in self._twin_links:
                    sc = s1.reshape(-1)[off:off + c].contiguous()
                    sh = t1.reshape(-1)[off:off + c].contiguous()
                    if prod.register_twin(key, sc, sh):
                        self._twin_links[key] = (prod, anchor)
            twins.append(t)
            off += c
        return twins if complete else None

    def rebuild_derived_caches(self):
        """(sige_amd.parallel) the cache tensors were rewritten in place: recompute the activated ScatterGather copy."""
        if self.sparse_main and self.preactivate:
            for cid, (_, _, s2, t2) in self.affine.items():
                if cid in self.scatter_gather.original_outputs:
                    keep, self.scatter_gather.cache_id = self.scatter_gather.cache_id, cid
                    self.scatter_gather.cache_activated(s2, t2)
                    self.scatter_gather.cache_id = keep

    def _shortcut_async(self, fn, x_ready: torch.Tensor):
        """Run `fn()` (the shortcut branch) on a side stream forked from the current one; returns (result, join).
        `join()` must be called on the current stream before the result is consumed.  hipGraph capture records the
        fork / join as graph edges, so the two branches run concurrently in a replay."""
        if not (self.overlap and x_ready.is_cuda and self.mode == "sparse"):
            return fn(), (lambda: None)
        cur = torch.cuda.current_stream(x_ready.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=x_ready.device)
        side = self._side
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            out = fn()

        def join():
            cur.wait_stream(side)
            out.record_stream(cur)  # (allocated on the side stream, consumed on the current one)

        return out, join

    def _pairing(self, t: torch.Tensor):
        """Context for [shortcut conv, conv1] of a sparse-mode block: on the GPU the two share one launch."""
        return paired_convs(t, enabled=self.pair and self.cin != self.cout and self.mode == "sparse")

    def forward(self, x, temb: Optional[torch.Tensor]) -> torch.Tensor:
        """`x` may be a pair (h, skip): the up path's torch.cat, which the dense
        sparse-mode blocks fold into their convs' two-pointer input."""
        pair = x if isinstance(x, (tuple, list)) else None
        if self.mode == "full":
            if pair and not self.plain and self._fast_full() and pair[0].is_cuda:
                return self._full(pair[0], temb, x2=pair[1])  # (the cat never exists: two-pointer convs, statistics per part)
            return self._full(torch.cat(pair, dim=1) if pair else x, temb)
        if self.mode in ("sparse", "profile"):
            if pair and not self.sparse_main and self.mode == "sparse":
                return self._sparse_dense(pair[0], pair[1])
            if pair and self.mode == "sparse" and self.cin != self.cout:
                return self._sparse(lazy_cat(pair[0], pair[1]))  # consumed by the block's two Gathers only
            return self._sparse(torch.cat(pair, dim=1) if pair else x)
        raise NotImplementedError("Unknown mode [%s]!!!" % self.mode)

    def _fast_full(self) -> bool:
        """The full pass on the library's kernels (dense.full_conv2d) rather than torch's: compute dtype "f16" / "f16x3", or
        dense.FULL_PASS_F32_NATIVE for exact fp32."""
        from ..nn import dense as _dense

        return _dense.fast_full_pass(self.conv1)

    def _shortcut(self, x, x2=None):
        if self.cin == self.cout:
            return x if x2 is None else torch.cat([x, x2], dim=1)  # (identity shortcut of a concatenated input: both halves)
        if x2 is not None:  # (full mode, fast: the cat is read through two pointers)
            if self.sparse_shortcut:
                self.shortcut_gather.note_full_input(x.shape[2:])
            return full_conv2d(self.nin_shortcut, x, x2=x2)
        if self.sparse_shortcut:
            x = self.shortcut_gather(x)
        return full_conv2d(self.nin_shortcut, x) if self.mode == "full" else self.nin_shortcut(x)

    def _plain(self, x, temb):
        """Dense forward with stock GroupNorm and no caching (the "original model"
        baseline the speedup is quoted against)."""
        skip = x if self.cin == self.cout else self.nin_shortcut(x)
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h + temb.reshape(1, -1, 1, 1))))
        return h + skip

    def _full(self, x, temb, x2=None):
        if self.plain:
            return self._plain(x, temb)
        # (the full pass on the library's kernels -- dense.full_conv2d -- also takes the library's GroupNorm statistics; its
        #  convs leave the per-channel sums of their outputs, so that a norm never reads a tensor its producer just wrote)
        fast = self._fast_full()
        skip = self._shortcut(x, x2)
        if x2 is not None:
            if self.sparse_main:
                self.main_gather.note_full_input(x.shape[2:])
            h = x
        else:
            h = self.main_gather(x) if self.sparse_main else x  # records the input resolution
        s1, t1 = norm_affine(h, self.norm1, fast, x2=x2)
        h = full_conv2d(self.conv1, h, _as4(s1), _as4(t1), "swish", x2=x2, stats=fast)  # conv1(silu(cat(h, x2) * s1 + t1))
        if self.sparse_main:
            h = self.scatter_gather(h)
        te = temb.reshape(-1)
        if fast and h.is_cuda:
            # statistics of h + temb and the embedding folded into the shift in ONE launch (no h + temb tensor)
            s2, t2 = norm_affine(h, self.norm2, True, cbias=te.contiguous())
        else:
            s2, t2 = norm_affine(h + _as4(te), self.norm2, fast)
            t2 = t2 + te * s2  # fold the timestep-embedding add into the cached shift
        self._drop_twin_links()  # (twins written for the previous affine are stale)
        self.affine[self.cache_id] = tuple(_as4(v).contiguous() for v in (s1, t1, s2, t2))
        if self.sparse_main and self.preactivate:
            self.scatter_gather.cache_activated(_as4(s2), _as4(t2))
        # conv2(silu(h * s2 + t2)) + skip in one launch: a plain Scatter caches the SUM (sige/nn/scatter.py:31-37), and so does
        # ScatterWithBlockResidual (sige/nn/scatter.py:89-93), next to the shortcut
        if self.sparse_main and self.sparse_shortcut:
            if fast and h.is_cuda:
                h = full_conv2d(self.conv2, h, _as4(s2), _as4(t2), "swish", residual=skip, stats=True)
                return self.scatter(h, skip, x_is_sum=True)
            h = full_conv2d(self.conv2, h, _as4(s2), _as4(t2), "swish")  # conv2(silu(h * s2 + t2))
            return self.scatter(h, skip)
        h = full_conv2d(self.conv2, h, _as4(s2), _as4(t2), "swish", residual=skip, stats=fast)
        return self.scatter(h) if self.sparse_main else h

    def _sparse(self, x):
        s1, t1, s2, t2 = self.affine[self.cache_id]
        if self.sparse_main and self.preactivate and self.mode == "sparse":
            parts = list(x.parts) if hasattr(x, "parts") else [x]
            first = parts[0]
            tw = self._twin_inputs(parts, s1, t1)
            with self._pairing(first):
                skip, join = (self._shortcut_async(lambda: self._shortcut(x), first) if self.cin != self.cout
                              else (x, lambda: None))
                if tw is not None:  # the producers wrote SiLU(s1 * x + t1): conv1 stages raw values
                    xa = tw[0] if len(tw) == 1 else lazy_cat(tw[0], tw[1])
                    h = self.conv1(self.main_gather(xa, preactivated=True), out_affine=(s2, t2, "swish"))
                else:
                    h = self.conv1(self.main_gather(x, s1, t1), out_affine=(s2, t2, "swish"))
            tiles = self.scatter_gather(h, preactivated=True)
            join()
            return self._produced(self.scatter.forward_fused(self.conv2, tiles, skip))
        if not self.sparse_main:
            return self._sparse_dense(x, None)
        skip = self._shortcut(x)
        h = self.conv1(self.main_gather(x, s1, t1))
        if self.mode == "sparse":
            return self.scatter.forward_fused(self.conv2, self.scatter_gather(h, s2, t2), skip)
        return self.scatter(self.conv2(self.scatter_gather(h, s2, t2)), skip)

    def _sparse_dense(self, x, x2):
        """Dense block on the cached affine: 2-3 fused launches (shortcut 1x1, conv1, conv2+skip)."""
        s1, t1, s2, t2 = self.affine[self.cache_id]
        join = lambda: None  # noqa: E731
        if self.preactivate:
            tw = self._twin_inputs([x] if x2 is None else [x, x2], s1, t1)
            with self._pairing(x):
                if self.cin == self.cout:
                    skip = x if x2 is None else torch.cat([x, x2], dim=1)
                else:
                    skip, join = self._shortcut_async(lambda: fused_conv2d(self.nin_shortcut, x, x2=x2), x)
                if tw is not None:  # the producers wrote SiLU(s1 * x + t1): conv1 stages raw values
                    h = fused_conv2d(self.conv1, tw[0], None, None, "identity", x2=(tw[1] if len(tw) > 1 else None),
                                     out_affine=(s2, t2, "swish"))
                else:
                    h = fused_conv2d(self.conv1, x, s1, t1, "swish", x2=x2, out_affine=(s2, t2, "swish"))
            join()
            return self._produced(fused_conv2d(self.conv2, h, residual=skip, twins=self._my_twins()))
        if self.cin == self.cout:
            skip = x if x2 is None else torch.cat([x, x2], dim=1)
        else:
            skip, join = self._shortcut_async(lambda: fused_conv2d(self.nin_shortcut, x, x2=x2), x)
        join()
        h = fused_conv2d(self.conv1, x, s1, t1, "swish", x2=x2)
        return fused_conv2d(self.conv2, h, s2, t2, "swish", residual=skip)


class AttnBlock(SIGEModule, _TwinProducer):
    def __init__(self, cfg: DDPMConfig, ch: int, sparse: bool):
        super().__init__()
        self.ch = ch
        self.edit_batch = 1  # (sige_amd.stacked: E images stacked along H; attention is per image)
        self.twin_regs = {}
        self.quirk = cfg.reference_attn_quirk
        self.sparse = sparse and cfg.shortcut_block is not None
        Conv = SIGEConv2d if self.sparse else nn.Conv2d
        self.norm = nn.GroupNorm(cfg.groups, ch, eps=cfg.eps)
        self.qkv = Conv(ch, 3 * ch, 1, 1, 0)
        self.proj_out = Conv(ch, ch, 1, 1, 0)
        if self.sparse:
            self.gather1 = Gather(self.qkv, cfg.shortcut_block)
            self.scatter1 = Scatter(self.gather1)
            self.gather2 = Gather(self.proj_out, cfg.shortcut_block)
            self.scatter2 = Scatter(self.gather2)
        self.affine = {}
        self.plain = False

    def clear_cache(self):
        self.affine = {}

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.mode == "full" and self.plain:
            h = self.norm(x)
        elif self.mode == "full":
            h = self.gather1(x) if self.sparse else x
            s, t = norm_affine(h, self.norm, fast_full_pass(self.qkv))
            self.affine[self.cache_id] = (_as4(s).contiguous(), _as4(t).contiguous())
            if not self.sparse:  # dense block: qkv(h * s + t) in one launch where the conv's compute dtype allows, then the rest
                qkv = full_conv2d(self.qkv, h, _as4(s), _as4(t), "identity")
                return full_conv2d(self.proj_out, self._attention(qkv), residual=x)
            h = h * _as4(s) + _as4(t)
        else:
            s, t = self.affine[self.cache_id]
            if self.quirk:
                s, t = s[:, :1].expand_as(s).contiguous(), t[:, :1].expand_as(t).contiguous()
            if not self.sparse and self.mode == "sparse":
                return self._dense_sparse(x, s, t)
            h = self.gather1(x, s, t) if self.sparse else x * s + t
        plain = self.mode == "full" and self.plain
        qkv = self.qkv(h)
        if self.sparse and not plain:
            qkv = self.scatter1(qkv)
        b, _, hh, ww = qkv.shape
        q, k, v = qkv.reshape(b, 3, self.ch, hh * ww).unbind(1)
        attn = torch.softmax(torch.bmm(q.transpose(1, 2), k) * (self.ch ** -0.5), dim=2)  # [b, hw(q), hw(k)]
        h = torch.bmm(v, attn.transpose(1, 2)).reshape(b, self.ch, hh, ww)
        if self.sparse and not plain:
            h = self.gather2(h)
        h = self.proj_out(h)
        return self.scatter2(h, x) if (self.sparse and not plain) else h + x


    def _attention(self, qkv):
        if self.edit_batch > 1 and qkv.shape[0] == 1:
            # stacked edits: the tall image is E images -- tokens attend within their own image (the same bytes seen as batch E)
            from ..stacked import tall, untall

            return tall(self._attention(untall(qkv, self.edit_batch)))
        b, _, hh, ww = qkv.shape
        if qkv.is_cuda and qkv.dtype == torch.float32:
            from .. import hip

            if hip.is_cl(qkv):
                out = hip.attention_cl(qkv, self.ch ** -0.5)
                if out is not None:
                    return out
            if hip.attention_supported(self.ch, hh * ww):
                return hip.attention(qkv, self.ch ** -0.5)
        q, k, v = qkv.reshape(b, 3, self.ch, hh * ww).unbind(1)
        attn = torch.softmax(torch.bmm(q.transpose(1, 2), k) * (self.ch ** -0.5), dim=2)
        return torch.bmm(v, attn.transpose(1, 2)).reshape(b, self.ch, hh, ww)

    def _twins_ok(self) -> bool:
        return not self.sparse  # (the tiled form ends in a plain Scatter with a residual: not wired for twins)

    def _dense_sparse(self, x, s, t):
        qkv = fused_conv2d(self.qkv, x, s, t, "identity")
        return self._produced(fused_conv2d(self.proj_out, self._attention(qkv), residual=x, twins=self._my_twins()))


class Upsample(SIGEModule, _TwinProducer):
    """nearest x2 then a tiled 3x3 conv (always tiled, as in the reference)."""

    def _twin_scatter(self):
        return self.scatter

    def __init__(self, cfg: DDPMConfig, ch: int):
        super().__init__()
        self.conv = SIGEConv2d(ch, ch, 3, 1, 1)
        self.gather = Gather(self.conv, cfg.main_block)
        self.scatter = Scatter(self.gather)
        self.plain = False

    def forward(self, x):
        if self.mode == "sparse" and self.gather.fuses_upsample(x):
            # the upsampled tensor only feeds the gather: read the half-resolution one at (h/2, w/2) instead
            return self._produced(self.scatter.forward_fused(self.conv, self.gather(x, upsample2x=True)))
        if self.mode == "full" and not self.plain and x.is_cuda:
            from ..nn import dense as _dense

            if _dense.fast_full_pass(self.conv):
                # the full pass on the library's kernels: the conv reads the half-resolution tensor at (h/2, w/2) as well
                self.gather.note_full_input((2 * x.shape[2], 2 * x.shape[3]))
                return self.scatter(full_conv2d(self.conv, x, upsample2x=True, stats=True))
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        if self.mode == "sparse":
            return self._produced(self.scatter.forward_fused(self.conv, self.gather(x)))
        if self.plain and self.mode == "full":
            return self.conv(x)
        return self.scatter(self.conv(self.gather(x)))


class Downsample(SIGEModule, _TwinProducer):
    """3x3 stride-2 conv with (0,1,0,1) zero padding; tiled when `sparse` (the
    gather's zero fill past the bottom/right border IS the padding then)."""

    def _twin_scatter(self):
        return self.scatter if self.sparse else None

    def __init__(self, cfg: DDPMConfig, ch: int, sparse: bool):
        super().__init__()
        self.sparse = sparse
        self.plain = False
        self.conv = (SIGEConv2d if sparse else nn.Conv2d)(ch, ch, 3, 2, 0)
        if sparse:
            self.gather = Gather(self.conv, cfg.main_block)
            self.scatter = Scatter(self.gather)

    def forward(self, x):
        if not self.sparse and self.mode == "sparse":
            return self._produced(fused_conv2d(self.conv, x, pad_bottom_right=True, twins=self._my_twins()))
        if self.mode == "full" and not self.plain and x.is_cuda:
            from ..nn import dense as _dense

            if _dense.fast_full_pass(self.conv):
                # the full pass on the library's kernels: the padding is the tile kernel's zero fill, as in the sparse pass
                if not self.sparse:
                    return fused_conv2d(self.conv, x, pad_bottom_right=True)
                return self.scatter(fused_conv2d(self.conv, self.gather(x), pad_bottom_right=True))
        if not self.sparse or (self.plain and self.mode == "full"):
            return self.conv(F.pad(x, (0, 1, 0, 1)))
        x = self.gather(x)
        if self.mode == "full":
            x = F.pad(x, (0, 1, 0, 1))
        if self.mode == "sparse":
            return self._produced(self.scatter.forward_fused(self.conv, x))
        return self.scatter(self.conv(x))


class DDPMSparseUNet(SIGEModel):
    # set_compute_dtype("f16"): the convs that stay at fp32-level precision (split fp16 operands, "f16x3") so that the f16
    # path meets sige_amd.tolerance.F16_CRITERION at every edit ratio of BASELINE.json configs[4] (1 ... 20 %).  From the
    # per-layer error trace (tests/f16_error_trace.py, profiles/r3a_f16_error_trace.json): fp16 rounding errors made on
    # the down path and at the 16x16 / 8x8 levels are amplified by everything downstream; with them kept the worst
    # element is at 0.34 of the allowed error at 20 % edit (2.5x over it with every conv in plain fp16).
    F16_KEEP = ("down", "up.4", "up.5")
    # ... for edits above this fraction of the image; below it every conv in plain fp16 meets the criterion (worst element at
    # 0.11 / 0.15 / 0.24 of the allowed error at 1 / 2 / 5 % edit, 0.98 at 10 %: the same trace)
    F16_KEEP_ABOVE = 0.05

    def __init__(self, cfg: DDPMConfig = DDPMConfig()):
        super().__init__()
        self.cfg = cfg
        self.edit_batch = 1  # (sige_amd.stacked.edit_batch: E edits of one original stacked along H in every sparse-mode tensor)
        ch, mult = cfg.ch, tuple(cfg.ch_mult)
        self.ch, self.temb_ch = ch, 4 * ch
        self.num_levels = len(mult)
        self.resolution = cfg.resolution
        self.conv_in = nn.Conv2d(cfg.in_ch, ch, 3, 1, 1)
        temb_slices = []

        res = cfg.resolution
        in_mult = (1,) + mult
        self.down = nn.ModuleList()
        cur = ch
        for lvl in range(self.num_levels):
            stage = nn.Module()
            stage.block, stage.attn = nn.ModuleList(), nn.ModuleList()
            cur, cout = ch * in_mult[lvl], ch * mult[lvl]
            sparse = res >= cfg.sparse_threshold
            for _ in range(cfg.num_res_blocks):
                stage.block.append(ResBlock(cfg, cur, cout, sparse))
                temb_slices.append(cout)
                cur = cout
                if res in cfg.attn_resolutions:
                    stage.attn.append(AttnBlock(cfg, cur, sparse))
            if lvl != self.num_levels - 1:
                stage.downsample = Downsample(cfg, cur, sparse)
                res //= 2
            self.down.append(stage)

        self.mid = nn.Module()
        self.mid.block_1 = ResBlock(cfg, cur, cur, False)
        self.mid.attn_1 = AttnBlock(cfg, cur, False)
        self.mid.block_2 = ResBlock(cfg, cur, cur, False)
        temb_slices += [cur, cur]

        ups = []
        for lvl in reversed(range(self.num_levels)):
            stage = nn.Module()
            stage.block, stage.attn = nn.ModuleList(), nn.ModuleList()
            cout = ch * mult[lvl]
            sparse = res >= cfg.sparse_threshold
            for i in range(cfg.num_res_blocks + 1):
                skip = ch * (in_mult[lvl] if i == cfg.num_res_blocks else mult[lvl])
                stage.block.append(ResBlock(cfg, cur + skip, cout, sparse))
                temb_slices.append(cout)
                cur = cout
                if res in cfg.attn_resolutions:
                    stage.attn.append(AttnBlock(cfg, cur, sparse))
            if lvl != 0:
                stage.upsample = Upsample(cfg, cur)
                res *= 2
            ups.insert(0, stage)
        self.up = nn.ModuleList(ups)

        self.temb_slices = temb_slices
        self.temb = nn.Module()
        self.temb.dense = nn.ModuleList([nn.Linear(ch, self.temb_ch), nn.Linear(self.temb_ch, self.temb_ch),
                                         nn.Linear(self.temb_ch, sum(temb_slices))])
        self.norm_out = nn.GroupNorm(cfg.groups, cur, eps=cfg.eps)
        self.conv_out = nn.Conv2d(cur, cfg.out_ch, 3, 1, 1)

    def set_plain_dense(self, plain: bool):
        """full mode = stock dense U-Net (F.group_norm, no cache bookkeeping)."""
        for m in self.modules():
            if isinstance(m, (ResBlock, AttnBlock, Upsample, Downsample)):
                m.plain = plain
        self.conv_in._plain_model = plain  # (the stock baseline keeps torch's first / last conv and GroupNorm)

    def _temb(self, t):
        if self.mode != "full":
            return None  # folded into the cached shifts
        e = timestep_embedding(t, self.ch)
        e = F.silu(self.temb.dense[0](e))
        e = F.silu(self.temb.dense[1](e))
        return list(torch.split(self.temb.dense[2](e), self.temb_slices, dim=1))

    # ---- the first conv of a sparse pass, on the active windows only (round 6) ------------------------------------------------------
    # hs[0] = conv_in(x) is read by down[0].block[0] and, as a skip, by the last block of up[0] (sige_fused_unet.py:395-400, 419-424):
    # both are tiled at this resolution with the same index list, so a sparse forward only ever looks at conv_in's output through
    # those 6x6 Gather windows (which contain the 4x4 shortcut windows).  The reference computes it densely because it is cheap there;
    # here it is 33.5 MB written per forward (17 us, write-bandwidth-bound) for a 1.2 % edit that reads 6 % of them.  The output
    # buffer is persistent; what lies outside the active windows is stale and unread.
    SPARSE_CONV_IN = True

    def _conv_in(self, x, native_full):
        if self.mode != "sparse" and not native_full:
            return self.conv_in(x)
        if self.mode == "sparse" and self.SPARSE_CONV_IN and self.edit_batch == 1 and x.is_cuda:
            first, last = self.down[0].block[0], self.up[0].block[-1]
            g0, g1 = getattr(first, "main_gather", None), getattr(last, "main_gather", None)
            if (g0 is not None and g1 is not None and g0.active_indices is not None and g1.active_indices is not None
                    and tuple(g0.block_size) == tuple(g1.block_size) and tuple(g0.offset) == tuple(g1.offset)
                    and tuple(g0.model_stride) == (1, 1) and tuple(g1.model_stride) == (1, 1)
                    # (one index list for both: set_masks memoises the lists per (resolution, block, stride, offset) key -- no comparison
                    #  of contents here: that would be a device -> host sync inside a capture)
                    and g0.active_indices.data_ptr() == g1.active_indices.data_ptr()):
                buf = getattr(self, "_h0_buf", None)
                shape = (x.shape[0], self.conv_in.out_channels, x.shape[2], x.shape[3])
                if buf is None or tuple(buf.shape) != shape or buf.device != x.device:
                    buf = self._h0_buf = torch.zeros(shape, dtype=torch.float32, device=x.device).contiguous(memory_format=torch.channels_last)
                return input_conv2d(self.conv_in, x, tiles=(g0.active_indices, tuple(g0.block_size)), out=buf)
        return input_conv2d(self.conv_in, x)

    def forward(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        assert x.shape[2] == x.shape[3] == self.resolution
        temb = self._temb(t)
        nxt = (lambda: temb.pop(0)) if temb is not None else (lambda: None)

        from ..nn import dense as _dense

        # the full pass on the library's kernels (compute dtype "f16" / "f16x3", or dense.FULL_PASS_F32_NATIVE) also takes the
        # library's first / last conv and output norm -- the same launches the sparse pass uses
        native_full = self.mode == "full" and x.is_cuda and not getattr(self.conv_in, "_plain_model", False) and (
            getattr(self.conv_out, "compute_dtype", "f32") != "f32" or _dense.FULL_PASS_F32_NATIVE)
        h0 = self._conv_in(x, native_full)
        if x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
            h0 = h0.contiguous(memory_format=torch.channels_last)  # (MIOpen may hand back NCHW for 3 input channels)
        E = self.edit_batch
        if E > 1:
            # stacked edits (sige_amd/stacked.py): x is [E,3,H,W]; from here to the output norm every tensor is the tall image
            # [1,C,E*H,W] -- the same bytes -- and every sparse module works on it unchanged
            if self.mode != "sparse" or x.shape[0] != E:
                raise RuntimeError("stacked edits: sparse mode, one input image per edit")
            from ..stacked import tall, untall

            h0 = tall(h0)
        hs = [h0]
        for lvl, stage in enumerate(self.down):
            for i, block in enumerate(stage.block):
                h = block(hs[-1], nxt())
                if len(stage.attn):
                    h = stage.attn[i](h)
                hs.append(h)
            if lvl != self.num_levels - 1:
                hs.append(stage.downsample(hs[-1]))

        h = self.mid.block_1(hs[-1], nxt())
        h = self.mid.attn_1(h)
        h = self.mid.block_2(h, nxt())

        for lvl in reversed(range(self.num_levels)):
            stage = self.up[lvl]
            for i, block in enumerate(stage.block):
                h = block((h, hs.pop()), nxt())
                if len(stage.attn):
                    h = stage.attn[i](h)
            if lvl != 0:
                h = stage.upsample(h)
        if self.mode == "sparse" or native_full:
            if E > 1:
                h = untall(h, E)  # (the output norm and the last conv are per image)
            # the output norm is a TRUE GroupNorm of the edited activation (sige_fused_unet.py:430-432)
            so, to = group_norm_affine(h, self.norm_out)
            return fused_conv2d(self.conv_out, h, so, to, "swish")
        return self.conv_out(F.silu(self.norm_out(h)))

