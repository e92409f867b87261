"""Stable-Diffusion spatial transformer with sparse queries (BASELINE.json configs[3]) on sige_amd.nn.

The block of stable-diffusion/ldm/modules/sige_attention.py (`SIGESpatialTransformer`): GroupNorm -> 1x1 proj_in ->
[self-attention, cross-attention on the text context, GEGLU feed-forward] x depth -> 1x1 proj_out + residual, where in sparse
mode only the tokens of the ACTIVE 4x4 tiles are queries, while keys / values of the self-attention cover every token of the
(scattered) full feature map.  Parameter names follow the reference (`norm`, `proj_in`, `transformer_blocks.0.attn1.to_q`,
`...ff.net.0.proj`, `...ff.net.2`, `proj_out`), so its state dict loads here (tests/test_reference_models.py).

Tile <-> token glue.  The reference keeps everything NCHW and pays, per transformer, a full-tensor Scatter (clone + write)
plus `b c h w -> b (h w) c` permute copies of the full tensor and of the tiles in both directions
(sige_attention.py:151-176).  Channels-last makes all of that free: a channels-last [B,C,H,W] tensor IS the token matrix
[B,HW,C], and channels-last tiles [B*N,C,4,4] ARE the query tokens [B,N*16,C] -- both directions are views.  The Scatter
writes the tiles into a persistent full-size buffer (in-place form), and proj_in / proj_out are the fused gather -> conv and
conv -> scatter kernels.

Keys / values of the self-attention (MI355X-first option `sparse_kv`, same values): LayerNorm and the K / V projections are
per-token maps, and the full feature map differs from the original image's only on the active tiles.  So the full pass caches
K and V of every token, and a sparse pass projects ONLY the active tokens and scatters those rows into a persistent copy of
the cached K / V (two Scatter launches per block) instead of re-projecting all HW tokens (sige_attention.py:78, attention.py:
78-80).  The cross-attention's K / V depend on the text only and are cached as in the reference (sige_attention.py:35-42).
"""
from typing import Optional

import torch
from torch import nn
from torch.nn import functional as F

from ..nn import Gather, Scatter, SIGEConv2d, SIGEModule


# MI355X-first switches of this workload (values identical either way; bench.py --workload sd prints which are on):
#   NATIVE_ATTENTION  the attention core as ONE library launch per attention (sige_hip_attention_tokens_f32: heads as strides,
#                     online softmax, no score tensor in HBM) instead of rearrange x 3 / bmm / softmax / bmm / rearrange
#   NATIVE_LINEAR     the token linears (to_q / to_k / to_v / to_out, the GEGLU projection, the feed-forward's second layer) on
#                     the library's MFMA tile kernel -- 16 tokens are one channels-last 4 x 4 tile, a Linear is a 1 x 1 conv --
#                     instead of the generic GEMM library
#   BATCHED_QKV       the three bias-free projections of a self-attention's input as ONE strided-batched GEMM (x broadcast
#                     against the stacked weights [3, C, inner]: three dense outputs, one launch instead of three)
#   FUSED_TOKENS      (round 5) what sits between the GEMMs of a block as one library launch each (csrc/token_ops.hip): residual add
#                     + the projection's bias + the next LayerNorm; GEGLU's a * gelu(gate); the block's last residual add + bias --
#                     48 kernels fewer per forward of the SD v1 U-Net (16 blocks x 3), same arithmetic.  Round 5 measured it SLOWER
#                     (10.98 vs 10.84 ms per forward, profiles/r5m_bench_sd_fused_tokens.json) and left it off; round 6 found why --
#                     the add + LayerNorm kernel's loads sat behind exec-masked branches, one dependent round trip per element of a
#                     lane, and the elementwise kernels divided 64-bit indices by run-time values -- and with both fixed it is
#                     FASTER and on: 10.08 vs 10.24 ms (tools/sd_fused_tokens_ab.py, profiles/r6ac_sd_fused_tokens.json; the
#                     round-5 kernels in the same run: 10.46 vs 10.27)
NATIVE_ATTENTION = True
NATIVE_LINEAR = False
BATCHED_QKV = True
FUSED_TOKENS = True


def linear(lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """`lin(x)` for tokens x [B,N,C]; with NATIVE_LINEAR on fp32 GPU tokens (B*N a multiple of 16, channels multiples of 4) one
    launch of the stacked-block 1x1 conv over the tokens seen as channels-last 4x4 tiles (views on both sides)."""
    if (NATIVE_LINEAR and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.is_contiguous()
            and (x.shape[0] * x.shape[1]) % 16 == 0 and x.shape[2] % 4 == 0 and lin.out_features % 4 == 0):
        from .. import hip

        w = lin.weight
        key = (w.data_ptr(), w._version, tuple(w.shape), w.device)
        if getattr(lin, "_sige_packed_key", None) != key:
            lin._sige_packed = hip.conv_pack_weights(w.detach().reshape(w.shape[0], w.shape[1], 1, 1), 4, 4, (1, 1))
            lin._sige_packed_key = key
        if lin._sige_packed is not None:
            b, n, c = x.shape
            tiles = x.reshape(b * n // 16, 4, 4, c).permute(0, 3, 1, 2)  # [T,C,4,4] channels-last: the same bytes
            out = hip.block_conv_cl(tiles, lin._sige_packed, lin.bias, lin.out_features, (1, 1), (1, 1))
            if out is not None:
                return out.permute(0, 2, 3, 1).reshape(b, n, lin.out_features)
    return lin(x)


def _heads(t: torch.Tensor, h: int) -> torch.Tensor:
    b, n, c = t.shape
    return t.reshape(b, n, h, c // h).permute(0, 2, 1, 3).reshape(b * h, n, c // h)


def _merge(t: torch.Tensor, h: int) -> torch.Tensor:
    bh, n, d = t.shape
    return t.reshape(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, h * d)


class Attention(SIGEModule):
    """softmax(q k^T / sqrt(d)) v with separate bias-free q / k / v projections and an output projection.
    `cache_context=True`: the keys / values of a fixed context (the text embedding) are computed in full mode only."""

    def __init__(self, query_dim: int, context_dim: Optional[int], heads: int, dim_head: int, cache_context: bool = False):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.scale = heads, dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim or query_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim or query_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(0.0))
        self.cache_context = cache_context
        self.cached_k = self.cached_v = None

    def qkv(self, x):
        """(to_q(x), to_k(x), to_v(x)) for a self-attention whose queries, keys and values come from the same tokens x [B,n,C]."""
        if not (BATCHED_QKV and not NATIVE_LINEAR and x.is_cuda and self.to_k.in_features == self.to_q.in_features):
            return linear(self.to_q, x), linear(self.to_k, x), linear(self.to_v, x)
        ws = (self.to_q.weight, self.to_k.weight, self.to_v.weight)
        key = tuple((w.data_ptr(), w._version) for w in ws)
        if getattr(self, "_qkv_key", None) != key:
            self._qkv_w = torch.stack([w.detach().t() for w in ws]).contiguous()  # [3, C, inner]
            self._qkv_key = key
        b, n, c = x.shape
        out = torch.matmul(x.reshape(1, b * n, c), self._qkv_w)  # [3, B*n, inner]: one strided-batched GEMM
        return tuple(t.reshape(b, n, -1) for t in out.unbind(0))

    def attend(self, q, k, v, bias: bool = True):
        """`bias=False`: the output projection WITHOUT its bias (the caller's fused add + LayerNorm adds it)."""
        if NATIVE_ATTENTION and q.is_cuda and q.dtype == torch.float32:
            from .. import hip

            out = hip.attention_tokens(q, k, v, self.heads, self.scale)
            if out is not None:
                if not bias:
                    return F.linear(out, self.to_out[0].weight)
                return self.to_out[1](linear(self.to_out[0], out))
        if not bias:
            raise RuntimeError("attend(bias=False) needs the native attention path")
        q, k, v = _heads(q, self.heads), _heads(k, self.heads), _heads(v, self.heads)
        sim = torch.bmm(q, k.transpose(1, 2)) * self.scale
        return self.to_out[1](linear(self.to_out[0], _merge(torch.bmm(sim.softmax(dim=-1), v), self.heads)))

    def forward(self, x, context=None, bias: bool = True):
        context = x if context is None else context
        if self.cache_context and self.mode != "full":
            k, v = self.cached_k, self.cached_v
        else:
            k, v = linear(self.to_k, context), linear(self.to_v, context)
            if self.cache_context:
                self.cached_k, self.cached_v = k, v
        return self.attend(linear(self.to_q, x), k, v, bias=bias)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        a, gate = linear(self.proj, x).chunk(2, dim=-1)
        return a * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim))

    def forward(self, x):
        return linear(self.net[2], self.net[1](self.net[0](x)))


class TransformerBlock(SIGEModule):
    def __init__(self, dim, heads, dim_head, context_dim):
        super().__init__()
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.ff = FeedForward(dim)
        self.attn2 = Attention(dim, context_dim, heads, dim_head, cache_context=True)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)

    def _fused_ok(self, x) -> bool:
        return (FUSED_TOKENS and NATIVE_ATTENTION and not NATIVE_LINEAR and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3
                and x.shape[2] % 4 == 0 and x.shape[2] <= 2048 and x.is_contiguous())

    def _forward_fused(self, x, full_x, context, kv_scatter):
        """The sparse-mode block with the token helpers of csrc/token_ops.hip: per block LayerNorm, [q | k | v GEMM, K / V scatter,
        attention], out-projection GEMM, add + bias + LayerNorm, q GEMM, [attention], out-projection GEMM, add + bias + LayerNorm,
        GEGLU-projection GEMM, GEGLU, second feed-forward GEMM, add + bias: 6 GEMMs + 5 library launches between them (was 6 GEMMs + 3
        LayerNorms + 3 adds + GELU + multiply).  Same operations in the same order as forward()."""
        from .. import hip

        a1, a2 = self.attn1, self.attn2
        _, xn = hip.add_layer_norm_tokens(x, None, None, self.norm1)
        sk, sv, (hh, ww) = kv_scatter
        as_tiles = lambda t: t.reshape(-1, 4, 4, t.shape[2]).permute(0, 3, 1, 2)  # noqa: E731
        q_t, k_t, v_t = a1.qkv(xn)
        k = sk(as_tiles(k_t))
        v = sv(as_tiles(v_t))
        as_tokens = lambda t: t.permute(0, 2, 3, 1).reshape(t.shape[0], hh * ww, t.shape[1])  # noqa: E731
        d1 = a1.attend(q_t, as_tokens(k), as_tokens(v), bias=False)
        x, xn = hip.add_layer_norm_tokens(x, d1, a1.to_out[0].bias, self.norm2)
        d2 = a2(xn, context=context, bias=False)
        x, xn = hip.add_layer_norm_tokens(x, d2, a2.to_out[0].bias, self.norm3)
        ff = self.ff
        h = hip.geglu_tokens(linear(ff.net[0].proj, xn))
        return hip.add_bias_tokens(x, F.linear(h, ff.net[2].weight), ff.net[2].bias)

    def forward(self, x, full_x=None, context=None, kv_scatter=None):
        """x: query tokens [B,n,C]; full_x: all tokens [B,HW,C] (None: x itself); kv_scatter: (scatter_k, scatter_v, hw)
        -- Scatter modules of the enclosing transformer for the sparse K / V refresh, or None for the reference's form."""
        if kv_scatter is not None and self.mode == "sparse" and self._fused_ok(x):
            return self._forward_fused(x, full_x, context, kv_scatter)
        a1 = self.attn1
        xn = self.norm1(x)
        if kv_scatter is not None and self.mode == "full":
            ctx = xn if full_x is None else self.norm1(full_x)
            k, v = linear(a1.to_k, ctx), linear(a1.to_v, ctx)
            sk, sv, (hh, ww) = kv_scatter
            as_map = lambda t: t.reshape(t.shape[0], hh, ww, t.shape[2]).permute(0, 3, 1, 2)  # noqa: E731  [B,C,H,W] channels-last view
            sk(as_map(k)), sv(as_map(v))  # full mode: the Scatter modules remember K / V as their cached tensors
            x = a1.attend(linear(a1.to_q, xn), k, v) + x
        elif kv_scatter is not None and self.mode == "sparse":
            sk, sv, (hh, ww) = kv_scatter
            b, n, c = xn.shape
            as_tiles = lambda t: t.reshape(-1, 4, 4, t.shape[2]).permute(0, 3, 1, 2)  # noqa: E731  tokens -> [B*N,C,4,4] channels-last view
            q_t, k_t, v_t = a1.qkv(xn)
            k = sk(as_tiles(k_t))  # rows of the active tokens over the cached K of the original image
            v = sv(as_tiles(v_t))
            as_tokens = lambda t: t.permute(0, 2, 3, 1).reshape(t.shape[0], hh * ww, t.shape[1])  # noqa: E731
            x = a1.attend(q_t, as_tokens(k), as_tokens(v)) + x
        else:
            x = a1(xn, context=None if full_x is None else self.norm1(full_x)) + x
        x = self.attn2(self.norm2(x), context=context) + x
        return self.ff(self.norm3(x)) + x


def group_norm_affine(x: torch.Tensor, norm: nn.GroupNorm):
    """(normalised x, scale, shift) with per-sample per-channel scale / shift [B,C,1,1]: GroupNorm(x) == x*scale + shift
    (the contract of the reference's my_group_norm, stable-diffusion/ldm/modules/diffusionmodules/sige_model.py:12-33)."""
    b, c, h, w = x.shape
    g = norm.num_groups
    xg = x.reshape(b, g, -1)
    var, mean = torch.var_mean(xg, dim=2, unbiased=False, keepdim=True)
    std = torch.sqrt(var + norm.eps)
    y = ((xg - mean) / std).reshape(b, c, h, w)
    scale = (1 / std).reshape(b, g, 1, 1).repeat_interleave(c // g, dim=1)
    shift = (-mean / std).reshape(b, g, 1, 1).repeat_interleave(c // g, dim=1)
    if norm.affine:
        wt, bs = norm.weight.view(1, -1, 1, 1), norm.bias.view(1, -1, 1, 1)
        y = y * wt + bs
        scale = scale * wt
        shift = shift * wt + bs
    return y, scale, shift


class SpatialTransformer(SIGEModule):
    def __init__(self, in_channels: int, n_heads: int, d_head: int, depth: int = 1, context_dim: Optional[int] = None,
                 block_size: Optional[int] = 4, sparse_kv: bool = True):
        super().__init__()
        inner = n_heads * d_head
        self.in_channels, self.inner = in_channels, inner
        self.tiled = block_size is not None
        self.sparse_kv = sparse_kv and self.tiled
        Conv = SIGEConv2d if self.tiled else nn.Conv2d
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.proj_in = Conv(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([TransformerBlock(inner, n_heads, d_head, context_dim) for _ in range(depth)])
        self.proj_out = Conv(inner, in_channels, 1)
        if self.tiled:
            self.gather = Gather(self.proj_in, block_size)
            self.scatter1 = Scatter(self.gather)
            self.scatter2 = Scatter(self.gather)
            if self.sparse_kv:
                # K / V of the self-attention live on the token grid: the same 4x4 tiles, one Scatter pair per block
                self.kv_scatters = nn.ModuleList([nn.ModuleList([Scatter(self.gather), Scatter(self.gather)]) for _ in range(depth)])
        self.scale = self.shift = None

    def forward(self, x: torch.Tensor, context=None) -> torch.Tensor:
        b, c, h, w = x.shape
        x_in = x
        if self.mode == "full":
            if self.tiled:
                x = self.gather(x)
            x, self.scale, self.shift = group_norm_affine(x, self.norm)
        elif self.mode in ("sparse", "profile"):
            x = self.gather(x, self.scale, self.shift) if self.tiled else x * self.scale + self.shift
        else:
            raise NotImplementedError("Unknown mode [%s]!!!" % self.mode)
        x = self.proj_in(x)

        def tokens(t):  # [B,C,H,W] -> [B,HW,C]; a view for channels-last tensors
            return t.permute(0, 2, 3, 1).reshape(t.shape[0], t.shape[2] * t.shape[3], t.shape[1])

        if self.tiled:
            full_x = tokens(self.scatter1(x))
            if self.mode == "full":
                q = full_x
            else:  # tiles [B*N,C,4,4] -> [B, N*16, C]
                q = x.permute(0, 2, 3, 1).reshape(b, -1, x.shape[1])
        else:
            full_x, q = None, tokens(x)
        for i, blk in enumerate(self.transformer_blocks):
            kv = (self.kv_scatters[i][0], self.kv_scatters[i][1], (h, w)) if (self.sparse_kv and self.mode != "profile") else None
            q = blk(q, full_x=full_x, context=context, kv_scatter=kv)
        if self.tiled and self.mode != "full":
            bs = self.gather.block_size
            x = q.reshape(-1, bs[0], bs[1], q.shape[-1]).permute(0, 3, 1, 2)  # tokens -> tiles (channels-last view)
        else:
            x = q.reshape(b, h, w, q.shape[-1]).permute(0, 3, 1, 2)
        if self.tiled:
            return self.scatter2.forward_fused(self.proj_out, x, x_in) if self.mode == "sparse" else self.scatter2(self.proj_out(x), x_in)
        return self.proj_out(x) + x_in
