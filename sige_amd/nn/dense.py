"""Dense layers of a SIGE network through the tile kernels.

In sparse mode the reference still runs the low-resolution blocks densely, as
separate torch ops: `h * scale + shift`, `swish`, `nn.Conv2d`, `+ skip`, after a
`torch.cat` on the up path (sige_fused_unet.py:112-114,121-123,416).  On MI355X
that is ~10 launch-bound kernels per block.  `fused_conv2d` runs the same math
as ONE launch of the MFMA tile kernel with every tile active:

    out = conv(act(cat(x, x2) * scale + shift)) + residual

(zero padding = the gather's zero fill, so padded pixels are 0 AFTER the
affine/activation exactly like padding the activated tensor).  On tensors that
are not on the GPU it evaluates the identical expression with torch ops -- this
is the reference's own implementation of these dense layers, not a fallback of
the sparse path.
"""
from typing import Optional, Tuple

import torch
from torch import nn
from torch.nn import functional as F

_GEOMETRY = {
    # (kernel, stride, padding) -> (tile block, out tile, gather offset)
    ((3, 3), (1, 1), (1, 1)): ((6, 6), (4, 4), (1, 1)),
    ((1, 1), (1, 1), (0, 0)): ((4, 4), (4, 4), (0, 0)),
    ((3, 3), (2, 2), (0, 0)): ((5, 5), (2, 2), (0, 0)),  # with (0,1,0,1) zero padding supplied by the zero fill
}


def _tile_compute(compute: str) -> str:
    """Arithmetic of the TILE kernels (conv_mfma.hpp) for a conv whose compute dtype is `compute`: exact fp32, fp16 operands,
    or split fp16 operands (geometries without such a kernel -- stride 2 -- are packed for exact fp32 by conv_pack_weights)."""
    return compute if compute in ("f16", "f16x3") else "f32"


# Which dense layers take the dense-layer kernel (conv_wide.hpp) instead of the tile kernels with every tile active: those of at
# least this many flop, by kernel size -- in practice the convs of the FULL pass (4.8 ... 39 GFLOP each at 64^2 ... 256^2).
# Measured on MI355X, round 3:
#   * layer by layer with cold weights (tools/wide_bench.py, profiles/r3b_wide_bench.jsonl) the 64 x 64 blocks win on every 3x3
#     dense layer (3.6 GFLOP: 24 vs 51 us; 19 GFLOP: 73 vs 282 us) and lose on the 1x1s (two k-steps per chunk);
#   * inside the sparse forward they do not (tools/routing_bench.py, profiles/r3d_routing.jsonl): the tile kernels run a
#     residual block's 1x1 shortcut INSIDE the conv1 launch (conv_pair), the wide kernel cannot, and the ten extra launches
#     (~8 us each) eat what the dense 3x3s gain -- 1.456 vs 1.446 ms (split operands vs exact fp32, 1.2 % edit), 1.250 vs
#     1.034 ms with plain fp16 operands.  So the dense remainder of the sparse pass stays on the tile kernels.
WIDE_MIN_FLOP = {3: 4.0e9, 1: 2.0e9}
# ... for split fp16 operands ("f16x3"), its own threshold since round 4: a dense-layer launch now takes the block's 1x1 shortcut
# along (conv_wide_pair_kernel: the horizontal fusion the tile kernels had), so routing a conv1 here no longer costs a launch --
# measured again with tools/routing_bench.py (profiles/r4f_routing.jsonl)
WIDE_MIN_FLOP_X3 = {3: 4.0e9, 1: 2.0e9}
WIDE_MIN_FLOP_FULL_PASS = {3: 0.25e9, 1: 2.0e9}
# exact-fp32 form of the same kernel (v_mfma_f32_32x32x2_f32; matrix-bound).  Measured (profiles/r3g_wide_bench.jsonl,
# r3g_routing.jsonl): on the dense remainder it only ties the tile kernels layer by layer (3.6 GFLOP: 49.8 vs 51.6 us = 73 vs
# 70 TFLOP/s -- the f32-input MFMA is the limit for both) and loses the shortcut pairing in the forward (1.50 ... 1.67 vs 1.447
# ms), so the sparse pass does not use it; on the big layers of the full pass it reaches 109-127 TFLOP/s (0.70-0.81 of the fp32
# matrix peak; 283 -> 177 us), which FULL_PASS_F32_NATIVE turns on for an exact-fp32 full pass on the library's kernels.
WIDE_MIN_FLOP_F32 = {3: 1.0e30, 1: 1.0e30}
# ... with E stacked edits (sige_amd.stacked) a dense layer has E times the pixels: from 8 GFLOP per launch on the exact-fp32
# dense-layer kernel wins (profiles/r4_bench.json: batched_edits -- E = 8: 5.27 vs 5.58 ms per 8 edits, the routed layers at 128
# TFLOP/s = 0.81 of the fp32 MFMA peak; E = 16: 9.66 vs 10.31 ms; E = 4: a tie).  One image never reaches it (<= 3.6 GFLOP).
WIDE_MIN_FLOP_F32_STACKED = {3: 8.0e9, 1: 1.0e30}
FULL_PASS_F32_NATIVE = False
# Tile convs (conv_mfma.hpp) asked for split fp16 operands run them only above this many flop per launch; below, exact fp32.
# Measured (profiles/r3c_bench.json, DDPM-256 sparse forward, every tile conv on split operands against exact fp32): 1.2 %
# edit (0.4 GFLOP per launch) 555 vs 521 us over the 48 launches, 5 % 2.01 vs 1.92 ms per forward, 15 % (3.9 GFLOP) 2.47 vs
# 2.55 ms: the three-MFMA form wins once a launch is matrix-bound.
TILE_X3_MIN_FLOP = 2.0e9


def _wide_packed(conv: nn.Conv2d, compute: str):
    """Weights of `conv` packed for the dense-layer kernel on the fp16 matrix cores (None: no kernel for the shape)."""
    from .. import hip

    w = conv.weight
    key = (w.data_ptr(), w._version, tuple(w.shape), w.device)
    cache = conv.__dict__.setdefault("_sige_wide", {})
    entry = cache.get(compute)
    if entry is None or entry[0] != key:
        entry = (key, hip.wide_conv_pack_weights(_plain_weight(conv), compute))
        cache[compute] = entry
    return entry[1]


def _wide_conv(conv: nn.Conv2d, x, x2, scale, shift, activation_name, residual, out_affine, twins, made, upsample2x=False,
               min_flop=None, stats=False):
    """The conv as one launch of the dense-layer kernel (csrc/conv_wide.hpp) when the layer's compute dtype asks for the fp16
    matrix cores ("f16" / "f16x3") and the shape has a kernel; None otherwise (the caller goes on to the tile kernels)."""
    from .. import hip

    compute = getattr(conv, "compute_dtype", "f32")
    if compute == "f32" and min_flop is None:
        # (exact fp32 on v_mfma_f32_32x32x2_f32: the sparse pass's dense remainder)
        min_flop = WIDE_MIN_FLOP_F32
        if hip.get_edit_batch() > 1 and min_flop[3] >= 1.0e30:
            min_flop = WIDE_MIN_FLOP_F32_STACKED
    if compute not in ("f16", "f16x3", "f32") or not hip.is_cl(x) or (x2 is not None and not hip.is_cl(x2)):
        return None
    k = tuple(conv.kernel_size)
    if (k not in ((1, 1), (3, 3)) or tuple(conv.stride) != (1, 1) or tuple(conv.padding) != (k[0] // 2, k[1] // 2)
            or conv.groups != 1 or tuple(conv.dilation) != (1, 1) or getattr(conv, "padding_mode", "zeros") != "zeros"):
        return None
    C1, C2 = x.shape[1], 0 if x2 is None else x2.shape[1]
    pix = x.shape[0] * x.shape[2] * x.shape[3] * (4 if upsample2x else 1)
    if min_flop is None:
        min_flop = WIDE_MIN_FLOP_X3 if compute == "f16x3" else WIDE_MIN_FLOP
    if 2.0 * pix * conv.out_channels * (C1 + C2) * k[0] * k[1] < min_flop[k[0]]:
        return None  # (small layers are latency-bound: the tile kernels' 16 / 32-pixel blocks start up faster)
    if not hip.wide_conv_supported(C1, C2, conv.out_channels, k):
        if x2 is None or upsample2x or not hip.wide_conv_supported(C1 + C2, 0, conv.out_channels, k):
            return None
        x, x2, C1, C2 = torch.cat([x, x2], dim=1), None, C1 + C2, 0  # (a split that is not on a chunk boundary)
    packed = _wide_packed(conv, compute)
    if packed is None:
        return None
    B = x.shape[0]
    H, W = (2 * x.shape[2], 2 * x.shape[3]) if upsample2x else (x.shape[2], x.shape[3])
    tw = []
    if twins and out_affine is None:
        for key, (sc, sh) in list(twins.items())[:2]:
            tw.append((key, torch.empty((B, conv.out_channels, H, W), dtype=torch.float32, device=x.device,
                                        memory_format=torch.channels_last), sc, sh))
    out = hip.wide_conv_cl(x, x2, scale, shift, activation_name, packed, conv.bias, conv.out_channels, k, residual=residual,
                           out_affine=out_affine, twins=[(b, sc, sh) for _, b, sc, sh in tw] or None, upsample2x=upsample2x,
                           stats=stats)
    if out is not None and made is not None:
        made.update({k_: b for k_, b, _, _ in tw})
    return out


def fast_full_pass(conv: nn.Conv2d) -> bool:
    """Does the full pass run this conv on the library's kernels (compute dtype "f16" / "f16x3", or FULL_PASS_F32_NATIVE for
    exact fp32) rather than as the reference's torch conv?"""
    return getattr(conv, "compute_dtype", "f32") != "f32" or FULL_PASS_F32_NATIVE


def full_conv2d(conv: nn.Conv2d, x: torch.Tensor, scale=None, shift=None, activation_name: str = "identity",
                residual: Optional[torch.Tensor] = None, x2: Optional[torch.Tensor] = None, upsample2x: bool = False,
                stats: bool = False) -> torch.Tensor:
    """`conv(act(cat(x, x2) * scale + shift)) + residual` of the FULL pass (the pass that produces the caches: sige/nn/base.py:85-86).
    On a channels-last GPU tensor and a conv whose compute dtype is "f16" / "f16x3" (SIGEModel.set_compute_dtype) it is one
    launch of the dense-layer kernel with the affine + SiLU in its staging path; anywhere else exactly the torch
    expression the reference runs (a SIGEConv2d in full mode is nn.Conv2d.forward).
    `x2`: the second half of a torch.cat that then never exists; `upsample2x`: x is the half-resolution tensor, read as its
    nearest x2 upsampling (F.interpolate fused); `stats`: the launch also leaves the per-channel statistics of its output
    (hip.channel_stats(out)), from which group_norm_affine takes the next GroupNorm without a pass over the tensor."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4:
        # (no shortcut / conv1 pairing in the full pass: every 3x3 the kernel is at least as fast on takes it)
        out = None
        # (exact fp32: the reference's own torch conv, as it was -- unless FULL_PASS_F32_NATIVE asks for the exact-fp32 form of the
        #  dense-layer kernel)
        if fast_full_pass(conv):
            out = _wide_conv(conv, x, x2, scale, shift, activation_name, residual, None, None, None, upsample2x=upsample2x,
                             min_flop=WIDE_MIN_FLOP_FULL_PASS, stats=stats)
            if out is None and fusable(conv) and tuple(conv.stride) == (1, 1):
                # the small layers (1x1 shortcuts at 32^2 and below, the attention blocks' qkv / proj): the tile kernels with every
                # tile active, exactly as the sparse pass runs them -- affine, cat and residual inside the launch
                xs = F.interpolate(x if x2 is None else torch.cat([x, x2], 1), scale_factor=2.0, mode="nearest") if upsample2x else x
                return fused_conv2d(conv, xs, scale, shift, activation_name, x2=None if upsample2x else x2, residual=residual)
        if out is not None:
            return out
    if x2 is not None:
        x = torch.cat([x, x2], dim=1)
    if upsample2x:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    h = x
    if scale is not None:
        h = h * scale
    if shift is not None:
        h = h + shift
    h = _act(h, activation_name)
    h = nn.Conv2d.forward(conv, h)
    return h if residual is None else h + residual


def _packed(conv: nn.Conv2d, block, channels_last: bool = True):
    from .. import hip

    w = conv.weight
    compute = _tile_compute(getattr(conv, "compute_dtype", "f32")) if channels_last else "f32"  # (SIGEModel.set_compute_dtype)
    if compute == "f16x3":
        compute = "f32"  # (a dense layer too small for the wide kernel is below TILE_X3_MIN_FLOP as well)
    key = (w.data_ptr(), w._version, tuple(w.shape), block, w.device, compute)
    if getattr(conv, "_sige_packed_key", None) != key:
        conv._sige_packed = hip.conv_pack_weights(w, block[0], block[1], conv.stride, compute)
        conv._sige_packed_key = key
    if conv._sige_packed is None:
        raise RuntimeError("fused_conv2d: no matrix-core kernel for this conv geometry")
    return conv._sige_packed


def _plain_weight(conv: nn.Conv2d) -> torch.Tensor:
    """conv.weight in the dense [Cout,Cin,kH,kW] order the kernels index (a channels-last model holds it
    permuted); converted once per weight version, not once per forward."""
    w = conv.weight
    if w.is_contiguous():
        return w
    key = (w.data_ptr(), w._version, tuple(w.shape), w.device)
    if getattr(conv, "_sige_plain_key", None) != key:
        conv._sige_plain = w.detach().contiguous()
        conv._sige_plain_key = key
    return conv._sige_plain


def fusable(conv: nn.Conv2d) -> bool:
    geo = (tuple(conv.kernel_size), tuple(conv.stride), tuple(conv.padding))
    return geo in _GEOMETRY and conv.groups == 1 and tuple(conv.dilation) == (1, 1)


def _act(x, name):
    if name == "swish":
        return F.silu(x)
    if name == "identity":
        return x
    raise ValueError("Unknown activation: [%s]!!!" % name)


def fused_conv2d(conv: nn.Conv2d, x: torch.Tensor, scale: Optional[torch.Tensor] = None,
                 shift: Optional[torch.Tensor] = None, activation_name: str = "identity",
                 x2: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                 pad_bottom_right: bool = False, out_affine: Optional[tuple] = None,
                 twins: Optional[dict] = None) -> torch.Tensor:
    """conv(act(cat(x, x2) * scale + shift)) + residual.  scale/shift: [1|B, C, 1, 1].
    `pad_bottom_right`: the DDPM downsample's (0,1,0,1) zero padding (stride-2 convs).
    `out_affine` = (scale, shift, activation) of the CONSUMER, applied to the result in the kernel's
    epilogue (one launch, once per element): act(out * scale + shift).
    `twins` = {key: (scale [Cout], shift [Cout])} (at most two): where the launch can, it also writes
    SiLU(scale * result + shift) -- the activated input of a consumer's conv1 -- and returns them as
    `out._sige_twins[key]` (scatter.tag_twins); a consumer that finds no entry activates for itself."""
    made = {}
    out = _fused_conv2d(conv, x, scale, shift, activation_name, x2, residual, pad_bottom_right, out_affine, twins, made)
    if isinstance(out, tuple):  # (tensor, epilogue still to apply)
        out, (os_, oh_, oact) = out
        out = _act(out * os_.reshape(1, -1, 1, 1) + oh_.reshape(1, -1, 1, 1), oact)
    if not made and twins and out_affine is None:
        from . import scatter

        if scatter.EMULATE_TWINS:  # (tests only: see scatter.EMULATE_TWINS)
            made = scatter.emulated_twins(out, twins)
    out._sige_twins = made
    return out


def _fused_conv2d(conv, x, scale, shift, activation_name, x2, residual, pad_bottom_right, out_affine, twins=None, made=None):
    if x.is_cuda and x.dtype == torch.float32 and fusable(conv):
        from .. import hip

        block, out_tile, offset = _GEOMETRY[(tuple(conv.kernel_size), tuple(conv.stride), tuple(conv.padding))]
        if not pad_bottom_right:
            out = _wide_conv(conv, x, x2, scale, shift, activation_name, residual, out_affine, twins, made)
            if out is not None:
                return out
        if x2 is not None and not hip.cat_fusable(x.shape[0], x.shape[1], conv.kernel_size):
            x, x2 = torch.cat([x, x2], dim=1), None
        B, _, H, W = x.shape
        if conv.stride[0] == 2:
            if not pad_bottom_right:
                raise NotImplementedError("stride-2 fused conv expects the (0,1,0,1) padding of the DDPM downsample")
            Ho, Wo = (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1
        else:
            Ho, Wo = H, W
        if (hip.is_cl(x) and conv.out_channels <= 4 and tuple(conv.kernel_size) == (3, 3) and conv.stride[0] == 1
                and x2 is None and residual is None and out_affine is None):
            out = hip.conv3x3_small_cout_cl(x, _plain_weight(conv), conv.bias, scale, shift, activation_name)
            if out is not None:
                return out
        idx = hip.all_tiles(H, W, out_tile, conv.stride, offset, x.device)
        if hip.is_cl(x) and hip.cl_supported(x.shape[1], 0 if x2 is None else x2.shape[1], conv.out_channels):
            tw = []
            if twins and out_affine is None:  # (a twin is a function of the RAW result: the primary output must be it)
                for key, (sc, sh) in list(twins.items())[:2]:
                    tw.append((key, torch.empty((B, conv.out_channels, Ho, Wo), dtype=torch.float32, device=x.device,
                                                memory_format=torch.channels_last), sc, sh))
            out = hip.gather_conv_cl(x, x2, block, idx, scale, shift, activation_name, _packed(conv, block), conv.bias,
                                     conv.out_channels, conv.kernel_size, conv.stride,
                                     full=dict(offset=offset, out_res=(Ho, Wo), residual=residual), out_affine=out_affine,
                                     twins=[(b, sc, sh) for _, b, sc, sh in tw] or None)
            if out is not None:
                if made is not None:
                    made.update({k: b for k, b, _, _ in tw})
                return out
        out = hip.gather_conv_nchw(x.contiguous(), None if x2 is None else x2.contiguous(), block, idx,
                                    scale, shift, activation_name, _packed(conv, block, False), conv.bias,
                                    conv.out_channels, conv.kernel_size, conv.stride, offset, (Ho, Wo),
                                    None if residual is None else residual.contiguous())
        return out if out_affine is None else (out, out_affine)
    h = x if x2 is None else torch.cat([x, x2], dim=1)
    if scale is not None:
        h = h * scale
    if shift is not None:
        h = h + shift
    h = _act(h, activation_name)
    if pad_bottom_right:
        h = F.pad(h, (0, 1, 0, 1))
    h = conv(h)
    h = h if residual is None else h + residual
    return h if out_affine is None else (h, out_affine)


def input_conv2d(conv: nn.Conv2d, x: torch.Tensor, tiles=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`conv(x)` for the network's first layer (3x3 / padding 1, <= 3 input channels, e.g. DDPM's conv_in,
    sige_fused_unet.py:395).  On a channels-last GPU image: one thin-GEMM launch writing the channels-last
    result directly (libsige_hip.so); anything else is the plain `conv(x)`.
    `tiles` = (index list, block) with `out` a persistent channels-last buffer: the conv is evaluated only on those windows, in
    place (a sparse pass reads the first conv's output through Gather windows only); ignored where the launch does not apply."""
    if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] <= 3 and conv.groups == 1
            and tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1)
            and tuple(conv.dilation) == (1, 1) and conv.padding_mode == "zeros"
            and x.is_contiguous(memory_format=torch.channels_last) and (x.shape[1] == 1 or not x.is_contiguous())):
        from .. import hip

        if tiles is not None and out is not None:
            r = hip.conv3x3_small_cin_cl(x, _plain_weight(conv), conv.bias, tiles=tiles, out=out)
            if r is not None:
                return r
        r = hip.conv3x3_small_cin_cl(x, _plain_weight(conv), conv.bias)
        if r is not None:
            return r
    return conv(x)


def group_norm_affine(x: torch.Tensor, norm: nn.GroupNorm, channel_bias: Optional[torch.Tensor] = None,
                      x2: Optional[torch.Tensor] = None, make_stats: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """(scale, shift) as [B,C,1,1] with GroupNorm(x) == x*scale + shift.  `channel_bias` [C]: GroupNorm(x + channel_bias) ==
    x*scale + shift (a residual block's norm of h + temb without materialising the sum).  `x2`: the GroupNorm of
    torch.cat([x, x2], 1).  Tensors that carry the per-channel statistics their producer left (full_conv2d(stats=True)) are not
    read at all: one launch over the partial sums."""
    if x.is_cuda and x.dtype == torch.float32:
        from .. import hip

        parts = [hip.channel_stats(t) for t in ((x,) if x2 is None else (x, x2))]
        if make_stats:  # (the full pass: a tensor without statistics gets them in one pass, kept for its next consumer)
            parts = [p if p is not None else (hip.channel_stats_cl(t) if hip.is_cl(t) else None)
                     for p, t in zip(parts, (x,) if x2 is None else (x, x2))]
        if all(p is not None for p in parts):
            r = hip.group_norm_affine_from_stats(parts, norm.num_groups, norm.eps, norm.weight, norm.bias, channel_bias)
            if r is not None:
                return r
    if x2 is not None:
        x = torch.cat([x, x2], dim=1)
    if x.is_cuda and x.dtype == torch.float32:
        from .. import hip

        if hip.is_cl(x):
            r = hip.group_norm_affine_cl(x, norm.num_groups, norm.eps, norm.weight, norm.bias, channel_bias)
            if r is not None:
                return r
        if channel_bias is not None:
            sc, sh = group_norm_affine(x + channel_bias.reshape(1, -1, 1, 1), norm)
            return sc, sh + channel_bias.reshape(1, -1, 1, 1) * sc
        return hip.group_norm_affine(x, norm.num_groups, norm.eps, norm.weight, norm.bias)
    if channel_bias is not None:
        sc, sh = group_norm_affine(x + channel_bias.reshape(1, -1, 1, 1), norm)
        return sc, sh + channel_bias.reshape(1, -1, 1, 1) * sc
    B, C = x.shape[:2]
    g = norm.num_groups
    var, mean = torch.var_mean(x.reshape(B, g, -1), dim=2, unbiased=False)
    inv = torch.rsqrt(var + norm.eps).repeat_interleave(C // g, dim=1)
    mu = mean.repeat_interleave(C // g, dim=1)
    w = norm.weight if norm.weight is not None else torch.ones(C, device=x.device)
    b = norm.bias if norm.bias is not None else torch.zeros(C, device=x.device)
    scale = inv * w
    shift = b - mu * scale
    return scale.reshape(B, C, 1, 1), shift.reshape(B, C, 1, 1)
