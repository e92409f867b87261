"""Difference-mask helpers: mask -> active tile indices, dilation, pyramid.

API parity with sige/utils.py (reduce_mask :8-37, dilate_mask :40-71,
compute_difference_mask :74-85, downsample_mask :88-118).  Index tensors must be
bit-exact with the reference: int32 [N,2], row-major order of the candidate
grid, values stride*i - padding.
"""
from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch
from torch.nn import functional as F

IntPair = Union[int, Tuple[int, int]]


def _pair(v: IntPair) -> Tuple[int, int]:
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


def reduce_mask(
    mask: torch.Tensor,
    block_size: Optional[IntPair],
    stride: Optional[IntPair],
    padding: Optional[IntPair],
    verbose: bool = False,
) -> Optional[torch.Tensor]:
    """Active tiles of `mask` [H,W]: a tile (origin stride*i - padding, extent
    block_size) is active iff it contains a masked pixel.  The candidate grid is
    ((H+pad)//stride + 1) x ((W+pad)//stride + 1) -- the reference pads by a whole
    block on the bottom/right before pooling (sige/utils.py:27).

    GPU masks go through libsige_hip.so's ballot/compact kernel; CPU masks are
    reduced with integer torch ops (host logic, no float pooling)."""
    if block_size is None or stride is None or padding is None:
        return None
    block_size, stride, padding = _pair(block_size), _pair(stride), _pair(padding)
    if mask.is_cuda:
        from . import hip

        active_indices = hip.reduce_mask(mask, block_size, stride, padding)
        total = ((mask.shape[0] + padding[0]) // stride[0] + 1) * ((mask.shape[1] + padding[1]) // stride[1] + 1)
    else:
        H, W = mask.shape
        gh, gw = (H + padding[0]) // stride[0] + 1, (W + padding[1]) // stride[1] + 1
        # summed-area table of the zero-padded mask, then one window sum per candidate
        canvas = torch.zeros((padding[0] + H + block_size[0] + 1, padding[1] + W + block_size[1] + 1), dtype=torch.int64)
        canvas[padding[0] + 1:padding[0] + 1 + H, padding[1] + 1:padding[1] + 1 + W] = mask.to(torch.int64)
        sat = canvas.cumsum(0).cumsum(1)
        i0 = torch.arange(gh) * stride[0]
        j0 = torch.arange(gw) * stride[1]
        i1, j1 = i0 + block_size[0], j0 + block_size[1]
        win = sat[i1][:, j1] - sat[i0][:, j1] - sat[i1][:, j0] + sat[i0][:, j0]
        cells = torch.nonzero(win > 0)
        active_indices = torch.stack(
            (cells[:, 0] * stride[0] - padding[0], cells[:, 1] * stride[1] - padding[1]), dim=1
        ).to(torch.int32).contiguous()
        total = gh * gw
    if verbose:
        num_active = active_indices.shape[0]
        print("Block Sparsity: %d/%d=%.2f%%" % (num_active, total, 100 * num_active / total))
    return active_indices


def dilate_mask(mask: Union[torch.Tensor, np.ndarray], dilation: IntPair) -> Union[torch.Tensor, np.ndarray]:
    """OR the mask with itself shifted by 1..d along H then along W (of the
    ORIGINAL mask: a plus-shaped structuring element, sige/utils.py:57-61).
    [H,W] or [C,H,W]; torch or numpy; returns a new array."""
    d = _pair(dilation)
    if d[0] <= 0 and d[1] <= 0:
        return mask
    if isinstance(mask, torch.Tensor) and mask.is_cuda and mask.dim() == 2 and mask.dtype == torch.bool:
        # (bool only: the kernel reads "non-zero = set" and writes 0 / 1, while the slice-OR below keeps the VALUES of an
        #  integer mask -- a 0 / 255 uint8 mask stays on the generic path so CPU and GPU agree)
        from . import hip

        return hip.dilate_mask(mask, d)  # one kernel instead of 4 * d slice ops
    out = mask.clone() if isinstance(mask, torch.Tensor) else np.array(mask, copy=True)
    nd = out.ndim if isinstance(out, np.ndarray) else out.dim()
    if nd not in (2, 3):
        raise NotImplementedError("Unknown mask dimension [%d]!!!" % nd)
    hax = nd - 2  # H axis; W axis is the last one

    def shifted_or(axis: int, k: int):
        lo = [slice(None)] * nd
        hi = [slice(None)] * nd
        lo[axis], hi[axis] = slice(None, -k), slice(k, None)
        lo, hi = tuple(lo), tuple(hi)
        out[lo] |= mask[hi]
        out[hi] |= mask[lo]

    for k in range(1, d[0] + 1):
        shifted_or(hax, k)
    for k in range(1, d[1] + 1):
        shifted_or(nd - 1, k)
    return out


def compute_difference_mask(tensor1: torch.Tensor, tensor2: torch.Tensor, eps: float = 2e-2) -> torch.Tensor:
    """|a-b| > eps, reduced with `any` over channels -> [H,W] bool (sige/utils.py:74-85)."""
    if (tensor1.is_cuda and tensor2.is_cuda and tensor1.dtype == tensor2.dtype == torch.float32
            and tensor1.shape == tensor2.shape and tensor1.dim() in (2, 3, 4)):
        from . import hip

        if tensor1.dim() == 4:
            assert tensor1.shape[0] == 1
        return hip.difference_mask(tensor1, tensor2, eps)
    mask = torch.abs(tensor1 - tensor2) > eps
    if mask.dim() == 2:
        return mask
    if mask.dim() == 3:
        return torch.any(mask, 0)
    if mask.dim() == 4:
        assert mask.shape[0] == 1
        return torch.any(mask[0], 0)
    raise NotImplementedError("Unknown mask dimension [%d]!!!" % mask.dim())


def downsample_mask(
    mask: torch.Tensor,
    min_res: IntPair = 4,
    dilation: IntPair = 1,
    threshold: float = 0.3,
    eps: float = 1e-3,
) -> Dict[Tuple[int, int], torch.Tensor]:
    """Mask pyramid {(h,w): bool[h,w]}: repeatedly halve the running FLOAT mask
    with bilinear interpolation (align_corners=False), threshold at
    min(threshold, max - eps), dilate (sige/utils.py:88-118)."""
    assert mask.dim() == 2
    H, W = mask.shape
    min_h, min_w = _pair(min_res)
    if mask.is_cuda and mask.dtype == torch.bool:
        from . import hip

        # the whole pyramid in one launch; the loop below synchronises with the host once per level (level.max()).
        # bool only: the kernel's level 0 is 0.0 / 1.0; `mask.float()` of a 0 / 255 uint8 mask is not
        return hip.mask_pyramid(mask, (min_h, min_w), _pair(dilation), threshold, eps)
    level = mask.reshape(1, 1, H, W).float()
    h, w = H, W
    pyramid = {}
    while True:
        t = min(threshold, level.max() - eps)
        pyramid[(h, w)] = dilate_mask(level[0, 0] > t, dilation)
        h, w = h // 2, w // 2
        if h < min_h and w < min_w:
            return pyramid
        level = F.interpolate(level, (h, w), mode="bilinear", align_corners=False)
