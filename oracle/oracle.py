"""ctypes front-end of oracle/libsige_oracle.so (the plain-C restatement).

TEST INFRASTRUCTURE ONLY -- see oracle/sige_oracle.c.  The functions mirror the
reference's five native entry points (sige/cpu/pybind_cpu.cpp:5-12) with the
same positional signatures, on CPU torch tensors, so a test can use this module
wherever the reference would use ``sige.cpu``.  Extra helpers restate
sige/utils.py (reduce_mask, dilate_mask, downsample_mask) and the stacked-block
convolution (sige/nn/base.py:85-92).
"""
import ctypes
import os
import subprocess
from typing import Dict, Optional, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsige_oracle.so")
_lib = None

ACT = {"identity": 0, "swish": 1}


def build(verbose: bool = False) -> str:
    src = os.path.join(_HERE, "sige_oracle.c")
    if not os.path.isfile(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        cmd = ["gcc", "-O3", "-std=c99", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
               "-o", _LIB_PATH, src, "-lm"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_reduce_mask_i32.restype = ctypes.c_int
        _lib.oracle_version.restype = ctypes.c_int
    return _lib


def _f(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.dtype == torch.float32 and t.device.type == "cpu" and t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def _i(t: torch.Tensor):
    assert t.dtype == torch.int32 and t.device.type == "cpu" and t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def _dims(t: Optional[torch.Tensor]):
    if t is None:
        return (0, 0, 0, 0)
    assert t.dim() == 4
    return tuple(int(v) for v in t.shape)


def gather(x, bSizeH, bSizeW, activeIndices, scale=None, shift=None, activationName="identity",
           activationFirst=False):
    B, C, H, W = x.shape
    N = activeIndices.shape[0]
    out = torch.empty((B * N, C, bSizeH, bSizeW), dtype=torch.float32)
    lib().oracle_gather_f32(_f(x), B, C, H, W, bSizeH, bSizeW, _i(activeIndices), N,
                            _f(scale), *_dims(scale), _f(shift), *_dims(shift),
                            ACT[activationName], int(bool(activationFirst)), _f(out))
    return out


def scatter(x, y, offsetH, offsetW, strideH, strideW, activeIndices, residual=None):
    B, C, H, W = y.shape
    R, S = x.shape[2], x.shape[3]
    N = activeIndices.shape[0]
    out = torch.empty_like(y)
    lib().oracle_scatter_f32(_f(x), _f(y), B, C, H, W, R, S, offsetH, offsetW, strideH, strideW,
                             _i(activeIndices), N, _f(residual), *_dims(residual), _f(out))
    return out


def scatter_with_block_residual(x0, y0, x1, y1, offsetH, offsetW, strideH, strideW,
                                activeIndices0, activeIndices1):
    B, C, H, W = y0.shape
    out = torch.empty_like(y0)
    lib().oracle_scatter_with_block_residual_f32(
        _f(x0), _f(y0), _f(x1), _f(y1), B, C, H, W, x0.shape[2], x0.shape[3], x1.shape[2], x1.shape[3],
        offsetH, offsetW, strideH, strideW, _i(activeIndices0), activeIndices0.shape[0],
        _i(activeIndices1), activeIndices1.shape[0], _f(out))
    return out


def get_scatter_map(H, W, bSizeH, bSizeW, kSizeH, kSizeW, offsetH, offsetW, strideH, strideW, activeIndices):
    out = torch.empty((H, W, 3), dtype=torch.int32)
    lib().oracle_get_scatter_map_i32(H, W, bSizeH, bSizeW, kSizeH, kSizeW, offsetH, offsetW, strideH, strideW,
                                     _i(activeIndices), activeIndices.shape[0], _i(out))
    return out


def scatter_gather(x, y, bSizeH, bSizeW, activeIndices, scatterMap, scale=None, shift=None,
                   activationName="identity", activationFirst=False):
    B, C, H, W = y.shape
    N = activeIndices.shape[0]
    out = torch.empty((B * N, C, bSizeH, bSizeW), dtype=torch.float32)
    lib().oracle_scatter_gather_f32(_f(x), _f(y), B, C, H, W, x.shape[2], x.shape[3], bSizeH, bSizeW,
                                    _i(activeIndices), N, _i(scatterMap),
                                    _f(scale), *_dims(scale), _f(shift), *_dims(shift),
                                    ACT[activationName], int(bool(activationFirst)), _f(out))
    return out


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def reduce_mask(mask: torch.Tensor, block_size, stride, padding) -> torch.Tensor:
    bH, bW = _pair(block_size)
    sH, sW = _pair(stride)
    pH, pW = _pair(padding)
    H, W = mask.shape
    m = mask.to(torch.uint8).contiguous()
    n = lib().oracle_reduce_mask_i32(ctypes.c_void_p(m.data_ptr()), H, W, bH, bW, sH, sW, pH, pW, None, 0)
    idx = torch.empty((n, 2), dtype=torch.int32)
    lib().oracle_reduce_mask_i32(ctypes.c_void_p(m.data_ptr()), H, W, bH, bW, sH, sW, pH, pW, _i(idx), n)
    return idx


def dilate_mask(mask: torch.Tensor, dilation) -> torch.Tensor:
    dH, dW = _pair(dilation)
    H, W = mask.shape
    m = mask.to(torch.uint8).contiguous()
    out = torch.empty_like(m)
    lib().oracle_dilate_mask_u8(ctypes.c_void_p(m.data_ptr()), H, W, dH, dW, ctypes.c_void_p(out.data_ptr()))
    return out.to(torch.bool)


def downsample_mask(mask: torch.Tensor, min_res=4, dilation=1, threshold: float = 0.3,
                    eps: float = 1e-3) -> Dict[Tuple[int, int], torch.Tensor]:
    """sige/utils.py:88-118 restated on top of the C helpers."""
    H, W = mask.shape
    min_h, min_w = _pair(min_res)
    level = mask.to(torch.float32).contiguous()
    h, w = H, W
    masks = {}
    while True:
        bits = torch.empty((h, w), dtype=torch.uint8)
        lib().oracle_threshold_mask_f32(_f(level), h, w, ctypes.c_float(threshold), ctypes.c_float(eps),
                                        ctypes.c_void_p(bits.data_ptr()))
        masks[(h, w)] = dilate_mask(bits.to(torch.bool), dilation)
        h //= 2
        w //= 2
        if h < min_h and w < min_w:
            break
        nxt = torch.empty((h, w), dtype=torch.float32)
        lib().oracle_bilinear_resize_f32(_f(level), level.shape[0], level.shape[1], _f(nxt), h, w)
        level = nxt
    return masks


def block_conv(x, weight, bias, stride, groups=1):
    T, Cin, R, S = x.shape
    Cout, _, kH, kW = weight.shape
    sH, sW = _pair(stride)
    out = torch.empty((T, Cout, (R - kH) // sH + 1, (S - kW) // sW + 1), dtype=torch.float32)
    lib().oracle_block_conv_f32(_f(x), T, Cin, R, S, _f(weight.contiguous()), _f(bias), Cout, kH, kW,
                                sH, sW, groups, _f(out))
    return out


def conv2d_tiles(x, weight, bias, stride, dilation, groups):
    """The stacked-tile convolution exactly as the reference runs it: F.conv2d with padding 0
    (sige/nn/base.py:88-89).  sige_amd's SIGEConv2d has no conv of its own off the GPU; a test that
    registers this module as the "cpu" backend gets the reference's call through this hook."""
    return torch.nn.functional.conv2d(x, weight, bias, stride, (0, 0), dilation, groups)


def as_backend(native):
    """A backend object for sige_amd.runtime.register_backend made of `native`'s five reference functions
    (e.g. oracle/_ref, the reference's compiled sige/cpu) plus the conv2d_tiles hook above."""
    import types

    names = ("gather", "scatter", "scatter_with_block_residual", "scatter_gather", "get_scatter_map")
    return types.SimpleNamespace(conv2d_tiles=conv2d_tiles, **{n: getattr(native, n) for n in names})


def set_num_threads(n: int):
    """OpenMP thread count for the oracle's loops (libgomp is process-global)."""
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass
