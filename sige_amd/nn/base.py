"""Mode / mask / cache plumbing of the sparse-conv modules.

API parity with sige/nn/base.py:10-129 (SIGEModule, SIGEModuleWrapper,
SIGEConv2d, SIGEModel): same constructor arguments, attributes (`mode`,
`runtime`, `mask`, `timestamp`, `cache_id`, `sparse_update`, `devices`,
`supported_dtypes`) and methods, so reference model files work unchanged, incl.
the multiple-inheritance patterns `class X(nn.Conv2d, SIGEModule)` with
`call_super=False` and `class Net(SIGEModel, UNetModel)`.

What differs underneath: native functions come from the backend registry
(sige_amd.runtime) instead of `importlib.import_module("sige.<device>")`, and
SIGEConv2d's sparse mode runs the MFMA stacked-block convolution of
libsige_hip.so instead of F.conv2d.
"""
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn
from torch.nn import functional as F

from .. import runtime as _runtime
from . import deferred

MODES = ("full", "sparse", "profile")


class _RuntimeTable(dict):
    """`module.runtime[device_type]` -> callable or None, resolved lazily through
    the backend registry (so importing the package never needs the GPU library)."""

    def __init__(self, function_name: str):
        super().__init__()
        self.function_name = function_name

    def __missing__(self, device_type: str):
        fn = _runtime.resolve(device_type, self.function_name)
        if fn is not None:  # do not memoise "no backend": a test may register one later
            self[device_type] = fn
        return fn


class SIGEModule(nn.Module):
    def __init__(self, call_super: bool = True):
        if call_super:
            super(SIGEModule, self).__init__()
        self.devices: List[str] = ["cpu", "cuda", "mps"]
        self.supported_dtypes = [torch.float32]
        self.mode: str = "full"
        self.runtime: Dict = {}
        self.mask: Optional[torch.Tensor] = None
        self.timestamp = None
        self.cache_id = 0
        self.sparse_update = False
        self.cache_dtype = "f32"  # SIGEModel.set_cache_dtype: how Scatter / ScatterGather modules STORE their caches

    # -- mask / cache protocol driven by SIGEModel ---------------------------
    def set_mask(self, masks: Dict, cache: Dict, timestamp: int):
        self.timestamp = timestamp

    def set_cache_id(self, cache_id: int):
        self.cache_id = cache_id

    def clear_cache(self):
        pass

    def set_sparse_update(self, sparse_update: bool):
        self.sparse_update = sparse_update

    def set_mode(self, mode: str):
        self.mode = mode

    # -- native function lookup ---------------------------------------------------
    def load_runtime(self, function_name: str, runtime_dict: Dict = None):
        table = _RuntimeTable(function_name)
        if runtime_dict is None:
            self.runtime = table
        return table

    def native(self, table: Dict, x: torch.Tensor):
        """The native function for `x`'s device, or a loud error (no fallback)."""
        fn = table[x.device.type]
        if fn is None:
            raise RuntimeError(
                "[%s] no native backend for device type '%s': the sparse path runs on MI355X "
                "(torch-ROCm 'cuda' tensors) through libsige_hip.so only." % (type(self).__name__, x.device.type))
        return fn

    # -- argument checks (same exceptions as the reference) ------------------------
    def check_dtype(self, *args):
        for x in args:
            if x is not None:
                assert isinstance(x, torch.Tensor)
                if x.dtype not in self.supported_dtypes:
                    raise NotImplementedError(
                        "[%s] does not support dtype [%s]!!! "
                        "Currently supported dtype %s." % (self.__class__.__name__, x.dtype, str(self.supported_dtypes))
                    )

    def check_dim(self, *args):
        for x in args:
            if x is not None:
                assert isinstance(x, torch.Tensor)
                if x.dim() != 4:
                    raise NotImplementedError(
                        "[%s] does not support input with dim [%d]!!!" % (self.__class__.__name__, x.dim())
                    )


class SIGEModuleWrapper:
    """Holds a module without registering it as a child (sige/nn/base.py:75-77)."""

    def __init__(self, module: SIGEModule):
        self.module = module


class SIGEConv2d(nn.Conv2d, SIGEModule):
    """nn.Conv2d whose `sparse`/`profile` modes convolve pre-padded stacked tiles
    with padding 0 (sige/nn/base.py:80-92).

    The sparse mode runs libsige_hip.so's stacked-block conv on fp32 GPU tensors:
    the MFMA implicit GEMM for the tile geometries SIGE produces (3x3/s1 on 6x6,
    1x1 on 4x4, 3x3/s2 on 5x5), the direct vector kernel for any other
    groups / size / dilation.  There is no other path in the product: tiles on a
    device without a registered native backend raise (tests register the CPU oracle,
    whose `conv2d_tiles` is the reference's own F.conv2d call, sige/nn/base.py:88-89).
    """

    def __init__(self, *args, **kwargs):
        nn.Conv2d.__init__(self, *args, **kwargs)
        SIGEModule.__init__(self, call_super=False)
        self._packed = {}
        self._tiles_runtime = self.load_runtime("conv2d_tiles", {})
        # "f32": exact fp32 products (v_mfma_f32_*_f32); "f16": fp16 operands, fp32 accumulation on the 16x faster fp16
        # matrix path (SIGEModel.set_compute_dtype; channels-last tiles only -- BASELINE.json configs[4])
        self.compute_dtype = "f32"

    def _packed_weights(self, x: torch.Tensor, channels_last: bool = True):
        """Packed weights for tiles shaped like `x`; laid out for the f16 matrix path when `compute_dtype` asks for it
        and the consumer is a channels-last kernel."""
        from .. import hip

        from .dense import TILE_X3_MIN_FLOP, _tile_compute

        compute = _tile_compute(self.compute_dtype) if channels_last else "f32"
        if compute == "f16x3":
            # split operands pay off where the launch is matrix-bound; a handful of tiles is start-up-bound and the exact
            # fp32 kernel (two output-channel sub-blocks per workgroup, half the LDS stores) is as fast or faster
            ro, so = (x.shape[2] - self.kernel_size[0]) // self.stride[0] + 1, (x.shape[3] - self.kernel_size[1]) // self.stride[1] + 1
            flop = 2.0 * x.shape[0] * ro * so * self.out_channels * self.in_channels * self.kernel_size[0] * self.kernel_size[1]
            # (a residual block's 1x1 shortcut has exactly 1/9 of its conv1's flop: the same decision for both, so that the two
            #  still share a launch -- the pair kernels exist per operand form)
            if flop * (9 if tuple(self.kernel_size) == (1, 1) else 1) < TILE_X3_MIN_FLOP:
                compute = "f32"
        w = self.weight
        key = (w.data_ptr(), w._version, tuple(w.shape), x.shape[2], x.shape[3], w.device)
        entry = self._packed.get(compute)
        if entry is None or entry[0] != key:
            entry = (key, hip.conv_pack_weights(w, x.shape[2], x.shape[3], self.stride, compute))
            self._packed[compute] = entry
        return entry[1]

    def _block_conv(self, x: torch.Tensor, out_affine=None) -> torch.Tensor:
        from .. import hip

        if out_affine is not None:
            # the consumer's affine + activation: in the fused channels-last kernel's epilogue, else as torch ops
            spec = x.spec if isinstance(x, deferred.DeferredTiles) else None
            packed = self._packed_weights(x) if self.groups == 1 and tuple(self.dilation) == (1, 1) else None
            if (packed is not None and spec is not None and spec["kind"] == "gather" and spec.get("cl", False)
                    and self.out_channels % 4 == 0):
                out = hip.gather_conv_cl(spec["x"], spec.get("x2"), spec["block"], spec["idx"], spec["scale"], spec["shift"],
                                         spec["act"], packed, self.bias, self.out_channels, self.kernel_size, self.stride,
                                         out_affine=out_affine, upsample2x=spec.get("up", False))
                if out is not None:
                    return out
            out = self._block_conv(x)
            os_, oh_, oact = out_affine
            out = out * os_.reshape(1, -1, 1, 1) + oh_.reshape(1, -1, 1, 1)
            return F.silu(out) if oact == "swish" else out

        plain = tuple(self.dilation) == (1, 1)
        spec = x.spec if isinstance(x, deferred.DeferredTiles) else None
        mfma = self.groups == 1 and plain
        if mfma and spec is not None:
            # the producer of the tiles has not run: fuse it into the conv's prologue
            cl = spec.get("cl", False) and self.out_channels % 4 == 0
            packed = self._packed_weights(x, cl)
        if mfma and spec is not None and packed is not None:
            common = (packed, self.bias, self.out_channels, self.kernel_size, self.stride)
            if spec["kind"] == "gather":
                if cl:
                    out = hip.gather_conv_cl(spec["x"], spec.get("x2"), spec["block"], spec["idx"], spec["scale"],
                                             spec["shift"], spec["act"], *common, upsample2x=spec.get("up", False))
                elif spec.get("x2") is None and not spec.get("up", False):
                    out = hip.gather_conv(spec["x"], spec["block"], spec["idx"], spec["scale"], spec["shift"],
                                          spec["act"], *common)
                else:
                    out = None  # (the two-tensor input exists in the channels-last kernels only)
            else:
                f = hip.scatter_gather_conv_cl if cl else hip.scatter_gather_conv
                out = f(spec["x"], spec["y"], spec["block"], spec["idx"], spec["map"], spec["scale"], spec["shift"],
                        spec["act"], *common)
            if out is not None:
                return out  # (None: shape outside the fused kernel's limits -> two-kernel form below)
        x = deferred.resolve(x)
        if mfma:
            if hip.is_cl(x) and x.shape[1] % 4 == 0 and self.out_channels % 4 == 0:
                packed = self._packed_weights(x, True)
                out = None if packed is None else hip.block_conv_cl(x, packed, self.bias, self.out_channels, self.kernel_size,
                                                                    self.stride)
                if out is not None:
                    return out
            packed = self._packed_weights(x, False)
            if packed is not None:
                return hip.block_conv(x, packed, self.bias, self.out_channels, self.kernel_size, self.stride)
        return hip.block_conv_direct(x, self.weight, self.bias, self.stride, self.groups, self.dilation)

    def forward(self, x: torch.Tensor, out_affine=None) -> torch.Tensor:
        """`out_affine` (sparse mode only, not in the reference): (scale, shift, activation) applied to the
        output tiles -- the consumer's cached GroupNorm affine + SiLU, fused into the kernel's epilogue."""
        if self.mode == "full":
            if x.is_cuda and self.compute_dtype != "f32":
                from .dense import full_conv2d

                output = full_conv2d(self, x)  # (one launch on the fp16 matrix cores where the shape has a kernel)
            else:
                output = super(SIGEConv2d, self).forward(x)
        elif self.mode == "sparse":
            self.check_dtype(x)
            if x.is_cuda:
                output = self._block_conv(x, out_affine)
            else:
                # no product path off the GPU: whatever backend a test registered for this device type, or a loud error
                conv = self.native(self._tiles_runtime, x)
                output = conv(deferred.resolve(x), self.weight, self.bias, self.stride, self.dilation, self.groups)
                if out_affine is not None:
                    os_, oh_, oact = out_affine
                    output = output * os_.reshape(1, -1, 1, 1) + oh_.reshape(1, -1, 1, 1)
                    output = F.silu(output) if oact == "swish" else output
        elif self.mode == "profile":
            output = F.conv2d(x, self.weight, self.bias, self.stride, (0, 0), self.dilation, self.groups)
        else:
            raise NotImplementedError("Unknown mode: %s" % self.mode)
        return output


class SIGEModel(nn.Module):
    def __init__(self, call_super: bool = True):
        if call_super:
            super(SIGEModel, self).__init__()
        self.mode = "full"
        self.timestamp = 0

    def _sige_modules(self):
        return (m for m in self.modules() if isinstance(m, SIGEModule))

    def set_masks(self, masks: Dict[Tuple[int, int], torch.Tensor]):
        self.timestamp += 1
        cache = {}  # shared by all modules: one reduce_mask / scatter_map per distinct geometry
        self._prebuild_indices(masks, cache)
        for module in self._sige_modules():
            module.set_mask(masks, cache, self.timestamp)

    def _prebuild_indices(self, masks, cache):
        """GPU masks: the active-index lists of every distinct (resolution, tile geometry) of the network in one batch --
        all compaction kernels are launched back to back and their counts are read with ONE device -> host copy, instead
        of one `torch.nonzero`-style synchronisation per geometry (sige/utils.py:30 via sige/nn/gather.py:101-107)."""
        from .gather import Gather

        keys, reqs = [], []
        for m in self._sige_modules():
            if isinstance(m, Gather) and m.input_res is not None and not m.verbose:
                res = tuple(m.input_res)
                mask = masks.get(res)
                key = m.index_key(res)
                if mask is not None and mask.is_cuda and mask.dim() == 2 and key not in cache and key not in keys:
                    keys.append(key)
                    reqs.append((mask, m.block_size, m.block_stride, m.offset))
        if reqs:
            from .. import hip

            for key, idx in zip(keys, hip.reduce_mask_batch(reqs)):
                cache[key] = idx

    def set_mode(self, mode: str):
        self.mode = mode
        for module in self._sige_modules():
            module.set_mode(mode)

    def clear_cache(self):
        for module in self._sige_modules():
            module.clear_cache()

    def set_cache_id(self, cache_id: int):
        for module in self._sige_modules():
            module.set_cache_id(cache_id)

    def set_sparse_update(self, sparse_update: bool):
        for module in self._sige_modules():
            module.set_sparse_update(sparse_update)

    def set_compute_dtype(self, dtype: str, keep=None, edit_ratio: Optional[float] = None):
        """MI355X-first option (not in the reference, which is fp32-only: sige/nn/base.py:15,55-63).

        "f32" (default): exact fp32 products (v_mfma_f32_*_f32).
        "f16": fp16 operands, fp32 accumulation on the fp16 matrix cores (16x the f32-input rate) -- every tile conv
               (SIGEConv2d on channels-last tiles) and every dense layer routed through sige_amd.nn.dense.fused_conv2d;
               activations and caches stay fp32.  Stated tolerance: sige_amd.tolerance.F16_CRITERION.
        "f16x3": the fp16 matrix cores with every fp32 operand SPLIT into an fp16 hi + lo pair (three products per term,
               22-bit operands): fp32-level results (inside the fp32 path's 1e-3) -- dense layers and the full pass; tile
               convs, which have no such kernel, run exact fp32.

        `keep`: module-name prefixes of convs that stay at the higher precision when dtype is "f16" -- they run "f16x3".
        None = the model's own default (`F16_KEEP`, from its per-layer error trace: tests/f16_error_trace.py) -- applied only
        when `edit_ratio` (the fraction of the image the mask covers, if the caller knows it) is above the model's
        `F16_KEEP_ABOVE`: the fp16 rounding errors the trace attributes to those layers grow with the edited area, and small
        edits meet the criterion with every conv in plain fp16.  () = none."""
        if dtype not in ("f32", "f16", "f16x3"):
            raise ValueError("compute dtype must be 'f32', 'f16' or 'f16x3'")
        if keep is None:
            keep = ()
            if dtype == "f16" and (edit_ratio is None or edit_ratio > getattr(self, "F16_KEEP_ABOVE", 0.0)):
                keep = tuple(getattr(self, "F16_KEEP", ()))
        keep = tuple(keep)

        def kept(name):
            return any(name == k or name.startswith(k + ".") for k in keep)

        self.compute_policy = {"dtype": dtype, "keep": keep}
        for name, module in self.named_modules():
            if isinstance(module, nn.Conv2d):
                module.compute_dtype = "f16x3" if (dtype == "f16" and kept(name)) else dtype

    def set_cache_dtype(self, dtype: str):
        """MI355X-first option (not in the reference, whose caches are fp32: sige/nn/base.py:15,55-63): "f16" STORES the cached
        activations of the full pass -- Scatter / ScatterGather `original_outputs`, ScatterWithBlockResidual `original_outputs` /
        `original_residuals`, the activated ScatterGather copies -- as fp16 (channels-last GPU tensors): half the resident
        cache, half the bytes of the multi-GPU cache distribution with no conversion pass on either side.  Cached GroupNorm
        affines, activations, tiles and outputs stay fp32; the kernels that read a cache widen it exactly (include/sige_hip.h:
        the "_f16" / "_c16" entry points).  Takes effect at the next full-mode forward.  Rounding the cache to fp16 moves a sparse
        output by ~5e-4 ... 3e-3 (DDPM-256, 1 ... 20 % edit: profiles/r3_f16_cache_trace.json): inside the f16 criterion
        (sige_amd.tolerance), outside the fp32 path's 1e-3 at large edits -- an option of the "f16" compute mode."""
        if dtype not in ("f32", "f16"):
            raise ValueError("cache dtype must be 'f32' or 'f16'")
        self.cache_dtype = dtype
        for module in self._sige_modules():
            module.cache_dtype = dtype

    def set_scatter_inplace(self, inplace: bool):
        """MI355X-first option (not in the reference): Scatter / ScatterWithBlockResidual modules whose
        cache is channels-last keep a persistent output buffer and write only the covered pixels into
        it, instead of producing a fresh clone of the cached tensor per call (sige/cpu/scatter.cpp:83).
        Values are identical; the returned tensor is only valid until the module's next forward and
        must not be modified in place by the caller."""
        for module in self._sige_modules():
            if hasattr(module, "inplace"):
                module.inplace = inplace


def paired_convs(like, enabled: bool = True):
    """Context for the two independent convs at the head of a residual block -- the 1x1 shortcut, then conv1 (3x3, cached
    affine + SiLU), both gathering from the block's input: on the GPU (channels-last, fp32) they share ONE launch
    (sige_amd.hip.conv_pair / sige_hip_conv_pair_begin: horizontal fusion); anywhere else, and whenever the pair cannot be
    formed, each conv runs on its own.  Results do not depend on it.  `like`: a tensor (or deferred cat) on the convs' device."""
    import contextlib

    first = like.parts[0] if hasattr(like, "parts") else like
    if enabled and isinstance(first, torch.Tensor) and first.is_cuda:
        from .. import hip

        return hip.conv_pair(first)
    return contextlib.nullcontext()
