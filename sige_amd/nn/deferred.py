"""Deferred tile tensors: how gather -> conv fusion happens under an UNCHANGED
module API.

The reference's models call `Gather` (or `ScatterGather`) and then, as a separate
module, the conv that consumes the tiles (sige_fused_unet.py:118-125).  To fuse
the two into one MFMA kernel without touching that calling convention, the
sparse-mode Gather / ScatterGather return a `DeferredTiles`: a tensor subclass
with the right shape / dtype / device but no storage, remembering how to produce
its values.

  * consumed by `SIGEConv2d` -> the conv launches `gather_conv` /
    `scatter_gather_conv` (tiles staged straight into LDS, never written to HBM);
  * touched by ANY other torch operation (`tiles * (1 + gamma)`, `.view`,
    `torch.cat`, printing, ...) -> it materialises itself through the ordinary
    gather kernel first and the operation proceeds on the real tensor, so model
    code that post-processes tiles (GauGAN's SPADE modulation, SD attention)
    keeps working unchanged.
"""
from typing import Callable, Optional

import torch
from torch.utils._pytree import tree_map

FORCE_ON_CPU = False  # tests: exercise the mechanism without a GPU
FUSION = True         # tests / tools set this to False to run Gather and the conv as two kernels (A/B of the fusion)


def fusion_enabled() -> bool:
    return FUSION


_METADATA = None


def _metadata_funcs():
    global _METADATA
    if _METADATA is None:
        T = torch.Tensor
        _METADATA = {
            T.shape.__get__, T.dtype.__get__, T.device.__get__, T.ndim.__get__, T.is_cuda.__get__,
            T.requires_grad.__get__, T.layout.__get__, T.is_sparse.__get__, T.is_quantized.__get__,
            T.is_meta.__get__, T.names.__get__, T.grad_fn.__get__, T.is_leaf.__get__,
            T.size, T.dim, T.numel, T.nelement, T.stride, T.is_contiguous, T.is_floating_point, T.is_complex,
            T.element_size, T.get_device, T.__len__,
        }
    return _METADATA


class DeferredTiles(torch.Tensor):
    @staticmethod
    def __new__(cls, shape, dtype, device, thunk: Callable[[], torch.Tensor], spec: dict):
        shape = tuple(shape)
        strides = None
        if spec.get("cl", False) and len(shape) == 4 and shape[1] > 1 and shape[2] * shape[3] > 1:
            # the tiles will come out channels-last: advertise exactly those strides (layout queries on the pending
            # tensor -- is_contiguous(memory_format=...), stride() -- then agree with what materialisation yields)
            n, c, h, w = shape
            strides = (c * h * w, 1, w * c, c)
        t = torch.Tensor._make_wrapper_subclass(cls, shape, strides=strides, dtype=dtype, device=device, requires_grad=False)
        t._cl = strides is not None
        t._thunk = thunk
        t._spec = spec
        t._value = None
        return t

    @property
    def spec(self) -> Optional[dict]:
        """Fusion recipe, or None once the tiles have been materialised."""
        return self._spec if self._value is None else None

    def materialize(self) -> torch.Tensor:
        if self._value is None:
            self._value = self._thunk()
            self._thunk = None
            self._spec = None
        return self._value

    def __repr__(self):
        state = "pending" if self._value is None else "materialized"
        return "DeferredTiles(shape=%s, %s)" % (tuple(self.shape), state)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _metadata_funcs():
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        if func is torch.Tensor.contiguous and len(args) == 1 and not kwargs and not getattr(args[0], "_cl", False):
            return args[0]  # dense NCHW by construction (a channels-last one materialises and converts below)

        def real(a):
            return a.materialize() if isinstance(a, DeferredTiles) else a

        with torch._C.DisableTorchFunctionSubclass():
            return func(*tree_map(real, args), **tree_map(real, kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        def real(a):
            return a.materialize() if isinstance(a, DeferredTiles) else a

        return func(*tree_map(real, args), **tree_map(real, kwargs or {}))


class LazyCat(DeferredTiles):
    """`torch.cat([a, b], dim=1)` that has not run.  The up path of a U-Net concatenates the
    running activation with a skip tensor and hands the result to a block whose first ops are
    Gathers (sige_fused_unet.py:416): the fused gather -> conv kernels read the two tensors
    through two base pointers instead (conv_mfma.hpp), so the 2 x full-tensor copy of the cat
    never happens.  Any other consumer materialises the cat."""

    @staticmethod
    def __new__(cls, a: torch.Tensor, b: torch.Tensor):
        shape = (a.shape[0], a.shape[1] + b.shape[1], a.shape[2], a.shape[3])
        def _cl(p):
            return p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous()

        # channels-last strides are advertised only when BOTH parts are channels-last (torch.cat follows its inputs; a mixed
        # pair materialises as plain NCHW), and the materialised tensor is forced into the advertised format
        cl = _cl(a) and _cl(b)
        fmt = torch.channels_last if cl else torch.contiguous_format
        t = DeferredTiles.__new__(cls, shape, a.dtype, a.device,
                                  lambda: torch.cat([a, b], dim=1).contiguous(memory_format=fmt), dict(kind="cat", cl=cl))
        t.parts = (a, b)
        return t


def lazy_cat(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """cat([a, b], dim=1), deferred when both are GPU tensors of one layout (else eager)."""
    from .. import hip

    if (fusion_enabled() and a.is_cuda and b.is_cuda and a.dtype == b.dtype == torch.float32 and a.dim() == 4
            and a.shape[0] == b.shape[0] and a.shape[2:] == b.shape[2:] and hip.is_cl(a) == hip.is_cl(b)):
        return LazyCat(a, b)
    return torch.cat([a, b], dim=1)


def resolve(x: torch.Tensor) -> torch.Tensor:
    """The real tensor behind `x` (materialising a pending DeferredTiles / LazyCat)."""
    return x.materialize() if isinstance(x, DeferredTiles) else x


def defer_ok(x: torch.Tensor, scale, shift, activation_first: bool, sparse_update: bool,
             activation_name: str = "identity") -> bool:
    """Can the fused gather->conv kernels express this gather?  They stage
    `act(scale * x + shift)` with scale and shift either both absent (and no
    activation) or both present with ONE per-(batch, channel) shape [1|B, 1|C, 1, 1]
    -- what every caller in the reference passes (sige_fused_unet.py:111,118-120)."""
    if not fusion_enabled() or sparse_update or activation_first:
        return False
    if not (x.is_cuda or FORCE_ON_CPU):
        return False
    if (scale is None) != (shift is None):
        return False
    if scale is None:
        return activation_name == "identity"
    for t in (scale, shift):
        if t.dim() != 4 or t.shape[2] != 1 or t.shape[3] != 1:
            return False
    return tuple(scale.shape) == tuple(shift.shape)


def channels_last_ok(x: torch.Tensor, scale=None, shift=None, activation_first: bool = False, cache: bool = False) -> bool:
    """Take the channels-last (NHWC) kernels for this gather?  `x` must be a GPU tensor stored
    channels-last with C % 4 == 0, and the affine per-(batch, channel).  `cache`: `x` is a CACHED tensor, which may be
    stored as fp16 (SIGEModel.set_cache_dtype)."""
    if not x.is_cuda or x.dtype not in ((torch.float32, torch.float16) if cache else (torch.float32,)) or activation_first:
        return False
    from .. import hip

    if not (hip.is_cl(x) and x.shape[1] % 4 == 0):
        return False
    for t in (scale, shift):
        if t is not None and (t.dim() != 4 or t.shape[1] != x.shape[1] or t.shape[2] != 1 or t.shape[3] != 1):
            return False
    return True


def keep_layout(x: torch.Tensor) -> torch.Tensor:
    """Dense copy-free view of `x` for the cache: channels-last tensors stay channels-last,
    everything else becomes plain contiguous (what the reference stores)."""
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
        return x
    return x.contiguous()


def to_cache(x: torch.Tensor, cache_dtype: str = "f32") -> torch.Tensor:
    """What a module keeps of a full-mode output.  "f32": `keep_layout(x)` -- the reference's cache (sige/nn/scatter.py:31-37).
    "f16" (SIGEModel.set_cache_dtype, not in the reference): the same tensor ROUNDED to fp16, same layout -- half the resident
    bytes and half the bytes of a cache broadcast; the kernels that read caches widen on the fly (include/sige_hip.h "_f16")."""
    x = keep_layout(x)
    if cache_dtype == "f16" and x.dtype == torch.float32:
        return x.to(torch.float16, memory_format=torch.preserve_format)
    return x


def from_cache(c: torch.Tensor) -> torch.Tensor:
    """An fp32 view of a cached tensor for a path without an fp16 read kernel (NCHW / CPU-oracle paths: off the fast path)."""
    return c.float() if c.dtype == torch.float16 else c
