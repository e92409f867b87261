"""Stacked edits ("throughput mode"): E edited versions of ONE original image, each with its own mask, through ONE set of
launches (include/sige_hip.h: sige_hip_set_edit_batch).

The reference's batch dimension shares one mask (sige/cpu/gather.cpp:17-21: tile t of image b is index n of the ONE list), so E
different edits mean E forwards of ~100 launch-bound launches each.  Here the E images are stacked along H: a channels-last
activation [E,C,H,W] IS the tall image [1,C,E*H,W] (the same bytes), the E masks stacked the same way are one mask, the cached
tensors of the original are repeated E times (288 GB of HBM: 8 x 673 MB is nothing) -- and every sparse module then works on the
tall image UNCHANGED: one index list, one scatter map, one tile table, one persistent output per module, one launch per layer
that sees the tiles of all E edits.  The kernels only need to know where one image ends (a halo row across a seam is zero
padding): hip.set_edit_batch(E).  Whole-image ops (first / last conv, attention, the output GroupNorm) run with batch E on the
same memory.

    stacked.stack_caches(model, E)              # after the full pass on the original (B = 1)
    stacked.set_masks(model, [pyramid_of_edit_0, ..., pyramid_of_edit_{E-1}])
    with stacked.edit_batch(model, E):
        out = model(x_edits, t)                 # x_edits [E,3,H,W] -> out [E,3,H,W]
    stacked.unstack_caches(model)               # back to single edits
"""
from typing import Dict, List

import torch

from . import hip


def tall(x: torch.Tensor) -> torch.Tensor:
    """[E,C,H,W] channels-last -> [1,C,E*H,W] channels-last: a view of the same bytes."""
    E, C, H, W = x.shape
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    return x.permute(0, 2, 3, 1).reshape(1, E * H, W, C).permute(0, 3, 1, 2)


def untall(x: torch.Tensor, E: int) -> torch.Tensor:
    """[1,C,E*H,W] channels-last -> [E,C,H,W] channels-last: a view of the same bytes."""
    _, C, EH, W = x.shape
    if EH % E:
        raise ValueError("untall: height %d is not a multiple of the edit batch %d" % (EH, E))
    return x.permute(0, 2, 3, 1).reshape(E, EH // E, W, C).permute(0, 3, 1, 2)


def stack_masks(pyramids: List[Dict]) -> Dict:
    """Per-edit mask pyramids {(h, w): bool [h,w]} -> the pyramid of the tall image {(E*h, w): bool [E*h, w]}."""
    keys = list(pyramids[0])
    for p in pyramids:
        if list(p) != keys:
            raise ValueError("stack_masks: the pyramids must have the same levels")
    E = len(pyramids)
    return {(E * h, w): torch.cat([p[(h, w)] for p in pyramids], dim=0).contiguous() for (h, w) in keys}


def set_masks(model: torch.nn.Module, pyramids: List[Dict]) -> None:
    """`model.set_masks` for E edits: the stacked pyramid, with the library told about the seams while the index lists are built
    -- a candidate tile only looks at ITS image's mask rows, so the lists are exactly the per-edit lists one after the other (a
    window reaching into the neighbour's mask would activate a tile that the edit's own single forward leaves cached)."""
    E = len(pyramids)
    prev = hip.get_edit_batch()
    hip.set_edit_batch(E)
    try:
        model.set_masks(stack_masks(pyramids))
    finally:
        hip.set_edit_batch(prev)


def _cache_dicts(model):
    for m in model.modules():
        for name in ("original_outputs", "original_residuals", "activated_outputs"):
            d = getattr(m, name, None)
            if isinstance(d, dict):
                yield m, name, d


def stack_caches(model: torch.nn.Module, E: int) -> None:
    """Repeat every cached activation of `model` (the full pass of ONE original, B = 1) E times along H, and tell the Gather /
    Scatter modules their tensors are E times as tall.  The cached affines are per channel and stay as they are."""
    if getattr(model, "_sige_stacked", None):
        raise RuntimeError("stack_caches: the caches are stacked already (unstack_caches first)")
    saved = []
    for m, name, d in _cache_dicts(model):
        for k, t in list(d.items()):
            if t.dim() != 4 or t.shape[0] != 1:
                raise NotImplementedError("stack_caches: caches of one original image ([1,C,H,W]); got %s" % (tuple(t.shape),))
            saved.append((d, k, t))
            cl = t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()
            rep = t.repeat(1, 1, E, 1)
            d[k] = rep.contiguous(memory_format=torch.channels_last) if cl else rep.contiguous()
    res = []
    for m in model.modules():
        for name in ("input_res", "output_res"):
            r = m.__dict__.get(name)
            if r is not None and len(r) == 2:
                res.append((m, name, r))
                setattr(m, name, torch.Size((E * r[0], r[1])))
        bufs = getattr(m, "_out_bufs", None)
        if bufs is not None:
            bufs.invalidate()
        tw = getattr(m, "twins", None)
        if tw is not None and hasattr(tw, "invalidate"):
            tw.invalidate()
        drop = getattr(m, "_drop_twin_links", None)
        if drop is not None:
            drop()
    model.__dict__["_sige_stacked"] = (E, saved, res)


def unstack_caches(model: torch.nn.Module) -> None:
    st = model.__dict__.pop("_sige_stacked", None)
    if not st:
        return
    _, saved, res = st
    for d, k, t in saved:
        d[k] = t
    for m, name, r in res:
        setattr(m, name, r)
    for m in model.modules():
        bufs = getattr(m, "_out_bufs", None)
        if bufs is not None:
            bufs.invalidate()
        tw = getattr(m, "twins", None)
        if tw is not None and hasattr(tw, "invalidate"):
            tw.invalidate()
        drop = getattr(m, "_drop_twin_links", None)
        if drop is not None:
            drop()


class edit_batch:
    """`with edit_batch(model, E):` -- the library knows the seams (hip.set_edit_batch) and the model's whole-image ops run with
    batch E (model.edit_batch, read by the workload's forward)."""

    def __init__(self, model, E: int):
        self.model, self.E = model, int(E)

    def __enter__(self):
        self.prev = hip.get_edit_batch()
        self.prev_model = getattr(self.model, "edit_batch", 1)
        hip.set_edit_batch(self.E)
        for m in self.model.modules():
            if hasattr(m, "edit_batch"):
                m.edit_batch = self.E
        return self

    def __exit__(self, *exc):
        hip.set_edit_batch(self.prev)
        for m in self.model.modules():
            if hasattr(m, "edit_batch"):
                m.edit_batch = self.prev_model
        return False
