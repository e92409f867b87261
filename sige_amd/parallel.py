"""Multi-GPU use of the sparse path: one edited image per GPU, one process per
GPU, the original image's activation cache distributed with a single RCCL
broadcast over xGMI.

The reference has no multi-device path at all (no
torch.distributed / NCCL call sites).  What shards naturally is the unit of
work -- an edited image: every sparse forward only READS the cache of the
original image (Scatter/ScatterGather/ScatterWithBlockResidual `original_*`
dicts + the models' cached GroupNorm affines) and there is no exchange inside
the forward (halos come from the local cache).  So:

    flat = pack_caches(model)                  # the ORIGINAL cache tensors -> views of ONE buffer
    broadcast_cache(flat, src=0, model=model)  # one collective, ~0.67 GB for DDPM-256; derived caches
                                               # (activated copies, persistent outputs) are rebuilt locally
    my_edits = shard(range(num_edits))         # independent units, no collective

`torch.distributed` backend "nccl" is RCCL on ROCm; "gloo" works for CPU tests.
"""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist

_ALIGN = 64  # floats; keeps every view 256-byte aligned for 16-byte vector access
_PAD = 64 * 48  # floats: world sizes 1, 2, 3, 4, 6, 8, 12, 16 divide the padded buffer into 256-byte aligned chunks


def cache_slots(model: torch.nn.Module, derived: bool = False) -> List[Tuple]:
    """Where every cached tensor of `model` lives, in module order (identical on every rank that built the same model):
    ("dict", d, key, tuple-index-or-None) or ("attr", module, name, tuple-index-or-None).
    `derived=False` (default) lists only what the full pass produced -- Scatter / ScatterGather / ScatterWithBlockResidual
    `original_*` (incl. the K / V caches of the SD transformer, which are Scatter modules), the models' cached affines
    (`affine` dicts or tuples, `scale` / `shift`) and the cross-attention's `cached_k` / `cached_v` (38 big
    tensors / 673 MB for DDPM-256); `derived=True` adds the activated copies (`activated_outputs`), which every rank can
    recompute from those."""
    slots = []
    attrs = ("original_outputs", "original_residuals") + (("activated_outputs",) if derived else ())
    for m in model.modules():
        for attr in attrs:
            d = getattr(m, attr, None)
            if isinstance(d, dict):
                slots.extend(("dict", d, k, None) for k in sorted(d))
        aff = getattr(m, "affine", None)  # workload models keep (scale, shift, ...) tuples here
        if isinstance(aff, dict):
            for k in sorted(aff):
                if isinstance(aff[k], tuple):
                    slots.extend(("dict", aff, k, i) for i in range(len(aff[k])))
        elif isinstance(aff, tuple) and all(isinstance(t, torch.Tensor) for t in aff):
            slots.extend(("attr", m, "affine", i) for i in range(len(aff)))
        for name in ("scale", "shift", "cached_k", "cached_v"):
            t = m.__dict__.get(name)  # (plain attributes only: not parameters / buffers / sub-modules)
            if isinstance(t, torch.Tensor) and t.dtype == torch.float32:
                slots.append(("attr", m, name, None))
    return slots


def _get(slot):
    kind, holder, k, i = slot
    v = holder[k] if kind == "dict" else getattr(holder, k)
    return v if i is None else v[i]


def _set(slot, value):
    kind, holder, k, i = slot
    if i is not None:
        cur = list(holder[k] if kind == "dict" else getattr(holder, k))
        cur[i] = value
        value = tuple(cur)
    if kind == "dict":
        holder[k] = value
    else:
        setattr(holder, k, value)


def refresh_derived(model: torch.nn.Module) -> None:
    """After the cache tensors were rewritten behind the modules' backs (a collective into the packed buffer):
    re-copy every persistent scatter output (it was a copy of the old cache) and let the model recompute what it
    derives from the cache (`rebuild_derived_caches()` of its modules, e.g. the activated ScatterGather copies) --
    all in place, so that a hipGraph captured before keeps valid addresses."""
    for m in model.modules():
        fn = getattr(m, "refresh_outputs", None)
        if fn is not None:
            fn()
    for m in model.modules():
        fn = getattr(m, "rebuild_derived_caches", None)
        if fn is not None:
            fn()


def pack_caches(model: torch.nn.Module) -> torch.Tensor:
    """Copy every ORIGINAL cache tensor into one flat fp32 buffer and re-point the module
    caches at views of it.  Afterwards writing the buffer (e.g. by a broadcast)
    updates every cache in place (then call refresh_derived, or pass `model` to broadcast_cache)."""
    slots = cache_slots(model)
    if not slots:
        raise RuntimeError("pack_caches: no cached activations -- run the model in `full` mode first")
    sizes = [(_get(s).numel() + _ALIGN - 1) // _ALIGN * _ALIGN for s in slots]
    ref = _get(slots[0])
    total = (sum(sizes) + _PAD - 1) // _PAD * _PAD  # (see _PAD: distribute_cache splits the buffer evenly over the ranks)
    flat = torch.zeros(total, dtype=torch.float32, device=ref.device)
    off = 0
    for s, size in zip(slots, sizes):
        t = _get(s)
        if t.dtype != torch.float32:
            raise NotImplementedError("cache tensors are fp32 (got %s)" % t.dtype)
        if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous():
            # keep channels-last caches channels-last: an NHWC block of the flat buffer seen as [B,C,H,W]
            b, c, h, w = t.shape
            view = flat[off:off + t.numel()].view(b, h, w, c).permute(0, 3, 1, 2)
        else:
            view = flat[off:off + t.numel()].view(t.shape)  # (a non-dense view -- e.g. an expanded affine -- is stored dense)
        view.copy_(t)
        _set(s, view)
        off += size
    for m in model.modules():  # the cache tensors moved: persistent outputs are rebuilt on next use
        bufs = getattr(m, "_out_bufs", None)
        if bufs is not None:
            bufs.invalidate()
    return flat


def broadcast_cache(flat: torch.Tensor, src: int = 0, group=None, async_op: bool = False, model: torch.nn.Module = None):
    """ONE collective for the whole cache (RCCL broadcast over xGMI on MI355X).  With `model` (and not
    async) the receiving ranks' derived caches are refreshed right after; with `async_op=True` call
    `refresh_derived(model)` after waiting on the returned work handle."""
    work = dist.broadcast(flat, src=src, group=group, async_op=async_op)
    if model is not None and not async_op:
        refresh_derived(model)
    return work


def distribute_cache(flat: torch.Tensor, src: int = 0, method: str = "broadcast", group=None) -> None:
    """Rank `src`'s packed cache to every rank.

    "broadcast": one RCCL broadcast.  "scatter_allgather": rank `src` sends chunk r of the buffer to rank r (seven
    different xGMI links in parallel on an 8-GPU MI355X node), then one in-place all-gather, in which every link of the
    fully connected node carries 1/N of the buffer at the same time -- xGMI is point-to-point, so a broadcast's ring /
    tree moves the WHOLE buffer over each hop's single link (~4.4 ms vs ~1.3 ms ideal for 673 MB).
    `flat.numel()` must be a multiple of the world size for the second form (pack_caches pads it)."""
    world = dist.get_world_size(group)
    if world == 1:
        return
    if method == "broadcast":
        dist.broadcast(flat, src=src, group=group)
        return
    if method != "scatter_allgather":
        raise ValueError("unknown method %r" % method)
    if flat.numel() % world:
        raise RuntimeError("scatter_allgather: buffer of %d elements is not a multiple of the world size %d" % (flat.numel(), world))
    rank = dist.get_rank(group)
    chunk = flat.numel() // world
    mine = flat[rank * chunk:(rank + 1) * chunk]
    # scatter as point-to-point sends (rank `src` keeps its own chunk where it is: no self-send, no aliased buffers)
    if rank == src:
        ops = [dist.P2POp(dist.isend, flat[r * chunk:(r + 1) * chunk], r, group) for r in range(world) if r != src]
    else:
        ops = [dist.P2POp(dist.irecv, mine, src, group)]
    for work in dist.batch_isend_irecv(ops):
        work.wait()
    dist.all_gather_into_tensor(flat, mine, group=group)  # in place: rank r's input is chunk r of the output


def max_over_ranks(seconds: float, device=None, group=None) -> float:
    """The slowest rank's time (what a whole-job throughput is quoted on)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(seconds)
    v = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
    return float(v.item())


def shard(units: Sequence, rank: int = None, world: int = None) -> List:
    """Round-robin assignment of independent units (edited images) to ranks."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return [u for i, u in enumerate(units) if i % world == rank]
