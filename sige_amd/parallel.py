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
    return [s[:4] for s in _slots_with_modules(model, derived)]


def _slots_with_modules(model: torch.nn.Module, derived: bool = False) -> List[Tuple]:
    """cache_slots() with the owning module as a fifth element (the pipelined distribution refreshes per module)."""
    slots = []
    attrs = ("original_outputs", "original_residuals") + (("activated_outputs",) if derived else ())
    for m in model.modules():
        for attr in attrs:
            d = getattr(m, attr, None)
            if isinstance(d, dict):
                slots.extend(("dict", d, k, None, m) for k in sorted(d))
        aff = getattr(m, "affine", None)  # workload models keep (scale, shift, ...) tuples here
        if isinstance(aff, dict):
            for k in sorted(aff):
                if isinstance(aff[k], tuple):
                    slots.extend(("dict", aff, k, i, m) for i in range(len(aff[k])))
        elif isinstance(aff, tuple) and all(isinstance(t, torch.Tensor) for t in aff):
            slots.extend(("attr", m, "affine", i, m) for i in range(len(aff)))
        for name in ("scale", "shift", "cached_k", "cached_v"):
            t = m.__dict__.get(name)  # (plain attributes only: not parameters / buffers / sub-modules)
            if isinstance(t, torch.Tensor) and t.dtype == torch.float32:
                slots.append(("attr", m, name, None, m))
    return slots


def _get(slot):
    kind, holder, k, i = slot[:4]
    v = holder[k] if kind == "dict" else getattr(holder, k)
    return v if i is None else v[i]


def _set(slot, value):
    kind, holder, k, i = slot[:4]
    if i is not None:
        cur = list(holder[k] if kind == "dict" else getattr(holder, k))
        cur[i] = value
        value = tuple(cur)
    if kind == "dict":
        holder[k] = value
    else:
        setattr(holder, k, value)


def refresh_derived(model: torch.nn.Module, modules=None) -> None:
    """After the cache tensors were rewritten behind the modules' backs (a collective into the packed buffer):
    re-copy every persistent scatter output (it was a copy of the old cache) and let the model recompute what it
    derives from the cache (`rebuild_derived_caches()` of its modules, e.g. the activated ScatterGather copies) --
    all in place, so that a hipGraph captured before keeps valid addresses.  `modules`: only these (the pipelined
    distribution refreshes a module as soon as the chunks holding its caches have landed)."""
    mods = list(model.modules()) if modules is None else list(modules)
    for m in mods:
        fn = getattr(m, "refresh_outputs", None)
        if fn is not None:
            fn()
    for m in mods:
        fn = getattr(m, "rebuild_derived_caches", None)
        if fn is not None:
            fn()


def pack_caches(model: torch.nn.Module) -> torch.Tensor:
    """Copy every ORIGINAL cache tensor into one flat fp32 buffer and re-point the module
    caches at views of it.  Afterwards writing the buffer (e.g. by a broadcast)
    updates every cache in place (then call refresh_derived, or pass `model` to broadcast_cache)."""
    slots = _slots_with_modules(model)
    if not slots:
        raise RuntimeError("pack_caches: no cached activations -- run the model in `full` mode first")
    sizes = [(_get(s).numel() + _ALIGN - 1) // _ALIGN * _ALIGN for s in slots]
    ref = _get(slots[0])
    total = (sum(sizes) + _PAD - 1) // _PAD * _PAD  # (see _PAD: distribute_cache splits the buffer evenly over the ranks)
    flat = torch.zeros(total, dtype=torch.float32, device=ref.device)
    off = 0
    layout = []  # (owning module, first float, one past its last float) in module order
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
        layout.append((s[4], off, off + size))
        off += size
    for m in model.modules():  # the cache tensors moved: persistent outputs are rebuilt on next use
        bufs = getattr(m, "_out_bufs", None)
        if bufs is not None:
            bufs.invalidate()
    model.__dict__["_sige_cache_layout"] = (flat.data_ptr(), layout)
    return flat


def broadcast_cache(flat: torch.Tensor, src: int = 0, group=None, async_op: bool = False, model: torch.nn.Module = None):
    """ONE collective for the whole cache (RCCL broadcast over xGMI on MI355X).  With `model` (and not
    async) the receiving ranks' derived caches are refreshed right after; with `async_op=True` call
    `refresh_derived(model)` after waiting on the returned work handle."""
    work = dist.broadcast(flat, src=src, group=group, async_op=async_op)
    if model is not None and not async_op:
        refresh_derived(model)
    return work


def _global_rank(group, r: int) -> int:
    """P2P peers and broadcast sources are GLOBAL ranks; `src` / chunk owners here are ranks of `group`."""
    return r if group is None else dist.get_global_rank(group, r)


def distribute_cache(flat: torch.Tensor, src: int = 0, method: str = "broadcast", group=None,
                     model: torch.nn.Module = None) -> None:
    """Rank `src` (a rank of `group`)'s packed cache to every rank of the group; with `model`, the buffers derived from the
    cache (activated copies, persistent outputs) are refreshed afterwards -- for both methods.

    "broadcast": one RCCL broadcast.  "scatter_allgather": rank `src` sends chunk r of the buffer to rank r (seven
    different xGMI links in parallel on an 8-GPU MI355X node), then one in-place all-gather, in which every link of the
    fully connected node carries 1/N of the buffer at the same time -- xGMI is point-to-point, so a broadcast's ring /
    tree moves the WHOLE buffer over each hop's single link (~4.4 ms vs ~1.3 ms ideal for 673 MB).
    `flat.numel()` must be a multiple of the world size for the second form (pack_caches pads it)."""
    world = dist.get_world_size(group)
    if world == 1:
        return
    if method not in ("broadcast", "scatter_allgather"):
        raise ValueError("unknown method %r" % method)
    work = _issue(flat, src, method, group, world, async_op=False)
    assert work is None
    if model is not None:
        refresh_derived(model)


def _issue(buf: torch.Tensor, src: int, method: str, group, world: int, async_op: bool):
    """One collective over `buf` (a whole packed cache or one chunk of it); returns the work handle when async."""
    if method == "broadcast":
        return dist.broadcast(buf, src=_global_rank(group, src), group=group, async_op=async_op)
    if buf.numel() % world:
        raise RuntimeError("scatter_allgather: buffer of %d elements is not a multiple of the world size %d" % (buf.numel(), world))
    rank = dist.get_rank(group)
    chunk = buf.numel() // world
    mine = buf[rank * chunk:(rank + 1) * chunk]
    # scatter as point-to-point sends (rank `src` keeps its own chunk where it is: no self-send, no aliased buffers)
    if rank == src:
        ops = [dist.P2POp(dist.isend, buf[r * chunk:(r + 1) * chunk], _global_rank(group, r), group) for r in range(world) if r != src]
    else:
        ops = [dist.P2POp(dist.irecv, mine, _global_rank(group, src), group)]
    for work in dist.batch_isend_irecv(ops):
        work.wait()  # (RCCL: a stream dependency, not a host wait)
    return dist.all_gather_into_tensor(buf, mine, group=group, async_op=async_op)  # in place: rank r's input is chunk r of the output


def distribute_cache_pipelined(flat: torch.Tensor, model: torch.nn.Module, src: int = 0, method: str = "scatter_allgather",
                               n_chunks: int = 8, group=None) -> dict:
    """The same result as `distribute_cache(..., model=model)`, pipelined: the packed cache is cut into `n_chunks` pieces in
    MODULE ORDER (pack_caches lays the tensors out in the order the forward uses them), every piece's collective is issued
    asynchronously up front, and as soon as piece k has landed the buffers derived from the modules whose caches are
    complete (activated ScatterGather copies, persistent Scatter outputs, activated twins: ~0.65 GB of local copies for
    DDPM-256) are refreshed on the current stream -- while pieces k+1 ... are still moving over xGMI.  What remains exposed
    is the transfer itself plus the last piece's refresh.  Returns {"chunks": n, "refreshed": modules refreshed}."""
    world = dist.get_world_size(group)
    ptr, layout = model.__dict__.get("_sige_cache_layout", (None, None))
    if layout is None or ptr != flat.data_ptr():
        raise RuntimeError("distribute_cache_pipelined: `flat` is not the buffer pack_caches(model) returned")
    if world == 1:
        return {"chunks": 0, "refreshed": 0}
    if method not in ("broadcast", "scatter_allgather"):
        raise ValueError("unknown method %r" % method)
    total = flat.numel()
    gran = world * _ALIGN
    size = max(gran, ((total + n_chunks - 1) // n_chunks + gran - 1) // gran * gran)
    bounds = [(o, min(o + size, total)) for o in range(0, total, size)]
    # every module (and its ancestors: a block's derived caches read its children's) is ready once the chunk holding
    # the END of its last cache tensor has landed
    end = {}
    parents = {}
    for name, m in model.named_modules():
        for cname, c in m.named_children():
            parents[c] = m
    for m, _, e in layout:
        node = m
        while node is not None:
            end[node] = max(end.get(node, 0), e)
            node = parents.get(node)
    ready = [[] for _ in bounds]
    for m in model.modules():
        if getattr(m, "refresh_outputs", None) is None and getattr(m, "rebuild_derived_caches", None) is None:
            continue
        e = end.get(m, 0)
        k = next((i for i, (lo, hi) in enumerate(bounds) if e <= hi), len(bounds) - 1)
        ready[k].append(m)
    works = [_issue(flat[lo:hi], src, method, group, world, async_op=True) for lo, hi in bounds]
    n = 0
    for k, work in enumerate(works):
        if work is not None:
            work.wait()
        refresh_derived(model, ready[k])
        n += len(ready[k])
    return {"chunks": len(bounds), "refreshed": n}


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
