"""Multi-process path on CPU (gloo, world_size 2): rank 0 owns the original image's
cache, one broadcast distributes it, each rank runs its own edit."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(affine=False):
    from tests.test_host_logic import ResNet

    torch.manual_seed(11)
    net = ResNet(8, 12).eval()
    blk = net.block
    blk.s1, blk.t1 = torch.randn(1, 8, 1, 1), torch.randn(1, 8, 1, 1)
    blk.s2, blk.t2 = torch.randn(1, 12, 1, 1), torch.randn(1, 12, 1, 1)
    if affine:  # (what a workload model keeps per block: the cached GroupNorm affines travel with the activation caches)
        blk.affine = (blk.s1.clone(), blk.t1.clone(), blk.s2.clone(), blk.t2.clone())
    return net


def _inputs():
    g = torch.Generator().manual_seed(5)
    orig = torch.randn(1, 8, 32, 32, generator=g)
    edits = []
    for r in range(4):
        m = torch.zeros(32, 32, dtype=torch.bool)
        m[4 + 5 * r:9 + 5 * r, 6 + 3 * r:14 + 3 * r] = True
        edits.append((m, orig + torch.randn(1, 8, 32, 32, generator=g) * m))
    return orig, edits


def _sparse(net, mask, x):
    from sige_amd.utils import dilate_mask

    net.set_masks({(32, 32): dilate_mask(dilate_mask(mask, (2, 0)), (0, 2))})
    net.set_mode("sparse")
    return net(x)


def _worker(rank, world, port, out_dir, method="broadcast"):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from sige_amd import parallel, runtime

    runtime.register_backend("cpu", oracle)
    net = _build(affine=method.startswith(("f16wire:", "f16cache:")))
    if method.startswith("f16cache:"):
        net.set_cache_dtype("f16")  # (the caches are STORED as fp16: one fp16 buffer, nothing converted around the collective)
    orig, edits = _inputs()
    with torch.no_grad():
        net.set_mode("full")
        # rank 0 holds the true original; the others only need the cache SLOTS (shapes)
        net(orig if rank == 0 else torch.zeros_like(orig))
        flat = parallel.pack_caches(net)
        if method == "broadcast":
            parallel.broadcast_cache(flat, src=0, model=net)
        elif method.startswith("f16wire:"):
            # activations as fp16 on the wire, the cached affines in fp32; the source rounds its own cache the same way
            parallel._WIRE_SMALL = 1000  # (this toy network's activations are 8 192 elements: above, its affines 8 / 12: below)
            if rank != 0:
                flat.fill_(7.0)  # (nothing of the receivers' own values may survive)
            before = flat.clone()
            kind = method.split(":")[1]
            if kind == "pipelined":
                info = parallel.distribute_cache_pipelined(flat, net, src=0, method="scatter_allgather", n_chunks=3, wire_dtype=torch.float16)
                assert info["wire_bytes"] < flat.numel() * 4 * 0.6
            else:
                parallel.distribute_cache(flat, src=0, method=kind, model=net, wire_dtype=torch.float16)
            if rank == 0:
                assert not torch.equal(flat, before)  # (the source's own cache is rounded too)
        elif method.startswith("f16cache:"):
            assert flat.dtype == torch.float16
            if rank != 0:
                flat.fill_(3.0)
            kind = method.split(":")[1]
            if kind == "pipelined":
                info = parallel.distribute_cache_pipelined(flat, net, src=0, method="scatter_allgather", n_chunks=3)
                assert info["wire_bytes"] == flat.numel() * 2
            else:
                parallel.distribute_cache(flat, src=0, method=kind, model=net)
        elif method.startswith("pipelined:"):
            info = parallel.distribute_cache_pipelined(flat, net, src=0, method=method.split(":")[1], n_chunks=3)
            assert info["chunks"] >= 2 and info["refreshed"] >= 1
        elif method == "subgroup":
            # a group that is NOT the default one and whose rank 0 is global rank 1: `src` is a rank of the group
            grp = dist.new_group(ranks=[1, 0] if world == 2 else list(reversed(range(world))))
            src_in_group = dist.get_group_rank(grp, 0)
            parallel.distribute_cache(flat, src=src_in_group, method="scatter_allgather", group=grp, model=net)
        else:
            parallel.distribute_cache(flat, src=0, method=method, model=net)
        mine = parallel.shard(list(range(len(edits))))
        outs = {i: _sparse(net, *edits[i]) for i in mine}
    slowest = parallel.max_over_ranks(0.25 * (rank + 1))
    torch.save({"outs": outs, "flat_sum": float(flat.double().sum()) if flat.dtype == torch.float32 else parallel.checksum(flat),
                "slowest": slowest, "numel": flat.numel()},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("method", ["broadcast", "scatter_allgather", "pipelined:broadcast", "pipelined:scatter_allgather", "subgroup"])
def test_cache_broadcast_and_sharded_edits(tmp_path, method):
    """The multi-rank path of bench.py (pack -> one collective -> local refresh -> sharded edits -> max over ranks),
    both distribution methods, on gloo."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), method), nprocs=world, join=True)
    res = [torch.load(tmp_path / ("rank%d.pt" % r)) for r in range(world)]
    assert res[0]["flat_sum"] == res[1]["flat_sum"]  # identical caches after the collective
    assert res[0]["slowest"] == res[1]["slowest"] == 0.5  # MAX over ranks
    assert res[0]["numel"] % (64 * 48) == 0  # padded: 2 / 4 / 8 ranks get equal, aligned chunks
    assert sorted(res[0]["outs"]) == [0, 2] and sorted(res[1]["outs"]) == [1, 3]

    # single-process ground truth
    from oracle import oracle
    from sige_amd import runtime

    runtime.register_backend("cpu", oracle)
    try:
        net = _build()
        orig, edits = _inputs()
        with torch.no_grad():
            for i, (m, x) in enumerate(edits):
                net.set_mode("full")
                dense = net(x)
                net(orig)
                want = _sparse(net, m, x)
                got = res[i % world]["outs"][i]
                assert torch.equal(got, want)
                torch.testing.assert_close(got, dense, rtol=0, atol=1e-4)
    finally:
        runtime.unregister_backend("cpu")


def test_scatter_allgather_with_eight_ranks(tmp_path):
    """The driver's 8-GPU shape of the cache distribution: rank 0 sends seven different chunks (point to point), then one
    in-place all-gather -- chunk indexing and the padding of the packed buffer for a world size of 8."""
    world = 8
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "scatter_allgather"), nprocs=world, join=True)
    res = [torch.load(tmp_path / ("rank%d.pt" % r)) for r in range(world)]
    assert len({r["flat_sum"] for r in res}) == 1 and res[0]["flat_sum"] != 0.0
    assert res[0]["numel"] % world == 0 and {r["slowest"] for r in res} == {2.0}
    assert [sorted(r["outs"]) for r in res] == [[0], [1], [2], [3], [], [], [], []]


def test_pack_caches_keeps_results_and_aliases_buffer():
    from oracle import oracle
    from sige_amd import parallel, runtime

    runtime.register_backend("cpu", oracle)
    try:
        net = _build()
        orig, edits = _inputs()
        with torch.no_grad():
            net.set_mode("full")
            net(orig)
            before = _sparse(net, *edits[0])
            flat = parallel.pack_caches(net)
            after = _sparse(net, *edits[0])
            assert torch.equal(before, after)
            flat.zero_()  # the caches ARE the buffer now
            assert float(net.block.scatter.original_outputs[0].abs().sum()) == 0.0
    finally:
        runtime.unregister_backend("cpu")


def test_pack_caches_requires_full_pass():
    from sige_amd import parallel

    with pytest.raises(RuntimeError, match="full"):
        parallel.pack_caches(_build())


@pytest.mark.parametrize("kind", ["broadcast", "scatter_allgather", "pipelined"])
def test_cache_distribution_with_fp16_on_the_wire(tmp_path, kind):
    """`wire_dtype=torch.float16`: every rank -- the source included -- ends up with the source's cache rounded to fp16 (big
    tensors) resp. exact (tensors below 65 536 elements: the cached affines), so the ranks' outputs are bit-identical to a
    single process that rounds its cache the same way."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "f16wire:" + kind), nprocs=world, join=True)
    res = [torch.load(tmp_path / ("rank%d.pt" % r)) for r in range(world)]
    assert res[0]["flat_sum"] == res[1]["flat_sum"]
    from oracle import oracle
    from sige_amd import parallel, runtime

    runtime.register_backend("cpu", oracle)
    try:
        net = _build(affine=True)
        orig, edits = _inputs()
        with torch.no_grad():
            net.set_mode("full")
            net(orig)
            flat = parallel.pack_caches(net)
            exact = flat.clone()
            _, layout = net.__dict__["_sige_cache_layout"]
            assert any(n < 1000 for _, _, _, n in layout) and any(n >= 1000 for _, _, _, n in layout)
            for _, off, _, n in layout:
                if n >= 1000:
                    flat[off:off + n].copy_(flat[off:off + n].half().float())
            assert not torch.equal(flat, exact)  # (this network's caches are above the threshold: something was rounded)
            parallel.refresh_derived(net)
            assert float(flat.double().sum()) == res[0]["flat_sum"]
            for i, (m, x) in enumerate(edits):
                assert torch.equal(res[i % world]["outs"][i], _sparse(net, m, x))
    finally:
        runtime.unregister_backend("cpu")


@pytest.mark.parametrize("kind", ["broadcast", "scatter_allgather", "pipelined"])
def test_cache_distribution_of_an_fp16_stored_cache(tmp_path, kind):
    """SIGEModel.set_cache_dtype("f16"): the packed cache IS fp16 (fp16 activations as they are, the fp32 affines bit for bit in
    two elements each); one collective, no conversion on either side; every rank ends with identical bits and computes what a
    single process with the same fp16-stored cache computes."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "f16cache:" + kind), nprocs=world, join=True)
    res = [torch.load(tmp_path / ("rank%d.pt" % r)) for r in range(world)]
    assert res[0]["flat_sum"] == res[1]["flat_sum"]  # cache_identical_on_all_ranks
    from oracle import oracle
    from sige_amd import parallel, runtime

    runtime.register_backend("cpu", oracle)
    try:
        net = _build(affine=True)
        net.set_cache_dtype("f16")
        orig, edits = _inputs()
        with torch.no_grad():
            net.set_mode("full")
            net(orig)
            flat = parallel.pack_caches(net)
            assert flat.dtype == torch.float16 and parallel.checksum(flat) == res[0]["flat_sum"]
            for i, (m, x) in enumerate(edits):
                assert torch.equal(res[i % world]["outs"][i], _sparse(net, m, x))
    finally:
        runtime.unregister_backend("cpu")


# ---- which distribution does a job use?  measured once at start-up, the same decision on every rank --------
def _choose_worker(rank, world, port, out_dir):
    import time

    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sige_amd import parallel

    buf = torch.full((1024,), float(rank))
    log = []

    def broadcast():
        log.append("broadcast")
        dist.broadcast(buf, src=0)

    def slow_on_one_rank():  # a collective that crawls on ONE rank: the max over ranks is what the watchdog sees
        log.append("slow")
        if rank == world - 1:
            time.sleep(0.6)
        dist.broadcast(buf, src=0)

    def broken_on_one_rank():  # a method one rank cannot run (the others can): every rank must drop it
        log.append("broken")
        if rank == 1:
            raise RuntimeError("unsupported here")

    def recompute():
        log.append("recompute")
        time.sleep(0.05)
        buf.fill_(0.0)

    res = {}
    # 1: a healthy collective beats a slower recompute; the slow and the broken ones are dropped after ONE run
    res["healthy"] = parallel.choose_distribution({"slow": slow_on_one_rank, "broken": broken_on_one_rank, "broadcast": broadcast},
                                                  recompute=recompute, watchdog_s=0.3, sync=lambda: None)
    res["healthy_log"] = list(log)
    del log[:]
    # 2: every collective fails or trips the watchdog: fall back to recompute, and say why
    res["fallback"] = parallel.choose_distribution({"slow": slow_on_one_rank, "broken": broken_on_one_rank},
                                                   recompute=recompute, watchdog_s=0.3, sync=lambda: None)
    # 3: recompute wins on merit (no fallback reason) when it is simply faster
    res["merit"] = parallel.choose_distribution({"slow": slow_on_one_rank}, recompute=recompute, watchdog_s=5.0, sync=lambda: None)
    # 4: nothing works and there is no recompute: every rank raises
    try:
        parallel.choose_distribution({"broken": broken_on_one_rank}, recompute=None, watchdog_s=0.3, sync=lambda: None)
        res["nothing"] = "no error"
    except RuntimeError as e:
        res["nothing"] = str(e)
    torch.save(res, os.path.join(out_dir, "choose%d.pt" % rank))
    dist.destroy_process_group()


def test_choose_distribution_eight_ranks(tmp_path):
    world = 8
    mp.spawn(_choose_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / ("choose%d.pt" % r)) for r in range(world)]
    for key in ("healthy", "fallback", "merit"):
        assert len({r[key]["method_chosen"] for r in res}) == 1, key          # one decision
        assert all(r[key]["methods_ms"] == res[0][key]["methods_ms"] for r in res)  # from the same (max-over-ranks) numbers
    h = res[0]["healthy"]
    assert h["method_chosen"] == "broadcast" and h["fallback"] is None
    assert "watchdog" in h["errors"]["slow"] and h["methods_ms"]["slow"] >= 600 and h["methods_ms"]["broken"] is None
    assert h["methods_ms"]["broadcast"] < h["methods_ms"]["recompute"]
    # dropped candidates ran once, survivors `repeats` times -- on every rank, also the rank that could have run `broken`
    for r in res:
        assert r["healthy_log"] == ["slow", "broken", "broadcast", "broadcast", "recompute", "recompute"]
    f = res[0]["fallback"]
    assert f["method_chosen"] == "recompute" and "watchdog" in f["fallback"] and "broken" in f["fallback"]
    m = res[0]["merit"]
    assert m["method_chosen"] == "recompute" and m["fallback"] is None and m["methods_ms"]["slow"] >= 600
    assert all("no method worked" in r["nothing"] for r in res)


# ---- a collective that HANGS, and one that raises on a rank while the others are already inside it -------------------
def _hang_worker(rank, world, port, out_dir):
    import time

    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sige_amd import parallel

    buf = torch.full((1024,), float(rank))
    log = []

    def raises_inside():  # rank 1 fails AFTER the others have entered the collective: they wait for a peer that never comes
        log.append("raises_inside")
        if rank == 1:
            time.sleep(0.2)
            raise RuntimeError("died inside")
        dist.broadcast(buf, src=0)

    def never_tried():
        log.append("never_tried")
        dist.broadcast(buf, src=0)

    def recompute():
        log.append("recompute")
        buf.fill_(0.0)

    t0 = time.perf_counter()
    res = parallel.choose_distribution({"raises_inside": raises_inside, "never_tried": never_tried}, recompute=recompute,
                                       watchdog_s=0.3, sync=lambda: None, hang_timeout_s=1.5)
    res["seconds"] = time.perf_counter() - t0
    res["log"] = list(log)
    torch.save(res, os.path.join(out_dir, "hang%d.pt" % rank))
    os._exit(0)  # (the trial thread of the other ranks is still blocked inside gloo's broadcast: no orderly teardown)


def test_choose_distribution_survives_a_hung_collective(tmp_path):
    """The watchdog of round 5 was only read after a candidate returned.  Now: rank 1 raises while ranks 0 and 2 are inside the
    broadcast (which therefore never completes on them) -- every rank is out of the trial after hang_timeout_s, learns through
    the store that a rank hung, does not touch the communicator again (`never_tried` is skipped) and falls back to recompute."""
    world = 3
    ctx = mp.spawn(_hang_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=False)
    deadline = 60.0
    import time

    t0 = time.time()
    while time.time() - t0 < deadline and not all((tmp_path / ("hang%d.pt" % r)).exists() for r in range(world)):
        time.sleep(0.2)
    time.sleep(0.5)
    for p_ in ctx.processes:
        if p_.is_alive():
            p_.terminate()
    res = [torch.load(tmp_path / ("hang%d.pt" % r)) for r in range(world)]
    for r in res:
        assert r["method_chosen"] == "recompute" and r["poisoned"] is True
        assert r["methods_ms"]["raises_inside"] is None and r["methods_ms"]["never_tried"] is None
        assert "skipped" in r["errors"]["never_tried"]
        assert "hung" in r["fallback"] or "no return" in r["fallback"] or "died inside" in r["fallback"]
        assert r["log"] == ["raises_inside", "recompute", "recompute"]
        assert r["seconds"] < 20.0
    assert "died inside" in res[1]["errors"]["raises_inside"]
    assert "no return" in res[0]["errors"]["raises_inside"]
