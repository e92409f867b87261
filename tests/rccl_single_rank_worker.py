"""Worker of tests/test_gpu_rccl.py: a ONE-rank RCCL process group on cuda:0 running the collectives sige_amd/parallel.py issues,
with its dtypes and its buffer aliasing (the project's boxes have one GPU: this is as much of RCCL as can execute here; the
multi-rank logic runs over gloo in tests/test_parallel.py).  Prints one JSON line."""
import datetime
import json
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    port = sys.argv[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=60))
    from sige_amd import parallel
    from tests.test_parallel import _build, _inputs

    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    net = _build(affine=True).to(dev)
    blk = net.block  # (the toy block keeps its cached affines as plain attributes: .to() does not move them)
    blk.s1, blk.t1, blk.s2, blk.t2 = (t.to(dev) for t in (blk.s1, blk.t1, blk.s2, blk.t2))
    blk.affine = tuple(t.to(dev) for t in blk.affine)
    orig, _ = _inputs()
    with torch.no_grad():
        net.set_mode("full")
        net(orig.to(dev))
        flat = parallel.pack_caches(net)
    want = flat.clone()
    # what parallel._issue / broadcast_cache / max_over_ranks / _all_ok / choose_distribution call, on this backend
    parallel._issue(flat, 0, "broadcast", None, 1, False)
    w = parallel._issue(flat, 0, "broadcast", None, 1, True)
    w.wait()
    half = flat.to(torch.float16)
    dist.broadcast(half, src=0)
    dist.all_gather_into_tensor(flat, flat[0:flat.numel()])  # in place: rank r's input is chunk r of the output
    w = dist.all_gather_into_tensor(flat, flat[0:flat.numel()], async_op=True)
    w.wait()
    pieces = [flat[lo:lo + flat.numel() // 4] for lo in range(0, flat.numel() - flat.numel() % 4, flat.numel() // 4)]
    works = [dist.broadcast(p, src=0, async_op=True) for p in pieces]  # (distribute_cache_pipelined: chunks issued up front)
    for w in works:
        w.wait()
    v = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    i = torch.tensor([1], dtype=torch.int32, device=dev)
    dist.all_reduce(i, op=dist.ReduceOp.MIN)
    chk = flat.double().sum().reshape(1)
    dist.all_reduce(chk, op=dist.ReduceOp.MIN)
    dist.barrier()
    torch.cuda.synchronize()
    res["cache_unchanged"] = bool(torch.equal(flat, want))
    res["reductions"] = [float(v.item()), int(i.item())]
    # the start-up selection with a device: its bookkeeping tensors live on the GPU under this backend
    choice = parallel.choose_distribution({"broadcast": lambda: parallel._issue(flat, 0, "broadcast", None, 1, False),
                                           "raises": lambda: (_ for _ in ()).throw(RuntimeError("boom"))},
                                          recompute=lambda: None, device=dev)
    res["choice"] = {k: choice[k] for k in ("method_chosen", "methods_ms", "errors")}
    parallel.refresh_derived(net)
    dist.destroy_process_group()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
