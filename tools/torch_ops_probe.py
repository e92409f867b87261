#!/usr/bin/env python3
"""Which torch (aten) kernels are still inside a sparse forward?  A launch plan (sige_amd/plan.py) only sees library calls, so every
aten op that touches a GPU tensor in the steady-state forward is a launch the plan cannot record.  Runs one eager sparse forward
under a TorchDispatchMode and prints every non-view aten op with a GPU result, with the sige_amd source line that issued it.

    python tools/torch_ops_probe.py [--workload gaugan|sd] [--out gpurun_out/torch_ops.json]
"""
import argparse
import collections
import json
import os
import sys
import traceback

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

VIEWS = ("view", "permute", "slice", "select", "expand", "as_strided", "alias", "detach", "t.", "transpose", "unsqueeze", "squeeze",
         "split", "reshape", "_unsafe_view", "unbind", "narrow", "chunk", "lift_fresh", "_reshape_alias", "sym_", "is_", "size", "stride")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(v in name for v in VIEWS):
            return out
        flat = out if isinstance(out, (tuple, list)) else (out,)
        if any(isinstance(t, torch.Tensor) and t.is_cuda for t in flat):
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if "/sige_amd/" in fr.filename or "/benchlib/" in fr.filename:
                    where = "%s:%d %s" % (os.path.relpath(fr.filename, REPO), fr.lineno, (fr.line or "").strip()[:110])
                    break
            self.rows[(name, where)] += 1
        return out


def gaugan():
    import numpy as np

    from sige_amd.utils import compute_difference_mask, dilate_mask, downsample_mask
    from sige_amd.workloads.gaugan_spade import SPADEConfig, SpadeGenerator

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = SpadeGenerator(SPADEConfig()).eval().to(dev).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    rs = np.random.RandomState(3)
    lab0 = np.kron(rs.randint(0, 36, size=(32, 64)), np.ones((8, 8), dtype=np.int64))
    lab1 = lab0.copy()
    lab1[85:136, 128:256] = (lab0[85:136, 128:256] + 5) % 36
    oh = lambda l: torch.nn.functional.one_hot(torch.from_numpy(l), 36).permute(2, 0, 1)[None].float().to(dev).contiguous(memory_format=torch.channels_last)  # noqa: E731
    x0, x1 = oh(lab0), oh(lab1)
    with torch.no_grad():
        model.set_mode("full")
        model(x0)
        model.set_masks(downsample_mask(dilate_mask(compute_difference_mask(x0, x1), 1), (model.sh, model.sw), dilation=2))
        model.set_mode("sparse")
        for _ in range(3):
            model(x1)
    return lambda: model(x1)


def sd():
    from sige_amd.utils import downsample_mask
    from sige_amd.workloads.sd_unet import SDConfig, SDUNet

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = SDUNet(SDConfig()).eval().to(dev).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)  # noqa: E731
    g = torch.Generator().manual_seed(1)
    x0, noise = cl(torch.randn(2, 4, 64, 64, generator=g)), cl(torch.randn(2, 4, 64, 64, generator=g))
    ctx = torch.randn(2, 77, 768, generator=g).to(dev)
    ts = torch.full((2,), 500.0, device=dev)
    mask512 = torch.zeros(512, 512, dtype=torch.bool, device=dev)
    mask512[150:348, 120:318] = True
    masks = downsample_mask(mask512, min_res=8, dilation=1)
    x1 = cl(x0 + noise * masks[(64, 64)])
    with torch.no_grad():
        model.set_mode("full")
        model(x0, ts, context=ctx)
        model.set_masks(masks)
        model.set_mode("sparse")
        for _ in range(3):
            model(x1, ts, context=ctx)
    return lambda: model(x1, ts, context=ctx)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gaugan", choices=["gaugan", "sd"])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from sige_amd import hip

    fwd = {"gaugan": gaugan, "sd": sd}[a.workload]()
    log = Log()
    n0 = hip.launch_count()
    with torch.no_grad(), log:
        fwd()
    torch.cuda.synchronize()
    rows = [{"op": k[0], "where": k[1], "calls": v} for k, v in sorted(log.rows.items(), key=lambda kv: -kv[1])]
    res = {"workload": a.workload, "library_launches": hip.launch_count() - n0, "aten_ops_with_gpu_results": sum(r["calls"] for r in rows),
           "rows": rows}
    print(json.dumps(res, indent=1))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
