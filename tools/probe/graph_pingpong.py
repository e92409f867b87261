#!/usr/bin/env python3
"""Does the ~8.7 us a hipGraph replay spends before its first kernel (profiles/r6_kernel_sequence_*: gap_before of position 0) go away
when consecutive replays are different graph execs?  The benchmarked DDPM-256 sparse forward at a 1.2 % edit, timed four ways:
one graph replayed (what bench.py times), two captures of the same forward replayed alternately, one graph holding two forwards,
and the launch plan's run() (the calls re-issued from C, no graph).  ms per forward, median of --batches batches.  One JSON line."""
import argparse
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--batches", type=int, default=9)
    a = ap.parse_args()
    import bench
    from benchlib.common import capture, capture_fn
    from sige_amd import hip
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    hip.lib()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).eval().to(dev).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    x0, noise = bench.make_inputs()
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)  # noqa: E731
    t = torch.zeros(1, device=dev)

    def timed(fn, forwards_per_call):
        out = []
        for _ in range(a.batches):
            for _ in range(5):
                fn(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                fn(i)
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) * 1e3 / (a.steps * forwards_per_call))
        return round(statistics.median(out), 4)

    res = {}
    with torch.no_grad():
        model.set_mode("full")
        model(cl(x0), t)
        mask = bench.edit_mask(0.012)
        model.set_masks(downsample_mask(dilate_mask(mask.to(dev), 5), 8))
        model.set_mode("sparse")
        x1 = cl(x0 + noise * mask)
        model(x1, t)
        g1, out1 = capture(model, x1, t)
        g2, out2 = capture(model, x1, t)
        g1.replay(); g2.replay(); torch.cuda.synchronize()
        res["same_output"] = bool(torch.equal(out1, out2))
        gg, _ = capture_fn(lambda: (model(x1, t), model(x1, t))[1])
        res["one_graph"] = timed(lambda i: g1.replay(), 1)
        res["two_graphs_alternating"] = timed(lambda i: (g1 if i & 1 else g2).replay(), 1)
        res["two_forwards_per_graph"] = timed(lambda i: gg.replay(), 2)
        res["one_graph_again"] = timed(lambda i: g1.replay(), 1)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
