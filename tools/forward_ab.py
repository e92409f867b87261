#!/usr/bin/env python3
"""A/B of library builds on the benchmarked forward (select the build with SIGE_HIP_LIB): the DDPM-256 sparse forward as a hipGraph
replay at the given edit ratios, ms per forward = the median over --batches batches of --steps replays; also a checksum of the output
so that two builds can be compared for equality.  One JSON line."""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ratios", default="0.012,0.05,0.15")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batches", type=int, default=7)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    import bench
    from benchlib.common import capture, timed_replays
    from sige_amd import hip
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    hip.lib()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).eval().to(dev).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    if a.dtype != "f32":
        model.set_compute_dtype(a.dtype)
    x0, noise = bench.make_inputs()
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)  # noqa: E731
    t = torch.zeros(1, device=dev)
    res = {"lib": os.path.basename(hip.LIB_PATH), "tag": a.tag, "dtype": a.dtype, "rows": []}
    with torch.no_grad():
        model.set_mode("full")
        model(cl(x0), t)
        for r in [float(v) for v in a.ratios.split(",")]:
            mask = bench.edit_mask(r)
            model.set_masks(downsample_mask(dilate_mask(mask.to(dev), 5), 8))
            model.set_mode("sparse")
            x1 = cl(x0 + noise * mask)
            model(x1, t)
            n0 = hip.launch_count()
            model(x1, t)
            launches = hip.launch_count() - n0
            g, out = capture(model, x1, t)
            ms = [timed_replays(g, a.steps, 5, 1) * 1e3 / a.steps for _ in range(a.batches)]
            res["rows"].append({"ratio": r, "forward_ms": round(statistics.median(ms), 4), "min_ms": round(min(ms), 4), "launches": launches,
                                "checksum": float(out.double().abs().sum())})
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
