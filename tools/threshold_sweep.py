#!/usr/bin/env python3
"""The reference's `sparse_resolution_threshold` (diffusion/configs/church_ddpm256-sige.yml:29 = 64: resolutions below it
run DENSE convs on the cached affine, sige_fused_unet.py:300-368) swept on the benchmark network: 64 is BASELINE.json's
configuration and bench.py's headline; 32 / 16 tile the 32x32 / 16x16 levels as well.  Not the same function of the input
(below the threshold the reference recomputes every pixel, above it only the masked tiles), so this is a configuration study,
not a parity claim: printed are the hipGraph forward time, the library's launches per forward, and max |out - out_64|.

    python tools/threshold_sweep.py [--ratio 0.012]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import bench  # noqa: E402
from sige_amd import hip  # noqa: E402
from sige_amd.utils import dilate_mask, downsample_mask  # noqa: E402
from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ratios", default="0.012,0.05,0.15")
    a = ap.parse_args()
    dev = torch.device("cuda")
    cl = lambda v: v.to(dev).contiguous(memory_format=torch.channels_last)  # noqa: E731
    x0, noise = bench.make_inputs()
    t = torch.zeros(1, device=dev)
    for ratio in [float(r) for r in a.ratios.split(",")]:
        mask = bench.edit_mask(ratio)
        x1 = x0 + noise * mask
        ref = None
        for thr in (64, 32, 16):
            torch.manual_seed(0)
            model = DDPMSparseUNet(DDPMConfig(sparse_threshold=thr)).eval().to(dev).to(memory_format=torch.channels_last)
            model.set_scatter_inplace(True)
            with torch.no_grad():
                model.set_mode("full")
                model(cl(x0), t)
                model.set_masks(downsample_mask(dilate_mask(mask.to(dev), 5), 8))
                model.set_mode("sparse")
                model(cl(x1), t)
                n0 = hip.launch_count()
                model(cl(x1), t)
                launches = hip.launch_count() - n0
                g, out = bench.capture(model, cl(x1), t)
                dt = bench.timed_replays(g, 200, 20, 1)  # seconds for 200 replays
            o = out.float().cpu()
            ref = o if ref is None else ref
            print(json.dumps({"edit_ratio": ratio, "sparse_threshold": thr, "forward_ms": round(dt * 1e3 / 200, 4),
                              "launches_per_forward": int(launches), "max_abs_vs_threshold_64": float("%.3g" % (o - ref).abs().max().item())}),
                  flush=True)
            del model, g, out


if __name__ == "__main__":
    main()
