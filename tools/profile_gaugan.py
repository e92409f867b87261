#!/usr/bin/env python3
"""Profiling target: K hipGraph replays of the GauGAN SPADE generator's sparse (or dense) forward, after a 0.5 s idle gap
(tools/trace_summary.py isolates the burst):

    rocprofv3 --kernel-trace --output-format csv -d OUT -o g -- python tools/profile_gaugan.py --replays 30 [--mode dense]
"""
import argparse
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")

import torch  # noqa: E402

import bench  # noqa: E402
from sige_amd.utils import compute_difference_mask, dilate_mask, downsample_mask  # noqa: E402
from sige_amd.workloads.gaugan_spade import SPADEConfig, SpadeGenerator  # noqa: E402
from tests.golden.model_init import gaugan_labels, init_by_name  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replays", type=int, default=30)
    ap.add_argument("--mode", default="sparse", choices=["sparse", "dense"])
    ap.add_argument("--chain", action="store_true", help="module chain instead of the fused SPADE modulation")
    a = ap.parse_args()
    dev = torch.device("cuda")
    model = SpadeGenerator(SPADEConfig(fused=not a.chain)).eval()
    init_by_name(model)
    x0, x1 = gaugan_labels()
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)  # noqa: E731
    model = model.to(dev).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    x0, x1 = cl(x0), cl(x1)
    with torch.no_grad():
        model.set_mode("full")
        model(x0)
        if a.mode == "sparse":
            model.set_masks(downsample_mask(dilate_mask(compute_difference_mask(x0, x1), 1), (model.sh, model.sw), dilation=2))
            model.set_mode("sparse")
        ms, out, g = bench._replay_ms(lambda: model(x1), k=5, warm=2)
        torch.cuda.synchronize()
        time.sleep(0.5)
        for _ in range(a.replays):
            g.replay()
        torch.cuda.synchronize()
    print("done", a.mode, ms)


if __name__ == "__main__":
    main()
