#!/usr/bin/env python3
"""Profiling target: K hipGraph replays of the SD v1 U-Net sparse forward (bench.py --workload sd shapes), after a 0.5 s idle
gap:  rocprofv3 --kernel-trace --output-format csv -d OUT -o s -- python tools/profile_sd.py --replays 10"""
import argparse
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")

import torch  # noqa: E402

import bench  # noqa: E402
from sige_amd.utils import downsample_mask  # noqa: E402
from sige_amd.workloads.sd_unet import SDConfig, SDUNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replays", type=int, default=10)
    ap.add_argument("--mc", type=int, default=320)
    ap.add_argument("--no-tuned-gemms", action="store_true")
    a = ap.parse_args()
    if not a.no_tuned_gemms:  # (as bench.py --workload sd runs it)
        from sige_amd.workloads import gemm_tuning

        print("tuned token GEMMs:", gemm_tuning.enable_tuned_gemms())
    dev = torch.device("cuda")
    torch.backends.cudnn.benchmark = True  # (as bench.py --workload sd)
    torch.manual_seed(0)
    model = SDUNet(SDConfig(model_channels=a.mc)).eval().to(dev).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    gen = torch.Generator().manual_seed(1)
    cl = lambda t_: t_.to(dev).contiguous(memory_format=torch.channels_last)  # noqa: E731
    x0, noise = cl(torch.randn(2, 4, 64, 64, generator=gen)), cl(torch.randn(2, 4, 64, 64, generator=gen))
    ctx = torch.randn(2, 77, 768, generator=gen).to(dev)
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
        g, out = bench.capture_fn(lambda: model(x1, ts, context=ctx))
        g.replay()
        torch.cuda.synchronize()
        time.sleep(0.5)
        for _ in range(a.replays):
            g.replay()
        torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
