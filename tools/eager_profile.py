#!/usr/bin/env python3
"""Where does the host time of an EAGER sparse forward go?  (4.7 ms for 102 launches = ~45 us of Python per launch; a launch plan
issues the same launches from C in 1.4 ms.)  cProfile over 30 eager forwards of the DDPM-256 U-Net at 1.2 % edit.

    python tools/eager_profile.py [--out gpurun_out/eager_profile.txt]
"""
import argparse
import cProfile
import io
import os
import pstats
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    dev = torch.device("cuda:0")
    cl = lambda t_: t_.contiguous(memory_format=torch.channels_last)  # noqa: E731
    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).eval().to(dev).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    x0, noise = bench.make_inputs()
    x0, noise, t = cl(x0.to(dev)), cl(noise.to(dev)), torch.zeros(1, device=dev)
    mask = bench.square_mask(0.012).to(dev)
    with torch.no_grad():
        model.set_mode("full")
        model(x0, t)
        model.set_masks(downsample_mask(dilate_mask(mask, 5), 8))
        model.set_mode("sparse")
        x1 = x0 + noise * mask
        for _ in range(5):
            model(x1, t)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(30):
            model(x1, t)
        torch.cuda.synchronize()
        pr.disable()
    buf = io.StringIO()
    st = pstats.Stats(pr, stream=buf)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(35)
    text = buf.getvalue()
    print(text[:9000])
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
