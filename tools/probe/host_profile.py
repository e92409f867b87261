"""Where does the HOST time of an eager sparse forward go?  cProfile over 30 forwards (DDPM-256, 1.2 % edit).

    python tools/probe/host_profile.py [--sort tottime] [--top 45]"""
import argparse
import cProfile
import os
import pstats
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sort", default="tottime")
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    import bench
    from sige_amd import hip
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    hip.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).to(dev).eval().to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    x0, noise = (t.to(dev).contiguous(memory_format=torch.channels_last) for t in bench.make_inputs())
    t = torch.zeros(1, device=dev)
    with torch.no_grad():
        model.set_mode("full")
        model(x0, t)
        m = bench.edit_mask(0.012).to(dev)
        x1 = x0 + noise * m
        model.set_masks(downsample_mask(dilate_mask(m, 5), 8))
        model.set_mode("sparse")
        for _ in range(3):
            model(x1, t)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(30):
            model(x1, t)
        torch.cuda.synchronize()
        pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats(a.sort).print_stats(a.top)


if __name__ == "__main__":
    main()
