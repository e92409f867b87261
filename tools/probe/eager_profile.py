"""Where does the host time of an EAGER sparse forward go (4.4 ms for 102 launches = 43 us of Python per launch)?  cProfile over
20 eager forwards of the DDPM-256 workload; the top functions by own time and by cumulative time.

    python tools/probe/eager_profile.py [--out gpurun_out/eager_profile.txt]
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    args = ap.parse_args()
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
        for _ in range(5):
            model(x1, t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            model(x1, t)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 20 * 1e3
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(20):
            model(x1, t)
        torch.cuda.synchronize()
        pr.disable()
    buf = io.StringIO()
    buf.write("eager sparse forward: %.3f ms wall per forward (20 forwards, unprofiled)\n\n" % wall)
    for key in ("tottime", "cumulative"):
        ps = pstats.Stats(pr, stream=buf).strip_dirs().sort_stats(key)
        ps.print_stats(45)
    text = buf.getvalue()
    print(text[:12000])
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
