"""The benchmarked forward (DDPM-256, hipGraph replay) under pinned output-block shapes / K splits of the tile conv, in the measurement
build (dispatch knobs): is the per-launch policy of csrc/block_conv.hip (the largest block whose grid still covers the chip; K split
only below half a chip of blocks) still the best global choice on this round's kernels?

    python tools/block_policy_sweep.py [--ratios 0.012,0.05] [--out gpurun_out/block_policy.json]
"""
import argparse
import json
import os
import statistics
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("SIGE_HIP_LIB", os.path.join(REPO, "sige_amd", "lib", "libsige_hip_tuning.so"))

SETTINGS = [("default", {}), ("mt16_nb1", {"conv_tile_mt": 16, "conv_tile_nb": 1}), ("mt16_nb2", {"conv_tile_mt": 16, "conv_tile_nb": 2}),
            ("mt32_nb1", {"conv_tile_mt": 32, "conv_tile_nb": 1}), ("mt32_nb2", {"conv_tile_mt": 32, "conv_tile_nb": 2}),
            ("ksplit2", {"conv_ksplit": 2}), ("ksplit4", {"conv_ksplit": 4}), ("waves8", {"conv_waves": 8}), ("default_again", {})]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ratios", default="0.012,0.05")
    ap.add_argument("--out", default="")
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
    x0, noise = bench.make_inputs()
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)  # noqa: E731
    t = torch.zeros(1, device=dev)
    rows = []
    with torch.no_grad():
        model.set_mode("full")
        model(cl(x0), t)
        for r in [float(v) for v in a.ratios.split(",")]:
            mask = bench.edit_mask(r)
            model.set_masks(downsample_mask(dilate_mask(mask.to(dev), 5), 8))
            model.set_mode("sparse")
            x1 = cl(x0 + noise * mask)
            ref = None
            for name, knobs in SETTINGS:
                for k in ("conv_tile_mt", "conv_tile_nb", "conv_ksplit", "conv_waves"):
                    hip.tuning_set(k, knobs.get(k, 0))
                try:
                    model(x1, t)
                    n0 = hip.launch_count()
                    model(x1, t)
                    launches = hip.launch_count() - n0
                    g, out = capture(model, x1, t)
                    ms = [timed_replays(g, 50, 5, 1) * 1e3 / 50 for _ in range(5)]
                    o = out.float().clone()
                    if ref is None:
                        ref = o
                    rows.append({"ratio": r, "setting": name, "forward_ms": round(statistics.median(ms), 4), "launches": launches,
                                 "max_abs_vs_default": float((o - ref).abs().max())})
                except Exception as e:  # (a pinned shape a layer cannot take)
                    rows.append({"ratio": r, "setting": name, "error": str(e)[:200]})
                print(json.dumps(rows[-1]), flush=True)
            for k in ("conv_tile_mt", "conv_tile_nb", "conv_ksplit", "conv_waves"):
                hip.tuning_set(k, 0)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump({"rows": rows}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
