#!/usr/bin/env python3
"""Where should the tile conv v3 (csrc/conv_tile3.hpp) be routed?  The DDPM-256 sparse forward (hipGraph replay) per edit ratio and
per threshold of sige_amd.hip.TILE3_MIN_BLOCKS (v3 workgroups = tile pairs x 64-channel blocks a launch must have), and eight
stacked edits at 1.2 %.

    python tools/tile3_router_bench.py [--out gpurun_out/tile3_router.json]
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--thresholds", default="1000000,1024,512,256,150")
    a = ap.parse_args()
    from sige_amd import hip, stacked
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    dev = torch.device("cuda:0")
    cl = lambda t_: t_.contiguous(memory_format=torch.channels_last)  # noqa: E731
    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).eval().to(dev).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    x0, noise = bench.make_inputs()
    x0, noise, t = cl(x0.to(dev)), cl(noise.to(dev)), torch.zeros(1, device=dev)
    ths = [int(v) for v in a.thresholds.split(",")]
    keep = hip.TILE3_MIN_BLOCKS
    res = {"single": [], "stacked": []}

    def build_pyr(mk):
        return downsample_mask(dilate_mask(mk, 5), 8)

    with torch.no_grad():
        model.set_mode("full")
        model(x0, t)
        for ratio in (0.012, 0.05, 0.10, 0.15, 0.20):
            mask = bench.square_mask(ratio).to(dev)
            model.set_masks(build_pyr(mask))
            model.set_mode("sparse")
            x1 = x0 + noise * mask
            row = {"edit_ratio": ratio}
            for th in ths:
                hip.TILE3_MIN_BLOCKS = th
                model(x1, t)
                model(x1, t)
                n0 = hip.launch_count()
                model(x1, t)
                launches = hip.launch_count() - n0
                g, _ = bench.capture(model, x1, t)
                ms = bench.timed_replays(g, 30, 5, 1) * 1e3 / 30
                del g
                row[str(th)] = {"forward_ms": round(ms, 4), "launches": launches}
            res["single"].append(row)
        # eight stacked edits at 1.2 %
        E = 8
        model.clear_cache()
        model.set_mode("full")
        model(x0, t)
        mks = [bench.square_mask(0.012, top=(16 + 61 * e) % 208, left=(24 + 97 * e) % 208).to(dev) for e in range(E)]
        xe = cl(torch.cat([x0 + noise * mk for mk in mks], 0))
        stacked.stack_caches(model, E)
        try:
            stacked.set_masks(model, [build_pyr(mk) for mk in mks])
            model.set_mode("sparse")
            row = {"edits": E, "edit_ratio": 0.012}
            with stacked.edit_batch(model, E):
                for th in ths:
                    hip.TILE3_MIN_BLOCKS = th
                    g, _ = bench.capture(model, xe, t)
                    ms = bench.timed_replays(g, 20, 5, 1) * 1e3 / 20
                    del g
                    row[str(th)] = {"ms_per_launch_set": round(ms, 4), "ms_per_edit": round(ms / E, 4)}
            res["stacked"].append(row)
        finally:
            stacked.unstack_caches(model)
    hip.TILE3_MIN_BLOCKS = keep
    text = json.dumps(res, indent=1)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
