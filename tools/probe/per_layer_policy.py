#!/usr/bin/env python3
"""Upper bound of a per-LAYER choice of the tile conv's launch shape: every conv call of the benchmarked forward (DDPM-256, one
traced sparse forward) timed on its own, warm, under the dispatch policy of csrc/block_conv.hip and under each pinned (pixel block,
channel blocks) / K split / wave count of the measurement build.  Per call: the policy's time and the best pinned time; summed:
what a per-layer autotuner could take off the forward if every call kept its in-situ cost (it does not: calls are timed back to
back on warm operands, pairs are broken up -- so this is a BOUND on the sum of launches, not a forward time).

    SIGE_HIP_LIB=sige_amd/lib/libsige_hip_tuning.so python tools/probe/per_layer_policy.py --out gpurun_out/per_layer_policy.json
"""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ.setdefault("SIGE_HIP_LIB", os.path.join(REPO, "sige_amd", "lib", "libsige_hip_tuning.so"))

SETTINGS = [("policy", {}), ("mt16_nb1", {"conv_tile_mt": 16, "conv_tile_nb": 1}), ("mt16_nb2", {"conv_tile_mt": 16, "conv_tile_nb": 2}),
            ("mt32_nb1", {"conv_tile_mt": 32, "conv_tile_nb": 1}), ("mt32_nb2", {"conv_tile_mt": 32, "conv_tile_nb": 2}),
            ("ksplit2", {"conv_ksplit": 2}), ("ksplit4", {"conv_ksplit": 4}), ("waves8", {"conv_waves": 8})]
KNOBS = ("conv_tile_mt", "conv_tile_nb", "conv_ksplit", "conv_waves")
CONVS = ("gather_conv_cl", "scatter_gather_conv_cl", "scatter_gather_conv_scatter_cl", "block_conv_cl")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ratio", type=float, default=0.012)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import bench
    from benchlib.common import time_graph_of
    from sige_amd import hip
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    hip.lib()
    tracer = bench.Tracer(hip)
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
        mask = bench.edit_mask(a.ratio)
        model.set_masks(downsample_mask(dilate_mask(mask.to(dev), 5), 8))
        model.set_mode("sparse")
        x1 = cl(x0 + noise * mask)
        model(x1, t)
        tracer.log = []
        model(x1, t)
        trace, tracer.log = tracer.log, None
        seen = {}
        for name, args, kw, orig in trace:
            if name not in CONVS:
                continue
            key = bench.shape_key(name, args)
            if key in seen:
                seen[key]["count"] += 1
                continue
            rec = {"call": name, "count": 1, "us": {}}
            shapes = [tuple(v.shape) for v in args if isinstance(v, torch.Tensor)][:3]
            rec["shapes"] = [list(s) for s in shapes]
            for tag, knobs in SETTINGS:
                for k in KNOBS:
                    hip.tuning_set(k, 0)
                for k, v in knobs.items():
                    hip.tuning_set(k, v)
                try:
                    rec["us"][tag] = round(time_graph_of(lambda: orig(*args, **kw), reps=8), 2)
                except Exception as e:  # (a pinned shape this layer cannot take)
                    rec["us"][tag] = None
                    rec.setdefault("errors", {})[tag] = str(e)[:80]
            for k in KNOBS:
                hip.tuning_set(k, 0)
            seen[key] = rec
            rows.append(rec)
    pol = sum(r["us"]["policy"] * r["count"] for r in rows)
    best = sum(min(v for v in r["us"].values() if v is not None) * r["count"] for r in rows)
    wins = [r for r in rows if min(v for v in r["us"].values() if v is not None) < 0.97 * r["us"]["policy"]]
    res = {"ratio": a.ratio, "distinct_conv_calls": len(rows), "conv_calls": sum(r["count"] for r in rows),
           "sum_policy_us": round(pol, 1), "sum_best_pinned_us": round(best, 1), "calls_where_a_pin_wins_by_3pct": len(wins), "rows": rows}
    print(json.dumps({k: v for k, v in res.items() if k != "rows"}), flush=True)
    for r in wins:
        print(r["call"], r["shapes"], r["count"], r["us"], flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
