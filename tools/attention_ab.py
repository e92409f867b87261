"""The DDPM-256 sparse forward (hipGraph replay, 1.2 % edit) with the attention blocks' scores + softmax + values in ONE launch
(csrc/attention_fused.hip) against the two-launch form (csrc/nhwc_ops.hip), and the attention call alone at DDPM's two shapes.

    python tools/attention_ab.py [--out gpurun_out/attention_ab.json]
"""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def replay_ms(fn, k=50, warm=3):
    import bench

    g, out = bench.capture_fn(fn, warm=warm)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k, out


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
    res = {"forward": {}, "attention_call": {}}
    # the call alone: 8 back-to-back launches of one shape in a graph
    for C, hw in ((512, 16), (512, 8)):
        qkv = torch.randn(1, 3 * C, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
        row = {}
        outs = {}
        for fused in (False, True):
            hip.FUSED_ATTENTION = fused
            ms, o = replay_ms(lambda: [hip.attention_cl(qkv, C ** -0.5) for _ in range(8)][-1], k=100)
            row["one_launch_us" if fused else "two_launches_us"] = round(ms * 1e3 / 8, 2)
            outs[fused] = o.clone()
        row["max_abs_diff"] = float((outs[True] - outs[False]).abs().max())
        res["attention_call"]["C%d_%dx%d" % (C, hw, hw)] = row
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
        outs = {}
        for rep in range(2):  # (twice, alternating: the order of measurement is not the result)
            for fused in (False, True):
                hip.FUSED_ATTENTION = fused
                model(x1, t)
                model(x1, t)
                n0 = hip.launch_count()
                model(x1, t)
                launches = hip.launch_count() - n0
                ms, o = replay_ms(lambda: model(x1, t), k=100)
                key = "one_launch" if fused else "two_launches"
                res["forward"].setdefault(key, {"launches_per_forward": launches, "forward_ms": []})["forward_ms"].append(round(ms, 4))
                outs[fused] = o.clone()
        res["forward"]["max_abs_diff"] = float((outs[True] - outs[False]).abs().max())
    hip.FUSED_ATTENTION = True
    print(json.dumps(res))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
