#!/usr/bin/env python3
"""Per-kernel timing of the hot-path ops at DDPM-256 shapes.

hipEvent timing on torch's current stream (the stream every sige_amd.hip call
launches on), >= 4 rotating buffer sets so the 256 MiB Infinity Cache does not
serve the activations.  Prints one JSON line per (op, config) with the
algorithmic-byte / flop rates of SURVEY.md 8(d).
"""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from sige_amd import hip
from sige_amd.utils import reduce_mask


def square_mask(ratio, H=256, W=256, top=100, left=90):
    side = int(round((ratio ** 0.5) * H))
    m = torch.zeros(H, W, dtype=torch.bool)
    m[top:top + side, left:left + side] = True
    return m


def timeit(fn, sets, iters, warmup=10):
    for i in range(warmup):
        fn(sets[i % len(sets)])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(sets[i % len(sets)])
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--ratios", default="0.012,0.05,0.15")
    ap.add_argument("--channels", default="128,256")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--rot", type=int, default=6)
    a = ap.parse_args()
    dev = torch.device("cuda")
    H = a.res
    for ratio in [float(r) for r in a.ratios.split(",")]:
        mask = square_mask(ratio, H, H, int(100 * H / 256), int(90 * H / 256)).to(dev)
        idx6, idx4 = reduce_mask(mask, 6, 4, 1), reduce_mask(mask, 4, 4, 0)
        N6, N4 = idx6.shape[0], idx4.shape[0]
        smap = hip.get_scatter_map(H, H, 6, 6, 3, 3, 1, 1, 1, 1, idx6)
        t0 = hip.tile_table(idx6, (1, 1), (1, 1), (4, 4), (H, H))
        t1 = hip.tile_table(idx4, (0, 0), (1, 1), (4, 4), (H, H))
        for C in [int(c) for c in a.channels.split(",")]:
            sets = []
            for _ in range(a.rot):
                sets.append(dict(
                    x=torch.randn(1, C, H, H, device=dev), y=torch.randn(1, C, H, H, device=dev),
                    y1=torch.randn(1, C, H, H, device=dev),
                    t6=torch.randn(N6, C, 6, 6, device=dev), t4=torch.randn(N6, C, 4, 4, device=dev),
                    s4=torch.randn(N4, C, 4, 4, device=dev)))
            scale, shift = torch.randn(1, C, 1, 1, device=dev), torch.randn(1, C, 1, 1, device=dev)
            w3 = torch.randn(C, C, 3, 3, device=dev) / (3 * C ** 0.5)
            w1 = torch.randn(C, C, 1, 1, device=dev) / C ** 0.5
            bias = torch.randn(C, device=dev)
            p3 = hip.conv_pack_weights(w3, 6, 6, (1, 1))
            p1 = hip.conv_pack_weights(w1, 4, 4, (1, 1))
            e = 4
            full = 2 * e * C * H * H
            ops = {
                "gather6_swish": (lambda s: hip.gather(s["x"], 6, 6, idx6, scale, shift, "swish", False),
                                  2 * e * N6 * C * 36, 0),
                "gather4": (lambda s: hip.gather(s["x"], 4, 4, idx4), 2 * e * N4 * C * 16, 0),
                "scatter_gather_swish": (lambda s: hip.scatter_gather(s["t4"], s["y"], 6, 6, idx6, smap, scale, shift, "swish", False),
                                         2 * e * N6 * C * 36 + 12 * N6 * 36, 0),
                "scatter_2pass_res": (lambda s: hip.scatter(s["t4"], s["y"], 1, 1, 1, 1, idx6, s["x"]),
                                      full + e * N6 * C * 16 * 3, 0),
                "scatter_fused_res": (lambda s: hip.scatter_fused(s["t4"], s["y"], t0, N6, s["x"]),
                                      full + e * N6 * C * 16 * 3, 0),
                "swbr_2pass": (lambda s: hip.scatter_with_block_residual(s["t4"], s["y"], s["s4"], s["y1"], 1, 1, 1, 1, idx6, idx4),
                               full + 3 * e * N6 * C * 16 + 4 * e * N4 * C * 16, 0),
                "swbr_fused": (lambda s: hip.scatter_with_block_residual_fused(s["t4"], s["y"], s["s4"], s["y1"], t0, N6, t1, N4),
                               full + 3 * e * N6 * C * 16 + 4 * e * N4 * C * 16, 0),
                "clone(torch)": (lambda s: s["y"].clone(), full, 0),
                "conv3x3_mfma": (lambda s: hip.block_conv(s["t6"], p3, bias, C, (3, 3), (1, 1)), 0, 2 * N6 * 16 * C * C * 9),
                "conv3x3_miopen": (lambda s: torch.nn.functional.conv2d(s["t6"], w3, bias), 0, 2 * N6 * 16 * C * C * 9),
                "conv1x1_mfma": (lambda s: hip.block_conv(s["s4"], p1, bias, C, (1, 1), (1, 1)), 0, 2 * N4 * 16 * C * C),
                "conv1x1_miopen": (lambda s: torch.nn.functional.conv2d(s["s4"], w1, bias), 0, 2 * N4 * 16 * C * C),
            }
            for name, (fn, nbytes, flops) in ops.items():
                us = timeit(fn, sets, a.iters)
                rec = {"op": name, "ratio": ratio, "C": C, "res": H, "N6": N6, "N4": N4, "us": round(us, 2)}
                if nbytes:
                    rec["alg_GBps"] = round(nbytes / us / 1e3, 1)
                    rec["alg_MB"] = round(nbytes / 1e6, 2)
                if flops:
                    rec["TFLOPs"] = round(flops / us / 1e6, 2)
                    rec["GFLOP"] = round(flops / 1e9, 3)
                print(json.dumps(rec), flush=True)
            del sets


if __name__ == "__main__":
    main()
