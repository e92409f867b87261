"""Which kernel should a dense layer take under each compute dtype?  The DDPM-256 sparse forward (hipGraph replay) with the
dense-layer kernel (conv_wide.hpp) enabled / disabled for the dense 3x3 convs, per compute dtype and edit ratio.

    python tools/routing_bench.py [--out gpurun_out/routing.jsonl]
"""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--ratios", default="0.012,0.15")
    args = ap.parse_args()
    import bench
    from sige_amd import hip
    from sige_amd.nn import dense
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    hip.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).to(dev).eval().to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    x0, noise = (t.to(dev).contiguous(memory_format=torch.channels_last) for t in bench.make_inputs())
    t = torch.zeros(1, device=dev)
    default_min = dict(dense.WIDE_MIN_FLOP)
    default_min_x3 = dict(dense.WIDE_MIN_FLOP_X3)
    default_x3 = dense.TILE_X3_MIN_FLOP
    rows = []
    with torch.no_grad():
        model.set_mode("full")
        model(x0, t)
        for r in [float(v) for v in args.ratios.split(",")]:
            m = bench.edit_mask(r).to(dev)
            x1 = x0 + noise * m
            ref = None
            default_f32 = dict(dense.WIDE_MIN_FLOP_F32)
            for name, dtype, keep, wide, tile_x3 in (
                    ("f32 exact", "f32", (), True, default_x3),
                    ("f32 exact, dense 3x3 on the wide kernel (all)", "f32", (), "f32:0.25", default_x3),
                    ("f32 exact, dense 3x3 on the wide kernel (>= 1 GFLOP)", "f32", (), "f32:1.0", default_x3),
                    ("f32 exact, dense 3x3 on the wide kernel (>= 2 GFLOP)", "f32", (), "f32:2.0", default_x3),
                    ("f16 everywhere, dense 3x3 on the wide kernel", "f16", (), True, default_x3),
                    ("f16 everywhere, tile kernels only", "f16", (), False, default_x3),
                    ("f16 + F16_KEEP as f16x3", "f16", None, True, default_x3),
                    ("f16x3: wide dense + exact tiles below 2 GFLOP", "f16x3", (), True, default_x3),
                    ("f16x3: wide dense (>= 2 GFLOP) + exact tiles below 2 GFLOP", "f16x3", (), "x3:2.0", default_x3),
                    ("f16x3: wide dense (>= 1 GFLOP) + exact tiles below 2 GFLOP", "f16x3", (), "x3:1.0", default_x3),
                    ("f16x3: wide dense (>= 0.25 GFLOP) + exact tiles below 2 GFLOP", "f16x3", (), "x3:0.25", default_x3),
                    ("f16: wide dense (>= 1 GFLOP)", "f16", (), "f16:1.0", default_x3),
                    ("f16x3: wide dense + split-operand tiles everywhere", "f16x3", (), True, 0.0),
                    ("f16x3: tile kernels only (split operands)", "f16x3", (), False, 0.0)):
                dense.WIDE_MIN_FLOP_F32 = default_f32
                dense.WIDE_MIN_FLOP_X3 = default_min_x3
                over = None
                if isinstance(wide, str):
                    kind, val = wide.split(":")
                    over = {3: float(val) * 1e9, 1: 1e30 if kind == "f32" else 2.0e9}
                    if kind == "f32":
                        dense.WIDE_MIN_FLOP_F32 = over
                    elif kind == "x3":
                        dense.WIDE_MIN_FLOP_X3 = over
                    wide = True
                dense.WIDE_MIN_FLOP = (over if (over is not None and kind == "f16") else default_min) if wide else {1: 1e30, 3: 1e30}
                if not wide:
                    dense.WIDE_MIN_FLOP_X3 = {1: 1e30, 3: 1e30}
                dense.TILE_X3_MIN_FLOP = tile_x3
                model.set_compute_dtype(dtype, keep=keep)
                model.set_masks(downsample_mask(dilate_mask(m, 5), 8))
                model.set_mode("sparse")
                model(x1, t)
                model(x1, t)
                n0, p0 = hip.launch_count(), hip.conv_pairs_fused()
                model(x1, t)
                launches, pairs = hip.launch_count() - n0, hip.conv_pairs_fused() - p0
                g, out = bench.capture(model, x1, t)
                ms = bench.timed_replays(g, 100, 10, 1) * 1e3 / 100
                o = out.float().clone()
                if ref is None:
                    ref = o
                row = {"edit_ratio": r, "config": name, "forward_ms": round(ms, 4), "launches": launches, "pairs": pairs,
                       "max_abs_vs_f32": round(float((o - ref).abs().max()), 7)}
                print(json.dumps(row), flush=True)
                rows.append(row)
                del g, out
    dense.WIDE_MIN_FLOP = default_min
    dense.WIDE_MIN_FLOP_X3 = default_min_x3
    dense.WIDE_MIN_FLOP_F32 = default_f32
    dense.TILE_X3_MIN_FLOP = default_x3
    if args.out:
        with open(os.path.join(REPO, args.out), "w") as f:
            for row in rows:
                f.write(json.dumps(row) + "\n")


if __name__ == "__main__":
    main()
