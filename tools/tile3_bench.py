#!/usr/bin/env python3
"""Tile conv v3 (csrc/conv_tile3.hpp) against conv_mfma.hpp, launch by launch: the SIGE 3x3 convs of the DDPM-256 U-Net at the
bench's edit ratios (gather -> tiles with affine + SiLU, and scatter_gather -> full tensor with block residual), each as a
hipGraph of back-to-back launches (rotating nothing: weights L2-warm in both), and the whole sparse forward per ratio with the
router off / forced / automatic.

    python tools/tile3_bench.py [--out gpurun_out/tile3_bench.json]
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
    ap.add_argument("--compute", default="f32", choices=["f32", "f16"], help="operand form of both kernels (f16: round 6, Tile3Geo<2, WIDE_F16>)")
    ap.add_argument("--cache-dtype", default="f32", choices=["f32", "f16"], help="the forward's cache storage (bench.py --dtype f16 uses f16)")
    ap.add_argument("--skip-forward", action="store_true")
    ap.add_argument("--skip-layers", action="store_true")
    ap.add_argument("--ratios", default="0.012,0.05,0.10,0.15,0.20")
    a = ap.parse_args()
    from sige_amd import hip
    from sige_amd.utils import dilate_mask, downsample_mask, reduce_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    dev = torch.device("cuda:0")
    tunable = hasattr(hip.lib(), "sige_hip_tuning_set") or os.path.basename(hip.LIB_PATH).startswith("libsige_hip_tuning")
    cl = lambda t_: t_.contiguous(memory_format=torch.channels_last)  # noqa: E731
    res = {"layers": [], "forward": []}
    torch.manual_seed(0)
    ratios = tuple(float(v) for v in a.ratios.split(","))
    for ratio in (() if a.skip_layers else ratios):
        pyr = downsample_mask(dilate_mask(bench.square_mask(ratio).to(dev), 5), 8)
        for (R, C, Cout) in ((256, 128, 128), (128, 128, 128), (64, 256, 256)):
            m = pyr[(R, R)]
            idx = reduce_mask(m, 6, 4, 1)
            idx1 = reduce_mask(m, 4, 4, 0)
            N = idx.shape[0]
            x = cl(torch.randn(1, C, R, R, device=dev))
            y = cl(torch.randn(1, C, R, R, device=dev))
            w = torch.randn(Cout, C, 3, 3, device=dev) / (3 * C ** 0.5)
            bias = torch.randn(Cout, device=dev)
            sc, sh = torch.randn(1, C, 1, 1, device=dev), torch.randn(1, C, 1, 1, device=dev)
            packed = hip.conv_pack_weights(w, 6, 6, (1, 1), a.compute)
            smap = hip.get_scatter_map(R, R, 6, 6, 3, 3, 1, 1, 1, 1, idx)
            t4 = cl(torch.randn(N, C, 4, 4, device=dev))
            y1 = cl(torch.randn(1, Cout, R, R, device=dev))
            x1 = cl(torch.randn(idx1.shape[0], Cout, 4, 4, device=dev))
            table1 = hip.tile_table(idx1, (0, 0), (1, 1), (4, 4), (R, R))
            out = cl(torch.randn(1, Cout, R, R, device=dev))
            flop = 2.0 * N * 16 * Cout * C * 9
            row = {"edit_ratio": ratio, "resolution": R, "C": C, "Cout": Cout, "tiles": N, "GFLOP": round(flop / 1e9, 3),
                   "v3_workgroups": -(-N // 2) * (Cout // 64)}
            for name, fn in (("gather_affine_swish_to_tiles", lambda: hip.gather_conv_cl(x, None, (6, 6), idx, sc, sh, "swish", packed, bias, Cout, (3, 3), (1, 1))),
                             ("scatter_gather_to_full_block_residual", lambda: hip.scatter_gather_conv_scatter_cl(
                                 t4, y, (6, 6), idx, smap, None, None, "identity", packed, bias, Cout, (3, 3), (1, 1), out, residual=y1, x1=x1, table1=table1))):
                us = {}
                for tag, flag in (("conv_mfma", False), ("tile3", True)):
                    hip.TILE3 = flag
                    try:
                        us[tag] = round(bench.time_graph_of(fn, 20), 2)
                    finally:
                        hip.TILE3 = None
                if a.compute == "f16" and tunable:  # (4 tiles per workgroup: never / always)
                    hip.TILE3 = True
                    try:
                        hip.tuning_set("tile3_f16_tpw4_min", 0)
                        us["tile3"] = round(bench.time_graph_of(fn, 20), 2)
                        hip.tuning_set("tile3_f16_tpw4_min", 1)
                        us["tile3_tpw4"] = round(bench.time_graph_of(fn, 20), 2)
                    finally:
                        hip.TILE3 = None
                        hip.tuning_set("tile3_f16_tpw4_min", -1)
                row[name] = dict(us, TFLOPs_conv_mfma=round(flop / us["conv_mfma"] / 1e6, 1), TFLOPs_tile3=round(flop / us["tile3"] / 1e6, 1))
            res["layers"].append(row)
    # the whole forward
    model = DDPMSparseUNet(DDPMConfig()).eval().to(dev).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    if a.compute != "f32":
        model.set_compute_dtype(a.compute)
    if a.cache_dtype != "f32":
        model.set_cache_dtype(a.cache_dtype)
    res["compute"], res["cache_dtype"] = a.compute, a.cache_dtype
    x0, noise = bench.make_inputs()
    x0, noise, t = cl(x0.to(dev)), cl(noise.to(dev)), torch.zeros(1, device=dev)
    with torch.no_grad():
        model.set_mode("full")
        model(x0, t)
        for ratio in (() if a.skip_forward else ratios):
            mask = bench.square_mask(ratio).to(dev)
            model.set_masks(downsample_mask(dilate_mask(mask, 5), 8))
            model.set_mode("sparse")
            x1 = x0 + noise * mask
            row = {"edit_ratio": ratio}
            hip.TILE3_MIN_BLOCKS = 512
            for tag, flag, th, t4 in (("conv_mfma_only", False, None, None), ("router", None, None, None),
                                      ("sparse_from_32", None, None, ("tile3_f16_sparse_min", 32)),
                                      ("sparse_from_64", None, None, ("tile3_f16_sparse_min", 64)),
                                      ("sparse_from_96", None, None, ("tile3_f16_sparse_min", 96)),
                                      ("sparse_from_128", None, None, ("tile3_f16_sparse_min", 128)),
                                      ("sparse_from_64_pairs_64", None, None, ("tile3_f16_sparse_min", 64, "tile3_f16_pair_min", 64)),
                                      ("sparse_from_128_pairs_128", None, None, ("tile3_f16_sparse_min", 128, "tile3_f16_pair_min", 128))):
                if (th is not None or t4 is not None) and a.compute != "f16":
                    continue
                if t4 is not None and not tunable:
                    continue
                hip.TILE3 = flag
                keep_th = hip.TILE3_MIN_BLOCKS_F16
                if th is not None:
                    hip.TILE3_MIN_BLOCKS_F16 = th
                if t4 is not None:
                    for q in range(0, len(t4), 2):
                        hip.tuning_set(t4[q], t4[q + 1])
                try:
                    model(x1, t)
                    model(x1, t)
                    n0 = hip.launch_count()
                    model(x1, t)
                    launches = hip.launch_count() - n0
                    g, _ = bench.capture(model, x1, t)
                    ms = bench.timed_replays(g, 30, 5, 1) * 1e3 / 30
                    del g
                finally:
                    hip.TILE3 = None
                    hip.TILE3_MIN_BLOCKS_F16 = keep_th
                    if t4 is not None:
                        for q in range(0, len(t4), 2):
                            hip.tuning_set(t4[q], -1)
                row[tag] = {"forward_ms": round(ms, 4), "launches": launches}
            res["forward"].append(row)
    text = json.dumps(res, indent=1)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
