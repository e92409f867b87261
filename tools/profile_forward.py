#!/usr/bin/env python3
"""Profiling target: K hipGraph replays of one sparse (or dense) DDPM-256 U-Net
forward, separated from all set-up work by a 0.5 s idle gap so that
tools/trace_summary.py can isolate them in a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d OUT -o fwd -- \
        python tools/profile_forward.py --ratio 0.012 --replays 50
"""
import argparse
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")

import torch  # noqa: E402

import bench  # noqa: E402
import os as _os; _os.environ.setdefault("SIGE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "sige_amd", "lib", "libsige_hip_tuning.so"))  # noqa: E702 -- dispatch knobs exist only in the measurement build (python -m sige_amd.build --tuning)
from sige_amd.utils import dilate_mask, downsample_mask  # noqa: E402
from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ratio", type=float, default=0.012)
    ap.add_argument("--replays", type=int, default=50)
    ap.add_argument("--mode", default="sparse", choices=["sparse", "dense", "eager", "full"],
                    help="full = the cache-producing full pass (with --dtype f16x3: on the library's dense-layer kernel)")
    ap.add_argument("--layout", default="nhwc", choices=["nhwc", "nchw"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16", "f16x3"], help="arithmetic of the convs (bench.py --dtype)")
    ap.add_argument("--ksplit", type=int, default=0, help="pin the cross-workgroup K split of the tile convs (hip.conv_force_ksplit; 0 = automatic)")
    ap.add_argument("--tile3-min-blocks", type=int, default=None, help="sige_amd.hip.TILE3_MIN_BLOCKS (the tile conv v3's routing threshold)")
    ap.add_argument("--edits", type=int, default=1, help="E > 1: E edits of one original, each with its own mask, stacked into one forward "
                                                        "(sige_amd/stacked.py; sparse mode, channels-last)")
    ap.add_argument("--manifest", default="", help="eager mode: write the kernel-family sequence of one forward's conv launches here")
    a = ap.parse_args()
    dev = torch.device("cuda")
    if a.tile3_min_blocks is not None:
        from sige_amd import hip as _hip3

        _hip3.TILE3_MIN_BLOCKS = a.tile3_min_blocks
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).to(dev).eval()
    gen = torch.Generator().manual_seed(1)
    x0 = torch.randn(1, 3, 256, 256, generator=gen).to(dev)
    noise = torch.randn(1, 3, 256, 256, generator=gen).to(dev)
    t = torch.zeros(1, device=dev)
    if a.layout == "nhwc":
        model = model.to(memory_format=torch.channels_last)
        x0, noise = x0.contiguous(memory_format=torch.channels_last), noise.contiguous(memory_format=torch.channels_last)
        model.set_scatter_inplace(True)
    if a.ksplit:
        from sige_amd import hip as _hip

        _hip.conv_force_ksplit(a.ksplit)
    model.set_compute_dtype(a.dtype, edit_ratio=a.ratio)  # (the f16 precision policy depends on the edited area)
    mask = bench.square_mask(a.ratio).to(dev)
    x1 = x0 + noise * mask
    with torch.no_grad():
        model.set_mode("full")
        if a.mode == "dense":
            model.set_plain_dense(True)
        elif a.mode == "full":
            pass
        else:
            model(x0, t)
            model.set_masks(downsample_mask(dilate_mask(mask, 5), 8))
            model.set_mode("sparse")
        if a.mode == "eager":
            # (the PMC attribution matches counter rows to conv CALLS by launch order: every call must be its own kernel, so the
            #  shortcut / conv1 pairing -- the same two workgroup programs side by side in one launch -- is off for these passes)
            for m in model.modules():
                if hasattr(m, "pair"):
                    m.pair = False
            if a.manifest:
                import json

                from sige_amd import hip

                tracer = bench.Tracer(hip)
                model(x1, t)
                tracer.log = []
                model(x1, t)
                seq = [bench.op_cost(n, args, kw)[0] for n, args, kw, _ in tracer.log]
                tracer.log = None
                json.dump({"conv_families_in_launch_order": [f for f in seq if f.endswith("conv_mfma")],
                           "warmup_forwards": 5, "measured_forwards": a.replays, "source_hash": bench.source_hash()},
                          open(a.manifest, "w"))
            for _ in range(3):
                model(x1, t)
            torch.cuda.synchronize()
            time.sleep(0.5)
            for _ in range(a.replays):
                model(x1, t)
        elif a.edits > 1:
            # the masks of bench.py's batched_edits section: E squares of the same size at different places
            from sige_amd import stacked

            assert a.mode == "sparse" and a.layout == "nhwc"
            mks = [bench.square_mask(a.ratio, top=(16 + 61 * e) % 208, left=(24 + 97 * e) % 208).to(dev) for e in range(a.edits)]
            xe = torch.cat([x0 + noise * mk for mk in mks], 0).contiguous(memory_format=torch.channels_last)
            stacked.stack_caches(model, a.edits)
            stacked.set_masks(model, [downsample_mask(dilate_mask(mk, 5), 8) for mk in mks])
            with stacked.edit_batch(model, a.edits):
                g, _ = bench.capture(model, xe, t)
                g.replay()
                torch.cuda.synchronize()
                time.sleep(0.5)
                for _ in range(a.replays):
                    g.replay()
        else:
            if a.mode == "full":
                x1 = x0
            g, _ = bench.capture(model, x1, t)
            g.replay()
            torch.cuda.synchronize()
            time.sleep(0.5)
            for _ in range(a.replays):
                g.replay()
        torch.cuda.synchronize()
    print("done", a.mode, a.ratio, a.replays)


if __name__ == "__main__":
    main()
