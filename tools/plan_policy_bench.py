"""Which output block / workgroup shape should the tile conv take when a launch has MANY blocks?  The defaults were tuned at
1.2 % edit (every launch <= one block per CU).  Here: the DDPM-256 sparse forward (hipGraph replay) with many tiles per launch
-- one image at 5 % / 15 % edit, and E = 8 stacked edits at 1.2 % -- under the plan policies of the library:

    default                       32 x 64 blocks (NB = 2: ~300 registers = one 4-wave workgroup per CU) wherever they fill the chip
    nb1 >= N blocks               32 x 32 blocks for launches with at least N 32 x 64 blocks (two or three workgroups per CU)
    waves 8                       8-wave workgroups (two waves per SIMD inside ONE workgroup)

    python tools/plan_policy_bench.py [--out gpurun_out/plan_policy.json]
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
    ap.add_argument("--edits", type=int, default=8)
    ap.add_argument("--thresholds", type=lambda v: [int(x) for x in v.split(",")], default=[1024, 512, 256])
    ap.add_argument("--ratios", type=lambda v: [float(x) for x in v.split(",")], default=[0.05, 0.15])
    ap.add_argument("--waves8", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16", "f16x3"])
    args = ap.parse_args()
    import bench
    import os as _os; _os.environ.setdefault("SIGE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "sige_amd", "lib", "libsige_hip_tuning.so"))  # noqa: E702 -- dispatch knobs exist only in the measurement build (python -m sige_amd.build --tuning)
    from sige_amd import hip, stacked
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    hip.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).to(dev).eval().to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    x0, noise = (t.to(dev).contiguous(memory_format=torch.channels_last) for t in bench.make_inputs())
    t = torch.zeros(1, device=dev)

    def build_pyr(mk):
        return downsample_mask(dilate_mask(mk, 5), 8)

    policies = [("32 x 64 blocks (the default until round 4)", 0, 0)] + [("library default" if n < 0 else "32 x 32 blocks from %d blocks on" % n, n, 0) for n in args.thresholds] + ([("waves 8", 0, 8)] if args.waves8 else [])
    res = {"dtype": args.dtype, "cases": {}}

    def set_policy(nb1, waves):
        hip.conv_large_grid_nb1(nb1)
        hip.conv_force_waves(waves)

    def timed(fn, k=40):
        g, out = bench.capture_fn(fn, warm=2)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k, out

    with torch.no_grad():
        model.set_compute_dtype(args.dtype)
        model.set_mode("full")
        model(x0, t)
        # ---- one image, larger edits ----
        for ratio in args.ratios:
            if args.dtype == "f16":
                model.set_compute_dtype("f16", edit_ratio=ratio)
            m = bench.edit_mask(ratio).to(dev)
            x1 = x0 + noise * m
            model.set_masks(build_pyr(m))
            model.set_mode("sparse")
            rows, ref = {}, None
            for name, nb1, waves in policies:
                set_policy(nb1, waves)
                model(x1, t)
                model(x1, t)
                n0 = hip.launch_count()
                model(x1, t)
                nl = hip.launch_count() - n0
                ms, out = timed(lambda: model(x1, t))
                if ref is None:
                    ref = out.clone()
                rows[name] = {"forward_ms": round(ms, 4), "launches": nl, "max_abs_vs_first_row": float((out - ref).abs().max())}
            res["cases"]["one image, %g %% edit" % (ratio * 100)] = rows
        # ---- E stacked edits at 1.2 % ----
        E = args.edits
        set_policy(-1, 0)
        model.clear_cache()
        model.set_mode("full")
        model(x0, t)
        mks = [bench.square_mask(0.012, top=(16 + 61 * e) % 208, left=(24 + 97 * e) % 208).to(dev) for e in range(E)]
        xe = torch.cat([x0 + noise * mk for mk in mks], 0).contiguous(memory_format=torch.channels_last)
        stacked.stack_caches(model, E)
        try:
            stacked.set_masks(model, [build_pyr(mk) for mk in mks])
            model.set_mode("sparse")
            rows, ref = {}, None
            with stacked.edit_batch(model, E):
                for name, nb1, waves in policies:
                    set_policy(nb1, waves)
                    model(xe, t)
                    model(xe, t)
                    ms, out = timed(lambda: model(xe, t), k=20)
                    if ref is None:
                        ref = out.clone()
                    rows[name] = {"ms_per_launch_set": round(ms, 4), "forwards_per_s": round(E / ms * 1e3, 1),
                                  "max_abs_vs_first_row": float((out - ref).abs().max())}
            res["cases"]["%d stacked edits, 1.2 %% edit" % E] = rows
        finally:
            set_policy(-1, 0)
            stacked.unstack_caches(model)
    print(json.dumps(res, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
