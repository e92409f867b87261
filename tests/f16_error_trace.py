"""Per-layer error trace of the f16 path (BASELINE.json configs[4]) WITHOUT a GPU.

    python -m tests.f16_error_trace [--ratios 0.01,0.02,0.05,0.1,0.2] [--per-layer 0.2] [--out profiles/r3_f16_error_trace.json]

Test infrastructure (it drives the CPU oracle backend, so it lives under tests/).  The DDPM-256 U-Net runs on the
CPU oracle exactly as bench.py's parity leg does; "f16 compute" is emulated where the GPU kernels apply it: the input of a
conv (after the cached GroupNorm affine + SiLU) and its weights are rounded to fp16 (RNE), products and sums stay fp32 --
the arithmetic of v_mfma_f32_*_f16 up to fp32 summation order.  "f16 storage" additionally rounds every conv OUTPUT that
the GPU path would store as fp16 (tiles, full activations, caches).

It prints, per edit ratio, the numbers of sige_amd.tolerance.f16_check against the fp32 output; with --per-layer R it
switches ONE conv at a time to f16 at edit ratio R and records the output error each causes, then grows the set of convs
kept in fp32 (largest contribution first) until the criterion holds at every ratio.  The result is the default
`keep_f32` policy of SIGEModel.set_compute_dtype("f16") (sige_amd/nn/base.py: F16_KEEP_F32).
"""
import argparse
import json
import os
import sys

import torch
from torch import nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def f16_eligible(name: str, conv: nn.Conv2d) -> bool:
    """The convs the GPU runs through the f16 kernels: 3x3 / stride 1 and 1x1 tile convs and dense layers (the stride-2
    geometry, conv_in and conv_out have fp32 kernels only)."""
    k, s = tuple(conv.kernel_size), tuple(conv.stride)
    return name not in ("conv_in", "conv_out") and s == (1, 1) and k in ((3, 3), (1, 1))


class Emulator:
    def __init__(self, model, storage=False):
        self.model, self.storage = model, storage
        self.convs = {n: m for n, m in model.named_modules() if isinstance(m, nn.Conv2d) and f16_eligible(n, m)}
        self.active = set()
        self._orig = {}
        for n, m in self.convs.items():
            m.register_forward_pre_hook(self._pre(n))
            if storage:
                m.register_forward_hook(self._post(n))

    def _pre(self, name):
        def hook(mod, args):
            if name in self.active and isinstance(args[0], torch.Tensor) and args[0].dtype == torch.float32:
                return (args[0].half().float(),) + tuple(args[1:])
            return None
        return hook

    def _post(self, name):
        def hook(mod, args, out):
            if name in self.active and isinstance(out, torch.Tensor):
                return out.half().float()
            return None
        return hook

    def set_active(self, names):
        for n, w in self._orig.items():
            self.convs[n].weight.data = w
        self._orig = {}
        self.active = set(names)
        for n in self.active:
            m = self.convs[n]
            self._orig[n] = m.weight.data
            m.weight.data = m.weight.data.half().float()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ratios", default="0.01,0.02,0.05,0.1,0.2")
    ap.add_argument("--per-layer", type=float, default=0.0, help="edit ratio of the one-conv-at-a-time trace (0 = skip)")
    ap.add_argument("--storage", action="store_true", help="also round conv outputs to fp16 (f16 storage)")
    ap.add_argument("--keep", default="", help="comma-separated conv names kept in fp32 (prefix match)")
    ap.add_argument("--cache", action="store_true", help="fp16 CACHE only: the cached activations of the full pass are rounded to fp16 (what "
                    "parallel.distribute_cache(wire_dtype=torch.float16) leaves on every rank), the sparse pass computes in fp32 and, second "
                    "row, with the model's f16 policy")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import bench
    from oracle import oracle
    from sige_amd import runtime, tolerance
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    n_thr = min(32, os.cpu_count() or 1)
    torch.set_num_threads(n_thr)
    oracle.set_num_threads(n_thr)
    runtime.register_backend("cpu", oracle)
    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).eval()
    emu = Emulator(model, storage=args.storage)
    x0, noise = bench.make_inputs()
    t = torch.zeros(1)
    ratios = [float(v) for v in args.ratios.split(",")]
    keep = tuple(k for k in args.keep.split(",") if k)
    report = {"storage": args.storage, "criterion": tolerance.F16_CRITERION, "ratios": {}, "convs": len(emu.convs)}

    def kept(name, keep_):
        return any(name == k or name.startswith(k + ".") or name.startswith(k) for k in keep_)

    with torch.no_grad():
        model.set_mode("full")
        model(x0, t)  # the original image's cache: always the fp32 full pass
        ref, x1 = {}, {}
        for r in ratios:
            mask = bench.edit_mask(r)
            x1[r] = x0 + noise * mask
            model.set_masks(downsample_mask(dilate_mask(mask, 5), 8))
            model.set_mode("sparse")
            ref[r] = model(x1[r], t).clone()

        def run(r, names):
            mask = bench.edit_mask(r)
            model.set_masks(downsample_mask(dilate_mask(mask, 5), 8))
            emu.set_active(names)
            try:
                return model(x1[r], t).clone()
            finally:
                emu.set_active(())

        def sweep(keep_):
            names = [n for n in emu.convs if not kept(n, keep_)]
            rows = {}
            for r in ratios:
                rows["%g" % r] = tolerance.f16_check(run(r, names), ref[r])
            return rows

        if args.cache:
            from sige_amd import parallel

            rounded = 0
            for sl in parallel.cache_slots(model):
                v = parallel._get(sl)
                if v.numel() >= parallel._WIRE_SMALL:  # (the cached affines travel -- and stay -- fp32)
                    parallel._set(sl, v.half().float())
                    rounded += v.numel()
            parallel.refresh_derived(model)
            report["cache_elements_rounded_to_fp16"] = rounded
            report["fp16_cache_fp32_compute"] = {}
            for r in ratios:
                out = run(r, ())
                c = tolerance.f16_check(out, ref[r])
                c["meets_the_fp32_tolerance_1e-3"] = bool(c["max_abs"] <= 1e-3)
                report["fp16_cache_fp32_compute"]["%g" % r] = c
            print(json.dumps(report["fp16_cache_fp32_compute"], indent=1), flush=True)
            report["fp16_cache_f16_policy"] = {}
            for r in ratios:
                keep_r = tuple(getattr(model, "F16_KEEP", ())) if r > getattr(model, "F16_KEEP_ABOVE", 1.0) else ()
                names = [n for n in emu.convs if not kept(n, keep_r)]
                report["fp16_cache_f16_policy"]["%g" % r] = tolerance.f16_check(run(r, names), ref[r])
            print(json.dumps(report["fp16_cache_f16_policy"], indent=1), flush=True)
        report["ratios"] = sweep(keep)
        report["keep_f32"] = list(keep)
        print(json.dumps(report["ratios"], indent=1), flush=True)

        if args.per_layer > 0:
            r = args.per_layer
            contrib = {}
            for n in emu.convs:
                c = tolerance.f16_check(run(r, [n]), ref[r])
                contrib[n] = c["worst_over_allowed"]
                print("%-40s worst/allowed %.4f  max_abs %.5f" % (n, c["worst_over_allowed"], c["max_abs"]), flush=True)
            report["per_layer_at"] = r
            report["per_layer_worst_over_allowed"] = contrib
            order = sorted(contrib, key=contrib.get, reverse=True)
            keep_ = list(keep)
            rows = report["ratios"]
            while not all(v["ok"] for v in rows.values()) and order:
                keep_.append(order.pop(0))
                if len(keep_) % 4 == 0 or not order:
                    rows = sweep(tuple(keep_))
                    print("keep %d -> %s" % (len(keep_), {k: v["worst_over_allowed"] for k, v in rows.items()}), flush=True)
            report["greedy_keep_f32"] = keep_
            report["greedy_ratios"] = rows
    runtime.unregister_backend("cpu")
    if args.out:
        with open(os.path.join(REPO, args.out), "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
