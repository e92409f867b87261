"""Per-layer timing of the dense layers of the DDPM-256 U-Net: the tile kernel with every tile active (exact fp32 / fp16
operands) against the dense-layer kernel on the fp16 matrix cores (split fp16 operands "f16x3" / fp16 operands "f16").

    python tools/wide_bench.py [--out gpurun_out/wide_bench.jsonl] [--ksplit-sweep]

Layers: the dense remainder of the sparse pass (32x32 ... 8x8, sige_fused_unet.py:112-123) and the shapes of the full pass
(256x256 ... 64x64).  Each case is a hipGraph of 8 launches over 8 rotating input / weight sets (cold weights: a forward
reads every layer's weights from HBM once), timed with HIP events on the launch stream."""
import argparse
import json
import os
import sys

import torch
from torch import nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

LAYERS = [
    # name, k, C1, C2, Cout, res, affine+swish, residual
    ("32^2 conv 256->256", 3, 256, 0, 256, 32, True, True),
    ("32^2 conv cat 512->256", 3, 256, 256, 256, 32, True, False),
    ("32^2 conv cat 768->256", 3, 512, 256, 256, 32, True, False),
    ("32^2 1x1 cat 768->256", 1, 512, 256, 256, 32, False, False),
    ("16^2 conv 256->512", 3, 256, 0, 512, 16, True, False),
    ("16^2 conv 512->512", 3, 512, 0, 512, 16, True, True),
    ("16^2 conv cat 1024->512", 3, 512, 512, 512, 16, True, False),
    ("16^2 1x1 qkv 512->1536", 1, 512, 0, 1536, 16, True, False),
    ("16^2 1x1 proj 512->512", 1, 512, 0, 512, 16, False, True),
    ("16^2 1x1 cat 1024->512", 1, 512, 512, 512, 16, False, False),
    ("8^2 conv 512->512", 3, 512, 0, 512, 8, True, True),
    ("8^2 conv cat 1024->512", 3, 512, 512, 512, 8, True, False),
    ("256^2 conv 128->128 (full pass)", 3, 128, 0, 128, 256, True, False),
    ("256^2 conv cat 256->128 (full pass)", 3, 128, 128, 128, 256, True, False),
    ("128^2 conv 128->128 (full pass)", 3, 128, 0, 128, 128, True, False),
    ("64^2 conv 256->256 (full pass)", 3, 256, 0, 256, 64, True, False),
    ("64^2 conv cat 512->256 (full pass)", 3, 256, 256, 256, 64, True, False),
]


def time_graph(fn, nsets, reps=5):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(nsets):
            fn(i)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(nsets):
                fn(i)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    for _ in range(2):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / nsets)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--ksplit-sweep", action="store_true")
    args = ap.parse_args()
    import os as _os; _os.environ.setdefault("SIGE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "sige_amd", "lib", "libsige_hip_tuning.so"))  # noqa: E702 -- dispatch knobs exist only in the measurement build (python -m sige_amd.build --tuning)
    from sige_amd import hip
    from sige_amd.nn import dense
    from sige_amd.nn.dense import fused_conv2d

    hip.lib()
    dense.WIDE_MIN_FLOP = {1: 0.0, 3: 0.0}  # (time the dense-layer kernel on every layer, also below the routing threshold)
    dev = "cuda"
    rows = []
    for name, k, c1, c2, cout, res, aff, resid in LAYERS:
        torch.manual_seed(0)
        nsets = 8 if res <= 64 else 3
        cl = lambda t: t.contiguous(memory_format=torch.channels_last)  # noqa: E731
        convs = [nn.Conv2d(c1 + c2, cout, k, 1, k // 2).to(dev) for _ in range(nsets)]
        xs = [cl(torch.randn(1, c1, res, res, device=dev)) for _ in range(nsets)]
        x2s = [cl(torch.randn(1, c2, res, res, device=dev)) if c2 else None for _ in range(nsets)]
        s, t = torch.randn(1, c1 + c2, 1, 1, device=dev), torch.randn(1, c1 + c2, 1, 1, device=dev)
        rs = [cl(torch.randn(1, cout, res, res, device=dev)) if resid else None for _ in range(nsets)]
        flops = 2.0 * res * res * cout * (c1 + c2) * k * k
        row = {"layer": name, "GFLOP": round(flops / 1e9, 3), "weights_MB_fp32": round(cout * (c1 + c2) * k * k * 4 / 1e6, 2)}

        def run_as(compute):
            for c in convs:
                c.compute_dtype = compute

            def fn(i):
                with torch.no_grad():
                    return fused_conv2d(convs[i], xs[i], s if aff else None, t if aff else None, "swish" if aff else "identity",
                                        x2=x2s[i], residual=rs[i])
            return fn

        ref = None
        for compute in ("f32", "f16x3", "f16"):
            fn = run_as(compute)
            n0 = hip.launch_count()
            out = fn(0)
            row["launches_" + compute] = hip.launch_count() - n0
            if compute == "f32":
                ref = out.clone()
            else:
                row["max_abs_vs_f32_" + compute] = float((out - ref).abs().max())
            us = time_graph(fn, nsets)
            row["us_" + compute] = round(us, 2)
            row["TFLOPs_" + compute] = round(flops / us / 1e6, 1)
        dense.WIDE_MIN_FLOP_F32 = {1: 0.0, 3: 0.0}   # exact fp32 on the dense-layer kernel
        try:
            row["us_f32_wide"] = round(time_graph(run_as("f32"), nsets), 2)
            row["TFLOPs_f32_wide"] = round(flops / row["us_f32_wide"] / 1e6, 1)
        finally:
            dense.WIDE_MIN_FLOP_F32 = {1: 1e30, 3: 1e30}
        if args.ksplit_sweep and res <= 32:
            for c in convs:
                c.compute_dtype = "f16x3"
            sw = {}
            for ks in (1, 2, 4, 8, 16):
                hip.wide_conv_force_ksplit(ks)
                try:
                    sw[str(ks)] = round(time_graph(run_as("f16x3"), nsets), 2)
                finally:
                    hip.wide_conv_force_ksplit(0)
            row["us_f16x3_by_ksplit"] = sw
        print(json.dumps(row), flush=True)
        rows.append(row)
        del convs, xs, x2s, rs
        torch.cuda.empty_cache()
    if args.out:
        os.makedirs(os.path.dirname(os.path.join(REPO, args.out)) or ".", exist_ok=True)
        with open(os.path.join(REPO, args.out), "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
