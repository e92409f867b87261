"""`sige_hip_attention_tokens_f32` alone at Stable Diffusion's shapes (15 % edit of a 64 x 64 latent, batch 2: self-attention
over all HW keys from the active tiles' queries, cross-attention over 77 text tokens), a graph of 8 back-to-back launches per shape;
checked against softmax(q k^T) v in fp64.  SIGE_HIP_LIB selects the build.

    python tools/attention_tokens_bench.py [--tag name] >> gpurun_out/attention_tokens.jsonl
"""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

# (name, B, Nq, Nk, C, heads)
SHAPES = [("self_64", 2, 1008, 4096, 320, 8), ("self_32", 2, 160, 1024, 640, 8), ("self_16", 2, 48, 256, 1280, 8),
          ("cross_64", 2, 1008, 77, 320, 8), ("cross_32", 2, 160, 77, 640, 8), ("self_64_b1", 1, 1008, 4096, 320, 8),
          ("self_64_dense", 2, 4096, 4096, 320, 8)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="")
    ap.add_argument("--eager", type=int, default=0, help="N eager launches per shape and nothing else (for rocprofv3 --pmc)")
    ap.add_argument("--only", default="", help="comma-separated shape names")
    ap.add_argument("--form", type=int, default=None, help="attention_form knob (needs SIGE_HIP_LIB=.../libsige_hip_tuning.so)")
    args = ap.parse_args()
    import bench
    from sige_amd import hip

    dev = torch.device("cuda", 0)
    if args.form is not None:
        hip.tuning_set("attention_form", args.form)
    torch.manual_seed(0)
    rows = []
    for name, B, Nq, Nk, C, heads in SHAPES:
        q, k, v = (torch.randn(B, n, C, device=dev) for n in (Nq, Nk, Nk))
        scale = (C // heads) ** -0.5
        if args.only and name not in args.only.split(","):
            continue
        if args.eager:
            for _ in range(args.eager):
                hip.attention_tokens(q, k, v, heads, scale)
            torch.cuda.synchronize()
            continue
        g, out = bench.capture_fn(lambda: [hip.attention_tokens(q, k, v, heads, scale) for _ in range(8)][-1], warm=2)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 160 * 1e3
        d = C // heads
        split = lambda t: t.double().view(B, -1, heads, d).transpose(1, 2)
        want = torch.softmax(split(q) @ split(k).transpose(-1, -2) * scale, -1) @ split(v)
        err = float((out.double() - want.transpose(1, 2).reshape(B, Nq, C)).abs().max())
        flop = 4.0 * B * heads * Nq * Nk * d
        rows.append({"shape": name, "B": B, "Nq": Nq, "Nk": Nk, "C": C, "heads": heads, "us": round(us, 2),
                     "tflops": round(flop / us * 1e-6, 1), "max_abs_err_vs_f64": err})
    print(json.dumps({"lib": os.path.basename(hip.LIB_PATH), "tag": args.tag, "rows": rows}))


if __name__ == "__main__":
    main()
