"""Does it pay to pull the NEXT layers' weights into the Infinity Cache while a layer runs?  A sparse DDPM-256 forward touches
455 MB of packed weights once each -- more than the 256 MB Infinity Cache holds, so every launch streams its weights from HBM
(tools/cold_weights.py: +0.4 ... +4 us per launch against warm weights) while the forward as a whole uses a twentieth of the
HBM bandwidth.  Here: the same hipGraph with a side branch that reads the weights of the conv call D calls ahead (a strided torch
reduction on a second stream, gated by an event of the main stream so that it runs D calls ahead, not at the start).

    python tools/probe/prefetch_probe.py [--out gpurun_out/prefetch_probe.json]
"""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--ratio", type=float, default=0.012)
    args = ap.parse_args()
    import bench
    from sige_amd import hip
    from sige_amd.utils import dilate_mask, downsample_mask
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    hip.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig()).to(dev).eval().to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    x0, noise = (t.to(dev).contiguous(memory_format=torch.channels_last) for t in bench.make_inputs())
    t = torch.zeros(1, device=dev)
    res = {"edit_ratio": args.ratio, "rows": {}}
    orig_conv_fn = hip._conv_fn
    with torch.no_grad():
        model.set_mode("full")
        model(x0, t)
        m = bench.edit_mask(args.ratio).to(dev)
        x1 = x0 + noise * m
        model.set_masks(downsample_mask(dilate_mask(m, 5), 8))
        model.set_mode("sparse")
        for _ in range(3):
            model(x1, t)
        # the weights of one forward, in call order
        order = []

        def logging_conv_fn(name, packed):
            order.append(packed)
            return orig_conv_fn(name, packed)

        hip._conv_fn = logging_conv_fn
        model(x1, t)
        hip._conv_fn = orig_conv_fn
        res["conv_calls"] = len(order)
        res["weight_MB"] = round(sum(p.numel() * 4 for p in order) / 1e6, 1)

        def timed(fn, k=60):
            g, out = bench.capture_fn(fn, warm=2)
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

        ms0, ref = timed(lambda: model(x1, t))
        res["rows"]["no prefetch"] = {"forward_ms": round(ms0, 4)}
        side = torch.cuda.Stream()
        sink = torch.zeros(len(order) + 8, device=dev)
        for dist, stride in ((1, 32), (2, 32), (3, 32), (2, 16), (4, 32)):
            state = {"i": 0}

            def prefetching_conv_fn(name, packed):
                i = state["i"]
                state["i"] += 1
                j = i + dist
                if j < len(order):
                    main = torch.cuda.current_stream()
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        w = order[j].as_subclass(torch.Tensor).view(-1)
                        sink[j] = w[::stride].sum()  # one element per `stride` floats: every 128-byte (64-byte) line is fetched
                return orig_conv_fn(name, packed)

            def fwd():
                state["i"] = 0
                hip._conv_fn = prefetching_conv_fn
                try:
                    out = model(x1, t)
                finally:
                    hip._conv_fn = orig_conv_fn
                torch.cuda.current_stream().wait_stream(side)  # (join the side branch: the graph ends when both do)
                return out

            side.wait_stream(torch.cuda.current_stream())
            ms, out = timed(fwd)
            res["rows"]["prefetch %d calls ahead, one read per %d B" % (dist, stride * 4)] = {
                "forward_ms": round(ms, 4), "max_abs_vs_no_prefetch": float((out - ref).abs().max())}
        ms1, _ = timed(lambda: model(x1, t))
        res["rows"]["no prefetch (again)"] = {"forward_ms": round(ms1, 4)}
    print(json.dumps(res, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
