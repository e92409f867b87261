"""How long does re-capturing the sparse forward's hipGraph take after a mask change, and what is it spent on?

    python tools/probe/recapture_probe.py

(a) torch.cuda.graph(...) as bench.py's capture(): a new private memory pool per capture, gc.collect() + empty_cache() on entry;
(b) CUDAGraph.capture_begin / capture_end with ONE pool handle and ONE capture stream for every capture, the previous graph
    destroyed first -- its blocks go back to the pool and are handed out again instead of hipMalloc'ed."""
import gc
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def main():
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
    stream, pool = torch.cuda.Stream(), torch.cuda.graph_pool_handle()
    # (a pool lives as long as a graph uses it: a one-kernel graph captured into it, kept for the life of the process, keeps the
    #  blocks of destroyed graphs cached in the pool)
    keeper = torch.cuda.CUDAGraph()
    dummy = torch.zeros(8, device=dev)
    with torch.cuda.stream(stream):
        keeper.capture_begin(pool=pool)
        dummy.add_(1.0)
        keeper.capture_end()
    torch.cuda.synchronize()
    with torch.no_grad():
        model.set_mode("full")
        model(x0, t)

        def new_mask(i):
            m = bench.edit_mask(0.012 + 0.003 * i).to(dev)
            x1 = x0 + noise * m
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.set_masks(downsample_mask(dilate_mask(m, 5), 8))
            model.set_mode("sparse")
            torch.cuda.synchronize()
            return x1, (time.perf_counter() - t0) * 1e3

        def cap_a(x1):
            g = torch.cuda.CUDAGraph()
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                out = model(x1, t)
            torch.cuda.current_stream().wait_stream(stream)
            return g, out

        def cap_b(x1):
            g = torch.cuda.CUDAGraph()
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                g.capture_begin(pool=pool, capture_error_mode="thread_local")
                try:
                    out = model(x1, t)
                finally:
                    g.capture_end()
            torch.cuda.current_stream().wait_stream(stream)
            return g, out

        for name, cap, eager_first in (("a: torch.cuda.graph, new pool", cap_a, True), ("b: shared pool, no gc / empty_cache", cap_b, True),
                                       ("b without the eager forward (capture is the first forward)", cap_b, False)):
            rows = []
            g = None
            for i in range(6):
                x1, ms_mask = new_mask(i + (0 if cap is cap_a else 7))
                t0 = time.perf_counter()
                if eager_first:
                    ref = model(x1, t).clone()
                    torch.cuda.synchronize()
                t1 = time.perf_counter()
                del g
                g, out = cap(x1)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                g.replay()
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                ok = bool(torch.equal(out, ref)) if eager_first else None
                rows.append((ms_mask, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, ok))
            med = lambda k: sorted(r[k] for r in rows[1:])[len(rows[1:]) // 2]  # noqa: E731
            print(json.dumps({"case": name, "set_masks_ms": round(med(0), 3), "eager_forward_ms": round(med(1), 3), "capture_ms": round(med(2), 3),
                              "first_replay_ms": round(med(3), 3), "replay_equals_eager": [r[4] for r in rows]}), flush=True)
            del g
            gc.collect()
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
