#!/usr/bin/env python3
"""The ~85 ms host stalls seen between synchronisation points on this box (profiles/r5*_gaugan_latency*.json): neither code-object
loading, nor Python's GC, nor the interrupt wait path.  How often does ONE small launch + one wait stall, per way of waiting?

    python tools/sync_spike_probe.py [--n 2000] [--out gpurun_out/sync_spikes.json]
"""
import argparse
import json
import os
import sys
import time

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    x = torch.zeros(1 << 16, device=dev)
    big = torch.zeros(1 << 24, device=dev)
    torch.cuda.synchronize()
    res = {}

    def run(name, launch, wait, n=a.n):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            launch()
            wait()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts_sorted = sorted(ts)
        res[name] = {"n": n, "median_ms": round(ts_sorted[n // 2], 4), "p99_ms": round(ts_sorted[int(n * 0.99)], 4), "max_ms": round(ts_sorted[-1], 3),
                     "over_10ms": sum(1 for t in ts if t > 10.0), "total_s": round(sum(ts) / 1e3, 3),
                     "spike_positions": [i for i, t in enumerate(ts) if t > 10.0][:20]}

    ev = torch.cuda.Event()

    def ev_poll():
        ev.record()
        while not ev.query():
            pass

    def ev_sync():
        ev.record()
        ev.synchronize()

    small = lambda: x.add_(1.0)  # noqa: E731
    run("small_kernel+device_synchronize", small, torch.cuda.synchronize)
    run("small_kernel+stream_synchronize", small, lambda: torch.cuda.current_stream().synchronize())
    run("small_kernel+event_synchronize", small, ev_sync)
    run("small_kernel+event_query_poll", small, ev_poll)
    run("small_kernel+item", small, lambda: x[0].item())
    # a fresh allocation of a new size every time (what set_masks / a first forward under a new tile count does)
    sizes = [1000 + 37 * i for i in range(a.n)]
    it = iter(sizes)
    run("new_size_alloc+kernel+device_synchronize", lambda: torch.empty(next(it) * 64, device=dev).fill_(1.0), torch.cuda.synchronize)
    run("16M_kernel+device_synchronize", lambda: big.add_(1.0), torch.cuda.synchronize, n=500)
    # 200 launches per wait (a forward's worth)
    run("200_small_kernels+device_synchronize", lambda: [x.add_(1.0) for _ in range(200)], torch.cuda.synchronize, n=300)
    text = json.dumps(res, indent=1)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
