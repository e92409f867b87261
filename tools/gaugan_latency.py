#!/usr/bin/env python3
"""Where do the ~40 ms of a GauGAN edit go?  (`difference_mask_and_set_masks` + the first eager forward of a
NEW mask took 44.6 ms against a 6.67 ms dense forward; the one-off moved between set_masks and the first forward from run to run.)

Per edit and phase, wall time with a device synchronisation after each phase: difference mask, mask pyramid, set_masks, the first
eager forward, a second eager forward under the same mask; TWO passes over the same five edits (a first-use effect -- a code object
HIP loads on the first launch of one of its kernels, allocator growth -- is gone in the second pass, a per-mask cost is not).
SIGE_HIP_NO_PRELOAD=1 switches sige_hip_preload (every code object loaded at the first launch on a device) off.

    python tools/gaugan_latency.py [--out gpurun_out/gaugan_latency.json]
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def labels(dy=0, dx=0):
    rs = np.random.RandomState(3)
    coarse = rs.randint(0, 36, size=(32, 64))
    lab0 = np.kron(coarse, np.ones((8, 8), dtype=np.int64))
    lab1 = lab0.copy()
    lab1[85 + dy:136 + dy, 128 + dx:256 + dx] = (lab0[85 + dy:136 + dy, 128 + dx:256 + dx] + 5) % 36
    oh = lambda l: torch.nn.functional.one_hot(torch.from_numpy(l), 36).permute(2, 0, 1)[None].float().contiguous()  # noqa: E731
    return oh(lab0), oh(lab1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--gc", default="default", choices=["default", "freeze", "off"],
                    help="Python's cyclic collector during the edits: default | freeze (gc.collect(); gc.freeze() after the full pass: "
                         "what exists then is never traversed again) | off (gc.disable())")
    ap.add_argument("--host-inputs", action="store_true", help="build every edit's label map on the host inside the loop (pageable upload)")
    ap.add_argument("--spin", action="store_true", help="hipSetDeviceFlags(hipDeviceScheduleSpin) before the context exists: host "
                                                        "waits poll instead of sleeping on an interrupt")
    a = ap.parse_args()
    import gc

    if a.spin:
        import ctypes

        rc = ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(1)  # hipDeviceScheduleSpin
        print("hipSetDeviceFlags(spin) ->", rc, file=sys.stderr)
    from sige_amd import hip
    from sige_amd.utils import compute_difference_mask, dilate_mask, downsample_mask
    from sige_amd.workloads.gaugan_spade import SPADEConfig, SpadeGenerator

    dev = torch.device("cuda:0")
    t0 = time.perf_counter()
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    t_ctx = time.perf_counter() - t0
    t0 = time.perf_counter()
    n_units = hip.preload(0)
    t_pre = time.perf_counter() - t0
    torch.manual_seed(0)
    model = SpadeGenerator(SPADEConfig()).eval().to(dev).to(memory_format=torch.channels_last)
    model.set_scatter_inplace(True)
    cl = lambda t_: t_.to(dev).contiguous(memory_format=torch.channels_last)  # noqa: E731
    x0 = cl(labels()[0])
    sync = torch.cuda.synchronize
    rows = []
    with torch.no_grad():
        model.set_mode("full")
        t0 = time.perf_counter()
        model(x0)
        sync()
        t_full = time.perf_counter() - t0
        edits = ((20, 40), (-40, -60), (60, 120), (0, -100), (35, 10))
        # every edited label map resident on the GPU BEFORE the loop (default).  --host-inputs: built with numpy and uploaded from
        # pageable memory inside the loop, as round 4's bench did -- the driver registers such a buffer as a userptr, and when
        # Python frees it the MMU notifier evicts this process's queues; the restore is scheduled ~100 ms later (amdgpu KFD): the
        # ~85 ms stalls of profiles/r5b_gaugan_latency_*.json, at a random point of the next few milliseconds of GPU work
        resident = None if a.host_inputs else {e: cl(labels(*e)[1]) for e in edits}
        sync()
        if a.gc == "freeze":
            gc.collect()
            gc.freeze()
        elif a.gc == "off":
            gc.collect()
            gc.disable()
        for rnd in range(2):
            for dy, dx in edits:
                xi = cl(labels(dy, dx)[1]) if resident is None else resident[(dy, dx)]
                sync()
                n0 = hip.launch_count()
                g0 = [g_["collections"] for g_ in gc.get_stats()]
                ts = [time.perf_counter()]
                d = compute_difference_mask(x0, xi)
                sync(); ts.append(time.perf_counter())
                masks = downsample_mask(dilate_mask(d, 1), (model.sh, model.sw), dilation=2)
                sync(); ts.append(time.perf_counter())
                model.set_masks(masks)
                model.set_mode("sparse")
                sync(); ts.append(time.perf_counter())
                model(xi)
                sync(); ts.append(time.perf_counter())
                model(xi)
                sync(); ts.append(time.perf_counter())
                ms = [round((b - a_) * 1e3, 3) for a_, b in zip(ts, ts[1:])]
                rows.append({"pass": rnd, "edit": [dy, dx], "difference_mask": ms[0], "mask_pyramid": ms[1], "set_masks": ms[2],
                             "first_forward": ms[3], "second_forward": ms[4], "to_first_output": round(sum(ms[:4]), 3),
                             "library_launches": hip.launch_count() - n0,
                             "gc_collections_gen0_1_2": [g_["collections"] - b for g_, b in zip(gc.get_stats(), g0)],
                             "reserved_MB": round(torch.cuda.memory_reserved() / 2 ** 20, 1)})
    res = {"host_inputs": a.host_inputs, "spin": a.spin, "HSA_ENABLE_INTERRUPT": os.environ.get("HSA_ENABLE_INTERRUPT"), "gc": a.gc, "gc_objects": len(gc.get_objects()), "preload": not os.environ.get("SIGE_HIP_NO_PRELOAD"), "preload_units": n_units, "preload_ms": round(t_pre * 1e3, 1),
           "context_ms": round(t_ctx * 1e3, 1), "full_forward_first_ms": round(t_full * 1e3, 1), "rows": rows}
    text = json.dumps(res, indent=1)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
