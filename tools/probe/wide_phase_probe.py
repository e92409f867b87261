#!/usr/bin/env python3
"""Where does a launch of the dense-layer conv (conv_wide.hpp) spend its time?  Phase timestamps from INSIDE the kernel.

    SIGE_PROBE_WIDE=1 SIGE_PROBE_TAG=_wide python -m sige_amd.build --probe     # lib/libsige_hip_probe_wide.so
    python tools/probe/wide_phase_probe.py                                      # on the GPU box

Stamps (lane 0 of every workgroup, s_memtime = 100 MHz): 0 entry | 1 prologue done | 2 K loop done (wave 0) | 3 all waves done |
4 accumulators in LDS | 5 output stores issued | 6 statistics written.  Printed per case: median ticks of each phase, the
workgroup's lifetime, the device-side span of the launch and the hipGraph-timed launch."""
import ctypes
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ.setdefault("SIGE_HIP_LIB", os.path.join(REPO, "sige_amd", "lib", "libsige_hip_probe_wide.so"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from sige_amd import hip  # noqa: E402

dev = torch.device("cuda")
cl = lambda t: t.contiguous(memory_format=torch.channels_last)  # noqa: E731
raw = hip.lib().handle
raw.sige_hip_wide_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
raw.sige_hip_wide_probe_clear.argtypes = []


def probe(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    assert raw.sige_hip_wide_probe_clear() == 0
    fn()  # (the buffer pointer is picked up by this launch)
    torch.cuda.synchronize()
    assert raw.sige_hip_wide_probe_clear() == 0
    fn()
    torch.cuda.synchronize()
    buf = np.zeros((4096, 8), dtype=np.uint64)
    assert raw.sige_hip_wide_probe_read(buf.ctypes.data, 4096) == 0
    st = buf[buf[:, 0] > 0].astype(np.int64)
    us = bench.time_graph_of(fn, reps=8)
    return st, us


def main():
    torch.manual_seed(0)
    cases = [("256^2 128->128", 128, 0, 128, 256), ("256^2 cat 256->128", 128, 128, 128, 256), ("64^2 cat 512->256", 256, 256, 256, 64),
             ("128^2 128->128", 128, 0, 128, 128)]
    for name, c1, c2, cout, res in cases:
        x = cl(torch.randn(1, c1, res, res, device=dev))
        x2 = cl(torch.randn(1, c2, res, res, device=dev)) if c2 else None
        w = torch.randn(cout, c1 + c2, 3, 3, device=dev) / (3 * (c1 + c2) ** 0.5)
        b = torch.randn(cout, device=dev)
        sc, sh = torch.randn(1, c1 + c2, 1, 1, device=dev), torch.randn(1, c1 + c2, 1, 1, device=dev)
        for compute in ("f16x3", "f16", "f32"):
            packed = hip.wide_conv_pack_weights(w, compute)
            for stats in (False, True):
                fn = lambda: hip.wide_conv_cl(x, x2, sc, sh, "swish", packed, b, cout, (3, 3), stats=stats)  # noqa: E731
                st, us = probe(fn)
                if len(st) == 0:
                    print(json.dumps({"case": name, "compute": compute, "error": "no stamps"}))
                    continue
                d = np.diff(st[:, :7], axis=1)
                span = int(st[:, 6].max() - st[:, 0].min())
                row = {"case": name, "compute": compute, "stats": stats, "workgroups_stamped": int(len(st)), "graph_launch_us": round(us, 2),
                       "span_ticks": span, "us_per_tick": round(us / max(span, 1), 4),
                       "median_ticks": {"0-1 prologue": int(np.median(d[:, 0])), "1-2 K loop": int(np.median(d[:, 1])), "2-3 wait for the other waves": int(np.median(d[:, 2])),
                                        "3-4 accumulators -> LDS": int(np.median(d[:, 3])), "4-5 epilogue": int(np.median(d[:, 4])),
                                        "5-6 statistics": int(np.median(d[:, 5])), "0-6 workgroup": int(np.median(st[:, 6] - st[:, 0]))}}
                print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
