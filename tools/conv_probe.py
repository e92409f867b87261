#!/usr/bin/env python3
"""Where does the conv kernel's time go?  Times the same conv with each staging
source (tile slab / gather raw / gather affine / gather affine+SiLU) and each
output block, and (--pmc) just runs each variant N times for rocprofv3 --pmc."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import os as _os; _os.environ.setdefault("SIGE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "sige_amd", "lib", "libsige_hip_tuning.so"))  # noqa: E702 -- dispatch knobs exist only in the measurement build (python -m sige_amd.build --tuning)
from sige_amd import hip
from tools.conv_bench import graph_time

CASES = [  # name, T, cin, cout, res
    ("s256 T124 128->128", 124, 128, 128, 256),
    ("s64 T18 256->256", 18, 256, 256, 64),
    ("d16 T16 512->512", 16, 512, 512, 16),
    ("d32 T64 256->256", 64, 256, 256, 32),
    ("d8 T4 512->512", 4, 512, 512, 8),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--tiles", default="16x1,16x2,32x1,32x2")
    ap.add_argument("--n", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    for name, T, cin, cout, res in CASES:
        n_side = res // 4
        w = max(1, int(round(T ** 0.5)))
        idx = torch.tensor([[min(i // w, n_side - 1) * 4 - 1, min(i % w, n_side - 1) * 4 - 1] for i in range(T)],
                           dtype=torch.int32, device=dev)
        xs = [torch.randn(1, cin, res, res, device=dev) for _ in range(4)]
        ts = [torch.randn(T, cin, 6, 6, device=dev) for _ in range(4)]
        xcl = [t.contiguous(memory_format=torch.channels_last) for t in xs]
        tcl = [t.contiguous(memory_format=torch.channels_last) for t in ts]
        wgt = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
        bias = torch.randn(cout, device=dev)
        sc, sh = torch.randn(1, cin, 1, 1, device=dev), torch.randn(1, cin, 1, 1, device=dev)
        packed = hip.conv_pack_weights(wgt, 6, 6, (1, 1))
        variants = {
            "tiles": lambda i: hip.block_conv(ts[i], packed, bias, cout, (3, 3), (1, 1)),
            "gather_raw": lambda i: hip.gather_conv(xs[i], (6, 6), idx, None, None, "identity", packed, bias, cout, (3, 3), (1, 1)),
            "gather_affine": lambda i: hip.gather_conv(xs[i], (6, 6), idx, sc, sh, "identity", packed, bias, cout, (3, 3), (1, 1)),
            "gather_swish": lambda i: hip.gather_conv(xs[i], (6, 6), idx, sc, sh, "swish", packed, bias, cout, (3, 3), (1, 1)),
            "cl_tiles": lambda i: hip.block_conv_cl(tcl[i], packed, bias, cout, (3, 3), (1, 1)),
            "cl_gather_raw": lambda i: hip.gather_conv_cl(xcl[i], None, (6, 6), idx, None, None, "identity", packed, bias, cout, (3, 3), (1, 1)),
            "cl_gather_swish": lambda i: hip.gather_conv_cl(xcl[i], None, (6, 6), idx, sc, sh, "swish", packed, bias, cout, (3, 3), (1, 1)),
        }
        flop = 2.0 * T * 16 * cout * cin * 9
        for tile in a.tiles.split(","):
            mt, nb = [int(v) for v in tile.split("x")]
            hip.conv_force_tile(mt, nb)
            for vname, fn in variants.items():
                if a.pmc:
                    for i in range(a.n):
                        fn(i % 4)
                    torch.cuda.synchronize()
                else:
                    us = graph_time(fn, 4)
                    print(json.dumps(dict(case=name, tile=tile, src=vname, us=round(us, 2), TFLOPs=round(flop / us / 1e6, 1),
                                          ideal_us=round(flop / 157.3e6, 2))), flush=True)
        hip.conv_force_tile(0, 0)


if __name__ == "__main__":
    main()
