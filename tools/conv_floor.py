#!/usr/bin/env python3
"""Fixed cost vs per-chunk cost of the fused conv: time vs Cin at a few tile counts (channels-last)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os; _os.environ.setdefault("SIGE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "sige_amd", "lib", "libsige_hip_tuning.so"))  # noqa: E702 -- dispatch knobs exist only in the measurement build (python -m sige_amd.build --tuning)
from sige_amd import hip
from tools.conv_bench import graph_time

dev = torch.device("cuda")
torch.manual_seed(0)
for T, res in ((4, 8), (16, 16), (64, 32)):
    n_side = res // 4
    idx = torch.tensor([[(i // n_side) * 4 - 1, (i % n_side) * 4 - 1] for i in range(T)], dtype=torch.int32, device=dev)
    for cout in (512,):
        for cin in (128, 512, 1024):
            for mode in ("swish",):
                xs = [torch.randn(1, cin, res, res, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(4)]
                ws = [torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5) for _ in range(4)]
                packs = [hip.conv_pack_weights(w, 6, 6, (1, 1)) for w in ws]
                bias = torch.randn(cout, device=dev)
                sc, sh = (torch.randn(1, cin, 1, 1, device=dev), torch.randn(1, cin, 1, 1, device=dev)) if mode == "swish" else (None, None)
                for tile in ("16x1w4", "16x2w4", "32x1w4", "16x1w8", "16x2w8", "32x1w8", "32x2w8"):
                    mt, nb = [int(v) for v in tile[:-2].split("x")]
                    hip.conv_force_tile(mt, nb)
                    hip.conv_force_waves(int(tile[-1]))
                    hip.KSPLIT = False
                    fn = lambda i: hip.gather_conv_cl(xs[i], None, (6, 6), idx, sc, sh, "swish" if mode == "swish" else "identity",
                                                      packs[i], bias, cout, (3, 3), (1, 1))
                    us = graph_time(fn, 4)
                    # same weights every call (L2 / MALL resident) for comparison
                    fn2 = lambda i: hip.gather_conv_cl(xs[i], None, (6, 6), idx, sc, sh, "swish" if mode == "swish" else "identity",
                                                       packs[0], bias, cout, (3, 3), (1, 1))
                    us2 = graph_time(fn2, 4)
                    flop = 2.0 * T * 16 * cout * cin * 9
                    print(json.dumps(dict(T=T, cin=cin, cout=cout, mode=mode, tile=tile, us_rot_w=round(us, 2), us_same_w=round(us2, 2),
                                          ideal_us=round(flop / 157.3e6, 2))), flush=True)
hip.conv_force_tile(0, 0)
hip.conv_force_waves(0)
