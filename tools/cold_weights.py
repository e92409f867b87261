#!/usr/bin/env python3
"""How much does a tile-conv launch pay for weights that are not in any cache?  In the U-Net every layer's weights are
touched once per forward (455 MB per forward > 256 MB Infinity Cache), so each launch streams them from HBM.  Here: a
hipGraph of back-to-back launches of ONE layer shape cycling over n different weight tensors -- n = 1: weights stay in
L2 / Infinity Cache; n large: every launch reads cold weights.  Prints us per launch for both."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from sige_amd import hip  # noqa: E402
from tools.conv_bench import graph_time  # noqa: E402

dev = torch.device("cuda")
cl = lambda t: t.contiguous(memory_format=torch.channels_last)  # noqa: E731


def case(name, res, cin, cout, k, T, nsets_list=(1, 4, 96)):
    block = (6, 6) if k == 3 else (4, 4)
    off = (1, 1) if k == 3 else (0, 0)
    if T == 0:
        idx = hip.all_tiles(res, res, (4, 4), (1, 1), off, dev)
        full = dict(offset=off, out_res=(res, res), residual=None)
    else:
        n = res // 4
        cells = torch.randperm(n * n)[:T]
        idx = (torch.stack([cells // n * 4, cells % n * 4], 1).int() - off[0]).contiguous().to(dev)
        full = None
    xs = [cl(torch.randn(1, cin, res, res, device=dev)) for _ in range(4)]
    sc, sh = torch.randn(1, cin, 1, 1, device=dev), torch.randn(1, cin, 1, 1, device=dev)
    bias = torch.randn(cout, device=dev)
    out = {}
    for nsets in nsets_list:
        if nsets * cout * cin * k * k * 4 > 6e9:
            continue
        packed = [hip.conv_pack_weights(torch.randn(cout, cin, k, k, device=dev) / (k * cin ** 0.5), block[0], block[1], (1, 1))
                  for _ in range(nsets)]
        fn = lambda i: hip.gather_conv_cl(xs[i % 4], None, block, idx, sc, sh, "swish", packed[i % nsets], bias, cout,  # noqa: E731
                                          (k, k), (1, 1), full=full)
        out["weight_sets_%d" % nsets] = round(graph_time(fn, 96, reps=96, iters=5), 2)
        del packed
    print(json.dumps({"case": name, "weights_MB": round(cout * cin * k * k * 4 / 1e6, 2), "us_per_launch": out}), flush=True)


def main():
    torch.manual_seed(0)
    case("dense 32x32 512->256 k3", 32, 512, 256, 3, 0)
    case("dense 32x32 256->256 k3", 32, 256, 256, 3, 0)
    case("dense 16x16 1024->512 k3", 16, 1024, 512, 3, 0)
    case("dense 8x8 1024->512 k3", 8, 1024, 512, 3, 0)
    case("SIGE 64x64 T=18 256->256 k3", 64, 256, 256, 3, 18)
    case("SIGE 256x256 T=124 128->128 k3", 256, 128, 128, 3, 124)
    case("dense 16x16 1024->512 k1", 16, 1024, 512, 1, 0)


if __name__ == "__main__":
    main()
