#!/usr/bin/env python3
"""Per-layer timing of the MFMA stacked-block conv at the DDPM-256 shapes.

For every distinct conv of the sparse forward (tools/ddpm_conv_shapes.json: the
SIGE layers at a given edit ratio) and of the dense remainder (resolutions < 64),
times the fused gather->conv kernel for each output-block shape
(auto / 16x16 / 16x32 / 32x32 / 32x64) as a hipGraph of back-to-back launches
with rotating inputs (HIP events on the launch stream), checks the result against
torch's conv, and prints one JSON line per (layer, tile).

    python tools/conv_bench.py --ratio 0.012 [--dense] [--tiles auto,16x1,...]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.nn.functional as F

import os as _os; _os.environ.setdefault("SIGE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "sige_amd", "lib", "libsige_hip_tuning.so"))  # noqa: E702 -- dispatch knobs exist only in the measurement build (python -m sige_amd.build --tuning)
from sige_amd import hip

PEAK_TF = 157.3
GEO = {(3, 1): ((6, 6), (4, 4), (1, 1)), (1, 1): ((4, 4), (4, 4), (0, 0)), (3, 2): ((5, 5), (2, 2), (0, 0))}


def graph_time(fn, nsets, reps=20, iters=10):
    """us per call of fn(i) from a captured graph of `reps` calls."""
    for i in range(3):
        fn(i % nsets)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(2):
            fn(i % nsets)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for i in range(reps):
                fn(i % nsets)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (iters * reps)


CL = False


def sparse_case(name, T, cin, cout, k, stride, R, res, dev, nsets=4):
    """Gather-fused conv over T active tiles of a [1,cin,res,res] activation."""
    block, out_tile, offset = GEO[(k, stride)]
    pitch = out_tile[0] * stride
    n_side = res // pitch
    # T tiles in a compact square-ish patch
    w = max(1, int(round(T ** 0.5)))
    coords = [(i // w, i % w) for i in range(T)]
    idx = torch.tensor([[min(r, n_side - 1) * pitch - offset[0], min(c, n_side - 1) * pitch - offset[1]] for r, c in coords],
                       dtype=torch.int32, device=dev)
    xs = [torch.randn(1, cin, res, res, device=dev) for _ in range(nsets)]
    if CL:
        xs = [x.contiguous(memory_format=torch.channels_last) for x in xs]
    wgt = torch.randn(cout, cin, k, k, device=dev) / (k * cin ** 0.5)
    bias = torch.randn(cout, device=dev)
    scale, shift = torch.randn(1, cin, 1, 1, device=dev), torch.randn(1, cin, 1, 1, device=dev)
    packed = hip.conv_pack_weights(wgt, block[0], block[1], (stride, stride))

    def run(i):
        if CL:
            return hip.gather_conv_cl(xs[i], None, block, idx, scale, shift, "swish", packed, bias, cout, (k, k), (stride, stride))
        return hip.gather_conv(xs[i], block, idx, scale, shift, "swish", packed, bias, cout, (k, k), (stride, stride))

    def check():
        got = run(0).contiguous()
        tiles = hip.gather(xs[0].contiguous(), block[0], block[1], idx, scale, shift, "swish", False)
        want = F.conv2d(tiles.double(), wgt.double(), bias.double(), stride).float()
        return (got - want).abs().max().item()

    flop = 2.0 * T * out_tile[0] * out_tile[1] * cout * cin * k * k
    return run, check, flop


def dense_case(name, res, c1, c2, cout, k, stride, dev, nsets=4):
    conv = torch.nn.Conv2d(c1 + c2, cout, k, stride, 0 if stride == 2 else k // 2).to(dev)
    from sige_amd.nn.dense import fused_conv2d

    fmt = torch.channels_last if CL else torch.contiguous_format
    xs = [torch.randn(1, c1, res, res, device=dev).contiguous(memory_format=fmt) for _ in range(nsets)]
    x2s = [torch.randn(1, c2, res, res, device=dev).contiguous(memory_format=fmt) for _ in range(nsets)] if c2 else [None] * nsets
    s, t = torch.randn(1, c1 + c2, 1, 1, device=dev), torch.randn(1, c1 + c2, 1, 1, device=dev)
    ro = res if stride == 1 else res // 2
    residual = torch.randn(1, cout, ro, ro, device=dev).contiguous(memory_format=fmt)

    def run(i):
        with torch.no_grad():
            return fused_conv2d(conv, xs[i], s, t, "swish", x2=x2s[i], residual=residual, pad_bottom_right=stride == 2)

    def check():
        with torch.no_grad():
            got = run(0)
            h = xs[0] if c2 == 0 else torch.cat([xs[0], x2s[0]], 1)
            h = F.silu(h * s + t)
            if stride == 2:
                h = F.pad(h, (0, 1, 0, 1))
            want = F.conv2d(h.double(), conv.weight.double(), conv.bias.double(), stride, conv.padding).float() + residual
            return (got.contiguous() - want).abs().max().item()

    flop = 2.0 * ro * ro * cout * (c1 + c2) * k * k
    return run, check, flop


DENSE = [
    # name, res, c1, c2, cout, k, stride   (DDPM-256 church: levels 3-5 = 32^2 x256, 16^2 x512, 8^2 x512)
    ("d32.conv 256->256", 32, 256, 0, 256, 3, 1),
    ("d32.conv 512->256 (cat)", 32, 256, 256, 256, 3, 1),
    ("d32.nin 512->256 (cat)", 32, 256, 256, 256, 1, 1),
    ("d32.down 256->256 s2", 32, 256, 0, 256, 3, 2),
    ("d16.conv 256->512", 16, 256, 0, 512, 3, 1),
    ("d16.conv 512->512", 16, 512, 0, 512, 3, 1),
    ("d16.conv 1024->512 (cat)", 16, 512, 512, 512, 3, 1),
    ("d16.conv 768->512 (cat)", 16, 512, 256, 512, 3, 1),
    ("d16.nin 1024->512 (cat)", 16, 512, 512, 512, 1, 1),
    ("d16.qkv 512->1536", 16, 512, 0, 1536, 1, 1),
    ("d16.down 512->512 s2", 16, 512, 0, 512, 3, 2),
    ("d8.conv 512->512", 8, 512, 0, 512, 3, 1),
    ("d8.conv 1024->512 (cat)", 8, 512, 512, 512, 3, 1),
    ("d8.nin 1024->512 (cat)", 8, 512, 512, 512, 1, 1),
    ("d256.conv_out 128->3", 256, 128, 0, 3, 3, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ratio", default="0.012")
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--sparse", action="store_true")
    ap.add_argument("--tiles", default="auto,16x1,16x2,32x1,32x2")
    ap.add_argument("--waves", default="0", help="waves per workgroup to sweep, e.g. 0,4,8 (0 = automatic)")
    ap.add_argument("--ksplit", default="0", help="forced K splits to sweep, e.g. 0,2,4 (0 = automatic)")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cl", action="store_true", help="dense cases in channels-last")
    a = ap.parse_args()
    global CL
    CL = a.cl
    if not a.dense and not a.sparse:
        a.dense = a.sparse = True
    dev = torch.device("cuda")
    torch.manual_seed(0)
    cases = []
    if a.sparse:
        shapes = json.load(open(os.path.join(os.path.dirname(__file__), "ddpm_conv_shapes.json")))[a.ratio]
        seen = {}
        for name, T, cin, cout, k, stride, R in shapes:
            key = (T, cin, cout, k, stride)
            seen.setdefault(key, [name, 0])[1] += 1
        lvl_res = {"0": 256, "1": 128, "2": 64, "3": 32, "4": 16, "5": 8}
        for (T, cin, cout, k, stride), (name, count) in seen.items():
            lvl = name.split(".")[1]
            res = lvl_res[lvl] * (2 if "upsample" in name else 1)
            cases.append(("sparse", "%s x%d" % (name, count), count,
                          lambda T=T, cin=cin, cout=cout, k=k, stride=stride, res=res, name=name:
                          sparse_case(name, T, cin, cout, k, stride, None, res, dev), "T=%d %d->%d k%d s%d" % (T, cin, cout, k, stride)))
    if a.dense:
        for name, res, c1, c2, cout, k, stride in DENSE:
            cases.append(("dense", name, 1,
                          lambda res=res, c1=c1, c2=c2, cout=cout, k=k, stride=stride, name=name:
                          dense_case(name, res, c1, c2, cout, k, stride, dev), "res=%d %d+%d->%d k%d s%d" % (res, c1, c2, cout, k, stride)))
    tiles = a.tiles.split(",")
    for kind, name, count, make, desc in cases:
        run, check, flop = make()
        best = None
        for tile in tiles:
            if tile == "auto":
                hip.conv_force_tile(0, 0)
            else:
                mt, nb = tile.split("x")
                hip.conv_force_tile(int(mt), int(nb))
            for waves in [int(v) for v in a.waves.split(",")]:
                for ks in [int(v) for v in a.ksplit.split(",")]:
                    hip.conv_force_waves(waves)
                    hip.conv_force_ksplit(ks)
                    err = check()
                    us = graph_time(run, 4, reps=a.reps)
                    rec = dict(kind=kind, layer=name, shape=desc, tile=tile, waves=waves, ksplit=ks, us=round(us, 2),
                               GFLOP=round(flop / 1e9, 4), TFLOPs=round(flop / us / 1e6, 2),
                               frac=round(flop / us / 1e6 / PEAK_TF, 3), max_err=float("%.2e" % err))
                    print(json.dumps(rec), flush=True)
        hip.conv_force_tile(0, 0)
        hip.conv_force_waves(0)
        hip.conv_force_ksplit(0)


if __name__ == "__main__":
    main()
