#!/usr/bin/env python3
"""Where does a tile-conv launch spend its time?  Phase timestamps from INSIDE the kernel (s_memtime of lane 0 of every
workgroup) for the shapes that dominate the DDPM-256 sparse forward, fp32 and f16 compute.

    python -m sige_amd.build --probe        # lib/libsige_hip_probe.so: conv kernels compiled with -DSIGE_CONV_PROBE
    python tools/conv_phase_probe.py        # on the GPU box

Stamps: 0 entry | 1 prologue loads issued (index / map round trips done) | 2 first chunk in LDS (data arrived) |
3 after the prologue barrier | 4 K loop done | 5 stores issued.  Printed per case: median over workgroups of each phase and
of entry -> stores, the device-side span (first entry -> last store stamp) and the hipGraph-timed launch (HIP events),
whose difference to the span is launch / dispatch / store-drain time.
"""
import ctypes
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("SIGE_HIP_LIB", os.path.join(REPO, "sige_amd", "lib", "libsige_hip_probe.so"))
CASES = os.environ.get("SIGE_PROBE_CASES", "")  # e.g. "f32:A,B,D": compute types and case letters to run (default: all)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import os as _os; _os.environ.setdefault("SIGE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "sige_amd", "lib", "libsige_hip_tuning.so"))  # noqa: E702 -- dispatch knobs exist only in the measurement build (python -m sige_amd.build --tuning)
from sige_amd import hip  # noqa: E402
from sige_amd.utils import dilate_mask, reduce_mask  # noqa: E402

dev = torch.device("cuda")
cl = lambda t: t.contiguous(memory_format=torch.channels_last)  # noqa: E731
raw = hip.lib().handle
raw.sige_hip_conv_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
raw.sige_hip_conv_probe_clear.argtypes = []


def probe(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    raw.sige_hip_conv_probe_clear()
    fn()
    torch.cuda.synchronize()
    buf = np.zeros((4096, 8), dtype=np.uint64)
    assert raw.sige_hip_conv_probe_read(buf.ctypes.data, 4096) == 0
    st = buf[buf[:, 0] > 0].astype(np.int64)
    us = bench.time_graph_of(fn, reps=8)
    return st, us


def report(name, fn):
    if CASES:
        comp, letter = name.split()[0], name.split()[1]
        want_c, _, want_l = CASES.partition(":")
        if comp not in want_c.split("+") or (want_l and letter not in want_l.split(",")):
            return
    st, us = probe(fn)
    if len(st) == 0:
        print(json.dumps({"case": name, "error": "no stamps"}))
        return
    d = np.diff(st[:, :6], axis=1)
    pre = {"0-6 first kernargs": int(np.median(st[:, 6] - st[:, 0])), "6-7 slot arithmetic + index requests": int(np.median(st[:, 7] - st[:, 6])),
           "7-1 weight / index-dependent / data loads issued": int(np.median(st[:, 1] - st[:, 7]))}
    span = int(st[:, 5].max() - st[:, 0].min())
    tick_us = us / max(span, 1)  # upper bound on the tick length: the span is shorter than the launch
    row = {"lib": os.path.basename(os.environ["SIGE_HIP_LIB"]), "case": name, "workgroups": int(len(st)), "graph_launch_us": round(us, 2),
           "median_ticks": {"0-1 idx/map trips": int(np.median(d[:, 0])), "1-2 data arrives -> LDS": int(np.median(d[:, 1])),
                            "2-3 barrier": int(np.median(d[:, 2])), "3-4 K loop": int(np.median(d[:, 3])),
                            "4-5 reduce + epilogue": int(np.median(d[:, 4])), "0-5 total": int(np.median(st[:, 5] - st[:, 0]))},
           "prologue_split_ticks": pre}
    print(json.dumps(row), flush=True)


def main():
    torch.manual_seed(0)
    for compute in ("f32", "f16"):
        # A: dense 32x32 layer, gather (affine + SiLU) -> conv -> full tensor + residual
        C = 256
        x = cl(torch.randn(1, C, 32, 32, device=dev))
        w = torch.randn(C, C, 3, 3, device=dev) / 48
        b = torch.randn(C, device=dev)
        sc, sh = torch.randn(1, C, 1, 1, device=dev), torch.randn(1, C, 1, 1, device=dev)
        res = cl(torch.randn(1, C, 32, 32, device=dev))
        idx = hip.all_tiles(32, 32, (4, 4), (1, 1), (1, 1), dev)
        p = hip.conv_pack_weights(w, 6, 6, (1, 1), compute)
        report("%s A dense 32x32 C256 gather+swish -> full (+res)" % compute,
               lambda: hip.gather_conv_cl(x, None, (6, 6), idx, sc, sh, "swish", p, b, C, (3, 3), (1, 1),
                                          full=dict(offset=(1, 1), out_res=(32, 32), residual=res)))
        report("%s A' same, raw staging" % compute,
               lambda: hip.gather_conv_cl(x, None, (6, 6), idx, None, None, "identity", p, b, C, (3, 3), (1, 1),
                                          full=dict(offset=(1, 1), out_res=(32, 32), residual=res)))
        # F: dense 16x16, fused cat 512 + 512 -> 512 (K = 9216): the deep-K layers of the up path, 4 and 8 waves per workgroup
        Cf = 512
        xf, xf2 = cl(torch.randn(1, Cf, 16, 16, device=dev)), cl(torch.randn(1, Cf, 16, 16, device=dev))
        wf = torch.randn(Cf, 2 * Cf, 3, 3, device=dev) / 96
        bf = torch.randn(Cf, device=dev)
        scf, shf = torch.randn(1, 2 * Cf, 1, 1, device=dev), torch.randn(1, 2 * Cf, 1, 1, device=dev)
        idxf = hip.all_tiles(16, 16, (4, 4), (1, 1), (1, 1), dev)
        pf = hip.conv_pack_weights(wf, 6, 6, (1, 1), compute)
        for waves in ((4, 8) if compute == "f32" else (4,)):
            hip.conv_force_waves(waves)
            report("%s F%d dense 16x16 cat 512+512->512 gather+swish -> full, %d waves" % (compute, waves, waves),
                   lambda: hip.gather_conv_cl(xf, xf2, (6, 6), idxf, scf, shf, "swish", pf, bf, Cf, (3, 3), (1, 1),
                                              full=dict(offset=(1, 1), out_res=(16, 16), residual=None)))
            report("%s G%d same, raw staging, %d waves" % (compute, waves, waves),
                   lambda: hip.gather_conv_cl(xf, xf2, (6, 6), idxf, None, None, "identity", pf, bf, Cf, (3, 3), (1, 1),
                                              full=dict(offset=(1, 1), out_res=(16, 16), residual=None)))
        hip.conv_force_waves(0)
        # B: SIGE 64x64 conv2: scatter_gather -> conv -> scatter with block residual
        m = torch.zeros(64, 64, dtype=torch.bool, device=dev)
        m[25:32, 22:29] = True
        m = dilate_mask(dilate_mask(m, (2, 0)), (0, 2))
        i6, i4 = reduce_mask(m, 6, 4, 1), reduce_mask(m, 4, 4, 0)
        smap = hip.get_scatter_map(64, 64, 6, 6, 3, 3, 1, 1, 1, 1, i6)
        t1 = hip.tile_table(i4, (0, 0), (1, 1), (4, 4), (64, 64))
        y, y1 = cl(torch.randn(1, C, 64, 64, device=dev)), cl(torch.randn(1, C, 64, 64, device=dev))
        t4 = cl(torch.randn(i6.shape[0], C, 4, 4, device=dev))
        x1 = cl(torch.randn(i4.shape[0], C, 4, 4, device=dev))
        out = y.clone(memory_format=torch.preserve_format)
        report("%s B SIGE 64x64 C256 T=%d scatter_gather -> conv -> scatter (block residual)" % (compute, i6.shape[0]),
               lambda: hip.scatter_gather_conv_scatter_cl(t4, y, (6, 6), i6, smap, None, None, "identity", p, b, C, (3, 3), (1, 1),
                                                          out, residual=y1, x1=x1, table1=t1))
        report("%s B' same tiles, scatter_gather -> conv -> tiles" % compute,
               lambda: hip.scatter_gather_conv_cl(t4, y, (6, 6), i6, smap, None, None, "identity", p, b, C, (3, 3), (1, 1)))
        # C: SIGE 256x256 conv1: gather (affine + SiLU) -> conv -> tiles with the consumer's affine in the epilogue
        C2 = 128
        m = dilate_mask(bench.square_mask(0.012).to(dev), 5)
        i6 = reduce_mask(m, 6, 4, 1)
        xb = cl(torch.randn(1, C2, 256, 256, device=dev))
        w2 = torch.randn(C2, C2, 3, 3, device=dev) / 34
        b2 = torch.randn(C2, device=dev)
        s2, h2 = torch.randn(1, C2, 1, 1, device=dev), torch.randn(1, C2, 1, 1, device=dev)
        p2 = hip.conv_pack_weights(w2, 6, 6, (1, 1), compute)
        report("%s C SIGE 256x256 C128 T=%d gather+swish -> conv -> tiles" % (compute, i6.shape[0]),
               lambda: hip.gather_conv_cl(xb, None, (6, 6), i6, s2, h2, "swish", p2, b2, C2, (3, 3), (1, 1),
                                          out_affine=(s2.reshape(-1), h2.reshape(-1), "swish")))
        tiles = cl(torch.randn(i6.shape[0], C2, 6, 6, device=dev))
        report("%s D tile slab T=%d C128 -> conv -> tiles (no gather)" % (compute, i6.shape[0]),
               lambda: hip.block_conv_cl(tiles, p2, b2, C2, (3, 3), (1, 1)))
        # E: 1x1 shortcut
        i4 = reduce_mask(m, 4, 4, 0)
        w1 = torch.randn(C2, C2, 1, 1, device=dev) / 11
        p1 = hip.conv_pack_weights(w1, 4, 4, (1, 1), compute)
        report("%s E SIGE 256x256 1x1 C128 T=%d gather raw -> conv -> tiles" % (compute, i4.shape[0]),
               lambda: hip.gather_conv_cl(xb, None, (4, 4), i4, None, None, "identity", p1, b2, C2, (1, 1), (1, 1)))


if __name__ == "__main__":
    main()
