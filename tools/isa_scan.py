#!/usr/bin/env python3
"""Static scan of every kernel of libsige_hip.so (no GPU): registers, spills, scratch, and how many
`load; s_waitcnt vmcnt(0)` serial points / scalar-argument fetches its ISA has (tools/isa_report.py per kernel).

    python tools/isa_scan.py > profiles/<snapshot>_isa_scan.txt
"""
import sys, os, re, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tools'))
import isa_report as R
out = []
for tu in ["conv_k3s1_nhwc.hip", "conv_k1_nhwc.hip", "conv_k3s2_nhwc.hip", "conv_k3s1_nhwc_w8.hip", "conv_k1_nhwc_w8.hip",
           "conv_pair_nhwc_t4.hip", "conv_pair_nhwc_f4.hip", "conv_pair_nhwc_f8.hip", "block_conv.hip", "mask_pipeline.hip", "nhwc_ops.hip", "group_norm.hip", "attention.hip", "conv_out.hip", "conv_in.hip", "gather.hip", "scatter.hip", "reduce_mask.hip"]:
    with tempfile.TemporaryDirectory() as d:
        asm = R.compile_to_asm(os.path.join(REPO, 'sige_amd', 'csrc', tu), d)
    res = R.resources(asm)
    names = R.demangle([n for n, _ in res])
    for (mn, r), nm in zip(res, names):
        try:
            n_lines, sk = R.skeleton(asm, mn)
        except SystemExit:
            continue
        serial = sum(1 for l in sk if "<-- serial" in l)
        vm0 = sum(1 for l in sk if "vmcnt(0)" in l)
        sl = sum(1 for l in sk if re.search(r"\bs_load", l))
        lg = sum(1 for l in sk if "lgkmcnt(0)" in l)
        out.append((tu, nm, r, n_lines, serial, vm0, sl, lg))
print("%-22s %5s %5s %7s %6s %6s %7s %6s  kernel" % ("file", "vgpr", "spill", "scratch", "serial", "vmcnt0", "s_loads", "lgkm0"))
for tu, nm, r, n_lines, serial, vm0, sl, lg in out:
    short = nm.replace("sige::", "").replace("void ", "")
    short = re.sub(r"\(.*", "", short)[:90]
    print("%-22s %5s %5s %7s %6d %6d %7d %6d  %s" % (tu, r["vgpr"], r["spill"], r["scratch"], serial, vm0, sl, lg, short))
