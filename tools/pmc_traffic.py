#!/usr/bin/env python3
"""Fabric-side bytes per launch per kernel family from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE)
of `tools/profile_forward.py --mode eager --replays R`, reduced by tools/pmc_summary.py.
Usage: pmc_traffic.py FETCH.csv WRITE.csv OUT.json [forwards_profiled=7]"""
import csv
import json
import re
import sys

f = list(csv.DictReader(open(sys.argv[1])))
w = list(csv.DictReader(open(sys.argv[2])))
FORWARDS = int(sys.argv[4]) if len(sys.argv) > 4 else 7  # 3 warm-up + 4 measured sparse forwards


def fam(k):
    m = re.search(r"conv_mfma_kernel<sige::ConvGeo<(\d), (\d), (\d), (\d+)>, (\d), (\d), (\d), (\d), (\d), (\d)>", k)
    if m:  # template args: G, NB, SRC, MODE, DST, LAYOUT, W -- DST 0 = tiles; SRC 2 (scatter_gather) is always a SIGE layer
        return "block_conv_mfma" if (m.group(8) == "0" or m.group(6) == "2") else "dense_or_fused_scatter_conv_mfma"
    for n in ("scatter_tiles_nhwc", "conv_out_nhwc", "gn_partial_nhwc", "attn_apply_nhwc", "attn_scores_nhwc", "splitk_reduce"):
        if n in k:
            return n
    return None


agg = {}
for rows, key in ((f, "FETCH_SIZE"), (w, "WRITE_SIZE")):
    for r in rows:
        fm = fam(r["kernel"])
        if not fm:
            continue
        d = agg.setdefault(fm, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n_FETCH_SIZE": 0, "n_WRITE_SIZE": 0})
        d[key] += float(r[key]) * int(r["dispatches"])
        d["n_" + key] += int(r["dispatches"])
out = {}
for k, d in agg.items():
    n, nw = d["n_FETCH_SIZE"], max(1, d["n_WRITE_SIZE"])
    rd, wr = 2 * d["FETCH_SIZE"] / n / 1e3, d["WRITE_SIZE"] / nw / 1e3
    out[k] = {"dispatches_profiled": n, "launches_per_forward": round(n / FORWARDS, 1),
              "FETCH_SIZE_KB_raw_per_launch": round(d["FETCH_SIZE"] / n, 1), "WRITE_SIZE_KB_raw_per_launch": round(d["WRITE_SIZE"] / nw, 1),
              "hbm_read_MB_per_launch_corrected_x2": round(rd, 2), "hbm_write_MB_per_launch": round(wr, 2),
              "traffic_MB_per_launch": round(rd + wr, 2)}
meta = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python "
                  "tools/profile_forward.py --mode eager --replays 4 (NHWC, in-place scatter, 1.2% edit)",
        "correction": "FETCH_SIZE counted in KB and doubled (MI355X_MICROARCH.md: gfx950 reports 1/2 of wide coalesced reads); "
                      "WRITE_SIZE in KB, uncalibrated",
        "note": "L2-miss traffic towards the fabric: Infinity-Cache hits are included, so this is an upper bound on HBM bytes. "
                "block_conv_mfma = the SIGE convs that write tiles or are fed by scatter_gather; gather-fed convs writing a full "
                "tensor (dense remainder, and the conv -> scatter fused up/downsample convs) are counted together",
        "families": out}
json.dump(meta, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
