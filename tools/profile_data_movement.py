#!/usr/bin/env python3
"""Profiling target: the standalone data-movement launches of bench.py's `data_movement` table (gather, scatter_gather, scatter
out-of-place / in-place; NCHW and channels-last; 15 % / C 256 / B 2 and 1.2 % / C 128 / B 1), once more under rocprofv3 so that
every row of that table has a profiler row next to it:

    rocprofv3 --kernel-trace --output-format csv -d OUT -o dm -- python tools/profile_data_movement.py
    python tools/trace_summary.py OUT/*kernel_trace.csv --replays 1 --by-grid --gap-ms 100 --out profiles/r2z_kerneltrace_data_movement.csv

and, for the COUNTER bytes of every row (two passes, one counter each, --kernel-trace only):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d F -o pmc -- python tools/profile_data_movement.py --pmc-manifest M.json
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d W -o pmc -- python tools/profile_data_movement.py --pmc-manifest M.json
    python tools/pmc_data_movement.py F/*counter_collection.csv W/*counter_collection.csv M.json profiles/pmc_data_movement.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import bench  # noqa: E402
from sige_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pmc-manifest", default="", help="counter mode: three eager launches per row, no timing; write which library launches they were")
a = ap.parse_args()
hip.lib()
dev = torch.device("cuda")
torch.zeros(1, device=dev)
torch.cuda.synchronize()
time.sleep(0.3)
if a.pmc_manifest:
    rows = []
    bench.data_movement_rooflines(hip, dev, pmc_manifest=rows)
    torch.cuda.synchronize()
    json.dump({"rows": rows, "source_hash": bench.source_hash(), "total_library_launches": hip.launch_count()}, open(a.pmc_manifest, "w"), indent=1)
else:
    res = bench.data_movement_rooflines(hip, dev)
    torch.cuda.synchronize()
    print(json.dumps(res["data_movement"]))
