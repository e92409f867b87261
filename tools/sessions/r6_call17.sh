#!/bin/bash
# round 6, GPU session 17: write-through stores in the standalone channels-last data-movement kernels (gather / scatter_gather / scatter /
# SPADE modulation): the data_movement rows of the bench, then the tests of those kernels
mkdir -p gpurun_out/r6p
cd /root/repo
export TMPDIR=/tmp
timeout 900 python bench.py --no-extras --cpu-seconds 0 --sweep '' --no-dynamic 2> gpurun_out/r6p/bench.err | tail -1 > gpurun_out/r6p/bench.json
cp bench_detail.json gpurun_out/r6p/bench_detail.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6p/bench_detail.json"))
print("forward_ms", d.get("forward_ms"))
for k in ("roofline_hbm", "roofline_gather", "roofline_scatter_gather"):
    r = d.get(k) or {}
    print(k, {q: r.get(q) for q in ("us", "frac", "frac_alg", "bytes")})
for row in d.get("data_movement", [])[:40]:
    print({q: row.get(q) for q in ("op", "layout", "us", "alg_GBps", "frac_of_hbm_peak")} if isinstance(row, dict) else row)
PY
timeout 900 python -m pytest tests -x -q -m gpu -k "golden_cases or cl or spade or gaugan or scatter or gather" > gpurun_out/r6p/pytest_subset.log 2>&1
tail -n 3 gpurun_out/r6p/pytest_subset.log
