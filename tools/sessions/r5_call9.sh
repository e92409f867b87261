#!/bin/bash
# round 5, GPU session 9: where to route the tile conv v3 (thresholds x edit ratios, stacked edits)
mkdir -p gpurun_out/r5i
cd /root/repo
export TMPDIR=/tmp
timeout 900 python tools/tile3_router_bench.py --out gpurun_out/r5i/tile3_router.json > gpurun_out/r5i/router.log 2>&1
tail -n 3 gpurun_out/r5i/router.log
