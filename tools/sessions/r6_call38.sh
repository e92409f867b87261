#!/bin/bash
# round 6, GPU session 38: the forward under pinned block shapes / K splits of the tile conv (measurement build)
mkdir -p gpurun_out/r6am
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python tools/block_policy_sweep.py --out gpurun_out/r6am/block_policy.json 2> gpurun_out/r6am/err.log | cut -c 1-200
tail -n 3 gpurun_out/r6am/err.log
