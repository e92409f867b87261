#!/bin/bash
# round 4, GPU session 15: weights of the next layers prefetched into the Infinity Cache on a side stream of the graph (probe)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4p; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/probe/prefetch_probe.py --out $OUT/prefetch_probe.json > $OUT/prefetch_probe.log 2>&1
echo "prefetch probe rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; tail -40 $OUT/prefetch_probe.log
