#!/bin/bash
# round 4, GPU session 3: two debug probes (stacked edits vs single edits per module; which K-split launches mismatch)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4c; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/probe/stacked_debug.py > $OUT/stacked_debug.log 2>&1
echo "stacked debug rc=$?" >> $OUT/summary.txt
timeout 600 python tools/probe/ksplit_stress_debug.py > $OUT/ksplit_debug.log 2>&1
echo "ksplit debug rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; grep -v Warn $OUT/stacked_debug.log | tail -20; grep -v Warn $OUT/ksplit_debug.log | tail -20
