#!/bin/bash
# round 4, GPU session 11: plan policy of the tile conv on grids with many blocks (large edits, stacked edits)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4k; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/plan_policy_bench.py --fills 0:0,448:0,448:1,336:1,672:1,0:1 --ratios 0.012,0.05 --no-stacked --out $OUT/plan_fill.json > $OUT/plan_policy.log 2>&1
echo "plan policy rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; tail -80 $OUT/plan_policy.log
