#!/bin/bash
# round 4, GPU session 11: plan policy of the tile conv on grids with many blocks (large edits, stacked edits)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4k; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/plan_policy_bench.py --thresholds=-1,128 --ratios 0.012,0.05,0.15 --out $OUT/plan_policy_default.json > $OUT/plan_policy.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_channels_last.py tests/test_gpu_round2.py -q -m gpu -x --tb=short > $OUT/pytest_conv.log 2>&1
echo "conv tests rc=$?" >> $OUT/summary.txt
tail -3 $OUT/pytest_conv.log
echo "plan policy rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; tail -80 $OUT/plan_policy.log
