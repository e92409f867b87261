#!/bin/bash
# round 4, GPU session 18: stacked-mode default routing (dense layers >= 8 GFLOP on the dense-layer kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4t; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -q -m gpu --tb=short -k "stacked or plan" > $OUT/pytest_stacked.log 2>&1
echo "stacked tests rc=$?"; tail -3 $OUT/pytest_stacked.log
timeout 600 python tools/plan_policy_bench.py --thresholds=-1 --ratios 0.012 --out $OUT/stacked_default.json > $OUT/stacked_default.log 2>&1
echo "bench rc=$?"; tail -30 $OUT/stacked_default.log | grep -A4 "stacked"
