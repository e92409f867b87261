#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3g; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "wide_conv or affine_act" > $OUT/pytest_wide.log 2>&1
echo "wide tests rc=$?" >> $OUT/summary.txt
timeout 600 python tools/wide_bench.py --out gpurun_out/r3g/wide_bench.jsonl > $OUT/wide_bench.log 2>&1
echo "wide bench rc=$?" >> $OUT/summary.txt
timeout 900 python tools/routing_bench.py --out gpurun_out/r3g/routing.jsonl > $OUT/routing.log 2>&1
echo "routing rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; tail -3 $OUT/pytest_wide.log; grep -v Warn $OUT/routing.log | tail -22
