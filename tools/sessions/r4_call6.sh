#!/bin/bash
# round 4, GPU session 6: round-4 tests (dense-layer conv + shortcut pairing, stacked plan), routing bench with the pairing
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4f; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -q -m gpu --tb=short > $OUT/pytest_round4.log 2>&1
echo "round4 tests rc=$?" >> $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -q -m gpu --tb=short -k "wide or pair or ddpm" > $OUT/pytest_r23.log 2>&1
echo "round2/3 subset rc=$?" >> $OUT/summary.txt
timeout 900 python tools/routing_bench.py --out gpurun_out/r4f/routing.jsonl > $OUT/routing.log 2>&1
echo "routing rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; tail -25 $OUT/pytest_round4.log; tail -4 $OUT/pytest_r23.log; grep -v Warn $OUT/routing.log | tail -32
