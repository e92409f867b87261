#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3f; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_all.log 2>&1
echo "all gpu tests rc=$?" >> $OUT/summary.txt
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; tail -5 $OUT/pytest_all.log; tail -2 $OUT/smoke.log; tail -3 $OUT/bench.err
