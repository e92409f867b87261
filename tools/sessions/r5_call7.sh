#!/bin/bash
# round 5, GPU session 7: the round's evidence on the current kernel sources, part 1 -- whole GPU suite, smoke, rocprofv3 traces + PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short > $OUT/r5_pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?" > $OUT/r5_summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r5_smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/r5_summary.txt
timeout 1500 bash tools/gpu_profile_round.sh r5 > $OUT/r5_profile_round.log 2>&1
echo "profile round rc=$?" >> $OUT/r5_summary.txt
cat $OUT/r5_summary.txt; tail -n 5 $OUT/r5_pytest_gpu.log; tail -n 15 $OUT/r5_profile_round.log; head -n 30 $OUT/r5_pmc_traffic.txt
