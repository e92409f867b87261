#!/bin/bash
# round 4, GPU session 16: the whole GPU suite and smoke() at HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x --tb=short > $OUT/r4_pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r4_smoke.log 2>&1
echo "smoke rc=$?"
tail -3 $OUT/r4_pytest_gpu.log; tail -2 $OUT/r4_smoke.log
