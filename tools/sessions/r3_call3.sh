#!/bin/bash
# round 3, GPU session 3: the whole -m gpu suite on the new build, then the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "f16x3_tile or f16x3_fused or window" > gpurun_out/r3c/pytest_new.log 2>&1
echo "new tests rc=$?" >> gpurun_out/r3c/summary.txt
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r3c/pytest_all.log 2>&1
echo "all gpu tests rc=$?" >> gpurun_out/r3c/summary.txt
timeout 1200 python bench.py > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err
echo "bench rc=$?" >> gpurun_out/r3c/summary.txt
cat gpurun_out/r3c/summary.txt
tail -4 gpurun_out/r3c/pytest_new.log
tail -6 gpurun_out/r3c/pytest_all.log
tail -3 gpurun_out/r3c/bench.err
