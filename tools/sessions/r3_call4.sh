#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_channels_last.py tests/test_gpu_round2.py -q -m gpu -k "window or channels_last_equals_nchw or benchmarked_forward or f16_policy or f16x3" > gpurun_out/r3d/pytest_sel.log 2>&1
echo "selected tests rc=$?" >> gpurun_out/r3d/summary.txt
timeout 900 python tools/routing_bench.py --out gpurun_out/r3d/routing.jsonl > gpurun_out/r3d/routing.log 2>&1
echo "routing rc=$?" >> gpurun_out/r3d/summary.txt
timeout 600 python tools/profile_data_movement.py > gpurun_out/r3d/data_movement.log 2>&1
echo "data movement rc=$?" >> gpurun_out/r3d/summary.txt
cat gpurun_out/r3d/summary.txt; tail -5 gpurun_out/r3d/pytest_sel.log; cat gpurun_out/r3d/routing.log | tail -20; tail -5 gpurun_out/r3d/data_movement.log
