#!/bin/bash
# round 6, GPU session 11: norm_out from tile differences (delta GroupNorm): tests, forward A/B (f32 and f16 configs)
mkdir -p gpurun_out/r6k
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q > gpurun_out/r6k/pytest_round6.log 2>&1
tail -n 12 gpurun_out/r6k/pytest_round6.log
timeout 300 python tools/forward_ab.py --tag delta-norm-out > gpurun_out/r6k/forward_ab.jsonl 2> gpurun_out/r6k/forward_ab.err
cat gpurun_out/r6k/forward_ab.jsonl; tail -n 2 gpurun_out/r6k/forward_ab.err
timeout 900 python -m pytest tests -x -q -m gpu -k "plan or stacked or benchmarked_forward or multi_step or sparse_update or inplace" > gpurun_out/r6k/pytest_subset.log 2>&1
tail -n 5 gpurun_out/r6k/pytest_subset.log
