#!/bin/bash
# round 6, GPU session 21: conv_in on the active windows only: tests, forward A/B, plan / stacked / multi-step tests
mkdir -p gpurun_out/r6u
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q > gpurun_out/r6u/pytest_round6.log 2>&1
tail -n 8 gpurun_out/r6u/pytest_round6.log
timeout 300 python tools/forward_ab.py --tag sparse-conv-in > gpurun_out/r6u/forward_ab.jsonl 2> gpurun_out/r6u/forward_ab.err
timeout 300 python tools/forward_ab.py --tag sparse-conv-in --dtype f16 >> gpurun_out/r6u/forward_ab.jsonl 2>> gpurun_out/r6u/forward_ab.err
cat gpurun_out/r6u/forward_ab.jsonl; tail -n 2 gpurun_out/r6u/forward_ab.err
timeout 1200 python -m pytest tests -x -q -m gpu -k "plan or stacked or benchmarked_forward or multi_step or sparse_update or inplace or ddpm or example" > gpurun_out/r6u/pytest_subset.log 2>&1
tail -n 4 gpurun_out/r6u/pytest_subset.log
