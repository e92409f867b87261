#!/bin/bash
# round 3, GPU session 2: wide kernel v2 numerics + per-layer timing, all round-3 model tests, the full bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3b
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "wide_conv" > gpurun_out/r3b/pytest_wide.log 2>&1
echo "wide tests rc=$?" >> gpurun_out/r3b/summary.txt
timeout 600 python tools/wide_bench.py --out gpurun_out/r3b/wide_bench.jsonl --ksplit-sweep > gpurun_out/r3b/wide_bench.log 2>&1
echo "wide bench rc=$?" >> gpurun_out/r3b/summary.txt
timeout 1200 python -m pytest tests/test_gpu_round3.py -q -m gpu -k "not wide_conv" > gpurun_out/r3b/pytest_models.log 2>&1
echo "model tests rc=$?" >> gpurun_out/r3b/summary.txt
timeout 900 python bench.py --no-extras > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err
echo "bench rc=$?" >> gpurun_out/r3b/summary.txt
timeout 900 python bench.py --workload sd --steps 20 --warmup 3 > gpurun_out/r3b/bench_sd.json 2> gpurun_out/r3b/bench_sd.err
echo "bench sd rc=$?" >> gpurun_out/r3b/summary.txt
cat gpurun_out/r3b/summary.txt
tail -3 gpurun_out/r3b/pytest_wide.log
tail -8 gpurun_out/r3b/pytest_models.log
tail -3 gpurun_out/r3b/bench.err
