#!/bin/bash
# round 6, GPU session 1: the driver's own command on a fresh box (pytest -x -q -m gpu, new collection order), then the whole
# suite without -x, the margin distribution of the test that failed in GPUTEST_r05, the default bench line
mkdir -p gpurun_out/r6a
cd /root/repo
export TMPDIR=/tmp
rm -f gpurun_out/test_margins.jsonl
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6a/pytest_gpu_x.log 2>&1
echo "pytest -x rc $?"; tail -n 3 gpurun_out/r6a/pytest_gpu_x.log
cp gpurun_out/test_margins.jsonl gpurun_out/r6a/test_margins.jsonl 2>/dev/null
timeout 900 python tools/twins_margin.py --reps 8 --out gpurun_out/r6a/twins_margin.jsonl > gpurun_out/r6a/twins_margin.log 2>&1
tail -n 7 gpurun_out/r6a/twins_margin.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r6a/bench.err | tail -1 > gpurun_out/r6a/bench.json
cp bench_detail.json gpurun_out/r6a/bench_detail.json
cat gpurun_out/r6a/bench.json | cut -c 1-1500
