#!/bin/bash
# round 5, GPU session 16: RCCL with one rank (the collectives of sige_amd/parallel.py), bench stdout = the result lines only
mkdir -p gpurun_out/r5r
cd /root/repo
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_rccl.py -m gpu -q --tb=short > gpurun_out/r5r/pytest.log 2>&1; echo "pytest rc=$?"
tail -n 5 gpurun_out/r5r/pytest.log
timeout 600 python bench.py --no-extras --cpu-seconds 1 --steps 20 --warmup 5 > gpurun_out/r5r/bench.out 2> gpurun_out/r5r/bench.err; echo "bench rc=$?"
wc -l gpurun_out/r5r/bench.out; tail -n 1 gpurun_out/r5r/bench.out | cut -c1-300
timeout 600 python bench.py --gpus 2 --oversubscribe --backend gloo --steps 10 --warmup 3 --no-extras --cpu-seconds 1 > gpurun_out/r5r/bench2.out 2> gpurun_out/r5r/bench2.err; echo "bench2 rc=$?"
wc -l gpurun_out/r5r/bench2.out; tail -n 1 gpurun_out/r5r/bench2.out | cut -c1-300
