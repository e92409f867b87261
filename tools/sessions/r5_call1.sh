#!/bin/bash
# round 5, GPU session 1: the GPU suite after the boundary refactor, the bench line contract, where GauGAN's 40 ms go
mkdir -p gpurun_out/r5a
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5a/pytest.log
tail -5 gpurun_out/r5a/pytest.log
timeout 200 python tools/gaugan_latency.py --out gpurun_out/r5a/gaugan_latency_preload.json > gpurun_out/r5a/gl1.log 2>&1
SIGE_HIP_NO_PRELOAD=1 timeout 200 python tools/gaugan_latency.py --out gpurun_out/r5a/gaugan_latency_nopreload.json > gpurun_out/r5a/gl2.log 2>&1
tail -3 gpurun_out/r5a/gl1.log gpurun_out/r5a/gl2.log
timeout 600 python bench.py > gpurun_out/r5a/bench.out 2> gpurun_out/r5a/bench.err; echo "bench rc=$?"
tail -c 4200 gpurun_out/r5a/bench.out
cp bench_detail.json gpurun_out/r5a/ 2>/dev/null
