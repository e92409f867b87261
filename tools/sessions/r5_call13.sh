#!/bin/bash
# round 5, GPU session 13: token helpers of the SD transformer -- unit tests, the SD parity tests, the SD bench line
mkdir -p gpurun_out/r5m
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_models_golden.py tests/test_gpu_round3.py -m gpu -q --tb=short -k "token or sd or SD" > gpurun_out/r5m/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5m/pytest.log
tail -n 25 gpurun_out/r5m/pytest.log
timeout 600 python bench.py --workload sd --steps 20 --warmup 5 > gpurun_out/r5m/bench_sd.out 2> gpurun_out/r5m/bench_sd.err; echo "bench rc=$?"
tail -n 1 gpurun_out/r5m/bench_sd.out | cut -c1-1500
timeout 300 python tools/torch_ops_probe.py --workload sd --out gpurun_out/r5m/torch_ops_sd.json > gpurun_out/r5m/probe.log 2>&1; head -c 400 gpurun_out/r5m/torch_ops_sd.json
