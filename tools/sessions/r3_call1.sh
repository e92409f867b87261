#!/bin/bash
# round 3, GPU session 1: new kernel numerics + per-layer timing + twins validation + a first bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/r3a/device.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "wide_conv" > gpurun_out/r3a/pytest_wide.log 2>&1
echo "wide tests rc=$?" >> gpurun_out/r3a/summary.txt
timeout 600 python tools/wide_bench.py --out gpurun_out/r3a/wide_bench.jsonl --ksplit-sweep > gpurun_out/r3a/wide_bench.log 2>&1
echo "wide bench rc=$?" >> gpurun_out/r3a/summary.txt
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "not wide_conv" > gpurun_out/r3a/pytest_models.log 2>&1
echo "model tests rc=$?" >> gpurun_out/r3a/summary.txt
timeout 600 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "twin" > gpurun_out/r3a/pytest_twins.log 2>&1
echo "twins tests rc=$?" >> gpurun_out/r3a/summary.txt
timeout 600 python bench.py --no-extras --cpu-seconds 0 --f16-sweep "" --sweep "" > gpurun_out/r3a/bench_f32.json 2> gpurun_out/r3a/bench_f32.err
echo "bench rc=$?" >> gpurun_out/r3a/summary.txt
cat gpurun_out/r3a/summary.txt
tail -5 gpurun_out/r3a/pytest_wide.log
tail -5 gpurun_out/r3a/pytest_models.log
