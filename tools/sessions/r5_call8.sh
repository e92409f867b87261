#!/bin/bash
# round 5, GPU session 8: tile conv v3 -- parity tests and the launch-by-launch / forward comparison with conv_mfma.hpp
mkdir -p gpurun_out/r5h
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q --tb=short -k "tile_conv3 or tile3" > gpurun_out/r5h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5h/pytest.log
tail -n 40 gpurun_out/r5h/pytest.log
timeout 600 python tools/tile3_bench.py --out gpurun_out/r5h/tile3_bench.json > gpurun_out/r5h/tile3_bench.log 2>&1
tail -n 5 gpurun_out/r5h/tile3_bench.log
