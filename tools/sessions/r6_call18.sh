#!/bin/bash
# round 6, GPU session 18: the v3 scatter_gather source with the cached affine + SiLU in its staging path (SD's conv2): tests, SD forward
mkdir -p gpurun_out/r6q
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -x -q -k "tile_conv3" > gpurun_out/r6q/pytest_tile3.log 2>&1
tail -n 6 gpurun_out/r6q/pytest_tile3.log
timeout 900 python -m pytest tests -x -q -m gpu -k "sd_ or sd_unet or spatial_transformer" > gpurun_out/r6q/pytest_sd.log 2>&1
tail -n 3 gpurun_out/r6q/pytest_sd.log
timeout 900 python bench.py --workload sd --steps 20 --warmup 5 2> gpurun_out/r6q/bench_sd.err | tail -1 > gpurun_out/r6q/bench_sd.json
cp bench_detail_sd.json gpurun_out/r6q/bench_sd_detail.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6q/bench_sd_detail.json"))
print({k: d.get(k) for k in ("forward_ms", "dense_forward_ms", "speedup_vs_dense", "parity_max_abs", "parity_ok", "library_launches_per_forward")})
PY
tail -n 2 gpurun_out/r6q/bench_sd.err
