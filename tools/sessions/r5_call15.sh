#!/bin/bash
# round 5, GPU session 15: dense layers without an index list (the kernel computes the tile origins) -- parity tests, A/B of the headline.
# A RECORD, not a tool: the kernel change it measured gained nothing (profiles/r5p_bench_grid_origins_*.json) and was reverted; the
# SIGE_HIP_GRID_ORIGINS switch and the grid_origins test no longer exist.
mkdir -p gpurun_out/r5p
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "grid_origins or benchmarked_forward or workload_unet or stacked_edits_vs or launch_plan" > gpurun_out/r5p/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5p/pytest.log
tail -n 8 gpurun_out/r5p/pytest.log
for G in 1 0 1 0; do
  SIGE_HIP_GRID_ORIGINS=$G timeout 300 python bench.py --no-extras --cpu-seconds 1 --steps 30 --warmup 5 2> gpurun_out/r5p/bench_g$G.err | tail -1 > gpurun_out/r5p/bench_g$G.json
  python - <<PY
import json
d = json.load(open("/root/repo/gpurun_out/r5p/bench_g$G.json"))
print("grid_origins=$G", d.get("ms_per_step"), d.get("roofline", {}).get("frac"), d.get("parity_ok"), d.get("parity_max_abs"))
PY
done
SIGE_HIP_GRID_ORIGINS=1 timeout 300 python bench.py --dtype f16 --no-extras --cpu-seconds 1 2> gpurun_out/r5p/bench_f16.err | tail -1 > gpurun_out/r5p/bench_f16.json
python -c "
import json
d = json.load(open('/root/repo/gpurun_out/r5p/bench_f16.json')); print('f16', d.get('ms_per_step'), d.get('parity_ok'))"
