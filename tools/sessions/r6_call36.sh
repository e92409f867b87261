#!/bin/bash
# round 6, GPU session 36: is run_from_c = 5.2 ms (first plan run under a new mask) of the final evidence a one-off?  two more bench runs
mkdir -p gpurun_out/r6aj
cd /root/repo
export TMPDIR=/tmp
for i in 1 2; do
timeout 900 python bench.py > gpurun_out/r6aj/bench_$i.json 2> gpurun_out/r6aj/bench_$i.err
cp bench_detail.json gpurun_out/r6aj/bench_detail_$i.json 2>/dev/null || cp gpurun_out/bench_detail.json gpurun_out/r6aj/bench_detail_$i.json
python - <<PY
import json
d = json.load(open("gpurun_out/r6aj/bench_detail_$i.json"))["dynamic"]
print({k: d["mask_change_plan"][k] for k in ("bind_mask", "run_from_c", "capture", "first_replay", "to_first_output_ms")}, d["mask_change_parts_ms"])
PY
done
