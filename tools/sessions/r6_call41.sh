#!/bin/bash
# round 6, GPU session 41: tile numbers divided through a float reciprocal (tile_div) + the stacked-edit row window branch-free in the
# tile conv's slot set-up, against the library of the previous commit (lib/libsige_hip_prev.so); then the GPU suite on the new library
mkdir -p gpurun_out/r6ao
cd /root/repo
export TMPDIR=/tmp
L=$PWD/sige_amd/lib
for rep in 1 2; do
timeout 300 python tools/forward_ab.py --tag tile-div >> gpurun_out/r6ao/forward_ab.jsonl 2>> gpurun_out/r6ao/err.log
SIGE_HIP_LIB=$L/libsige_hip_prev.so timeout 300 python tools/forward_ab.py --tag previous >> gpurun_out/r6ao/forward_ab.jsonl 2>> gpurun_out/r6ao/err.log
done
timeout 300 python tools/forward_ab.py --tag tile-div --dtype f16 >> gpurun_out/r6ao/forward_ab.jsonl 2>> gpurun_out/r6ao/err.log
SIGE_HIP_LIB=$L/libsige_hip_prev.so timeout 300 python tools/forward_ab.py --tag previous --dtype f16 >> gpurun_out/r6ao/forward_ab.jsonl 2>> gpurun_out/r6ao/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r6ao/forward_ab.jsonl"):
    d = json.loads(l)
    print(d["tag"], d["dtype"], [(r["ratio"], r["forward_ms"]) for r in d["rows"]])
PY
timeout 1500 python -m pytest tests -x -q -m gpu --tb=short > gpurun_out/r6ao/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -n 3 gpurun_out/r6ao/pytest_gpu.log
tail -n 2 gpurun_out/r6ao/err.log
