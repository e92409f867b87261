#!/bin/bash
# round 6, GPU session 44: the tile-conv translation units compiled with -mllvm -amdgpu-sched-strategy=max-ilp (the kernels run one
# wave per SIMD by design: nothing to gain from the default strategy's occupancy target) against the same build without the flag
mkdir -p gpurun_out/r6aq
cd /root/repo
export TMPDIR=/tmp
L=$PWD/sige_amd/lib
for rep in 1 2; do
SIGE_HIP_LIB=$L/libsige_hip_ilp.so timeout 300 python tools/forward_ab.py --tag max-ilp >> gpurun_out/r6aq/forward_ab.jsonl 2>> gpurun_out/r6aq/err.log
SIGE_HIP_LIB=$L/libsige_hip_tuning.so timeout 300 python tools/forward_ab.py --tag default-strategy >> gpurun_out/r6aq/forward_ab.jsonl 2>> gpurun_out/r6aq/err.log
done
timeout 300 python tools/forward_ab.py --tag product-library >> gpurun_out/r6aq/forward_ab.jsonl 2>> gpurun_out/r6aq/err.log
SIGE_HIP_LIB=$L/libsige_hip_ilp.so timeout 300 python tools/forward_ab.py --tag max-ilp --dtype f16 >> gpurun_out/r6aq/forward_ab.jsonl 2>> gpurun_out/r6aq/err.log
SIGE_HIP_LIB=$L/libsige_hip_tuning.so timeout 300 python tools/forward_ab.py --tag default-strategy --dtype f16 >> gpurun_out/r6aq/forward_ab.jsonl 2>> gpurun_out/r6aq/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r6aq/forward_ab.jsonl"):
    d = json.loads(l)
    print(d["tag"], d["dtype"], [(r["ratio"], r["forward_ms"]) for r in d["rows"]])
PY
tail -n 2 gpurun_out/r6aq/err.log
