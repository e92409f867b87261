#!/bin/bash
# round 6, GPU session 51: the tile conv's K-split policy without round 3's "stay unsplit between 112 and 224 blocks" exception (the
# 8x8 layers split in two again: per call 11.7 -> 10.3 / 19.0 -> 14.1 us in tools/probe/per_layer_policy.py) against the previous library
mkdir -p gpurun_out/r6az
cd /root/repo
export TMPDIR=/tmp
L=$PWD/sige_amd/lib
for rep in 1 2; do
timeout 300 python tools/forward_ab.py --tag split-8x8 >> gpurun_out/r6az/forward_ab.jsonl 2>> gpurun_out/r6az/err.log
SIGE_HIP_LIB=$L/libsige_hip_prev.so timeout 300 python tools/forward_ab.py --tag previous >> gpurun_out/r6az/forward_ab.jsonl 2>> gpurun_out/r6az/err.log
done
timeout 300 python tools/forward_ab.py --tag split-8x8 --dtype f16 >> gpurun_out/r6az/forward_ab.jsonl 2>> gpurun_out/r6az/err.log
SIGE_HIP_LIB=$L/libsige_hip_prev.so timeout 300 python tools/forward_ab.py --tag previous --dtype f16 >> gpurun_out/r6az/forward_ab.jsonl 2>> gpurun_out/r6az/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r6az/forward_ab.jsonl"):
    d = json.loads(l)
    print(d["tag"], d["dtype"], [(r["ratio"], r["forward_ms"], r["launches"]) for r in d["rows"]])
PY
tail -n 2 gpurun_out/r6az/err.log
