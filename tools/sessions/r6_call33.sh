#!/bin/bash
# round 6, GPU session 33: conv_mfma's first activation chunk requested AHEAD of the weights of the first two chunks (bafter) against
# the committed order (weights first)
mkdir -p gpurun_out/r6ag
cd /root/repo
export TMPDIR=/tmp
L=$PWD/sige_amd/lib
for rep in 1 2; do
SIGE_HIP_LIB=$L/libsige_hip_bafter.so timeout 300 python tools/forward_ab.py --tag activations-first >> gpurun_out/r6ag/forward_ab.jsonl 2>> gpurun_out/r6ag/err.log
timeout 300 python tools/forward_ab.py --tag weights-first >> gpurun_out/r6ag/forward_ab.jsonl 2>> gpurun_out/r6ag/err.log
done
python - <<'PY'
import json
for l in open("gpurun_out/r6ag/forward_ab.jsonl"):
    d = json.loads(l)
    print(d["tag"], d["dtype"], [(r["ratio"], r["forward_ms"], r["checksum"]) for r in d["rows"]])
PY
SIGE_HIP_LIB=$L/libsige_hip_bafter.so timeout 600 python tools/sd_fused_tokens_ab.py --settings 1,1 --out gpurun_out/r6ag/sd_bafter.json 2>> gpurun_out/r6ag/err.log
tail -n 2 gpurun_out/r6ag/err.log
