#!/bin/bash
# round 6, GPU session 39: the epilogue's unit set-up (addresses, residual prefetch) behind the prologue's activation loads instead of
# ahead of them (eprelate) against the committed order
mkdir -p gpurun_out/r6an
cd /root/repo
export TMPDIR=/tmp
L=$PWD/sige_amd/lib
for rep in 1 2; do
SIGE_HIP_LIB=$L/libsige_hip_eprelate.so timeout 300 python tools/forward_ab.py --tag epre-late >> gpurun_out/r6an/forward_ab.jsonl 2>> gpurun_out/r6an/err.log
timeout 300 python tools/forward_ab.py --tag committed-order >> gpurun_out/r6an/forward_ab.jsonl 2>> gpurun_out/r6an/err.log
done
SIGE_HIP_LIB=$L/libsige_hip_eprelate.so timeout 300 python tools/forward_ab.py --tag epre-late --dtype f16 >> gpurun_out/r6an/forward_ab.jsonl 2>> gpurun_out/r6an/err.log
timeout 300 python tools/forward_ab.py --tag committed-order --dtype f16 >> gpurun_out/r6an/forward_ab.jsonl 2>> gpurun_out/r6an/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r6an/forward_ab.jsonl"):
    d = json.loads(l)
    print(d["tag"], d["dtype"], [(r["ratio"], r["forward_ms"]) for r in d["rows"]])
PY
SIGE_HIP_LIB=$L/libsige_hip_eprelate.so timeout 600 python tools/sd_fused_tokens_ab.py --settings 1,1 --out gpurun_out/r6an/sd_eprelate.json 2>> gpurun_out/r6an/err.log
timeout 600 python tools/sd_fused_tokens_ab.py --settings 1,1 --out gpurun_out/r6an/sd_committed.json 2>> gpurun_out/r6an/err.log
tail -n 2 gpurun_out/r6an/err.log
