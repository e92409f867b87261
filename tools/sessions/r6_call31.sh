#!/bin/bash
# round 6, GPU session 31: kernarg_touch in the remaining kernels (conv_in / conv_out / gn / attention / scatter / spade / tokens):
# DDPM forward, SD forward, the tests of the touched kernels
mkdir -p gpurun_out/r6ae
cd /root/repo
export TMPDIR=/tmp
L=$PWD/sige_amd/lib
for rep in 1 2; do
timeout 300 python tools/forward_ab.py --tag touch-everywhere >> gpurun_out/r6ae/forward_ab.jsonl 2>> gpurun_out/r6ae/err.log
SIGE_HIP_LIB=$L/libsige_hip_notouch.so timeout 300 python tools/forward_ab.py --tag no-touch >> gpurun_out/r6ae/forward_ab.jsonl 2>> gpurun_out/r6ae/err.log
done
python - <<'PY'
import json
for l in open("gpurun_out/r6ae/forward_ab.jsonl"):
    d = json.loads(l)
    print(d["tag"], d["dtype"], [(r["ratio"], r["forward_ms"]) for r in d["rows"]])
PY
timeout 600 python tools/sd_fused_tokens_ab.py --settings 1,1 --out gpurun_out/r6ae/sd_touch.json 2>> gpurun_out/r6ae/err.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6ae/pytest.log 2>&1; tail -n 3 gpurun_out/r6ae/pytest.log
