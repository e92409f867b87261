#!/bin/bash
# round 6, GPU session 30: every line of the kernel-argument segment fetched at the top of the conv kernels (kernarg_touch) against
# the same build without it: DDPM forward at three edit ratios (fp32, fp16), SD forward
mkdir -p gpurun_out/r6ad
cd /root/repo
export TMPDIR=/tmp
L=$PWD/sige_amd/lib
for rep in 1 2; do
timeout 300 python tools/forward_ab.py --tag kernarg-touch >> gpurun_out/r6ad/forward_ab.jsonl 2>> gpurun_out/r6ad/err.log
SIGE_HIP_LIB=$L/libsige_hip_notouch.so timeout 300 python tools/forward_ab.py --tag no-touch >> gpurun_out/r6ad/forward_ab.jsonl 2>> gpurun_out/r6ad/err.log
done
timeout 300 python tools/forward_ab.py --tag kernarg-touch --dtype f16 >> gpurun_out/r6ad/forward_ab.jsonl 2>> gpurun_out/r6ad/err.log
SIGE_HIP_LIB=$L/libsige_hip_notouch.so timeout 300 python tools/forward_ab.py --tag no-touch --dtype f16 >> gpurun_out/r6ad/forward_ab.jsonl 2>> gpurun_out/r6ad/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r6ad/forward_ab.jsonl"):
    d = json.loads(l)
    print(d["tag"], d["dtype"], [(r["ratio"], r["forward_ms"]) for r in d["rows"]])
PY
timeout 600 python tools/sd_fused_tokens_ab.py --settings 1,1 --out gpurun_out/r6ad/sd_touch.json 2>> gpurun_out/r6ad/err.log
SIGE_HIP_LIB=$L/libsige_hip_notouch.so timeout 600 python tools/sd_fused_tokens_ab.py --settings 1,1 --out gpurun_out/r6ad/sd_notouch.json 2>> gpurun_out/r6ad/err.log
tail -n 2 gpurun_out/r6ad/err.log
