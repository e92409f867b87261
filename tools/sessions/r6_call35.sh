#!/bin/bash
# round 6, GPU session 35: attn_apply_nhwc -- score rows requested ahead of the values, values as buffer loads with scalar key-step
# offsets, row reductions on DPP: attention tests, the call alone, the forward against the previous build
mkdir -p gpurun_out/r6ai
cd /root/repo
export TMPDIR=/tmp
L=$PWD/sige_amd/lib
timeout 900 python -m pytest tests -x -q -m gpu -k "attention or benchmarked_forward or ddpm" > gpurun_out/r6ai/pytest.log 2>&1; tail -n 3 gpurun_out/r6ai/pytest.log
for rep in 1 2; do
timeout 300 python tools/forward_ab.py --tag attn-apply >> gpurun_out/r6ai/forward_ab.jsonl 2>> gpurun_out/r6ai/err.log
SIGE_HIP_LIB=$L/libsige_hip_prev.so timeout 300 python tools/forward_ab.py --tag previous >> gpurun_out/r6ai/forward_ab.jsonl 2>> gpurun_out/r6ai/err.log
done
python - <<'PY'
import json
for l in open("gpurun_out/r6ai/forward_ab.jsonl"):
    d = json.loads(l)
    print(d["tag"], d["dtype"], [(r["ratio"], r["forward_ms"]) for r in d["rows"]])
PY
timeout 300 python tools/attention_ab.py --out gpurun_out/r6ai/attention_ab.json > /dev/null 2>> gpurun_out/r6ai/err.log
SIGE_HIP_LIB=$L/libsige_hip_prev.so timeout 300 python tools/attention_ab.py --out gpurun_out/r6ai/attention_ab_prev.json > /dev/null 2>> gpurun_out/r6ai/err.log
python - <<'PY'
import json
for f in ("attention_ab", "attention_ab_prev"):
    try:
        d = json.load(open("gpurun_out/r6ai/%s.json" % f)); print(f, d["attention_call"])
    except Exception as e:
        print(f, "failed", e)
PY
tail -n 2 gpurun_out/r6ai/err.log
