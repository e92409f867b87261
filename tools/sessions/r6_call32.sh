#!/bin/bash
# round 6, GPU session 32: the weights of the first two chunks requested right behind the index loads (the bias / out-affine loads,
# whose branches drew the index wait in front of the weight loads, moved behind them) against the previous order
mkdir -p gpurun_out/r6af
cd /root/repo
export TMPDIR=/tmp
L=$PWD/sige_amd/lib
for rep in 1 2; do
timeout 300 python tools/forward_ab.py --tag weights-behind-index >> gpurun_out/r6af/forward_ab.jsonl 2>> gpurun_out/r6af/err.log
SIGE_HIP_LIB=$L/libsige_hip_epvfirst.so timeout 300 python tools/forward_ab.py --tag previous-order >> gpurun_out/r6af/forward_ab.jsonl 2>> gpurun_out/r6af/err.log
done
timeout 300 python tools/forward_ab.py --tag weights-behind-index --dtype f16 >> gpurun_out/r6af/forward_ab.jsonl 2>> gpurun_out/r6af/err.log
SIGE_HIP_LIB=$L/libsige_hip_epvfirst.so timeout 300 python tools/forward_ab.py --tag previous-order --dtype f16 >> gpurun_out/r6af/forward_ab.jsonl 2>> gpurun_out/r6af/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r6af/forward_ab.jsonl"):
    d = json.loads(l)
    print(d["tag"], d["dtype"], [(r["ratio"], r["forward_ms"]) for r in d["rows"]])
PY
timeout 600 python tools/sd_fused_tokens_ab.py --settings 1,1 --out gpurun_out/r6af/sd_new.json 2>> gpurun_out/r6af/err.log
SIGE_HIP_LIB=$L/libsige_hip_epvfirst.so timeout 600 python tools/sd_fused_tokens_ab.py --settings 1,1 --out gpurun_out/r6af/sd_prev.json 2>> gpurun_out/r6af/err.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6af/pytest.log 2>&1; tail -n 3 gpurun_out/r6af/pytest.log
