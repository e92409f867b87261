#!/bin/bash
# round 6, GPU session 28: MFMA issue order in the transposed-score attention kernel: scores on one accumulator chain (A), values
# channel-tile-major (B), both (C), against two score chains + key-step-major (the committed order)
mkdir -p gpurun_out/r6ab
cd /root/repo
export TMPDIR=/tmp
L=$PWD/sige_amd/lib
for rep in 1 2; do
timeout 300 python tools/attention_tokens_bench.py --tag base --only self_64,self_32,self_64_dense >> gpurun_out/r6ab/attention_tokens.jsonl 2>> gpurun_out/r6ab/err.log
for v in attA attB attC; do
SIGE_HIP_LIB=$L/libsige_hip_$v.so timeout 300 python tools/attention_tokens_bench.py --tag $v --only self_64,self_32,self_64_dense >> gpurun_out/r6ab/attention_tokens.jsonl 2>> gpurun_out/r6ab/err.log
done; done
python - <<'PY'
import json
for l in open("gpurun_out/r6ab/attention_tokens.jsonl"):
    d = json.loads(l)
    print(d["tag"], [(r["shape"], r["us"], r["tflops"], "%.1e" % r["max_abs_err_vs_f64"]) for r in d["rows"]])
PY
tail -n 3 gpurun_out/r6ab/err.log
