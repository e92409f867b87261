#!/bin/bash
# round 6, GPU session 23: attention_tokens with the (batch, head) pairs dealt to the XCDs vs the plain order, at SD's shapes + SD forward
mkdir -p gpurun_out/r6w
cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
timeout 300 python tools/attention_tokens_bench.py --tag xcd-pairs >> gpurun_out/r6w/attention_tokens.jsonl 2>> gpurun_out/r6w/err.log
SIGE_HIP_LIB=$PWD/sige_amd/lib/libsige_hip_attplain.so timeout 300 python tools/attention_tokens_bench.py --tag plain-order >> gpurun_out/r6w/attention_tokens.jsonl 2>> gpurun_out/r6w/err.log
done
cat gpurun_out/r6w/attention_tokens.jsonl; tail -n 3 gpurun_out/r6w/err.log
timeout 600 python bench.py --workload sd --steps 20 --warmup 3 > gpurun_out/r6w/bench_sd_xcd.json 2> gpurun_out/r6w/bench_sd_xcd.err
SIGE_HIP_LIB=$PWD/sige_amd/lib/libsige_hip_attplain.so timeout 600 python bench.py --workload sd --steps 20 --warmup 3 > gpurun_out/r6w/bench_sd_plain.json 2> gpurun_out/r6w/bench_sd_plain.err
python - <<'PY'
import json
for f in ("xcd", "plain"):
    try:
        d = json.loads(open("gpurun_out/r6w/bench_sd_%s.json" % f).read().strip().splitlines()[-1])
        print(f, d.get("value"), d.get("forward_ms"), d.get("parity_ok"))
    except Exception as e:
        print(f, "failed", e)
PY
timeout 600 python -m pytest tests -x -q -m gpu -k "attention or sd_" > gpurun_out/r6w/pytest.log 2>&1; tail -n 2 gpurun_out/r6w/pytest.log
