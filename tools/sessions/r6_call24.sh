#!/bin/bash
# round 6, GPU session 24: attention_tokens -- branch-free buffer loads one key block ahead + (batch, head) pairs dealt to the XCDs,
# against the round-5 kernel (attold), the plain workgroup order (attplain) and no prefetch (attnopf); then the SD forward
mkdir -p gpurun_out/r6x
cd /root/repo
export TMPDIR=/tmp
L=$PWD/sige_amd/lib
for rep in 1 2; do
timeout 300 python tools/attention_tokens_bench.py --tag new >> gpurun_out/r6x/attention_tokens.jsonl 2>> gpurun_out/r6x/err.log
for v in attold attplain attnopf; do
SIGE_HIP_LIB=$L/libsige_hip_$v.so timeout 300 python tools/attention_tokens_bench.py --tag $v >> gpurun_out/r6x/attention_tokens.jsonl 2>> gpurun_out/r6x/err.log
done; done
python - <<'PY'
import json
for l in open("gpurun_out/r6x/attention_tokens.jsonl"):
    d = json.loads(l)
    print(d["tag"], [(r["shape"], r["us"], r["tflops"], "%.1e" % r["max_abs_err_vs_f64"]) for r in d["rows"]])
PY
tail -n 3 gpurun_out/r6x/err.log
timeout 600 python -m pytest tests -x -q -m gpu -k "attention or sd_" > gpurun_out/r6x/pytest.log 2>&1; tail -n 2 gpurun_out/r6x/pytest.log
timeout 600 python bench.py --workload sd --steps 20 --warmup 3 > gpurun_out/r6x/bench_sd_new.json 2> gpurun_out/r6x/bench_sd_new.err
SIGE_HIP_LIB=$L/libsige_hip_attold.so timeout 600 python bench.py --workload sd --steps 20 --warmup 3 > gpurun_out/r6x/bench_sd_old.json 2> gpurun_out/r6x/bench_sd_old.err
python - <<'PY'
import json
for f in ("new", "old"):
    try:
        d = json.loads(open("gpurun_out/r6x/bench_sd_%s.json" % f).read().strip().splitlines()[-1])
        print(f, d.get("value"), d.get("forward_ms"), d.get("parity_ok"))
    except Exception as e:
        print(f, "failed", e)
PY
