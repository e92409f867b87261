#!/bin/bash
# round 6, GPU session 26: attention_tokens with the score tile transposed (form 0) against forms 1 / 2 (tuning build) and the
# __shfl_xor variant of the row maximum; the permlane swap probe first
mkdir -p gpurun_out/r6z
cd /root/repo
export TMPDIR=/tmp
L=$PWD/sige_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe/permlane_swap_probe.hip -o /tmp/pp 2>/dev/null && timeout 60 /tmp/pp | tee gpurun_out/r6z/permlane_swap_probe.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "attention_tokens or sd_" > gpurun_out/r6z/pytest.log 2>&1; tail -n 3 gpurun_out/r6z/pytest.log
for rep in 1 2; do
timeout 300 python tools/attention_tokens_bench.py --tag transposed >> gpurun_out/r6z/attention_tokens.jsonl 2>> gpurun_out/r6z/err.log
SIGE_HIP_LIB=$L/libsige_hip_tuning.so timeout 300 python tools/attention_tokens_bench.py --tag form1 --form 1 >> gpurun_out/r6z/attention_tokens.jsonl 2>> gpurun_out/r6z/err.log
SIGE_HIP_LIB=$L/libsige_hip_attshfl.so timeout 300 python tools/attention_tokens_bench.py --tag transposed-shfl >> gpurun_out/r6z/attention_tokens.jsonl 2>> gpurun_out/r6z/err.log
done
python - <<'PY'
import json
for l in open("gpurun_out/r6z/attention_tokens.jsonl"):
    d = json.loads(l)
    print(d["tag"], [(r["shape"], r["us"], r["tflops"], "%.1e" % r["max_abs_err_vs_f64"]) for r in d["rows"]])
PY
tail -n 3 gpurun_out/r6z/err.log
timeout 600 python bench.py --workload sd --steps 20 --warmup 3 > gpurun_out/r6z/bench_sd.json 2> gpurun_out/r6z/bench_sd.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6z/bench_sd.json").read().strip().splitlines()[-1])
print(d.get("value"), d.get("forward_ms"), d.get("parity_ok"))
PY
