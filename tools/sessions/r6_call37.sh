#!/bin/bash
# round 6, GPU session 37: regular index lists (dense layers compute their tile origins instead of reading the all-tiles list):
# the new test, the whole suite, the forward with / without
mkdir -p gpurun_out/r6al
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q -k regular_index > gpurun_out/r6al/pytest_new.log 2>&1; tail -n 3 gpurun_out/r6al/pytest_new.log
for rep in 1 2; do
timeout 300 python tools/forward_ab.py --tag index-grid >> gpurun_out/r6al/forward_ab.jsonl 2>> gpurun_out/r6al/err.log
timeout 300 python tools/forward_ab.py --tag lists-read --no-index-grid >> gpurun_out/r6al/forward_ab.jsonl 2>> gpurun_out/r6al/err.log
done
timeout 300 python tools/forward_ab.py --tag index-grid --dtype f16 >> gpurun_out/r6al/forward_ab.jsonl 2>> gpurun_out/r6al/err.log
timeout 300 python tools/forward_ab.py --tag lists-read --no-index-grid --dtype f16 >> gpurun_out/r6al/forward_ab.jsonl 2>> gpurun_out/r6al/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r6al/forward_ab.jsonl"):
    d = json.loads(l)
    print(d["tag"], d["dtype"], [(r["ratio"], r["forward_ms"]) for r in d["rows"]])
PY
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6al/pytest.log 2>&1; tail -n 3 gpurun_out/r6al/pytest.log
tail -n 2 gpurun_out/r6al/err.log
