#!/bin/bash
# round 4, GPU session 1: launch plans + hardening tests, the whole GPU suite, the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round4.py -q -m gpu -x --tb=long > $OUT/pytest_round4.log 2>&1
echo "round4 tests rc=$?" >> $OUT/summary.txt
timeout 600 python -m pytest tests -q -m gpu --deselect tests/test_gpu_round4.py > $OUT/pytest_all.log 2>&1
echo "all gpu tests rc=$?" >> $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; tail -40 $OUT/pytest_round4.log; tail -5 $OUT/pytest_all.log
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4a/bench.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "forward_ms_eager", "parity_max_abs", "roofline")})
    print(json.dumps(d.get("dynamic", {}).get("mask_change_plan"), indent=1))
    print(d.get("dynamic", {}).get("mask_change_capture_first"))
except Exception as e:
    print("bench parse failed", e)
PY
tail -5 $OUT/bench.err
