#!/bin/bash
# round 4, GPU session 14: 64 x 32 output blocks (two M tiles per workgroup): the conv parity suites with the form forced on, then
# the forward under thresholds
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4o; mkdir -p $OUT
export TMPDIR=/tmp
SIGE_TEST_TWO_M_TILES=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_channels_last.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -m gpu -x --tb=short > $OUT/pytest_mb2.log 2>&1
echo "conv suites with two M tiles rc=$?" >> $OUT/summary.txt
timeout 900 python tools/plan_policy_bench.py --mb2 1,256,512,1024 --ratios 0.012,0.05,0.15 --out $OUT/plan_mb2.json > $OUT/plan_mb2.log 2>&1
echo "plan bench rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; tail -6 $OUT/pytest_mb2.log
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r4o/plan_mb2.json"))
    for c, rows in d["cases"].items():
        print(c)
        for k, v in rows.items():
            print("   %-40s" % k, v.get("forward_ms") or v.get("ms_per_launch_set"), v.get("launches", ""), v["max_abs_vs_first_row"])
except Exception as e:
    print("parse failed", e)
PY
tail -5 $OUT/plan_mb2.log
