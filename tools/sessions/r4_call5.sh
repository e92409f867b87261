#!/bin/bash
# round 4, GPU session 5: round-4 tests (stacked mask pipeline + plan), default bench line with E = 16 and more routings
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -q -m gpu --tb=short > $OUT/pytest_round4.log 2>&1
echo "round4 tests rc=$?" >> $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; tail -25 $OUT/pytest_round4.log
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4e/bench.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "parity_max_abs")})
    b = d.get("batched_edits") or {}
    for r in b.get("rows", []):
        print({k: r.get(k) for k in ("edits", "dense_route", "ms_per_edit", "forwards_per_s", "launches", "max_abs_vs_single_edit_forward", "block_conv_frac_of_mfma_peak", "dense_conv_TFLOPs")})
    print(b.get("speedup_forwards_per_s_vs_one_edit"), b.get("error"))
    print((d.get("gaugan") or {}).get("per_edit_latency_ms"))
    print([(r["edit_ratio"], r.get("fp16_flop_fraction")) for r in (d.get("f16_compute") or {}).get("sweep", [])])
except Exception as e:
    print("bench parse failed", e)
PY
tail -3 $OUT/bench.err
