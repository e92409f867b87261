#!/bin/bash
# round 4, GPU session 2: launch plans (fixed tests), hardening, fp16-stored caches; f16 bench line; 2-rank gloo runs
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4b; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -q -m gpu --tb=short > $OUT/pytest_round4.log 2>&1
echo "round4 tests rc=$?" >> $OUT/summary.txt
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_round4.py --tb=short > $OUT/pytest_all.log 2>&1
echo "all gpu tests rc=$?" >> $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/summary.txt
timeout 900 python bench.py --dtype f16 > $OUT/bench_f16.json 2> $OUT/bench_f16.err
echo "bench f16 rc=$?" >> $OUT/summary.txt
timeout 600 python bench.py --gpus 2 --oversubscribe --backend gloo --distribute broadcast --dtype f16 --steps 10 --cpu-seconds 0 --no-extras > $OUT/bench_2ranks_gloo_f16cache.json 2> $OUT/bench_2ranks.err
echo "bench 2 ranks gloo f16 cache rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; tail -30 $OUT/pytest_round4.log; tail -8 $OUT/pytest_all.log
python - <<'PY'
import json
for f in ("bench.json", "bench_f16.json", "bench_2ranks_gloo_f16cache.json"):
    try:
        d = json.loads(open("gpurun_out/r4b/" + f).read().strip().splitlines()[-1])
        print(f, {k: d.get(k) for k in ("value", "ms_per_step", "cache_bytes", "cache_dtype", "parity_ok", "parity_max_abs")})
        print(" f16:", {k: (d.get("f16_compute") or {}).get(k) for k in ("forward_ms", "parity_ok")})
        print(" mg:", d.get("multi_gpu"))
        print(" plan:", (d.get("dynamic") or {}).get("mask_change_plan"))
        print(" batched:", json.dumps(d.get("batched_edits"), indent=1)[:3000])
    except Exception as e:
        print(f, "parse failed", e)
PY
tail -5 $OUT/bench.err; tail -5 $OUT/bench_f16.err; tail -5 $OUT/bench_2ranks.err
