#!/bin/bash
# round 4, GPU session 7: SD attention core / token linears on the library (tests, golden fixtures, the SD bench line)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4g; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -q -m gpu --tb=short -k "attention or sd_transformer" > $OUT/pytest_round4.log 2>&1
echo "round4 attention tests rc=$?" >> $OUT/summary.txt
timeout 900 python -m pytest tests/test_models_golden.py tests/test_gpu_round3.py -q -m gpu --tb=short -k "sd" > $OUT/pytest_sd.log 2>&1
echo "sd golden tests rc=$?" >> $OUT/summary.txt
timeout 900 python bench.py --workload sd --steps 20 --warmup 5 > $OUT/bench_sd.json 2> $OUT/bench_sd.err
echo "bench sd rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; tail -25 $OUT/pytest_round4.log; tail -6 $OUT/pytest_sd.log
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4g/bench_sd.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "forward_ms", "dense_forward_ms", "speedup_vs_dense", "parity_max_abs", "parity_ok", "hip_kernel_launches_per_forward")})
    print(json.dumps(d.get("attention_routing"), indent=1))
    print(d.get("roofline"))
    print({k: v for k, v in (d.get("kernels") or {}).items()})
except Exception as e:
    print("bench parse failed", e)
PY
tail -4 $OUT/bench_sd.err
