#!/bin/bash
# round 4, GPU session 13: SD with the three self-attention projections as one batched GEMM
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4n; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_models_golden.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -m gpu --tb=short -k "sd" > $OUT/pytest_sd.log 2>&1
echo "sd tests rc=$?" >> $OUT/summary.txt
timeout 900 python bench.py --workload sd --steps 20 --warmup 5 > $OUT/bench_sd.json 2> $OUT/bench_sd.err
echo "bench sd rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; tail -4 $OUT/pytest_sd.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4n/bench_sd.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "forward_ms", "dense_forward_ms", "speedup_vs_dense", "parity_max_abs", "parity_ok")})
print(json.dumps(d.get("attention_routing"), indent=1))
PY
tail -3 $OUT/bench_sd.err
