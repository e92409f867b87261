#!/bin/bash
# round 6, GPU session 14: the two 2-rank gloo bench lines again with the hang timeout at 60 s (gloo's scatter + all-gather between two
# ranks on one GPU takes 14 s: slow -> dropped by the watchdog, not hung)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
TAG=r6
timeout 900 python bench.py --gpus 2 --oversubscribe --backend gloo --steps 20 --warmup 5 --no-extras --cpu-seconds 1 2> "$OUT/${TAG}_bench_2ranks.err" | grep '^{"metric"' | tail -1 > "$OUT/${TAG}_bench_2ranks_gloo.json"
cp bench_detail.json "$OUT/${TAG}_bench_2ranks_gloo_detail.json"
timeout 900 python bench.py --workload sd --gpus 2 --oversubscribe --backend gloo --steps 10 --warmup 3 2> "$OUT/${TAG}_bench_sd_2ranks.err" | grep '^{"metric"' | tail -1 > "$OUT/${TAG}_bench_sd_2ranks_gloo.json"
cp bench_detail_sd.json "$OUT/${TAG}_bench_sd_2ranks_gloo_detail.json"
wc -c $OUT/${TAG}_bench_2ranks_gloo.json $OUT/${TAG}_bench_sd_2ranks_gloo.json
python - <<'PY'
import json
for f in ("r6_bench_2ranks_gloo_detail", "r6_bench_sd_2ranks_gloo_detail"):
    d = json.load(open("gpurun_out/%s.json" % f))
    m = d.get("multi_gpu") or {}
    print(f, d.get("value"), {k: m.get(k) for k in ("method_chosen", "methods_ms", "errors", "fallback", "efficiency", "poisoned")})
PY
tail -n 3 $OUT/${TAG}_bench_sd_2ranks.err
