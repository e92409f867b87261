#!/bin/bash
# round 4, GPU session 12: the default and the f16 bench lines again, now that the traffic tables of these kernel sources are in profiles/
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --dump-calls $OUT/r4_calls_1p2pct_f32.json 2> $OUT/r4_bench.err | tail -1 > $OUT/r4_bench.json
timeout 600 python bench.py --dtype f16 --no-extras --cpu-seconds 1 2> $OUT/r4_bench_f16.err | tail -1 > $OUT/r4_bench_f16.json
python - <<'PY'
import json
for f in ("r4_bench", "r4_bench_f16"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, {k: d.get(k) for k in ("value", "forward_ms", "launches_per_forward", "parity_ok")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"].get("traffic_source", "")[:60])
PY
