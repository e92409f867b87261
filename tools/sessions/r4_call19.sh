#!/bin/bash
# round 4, GPU session 19: the default bench line at HEAD (as the driver runs it)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/r4_bench.err | tail -1 > $OUT/r4_bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "forward_ms", "launches_per_forward", "parity_ok", "forward_ms_eager_launch_plan")}, d["roofline"]["frac"], d["roofline"]["traffic"])
print(d["batched_edits"]["speedup_forwards_per_s_vs_one_edit"])
print([(r["edit_ratio"], r["forward_ms"], r["block_conv_frac_of_mfma_peak"]) for r in d["sweep"]])
PY
tail -2 $OUT/r4_bench.err
