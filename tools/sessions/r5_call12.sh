#!/bin/bash
# round 5, GPU session 12: tile conv v3 off by default -- the whole GPU suite again, then the bench lines (the traffic tables of session 11
# stay valid: no kernel source changed)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short > $OUT/r5_pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?" > $OUT/r5_summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r5_smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/r5_summary.txt
timeout 2400 bash tools/gpu_final_round.sh r5 > $OUT/r5_final_round.log 2>&1
echo "final round rc=$?" >> $OUT/r5_summary.txt
cat $OUT/r5_summary.txt; tail -n 4 $OUT/r5_pytest_gpu.log
python - <<'PY'
import json
for f in ("r5_bench", "r5_bench_f16", "r5_bench_2ranks_gloo", "r5_bench_sd", "r5_bench_sd_2ranks_gloo"):
    try:
        t = open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1]
        d = json.loads(t)
        print(f, len(t), {k: d.get(k) for k in ("value", "forward_ms", "launches_per_forward", "parity_ok")}, (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic"), (d.get("roofline_hbm") or {}).get("bytes"))
    except Exception as e:
        print(f, "parse failed", e)
print(json.load(open("gpurun_out/r5_bench_detail.json")).get("tile_conv3_opt_in"))
PY
