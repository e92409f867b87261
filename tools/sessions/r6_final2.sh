#!/bin/bash
# round 6, final GPU session repeated on the sources with the late epilogue set-up (call 40): the round's evidence on the FINAL kernel sources -- the driver's own test command, smoke, rocprofv3
# traces + PMC passes, then (with the fresh traffic tables in place) the bench lines and the traces of the other workloads
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
rm -f $OUT/test_margins.jsonl
timeout 1500 python -m pytest tests -x -q -m gpu --tb=short > $OUT/r6_pytest_gpu.log 2>&1
echo "pytest -x -q -m gpu rc=$?" > $OUT/r6_summary.txt
cp $OUT/test_margins.jsonl $OUT/r6_test_margins.jsonl 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r6_smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/r6_summary.txt
timeout 1500 bash tools/gpu_profile_round.sh r6 > $OUT/r6_profile_round.log 2>&1
echo "profile round rc=$?" >> $OUT/r6_summary.txt
[ -s $OUT/r6_pmc_traffic.json ] && cp $OUT/r6_pmc_traffic.json profiles/pmc_traffic.json
[ -s $OUT/r6_pmc_traffic_f16.json ] && cp $OUT/r6_pmc_traffic_f16.json profiles/pmc_traffic_f16.json
timeout 2400 bash tools/gpu_final_round.sh r6 > $OUT/r6_final_round.log 2>&1
echo "final round rc=$?" >> $OUT/r6_summary.txt
cp profiles/pmc_data_movement.json $OUT/r6_pmc_data_movement_table.json 2>/dev/null
cat $OUT/r6_summary.txt; tail -n 4 $OUT/r6_pytest_gpu.log; tail -n 12 $OUT/r6_final_round.log
python - <<'PY'
import json
for f in ("r6_bench", "r6_bench_f16", "r6_bench_2ranks_gloo", "r6_bench_sd", "r6_bench_sd_2ranks_gloo"):
    try:
        t = open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1]
        d = json.loads(t)
        print(f, len(t), {k: d.get(k) for k in ("value", "forward_ms", "launches_per_forward", "parity_ok")}, (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic"), (d.get("roofline_hbm") or {}).get("bytes"), (d.get("cpu_baseline") or {}).get("model"))
    except Exception as e:
        print(f, "parse failed", e)
PY
