#!/bin/bash
# round 4, GPU session 10: the round's evidence on the current kernel sources -- whole GPU suite, rocprofv3 traces + PMC passes
# (fp32 and f16 forward, data movement), the bench lines (default, f16, 2-rank gloo, SD, SD 2-rank gloo), SD / GauGAN traces
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x --tb=short > $OUT/r4_pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?" > $OUT/r4_summary.txt
timeout 1500 bash tools/gpu_profile_round.sh r4 > $OUT/r4_profile_round.log 2>&1
echo "profile round rc=$?" >> $OUT/r4_summary.txt
[ -s $OUT/r4_pmc_traffic.json ] && cp $OUT/r4_pmc_traffic.json profiles/pmc_traffic.json
[ -s $OUT/r4_pmc_traffic_f16.json ] && cp $OUT/r4_pmc_traffic_f16.json profiles/pmc_traffic_f16.json
timeout 2400 bash tools/gpu_final_round.sh r4 > $OUT/r4_final_round.log 2>&1
echo "final round rc=$?" >> $OUT/r4_summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r4_smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/r4_summary.txt
cat $OUT/r4_summary.txt; tail -5 $OUT/r4_pytest_gpu.log; tail -15 $OUT/r4_profile_round.log; tail -12 $OUT/r4_final_round.log; cat $OUT/r4_pmc_traffic.txt | head -30; cat $OUT/r4_pmc_data_movement.txt
python - <<'PY'
import json
for f in ("r4_bench", "r4_bench_f16", "r4_bench_2ranks_gloo", "r4_bench_sd", "r4_bench_sd_2ranks_gloo"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, {k: d.get(k) for k in ("value", "forward_ms", "launches_per_forward", "parity_ok", "cache_bytes")}, (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic"))
    except Exception as e:
        print(f, "parse failed", e)
PY
