#!/bin/bash
# round 6, GPU session 48 (run twice: the second time with MIOpen choosing its convs by measurement in the SD job too): the SD evidence re-taken with the tuned token GEMMs (sige_amd/workloads/gemm_tuning.py; kernel sources
# unchanged): the driver's test command first, then the SD bench lines (one rank, two gloo ranks on one GPU), the same line
# without the table, and the SD kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; TAG=r6
export TMPDIR=/tmp
rm -f $OUT/test_margins.jsonl
timeout 1500 python -m pytest tests -x -q -m gpu --tb=short > $OUT/r6_pytest_gpu.log 2>&1
echo "pytest -x -q -m gpu rc=$?" > $OUT/r6au_summary.txt
cp $OUT/test_margins.jsonl $OUT/r6_test_margins.jsonl 2>/dev/null
timeout 600 python bench.py --workload sd --steps 20 --warmup 5 2> "$OUT/${TAG}_bench_sd.err" | tail -1 > "$OUT/${TAG}_bench_sd.json"
cp bench_detail_sd.json "$OUT/${TAG}_bench_sd_detail.json"
timeout 600 python bench.py --workload sd --gpus 2 --oversubscribe --backend gloo --steps 10 --warmup 3 2> "$OUT/${TAG}_bench_sd_2ranks.err" | grep '^{"metric"' | tail -1 > "$OUT/${TAG}_bench_sd_2ranks_gloo.json"
cp bench_detail_sd.json "$OUT/${TAG}_bench_sd_2ranks_gloo_detail.json"
timeout 600 python bench.py --workload sd --steps 20 --warmup 5 --no-tuned-gemms --cpu-seconds 0 2> "$OUT/r6au_bench_sd_default_gemms.err" | tail -1 > "$OUT/r6au_bench_sd_default_gemms.json"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace_sd" -o sd -- python $ROOT/tools/profile_sd.py --replays 10 > "$OUT/${TAG}_trace_sd.log" 2>&1
cd "$ROOT"
T=$(ls "$OUT/${TAG}_trace_sd"/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$T" ] && python tools/trace_summary.py "$T" --replays 10 --out "$OUT/${TAG}_kerneltrace_sd_sparse.csv" > "$OUT/${TAG}_trace_summary_sd.txt" 2>&1
S=$(ls "$OUT/${TAG}_trace_sd"/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$S" ] && cp "$S" "$OUT/${TAG}_rocprofv3_kernel_stats_sd.csv"
rm -rf "$OUT/${TAG}_trace_sd"
cat $OUT/r6au_summary.txt; tail -n 3 $OUT/r6_pytest_gpu.log
python - <<'PY'
import json
for f in ("r6_bench_sd", "r6_bench_sd_2ranks_gloo", "r6au_bench_sd_default_gemms"):
    try:
        t = open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1]
        d = json.loads(t)
        print(f, len(t), {k: d.get(k) for k in ("value", "forward_ms", "dense_forward_ms", "speedup_vs_dense", "tuned_token_gemms", "parity_ok", "parity_max_abs")})
    except Exception as e:
        print(f, "parse failed", e)
PY
head -n 12 $OUT/r6_trace_summary_sd.txt | cut -c1-200
