#!/bin/bash
# round 4, GPU session 9: SD attention forms (DPP row reductions; 16 / 32 / 64 queries per workgroup), SD kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4i; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -q -m gpu --tb=short -k "attention or sd_transformer" > $OUT/pytest_round4.log 2>&1
echo "round4 attention tests rc=$?" >> $OUT/summary.txt
timeout 900 python bench.py --workload sd --steps 20 --warmup 5 > $OUT/bench_sd.json 2> $OUT/bench_sd.err
echo "bench sd rc=$?" >> $OUT/summary.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sd -o sd -- python $ROOT/tools/profile_sd.py --replays 10 > $OUT/trace_sd.log 2>&1
echo "sd trace rc=$?" >> $OUT/summary.txt
cd $ROOT
T=$(ls $OUT/trace_sd/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$T" ] && python tools/trace_summary.py "$T" --replays 10 --out $OUT/r4i_kerneltrace_sd_unet_sparse_15pct.csv > $OUT/trace_summary_sd.txt 2>&1
S=$(ls $OUT/trace_sd/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$S" ] && cp "$S" $OUT/r4i_rocprofv3_kernel_stats_sd.csv
rm -rf $OUT/trace_sd
cat $OUT/summary.txt; tail -12 $OUT/pytest_round4.log
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4i/bench_sd.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "forward_ms", "dense_forward_ms", "speedup_vs_dense", "parity_max_abs", "parity_ok", "hip_kernel_launches_per_forward")})
    print(json.dumps(d.get("attention_routing"), indent=1))
    print({k: v for k, v in (d.get("kernels") or {}).items()})
except Exception as e:
    print("sd bench parse failed", e)
PY
tail -4 $OUT/bench_sd.err; head -45 $OUT/trace_summary_sd.txt
