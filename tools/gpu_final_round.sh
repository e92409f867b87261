#!/bin/bash
# Second GPU-box session of a round (after tools/gpu_profile_round.sh <tag> and `cp gpurun_out/<tag>_pmc_traffic.json
# profiles/pmc_traffic.json`): the profiler rows behind bench.py's data_movement table, the default bench line, the f16
# bench line, the multi-rank code path on one GPU (gloo, oversubscribed) and the SD U-Net workload.
set -u
TAG=${1:-r2}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/${TAG}_dm" -o dm -- python $ROOT/tools/profile_data_movement.py > "$OUT/${TAG}_dm.log" 2>&1
cd "$ROOT"
T=$(ls "$OUT/${TAG}_dm"/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$T" ] && python tools/trace_summary.py "$T" --replays 1 --by-grid --gap-ms 100 --top 0 --out "$OUT/${TAG}_kerneltrace_data_movement.csv" > /dev/null 2>&1
rm -rf "$OUT/${TAG}_dm"
# counter bytes of the data-movement rows (two passes, one counter each); bench.py prints them when profiles/pmc_data_movement.json
# carries the hash of these sources -- so this comes BEFORE the bench lines
mkdir -p "$OUT/${TAG}_dmpmc"
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_dmpmc/fetch" -o pmc -- python $ROOT/tools/profile_data_movement.py --pmc-manifest "$OUT/${TAG}_dm_manifest.json" > "$OUT/${TAG}_dm_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_dmpmc/write" -o pmc -- python $ROOT/tools/profile_data_movement.py --pmc-manifest "$OUT/${TAG}_dm_manifest.json" > "$OUT/${TAG}_dm_write.log" 2>&1
cd "$ROOT"
python tools/pmc_data_movement.py "$(ls $OUT/${TAG}_dmpmc/fetch/*counter_collection.csv | head -1)" "$(ls $OUT/${TAG}_dmpmc/write/*counter_collection.csv | head -1)" "$OUT/${TAG}_dm_manifest.json" "$OUT/${TAG}_pmc_data_movement.json" > "$OUT/${TAG}_pmc_data_movement.txt" 2>&1
[ -s "$OUT/${TAG}_pmc_data_movement.json" ] && cp "$OUT/${TAG}_pmc_data_movement.json" profiles/pmc_data_movement.json
find "$OUT/${TAG}_dmpmc" -type f -delete; find "$OUT/${TAG}_dmpmc" -type d -empty -delete
# (since round 5 the LAST stdout line is the <= 4 KB contract line and the whole result is bench_detail*.json: both are kept)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> "$OUT/${TAG}_bench.err" | tail -1 > "$OUT/${TAG}_bench.json"
cp bench_detail.json "$OUT/${TAG}_bench_detail.json"
timeout 600 python bench.py --dtype f16 --no-extras --cpu-seconds 1 2> "$OUT/${TAG}_bench_f16.err" | tail -1 > "$OUT/${TAG}_bench_f16.json"
cp bench_detail.json "$OUT/${TAG}_bench_f16_detail.json"
timeout 600 python bench.py --gpus 2 --oversubscribe --backend gloo --steps 20 --warmup 5 --no-extras --cpu-seconds 1 2> "$OUT/${TAG}_bench_2ranks.err" | grep '^{"metric"' | tail -1 > "$OUT/${TAG}_bench_2ranks_gloo.json"
cp bench_detail.json "$OUT/${TAG}_bench_2ranks_gloo_detail.json"
timeout 600 python bench.py --workload sd --steps 20 --warmup 5 2> "$OUT/${TAG}_bench_sd.err" | tail -1 > "$OUT/${TAG}_bench_sd.json"
cp bench_detail_sd.json "$OUT/${TAG}_bench_sd_detail.json"
timeout 600 python bench.py --workload sd --gpus 2 --oversubscribe --backend gloo --steps 10 --warmup 3 2> "$OUT/${TAG}_bench_sd_2ranks.err" | grep '^{"metric"' | tail -1 > "$OUT/${TAG}_bench_sd_2ranks_gloo.json"
cp bench_detail_sd.json "$OUT/${TAG}_bench_sd_2ranks_gloo_detail.json"
# kernel traces of the two other workloads (10 hipGraph replays each)
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace_sd" -o sd -- python $ROOT/tools/profile_sd.py --replays 10 > "$OUT/${TAG}_trace_sd.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace_gaugan" -o gg -- python $ROOT/tools/profile_gaugan.py --replays 10 > "$OUT/${TAG}_trace_gaugan.log" 2>&1
cd "$ROOT"
for W in sd gaugan; do
  T=$(ls "$OUT/${TAG}_trace_$W"/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$T" ] && python tools/trace_summary.py "$T" --replays 10 --out "$OUT/${TAG}_kerneltrace_${W}_sparse.csv" > "$OUT/${TAG}_trace_summary_$W.txt" 2>&1
  S=$(ls "$OUT/${TAG}_trace_$W"/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$S" ] && cp "$S" "$OUT/${TAG}_rocprofv3_kernel_stats_$W.csv"
  rm -rf "$OUT/${TAG}_trace_$W"
done
wc -c "$OUT"/${TAG}_bench*.json
for e in "$OUT"/${TAG}_bench*.err; do tail -n 2 "$e"; done
