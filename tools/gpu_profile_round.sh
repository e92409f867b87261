#!/bin/bash
# One GPU-box session producing the round's measured evidence (run through gpurun from the repo root):
#   tools/gpu_profile_round.sh r2c
# Writes under gpurun_out/<tag>_*: the bench line, rocprofv3 --kernel-trace --stats of 50 hipGraph replays of the sparse
# forward (fp32 and f16 compute), and three PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters; each in its own run, kernel-trace
# only) of eager forwards with the conv-family manifest tools/pmc_traffic.py needs.
set -u
TAG=${1:-r2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PF="python $ROOT/tools/profile_forward.py"
for DT in f32 f16; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace_$DT" -o fwd -- $PF --replays 50 --dtype $DT > "$OUT/${TAG}_trace_$DT.log" 2>&1
done
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_fetch" -o pmc -- $PF --mode eager --replays 4 --manifest "$OUT/${TAG}_manifest.json" > "$OUT/${TAG}_pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_write" -o pmc -- $PF --mode eager --replays 4 > "$OUT/${TAG}_pmc_write.log" 2>&1
# the f16 forward runs other kernels (ConvGeoH tile convs, the dense-layer kernel): its own traffic pass
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_fetch_f16" -o pmc -- $PF --mode eager --replays 4 --dtype f16 --manifest "$OUT/${TAG}_manifest_f16.json" > "$OUT/${TAG}_pmc_fetch_f16.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_write_f16" -o pmc -- $PF --mode eager --replays 4 --dtype f16 > "$OUT/${TAG}_pmc_write_f16.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_sq" -o pmc -- $PF --mode eager --replays 4 > "$OUT/${TAG}_pmc_sq.log" 2>&1
cd "$ROOT"
# reduce on the box (gpurun copies at most 64 MiB back): per-replay kernel summaries, per-kernel counter means, the traffic
# table; the raw traces / counter tables are dropped
for DT in f32 f16; do
  T=$(ls "$OUT/${TAG}_trace_$DT"/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$T" ] && python tools/trace_summary.py "$T" --replays 50 --out "$OUT/${TAG}_kerneltrace_sparse_fwd_1p2pct_$DT.csv" > "$OUT/${TAG}_trace_summary_$DT.txt" 2>&1
  [ -n "$T" ] && python tools/trace_summary.py "$T" --replays 50 --by-grid --top 0 --out "$OUT/${TAG}_kerneltrace_by_grid_sparse_fwd_1p2pct_$DT.csv" --sequence "$OUT/${TAG}_kernel_sequence_sparse_fwd_1p2pct_$DT.csv" > /dev/null 2>&1
  S=$(ls "$OUT/${TAG}_trace_$DT"/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$S" ] && cp "$S" "$OUT/${TAG}_rocprofv3_kernel_stats_profile_forward_$DT.csv"
  rm -rf "$OUT/${TAG}_trace_$DT"
done
F=$(ls "$OUT/${TAG}_pmc_fetch"/*counter_collection.csv | head -1); W=$(ls "$OUT/${TAG}_pmc_write"/*counter_collection.csv | head -1)
Q=$(ls "$OUT/${TAG}_pmc_sq"/*counter_collection.csv | head -1)
python tools/pmc_traffic.py "$F" "$W" "$OUT/${TAG}_manifest.json" "$OUT/${TAG}_pmc_traffic.json" > "$OUT/${TAG}_pmc_traffic.txt" 2>&1
F16=$(ls "$OUT/${TAG}_pmc_fetch_f16"/*counter_collection.csv | head -1); W16=$(ls "$OUT/${TAG}_pmc_write_f16"/*counter_collection.csv | head -1)
python tools/pmc_traffic.py "$F16" "$W16" "$OUT/${TAG}_manifest_f16.json" "$OUT/${TAG}_pmc_traffic_f16.json" > "$OUT/${TAG}_pmc_traffic_f16.txt" 2>&1
grep -E "Counter_Name|sige::conv|sige::attn" "$F16" | gzip > "$OUT/${TAG}_pmc_fetch_rows_f16.csv.gz"
grep -E "Counter_Name|sige::conv|sige::attn" "$W16" | gzip > "$OUT/${TAG}_pmc_write_rows_f16.csv.gz"
python tools/pmc_summary.py "$F16" "$OUT/${TAG}_pmc_fetch_size_per_kernel_f16.csv" sige::
python tools/pmc_summary.py "$W16" "$OUT/${TAG}_pmc_write_size_per_kernel_f16.csv" sige::
rm -rf "$OUT/${TAG}_pmc_fetch_f16" "$OUT/${TAG}_pmc_write_f16"
# (the raw counter rows of the conv / attention kernels, so that the traffic tables can be re-derived without the GPU: tests/test_profiles.py)
grep -E "Counter_Name|sige::conv|sige::attn" "$F" | gzip > "$OUT/${TAG}_pmc_fetch_rows.csv.gz"
grep -E "Counter_Name|sige::conv|sige::attn" "$W" | gzip > "$OUT/${TAG}_pmc_write_rows.csv.gz"
python tools/pmc_summary.py "$F" "$OUT/${TAG}_pmc_fetch_size_per_kernel.csv" sige::
python tools/pmc_summary.py "$W" "$OUT/${TAG}_pmc_write_size_per_kernel.csv" sige::
python tools/pmc_summary.py "$Q" "$OUT/${TAG}_pmc_sq_counters_per_kernel.csv" sige::
rm -rf "$OUT/${TAG}_pmc_fetch" "$OUT/${TAG}_pmc_write" "$OUT/${TAG}_pmc_sq"
du -sh "$OUT"/${TAG}_* 2>/dev/null | tail -30
