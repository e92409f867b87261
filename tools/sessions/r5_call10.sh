#!/bin/bash
# round 5, GPU session 10: the tile conv v3 INSIDE the forward, launch by launch (kernel traces at 15 % edit, routed from 150 blocks / never)
mkdir -p gpurun_out/r5j
ROOT=/root/repo; OUT=$ROOT/gpurun_out/r5j
export TMPDIR=/tmp
cd /tmp
for TH in 150 100000000; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_$TH" -o fwd -- python $ROOT/tools/profile_forward.py --replays 30 --ratio 0.15 --tile3-min-blocks $TH > "$OUT/trace_$TH.log" 2>&1
  T=$(ls "$OUT/trace_$TH"/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$T" ] && python $ROOT/tools/trace_summary.py "$T" --replays 30 --by-grid --top 0 --out "$OUT/by_grid_$TH.csv" --sequence "$OUT/sequence_15pct_tile3_from_$TH.csv" > /dev/null 2>&1
  rm -rf "$OUT/trace_$TH"
done
ls -la $OUT
