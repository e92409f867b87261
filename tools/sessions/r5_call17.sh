#!/bin/bash
# round 5, GPU session 17: kernel trace of the stacked forward (E = 8 edits at 1.2 %), 20 hipGraph replays
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r5s_trace" -o fwd -- python $ROOT/tools/profile_forward.py --replays 20 --edits 8 > "$OUT/r5s_trace.log" 2>&1
cd "$ROOT"
T=$(ls "$OUT/r5s_trace"/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$T" ] && python tools/trace_summary.py "$T" --replays 20 --out "$OUT/r5s_kerneltrace_stacked_e8_1p2pct_f32.csv" > "$OUT/r5s_trace_summary_stacked_e8.txt" 2>&1
rm -rf "$OUT/r5s_trace"
tail -n 3 "$OUT/r5s_trace.log"; head -n 14 "$OUT/r5s_trace_summary_stacked_e8.txt" | cut -c1-200
