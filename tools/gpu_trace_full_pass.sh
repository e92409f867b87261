#!/bin/bash
# rocprofv3 kernel trace of the cache-producing FULL pass on the library's kernels (f16x3, and exact fp32 if $2 = f32 too)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); TAG=${1:-r3}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for DT in f16x3 $2; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_full_$DT -o full -- python $ROOT/tools/profile_forward.py --mode full --dtype $DT --replays 20 > $OUT/trace_full_$DT.log 2>&1
  T=$(ls $OUT/trace_full_$DT/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$T" ] && python $ROOT/tools/trace_summary.py "$T" --replays 20 --out $OUT/kerneltrace_full_pass_$DT.csv --top 45 > $OUT/trace_summary_full_$DT.txt 2>&1
  rm -rf $OUT/trace_full_$DT
done
head -50 $OUT/trace_summary_full_f16x3.txt
