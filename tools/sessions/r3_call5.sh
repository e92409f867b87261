#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_channels_last.py tests/test_gpu_round2.py -q -m gpu -k "routing or channels_last_equals_nchw or benchmarked_forward or f16_policy or f16x3" > $OUT/pytest_sel.log 2>&1
echo "selected tests rc=$?" >> $OUT/summary.txt
cd /tmp
for DT in f16x3 f32; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_full_$DT -o full -- python $ROOT/tools/profile_forward.py --mode full --dtype $DT --replays 20 > $OUT/trace_full_$DT.log 2>&1
  T=$(ls $OUT/trace_full_$DT/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$T" ] && python $ROOT/tools/trace_summary.py "$T" --replays 20 --out $OUT/kerneltrace_full_pass_$DT.csv --top 45 > $OUT/trace_summary_full_$DT.txt 2>&1
  rm -rf $OUT/trace_full_$DT
done
cd $ROOT
cat $OUT/summary.txt; tail -4 $OUT/pytest_sel.log; head -60 $OUT/trace_summary_full_f16x3.txt
