cd "$GRAFT_REPO_ROOT"; ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3q; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for KS in ${KSLIST:-0 1}; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t$KS -o fwd -- python $ROOT/tools/profile_forward.py --replays 50 --dtype f32 --ksplit $KS > $OUT/t$KS.log 2>&1
  T=$(ls $OUT/t$KS/*kernel_trace.csv | head -1)
  python $ROOT/tools/trace_summary.py "$T" --replays 50 --by-grid --top 0 --out $OUT/bygrid_ks$KS.csv --sequence $OUT/seq_ks$KS.csv > /dev/null 2>&1
  rm -rf $OUT/t$KS
done
ls $OUT
