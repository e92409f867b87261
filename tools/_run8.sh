export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o fwd -- python $R/tools/profile_forward.py --mode sparse --replays 50 > $R/gpurun_out/prof.log 2>&1
f=$(find /tmp/prof -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_summary.py $f --replays 50 --out $R/gpurun_out/r1i_sparse_trace.csv --top 70
