set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/bench.json
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o fwd -- python $R/tools/profile_forward.py --mode sparse --replays 50 > $R/gpurun_out/prof.log 2>&1
f=$(find /tmp/prof -name '*kernel_trace.csv' | head -1)
python $R/tools/trace_summary.py $f --replays 50 --out $R/gpurun_out/r1b_sparse_trace.csv --top 60
head -50 $R/gpurun_out/r1b_sparse_trace.csv
