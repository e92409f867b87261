export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
timeout 900 python bench.py --layout nchw --cpu-seconds 0 > gpurun_out/bench_nchw.json 2> gpurun_out/bench_nchw.err; echo "bench nchw rc=$?"
bash tools/_run10.sh > gpurun_out/run10.log 2>&1; tail -3 gpurun_out/run10.log
