export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -5 gpurun_out/bench.err
tail -c 2500 gpurun_out/bench.json
