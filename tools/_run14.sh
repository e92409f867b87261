export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "conv or ddpm or golden" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --sweep '' --no-roofline > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
