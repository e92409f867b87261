export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_channels_last.py -m gpu -q -x -k "attention or small_cout or ddpm" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --sweep '' --no-roofline > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
