export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
python tools/conv_floor.py > gpurun_out/conv_floor2.jsonl 2>/dev/null
timeout 900 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
