export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
timeout 600 python tools/conv_probe.py > gpurun_out/conv_probe2.jsonl 2> gpurun_out/conv_probe.err; echo "probe rc=$?"
timeout 900 python tools/conv_bench.py --ratio 0.012 > gpurun_out/conv_bench_r1d.jsonl 2> gpurun_out/conv_bench.err; echo "convbench rc=$?"
timeout 600 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/bench.json
