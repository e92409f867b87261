export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
