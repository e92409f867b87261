export TMPDIR=/tmp
mkdir -p gpurun_out
./tools/probe/mfma_probe > gpurun_out/mfma_probe.txt 2>&1
cat gpurun_out/mfma_probe.txt
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
