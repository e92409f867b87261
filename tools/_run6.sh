export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_channels_last.py -m gpu -q -x > gpurun_out/pytest_cl.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_cl.log
tail -25 gpurun_out/pytest_cl.log
