#!/bin/bash
# round 6, GPU session 7: loads past the last channel chunk go through empty windows (conv_mfma / conv_wide / conv_tile3): A/B against
# the previous build on the whole forward, then the whole GPU suite
mkdir -p gpurun_out/r6g
cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
  timeout 300 python tools/forward_ab.py --tag empty-windows >> gpurun_out/r6g/forward_ab.jsonl 2>> gpurun_out/r6g/forward_ab.err
  SIGE_HIP_LIB=$PWD/sige_amd/lib/libsige_hip_prev.so timeout 300 python tools/forward_ab.py --tag previous >> gpurun_out/r6g/forward_ab.jsonl 2>> gpurun_out/r6g/forward_ab.err
done
timeout 300 python tools/forward_ab.py --tag empty-windows --dtype f16 >> gpurun_out/r6g/forward_ab.jsonl 2>> gpurun_out/r6g/forward_ab.err
SIGE_HIP_LIB=$PWD/sige_amd/lib/libsige_hip_prev.so timeout 300 python tools/forward_ab.py --tag previous --dtype f16 >> gpurun_out/r6g/forward_ab.jsonl 2>> gpurun_out/r6g/forward_ab.err
cat gpurun_out/r6g/forward_ab.jsonl
tail -n 3 gpurun_out/r6g/forward_ab.err
rm -f gpurun_out/test_margins.jsonl
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6g/pytest_gpu_x.log 2>&1
echo "pytest -x rc $?"; tail -n 5 gpurun_out/r6g/pytest_gpu_x.log
cp gpurun_out/test_margins.jsonl gpurun_out/r6g/test_margins.jsonl 2>/dev/null
