#!/bin/bash
# round 6, GPU session 25: conv_in (weights staged with all loads in flight, 32-bit index arithmetic), gn_partial_nhwc and conv_out's
# weight loads branch-free -- tests of the three kernels, forward A/B against the previous build, head/tail bench
mkdir -p gpurun_out/r6y
cd /root/repo
export TMPDIR=/tmp
L=$PWD/sige_amd/lib
timeout 900 python -m pytest tests -x -q -m gpu -k "conv_in or small_cin or small_cout or conv_out or group_norm or gn_ or norm or benchmarked_forward or smoke" > gpurun_out/r6y/pytest.log 2>&1; tail -n 3 gpurun_out/r6y/pytest.log
for rep in 1 2; do
timeout 300 python tools/forward_ab.py --tag branch-free >> gpurun_out/r6y/forward_ab.jsonl 2>> gpurun_out/r6y/err.log
SIGE_HIP_LIB=$L/libsige_hip_prev.so timeout 300 python tools/forward_ab.py --tag previous >> gpurun_out/r6y/forward_ab.jsonl 2>> gpurun_out/r6y/err.log
done
timeout 300 python tools/forward_ab.py --tag branch-free --dtype f16 >> gpurun_out/r6y/forward_ab.jsonl 2>> gpurun_out/r6y/err.log
SIGE_HIP_LIB=$L/libsige_hip_prev.so timeout 300 python tools/forward_ab.py --tag previous --dtype f16 >> gpurun_out/r6y/forward_ab.jsonl 2>> gpurun_out/r6y/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r6y/forward_ab.jsonl"):
    d = json.loads(l)
    print(d["tag"], d["dtype"], [(r["ratio"], r["forward_ms"]) for r in d["rows"]])
PY
timeout 300 python tools/headtail_bench.py > gpurun_out/r6y/headtail_new.json 2>> gpurun_out/r6y/err.log
SIGE_HIP_LIB=$L/libsige_hip_prev.so timeout 300 python tools/headtail_bench.py > gpurun_out/r6y/headtail_prev.json 2>> gpurun_out/r6y/err.log
tail -c 1500 gpurun_out/r6y/headtail_new.json; echo; tail -c 1500 gpurun_out/r6y/headtail_prev.json; tail -n 3 gpurun_out/r6y/err.log
