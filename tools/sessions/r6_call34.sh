#!/bin/bash
# round 6, GPU session 34: the fp16-stored residual requested in the prologue without its conversion (no vmcnt(0) there), the label
# map resized once per size within a GauGAN forward, kernarg_touch out of the attention kernels: whole GPU suite, f16 forward A/B,
# GauGAN trace
mkdir -p gpurun_out/r6ah
cd /root/repo
export TMPDIR=/tmp
L=$PWD/sige_amd/lib
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6ah/pytest.log 2>&1; tail -n 3 gpurun_out/r6ah/pytest.log
for rep in 1 2; do
timeout 300 python tools/forward_ab.py --tag raw-residual --dtype f16 >> gpurun_out/r6ah/forward_ab.jsonl 2>> gpurun_out/r6ah/err.log
SIGE_HIP_LIB=$L/libsige_hip_prev.so timeout 300 python tools/forward_ab.py --tag previous --dtype f16 >> gpurun_out/r6ah/forward_ab.jsonl 2>> gpurun_out/r6ah/err.log
done
python - <<'PY'
import json
for l in open("gpurun_out/r6ah/forward_ab.jsonl"):
    d = json.loads(l)
    print(d["tag"], d["dtype"], [(r["ratio"], r["forward_ms"]) for r in d["rows"]])
PY
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gg -o gg -- python /root/repo/tools/profile_gaugan.py --replays 10 > /root/repo/gpurun_out/r6ah/trace_gaugan.log 2>&1
tail -n 4 /root/repo/gpurun_out/r6ah/trace_gaugan.log
f=$(ls /tmp/gg/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && grep -i "resize" "$f" | cut -c 1-200
