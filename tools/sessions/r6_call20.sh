#!/bin/bash
# round 6, GPU session 20: exact fp32: conv1s with a held shortcut routed to the v3 kernel (pair broken up) from 512 / 768 / 1024 workgroups
mkdir -p gpurun_out/r6t
cd /root/repo
export TMPDIR=/tmp
SIGE_HIP_LIB=$PWD/sige_amd/lib/libsige_hip_tuning.so timeout 1500 python tools/tile3_bench.py --skip-layers --ratios 0.05,0.10,0.15,0.20 --out gpurun_out/r6t/tile3_f32_pairs.json > gpurun_out/r6t/tile3_f32_pairs.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6t/tile3_f32_pairs.json"))
for r in d["forward"]:
    print(r["edit_ratio"], {k: (v["forward_ms"], v["launches"]) for k, v in r.items() if isinstance(v, dict)})
PY
tail -n 3 gpurun_out/r6t/tile3_f32_pairs.log
