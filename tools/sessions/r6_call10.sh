#!/bin/bash
# round 6, GPU session 10: fp16 tile conv v3 also for conv1s whose 1x1 shortcut is held for the pair kernel (thresholds)
mkdir -p gpurun_out/r6j
cd /root/repo
export TMPDIR=/tmp
SIGE_HIP_LIB=$PWD/sige_amd/lib/libsige_hip_tuning.so timeout 1200 python tools/tile3_bench.py --compute f16 --cache-dtype f16 --out gpurun_out/r6j/tile3_f16_pairs_bench.json > gpurun_out/r6j/tile3_f16_pairs_bench.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6j/tile3_f16_pairs_bench.json"))
for r in d["forward"]:
    print(r["edit_ratio"], {k: (v["forward_ms"], v["launches"]) for k, v in r.items() if isinstance(v, dict)})
PY
tail -n 3 gpurun_out/r6j/tile3_f16_pairs_bench.log
