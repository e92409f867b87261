#!/bin/bash
# round 6, GPU session 15: routing threshold of the fp16 tile conv v3 in its final form (3 workgroups per CU), pairs included
mkdir -p gpurun_out/r6n
cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
SIGE_HIP_LIB=$PWD/sige_amd/lib/libsige_hip_tuning.so timeout 1200 python tools/tile3_bench.py --compute f16 --cache-dtype f16 --skip-layers --out gpurun_out/r6n/tile3_f16_thresholds_$rep.json > gpurun_out/r6n/tile3_f16_thresholds_$rep.log 2>&1
done
python - <<'PY'
import json
for rep in (1, 2):
    d = json.load(open("gpurun_out/r6n/tile3_f16_thresholds_%d.json" % rep))
    for r in d["forward"]:
        print(rep, r["edit_ratio"], {k.replace("router_from_", "th"): v["forward_ms"] for k, v in r.items() if isinstance(v, dict)})
PY
