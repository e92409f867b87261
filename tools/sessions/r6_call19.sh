#!/bin/bash
# round 6, GPU session 19: the sparse-list routing rule for the exact-fp32 tile conv v3 (dense layers keep 512)
mkdir -p gpurun_out/r6r
cd /root/repo
export TMPDIR=/tmp
SIGE_HIP_LIB=$PWD/sige_amd/lib/libsige_hip_tuning.so timeout 1500 python tools/tile3_bench.py --skip-layers --out gpurun_out/r6r/tile3_f32_sparse_min.json > gpurun_out/r6r/tile3_f32_sparse_min.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6r/tile3_f32_sparse_min.json"))
for r in d["forward"]:
    print(r["edit_ratio"], {k.replace("sparse_from_", "s"): (v["forward_ms"], v["launches"]) for k, v in r.items() if isinstance(v, dict)})
PY
tail -n 3 gpurun_out/r6r/tile3_f32_sparse_min.log
