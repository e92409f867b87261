#!/bin/bash
# round 6, GPU session 9: fp16 tile conv v3 with 4 tiles per workgroup (thresholds), then the whole GPU suite on the new build
mkdir -p gpurun_out/r6i
cd /root/repo
export TMPDIR=/tmp
SIGE_HIP_LIB=$PWD/sige_amd/lib/libsige_hip_tuning.so timeout 1200 python tools/tile3_bench.py --compute f16 --cache-dtype f16 --out gpurun_out/r6i/tile3_f16_tpw4_bench.json > gpurun_out/r6i/tile3_f16_tpw4_bench.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6i/tile3_f16_tpw4_bench.json"))
for r in d["layers"]:
    print(r["edit_ratio"], r["resolution"], r["tiles"], r["v3_workgroups"], {k[:14]: (v["conv_mfma"], v["tile3"], v.get("tile3_tpw4")) for k, v in r.items() if isinstance(v, dict)})
for r in d["forward"]:
    print(r["edit_ratio"], {k: v["forward_ms"] for k, v in r.items() if isinstance(v, dict)})
PY
tail -n 3 gpurun_out/r6i/tile3_f16_tpw4_bench.log
rm -f gpurun_out/test_margins.jsonl
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6i/pytest_gpu_x.log 2>&1
echo "pytest -x rc $?"; tail -n 5 gpurun_out/r6i/pytest_gpu_x.log
