#!/bin/bash
# round 6, GPU session 12: fp16 tile conv v3 at three workgroups per CU (weight ring of 6 / 9 k-steps) against the default (two, 9)
mkdir -p gpurun_out/r6l
cd /root/repo
export TMPDIR=/tmp
for V in "" _t3h_occ3_rb6 _t3h_occ3_rb9; do
  L=$PWD/sige_amd/lib/libsige_hip$V.so
  [ -z "$V" ] && L=$PWD/sige_amd/lib/libsige_hip_tuning.so
  SIGE_HIP_LIB=$L timeout 900 python tools/tile3_bench.py --compute f16 --cache-dtype f16 --ratios 0.05,0.15,0.20 --out gpurun_out/r6l/tile3_f16${V:-_default}.json > gpurun_out/r6l/tile3_f16${V:-_default}.log 2>&1
done
python - <<'PY'
import json
for v in ("_default", "_t3h_occ3_rb6", "_t3h_occ3_rb9"):
    try:
        d = json.load(open("gpurun_out/r6l/tile3_f16%s.json" % v))
    except Exception as e:
        print(v, e); continue
    print(v)
    for r in d["layers"]:
        print("  ", r["edit_ratio"], r["resolution"], r["v3_workgroups"], {k[:14]: (v_["conv_mfma"], v_["tile3"]) for k, v_ in r.items() if isinstance(v_, dict)})
    for r in d["forward"]:
        print("  ", r["edit_ratio"], {k: v_["forward_ms"] for k, v_ in r.items() if isinstance(v_, dict) and k in ("conv_mfma_only", "router")})
PY
rm -f gpurun_out/test_margins.jsonl
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6l/pytest_gpu_x.log 2>&1
echo "pytest -x rc $?"; tail -n 4 gpurun_out/r6l/pytest_gpu_x.log
