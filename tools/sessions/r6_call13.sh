#!/bin/bash
# round 6, GPU session 13: fp16 tile conv v3: (workgroups per CU, weight ring) = (3, 6) vs (4, 3) vs (3, 3)
mkdir -p gpurun_out/r6m
cd /root/repo
export TMPDIR=/tmp
for V in _t3h_occ3_rb6 _t3h_occ4_rb3 _t3h_occ3_rb3; do
  SIGE_HIP_LIB=$PWD/sige_amd/lib/libsige_hip$V.so timeout 900 python tools/tile3_bench.py --compute f16 --cache-dtype f16 --ratios 0.05,0.15,0.20 --out gpurun_out/r6m/tile3_f16$V.json > gpurun_out/r6m/tile3_f16$V.log 2>&1
done
python - <<'PY'
import json
for v in ("_t3h_occ3_rb6", "_t3h_occ4_rb3", "_t3h_occ3_rb3"):
    try:
        d = json.load(open("gpurun_out/r6m/tile3_f16%s.json" % v))
    except Exception as e:
        print(v, e); continue
    print(v)
    for r in d["layers"]:
        print("  ", r["edit_ratio"], r["resolution"], r["v3_workgroups"], {k[:14]: (v_["conv_mfma"], v_["tile3"]) for k, v_ in r.items() if isinstance(v_, dict)})
    for r in d["forward"]:
        print("  ", r["edit_ratio"], {k: v_["forward_ms"] for k, v_ in r.items() if isinstance(v_, dict) and k in ("conv_mfma_only", "router")})
PY
