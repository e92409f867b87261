#!/bin/bash
# round 6, GPU session 8: the tile conv v3 on fp16 operands: its tests, launch by launch against conv_mfma.hpp's fp16 form, forward
mkdir -p gpurun_out/r6h
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q > gpurun_out/r6h/pytest_round6.log 2>&1
tail -n 15 gpurun_out/r6h/pytest_round6.log
timeout 900 python tools/tile3_bench.py --compute f16 --cache-dtype f16 --out gpurun_out/r6h/tile3_f16_bench.json > gpurun_out/r6h/tile3_f16_bench.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6h/tile3_f16_bench.json"))
for r in d["layers"]:
    print(r["edit_ratio"], r["resolution"], r["tiles"], r["v3_workgroups"], {k: (v["conv_mfma"], v["tile3"]) for k, v in r.items() if isinstance(v, dict)})
for r in d["forward"]:
    print(r)
PY
tail -n 5 gpurun_out/r6h/tile3_f16_bench.log
