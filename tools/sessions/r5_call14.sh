#!/bin/bash
# round 5, GPU session 14: the SPADE generator in stacked mode (E edited label maps in one forward) -- tests + the bench section
mkdir -p gpurun_out/r5n
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_models_golden.py -m gpu -q --tb=short -x -k "gaugan or stacked or spade or split or resize or GauGAN" > gpurun_out/r5n/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5n/pytest.log
tail -n 30 gpurun_out/r5n/pytest.log
timeout 600 python - > gpurun_out/r5n/gaugan_section.json 2> gpurun_out/r5n/gaugan_section.err <<'PY'
import json, sys
sys.path.insert(0, "/root/repo")
import torch
from benchlib.gaugan import gaugan_section
r = gaugan_section(torch.device("cuda:0"), cpu_parity=False)
print(json.dumps(r))
PY
echo "section rc=$?"
python - <<'PY'
import json
d = json.load(open("/root/repo/gpurun_out/r5n/gaugan_section.json"))
print(json.dumps({k: d.get(k) for k in ("fused_spade_modulation", "batched_edits")}, indent=1)[:3000])
print(json.dumps(d.get("per_edit_latency_plan_ms", {}).get("to_first_output")), json.dumps(d.get("per_edit_latency_ms", {}).get("to_first_output")))
PY
tail -n 5 gpurun_out/r5n/gaugan_section.err
