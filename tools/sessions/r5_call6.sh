#!/bin/bash
# round 5, GPU session 6: GauGAN launch plan with keyed slab convs
mkdir -p gpurun_out/r5f
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_models_golden.py -m gpu -q --tb=short -k "gaugan or spade" > gpurun_out/r5f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5f/pytest.log
tail -n 25 gpurun_out/r5f/pytest.log
timeout 300 python - > gpurun_out/r5f/gaugan_section.json 2> gpurun_out/r5f/gaugan_section.err <<'PY'
import json, sys
sys.path.insert(0, "/root/repo")
import torch
from benchlib.gaugan import gaugan_section
print(json.dumps(gaugan_section(torch.device("cuda:0"), cpu_parity=False)))
PY
cat gpurun_out/r5f/gaugan_section.json | cut -c1-3000; tail -n 5 gpurun_out/r5f/gaugan_section.err
