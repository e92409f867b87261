#!/bin/bash
# round 6, GPU session 29: 32-bit index arithmetic in the elementwise tile kernels (gather / scatter_gather / scatter / spade /
# token ops) and the branch-free add + LayerNorm: the whole GPU suite, SD with the fused token ops on / off, SD and GauGAN against
# the previous build
mkdir -p gpurun_out/r6ac
cd /root/repo
export TMPDIR=/tmp
L=$PWD/sige_amd/lib
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6ac/pytest.log 2>&1; tail -n 3 gpurun_out/r6ac/pytest.log
timeout 600 python tools/sd_fused_tokens_ab.py --out gpurun_out/r6ac/sd_fused_tokens.json 2> gpurun_out/r6ac/err.log
SIGE_HIP_LIB=$L/libsige_hip_prev.so timeout 600 python tools/sd_fused_tokens_ab.py --settings 0,1 --out gpurun_out/r6ac/sd_fused_tokens_prev.json 2>> gpurun_out/r6ac/err.log
timeout 600 python bench.py --workload gaugan --steps 20 --warmup 3 > gpurun_out/r6ac/bench_gaugan.json 2> gpurun_out/r6ac/bench_gaugan.err
SIGE_HIP_LIB=$L/libsige_hip_prev.so timeout 600 python bench.py --workload gaugan --steps 20 --warmup 3 > gpurun_out/r6ac/bench_gaugan_prev.json 2> gpurun_out/r6ac/bench_gaugan_prev.err
python - <<'PY'
import json
for f in ("bench_gaugan", "bench_gaugan_prev"):
    try:
        d = json.loads(open("gpurun_out/r6ac/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d.get("value"), d.get("forward_ms"), d.get("ms_per_step"), d.get("parity_ok"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -n 3 gpurun_out/r6ac/err.log
