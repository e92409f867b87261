#!/bin/bash
# round 6, GPU session 47: SD's token GEMMs (2.3 ms of the 10.07 ms forward, torch -> hipBLASLt / rocBLAS) with PyTorch's TunableOp
# choosing the solution per shape: the SD bench line as it is, with tuning on (results to a csv), and from the csv without tuning
mkdir -p gpurun_out/r6at
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6at
timeout 600 python bench.py --workload sd --steps 20 --warmup 5 2> $O/sd_default.err | tail -1 > $O/sd_default.json
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$PWD/$O/tunableop.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5 \
  timeout 1200 python bench.py --workload sd --steps 20 --warmup 5 2> $O/sd_tuning.err | tail -1 > $O/sd_tuning.json
ls -la $O
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=$PWD/$O/tunableop.csv \
  timeout 600 python bench.py --workload sd --steps 20 --warmup 5 2> $O/sd_tuned.err | tail -1 > $O/sd_tuned.json
python - <<'PY'
import json
for f in ("sd_default", "sd_tuning", "sd_tuned"):
    try:
        d = json.loads(open("gpurun_out/r6at/%s.json" % f).read())
        print(f, d.get("value"), d.get("forward_ms"), d.get("parity_ok"), d.get("parity_max_abs"))
    except Exception as e:
        print(f, "failed", e)
PY
wc -l $O/tunableop*.csv; tail -n 3 $O/sd_tuning.err
