#!/bin/bash
# round 6, GPU session 49: the TunableOp table of SD's token GEMMs re-measured with 100 ms per candidate instead of 30 (is the shipped
# table's choice noise-limited?), then the SD line from the new table and from the shipped one
mkdir -p gpurun_out/r6ax
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r6ax
T0=$SECONDS
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$PWD/$O/tunableop.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=100 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=10 \
  timeout 1800 python bench.py --workload sd --steps 20 --warmup 5 --no-tuned-gemms --cpu-seconds 0 2> $O/sd_tuning.err | tail -1 > $O/sd_tuning.json
echo "tuning run: $((SECONDS - T0)) s"
cp $O/tunableop0.csv $O/table_100ms.csv
for rep in 1 2; do
python - <<'PY'
import json, os, subprocess, sys
env = dict(os.environ)
for tag, table in (("table_100ms", os.path.abspath("gpurun_out/r6ax/table_100ms.csv")), ("shipped_table", "")):
    code = ("import sys, runpy\n"
            "from sige_amd.workloads import gemm_tuning\n"
            + ("gemm_tuning.TABLE = %r\n" % table if table else "")
            + "sys.argv = ['bench.py', '--workload', 'sd', '--steps', '20', '--warmup', '5', '--cpu-seconds', '0']\n"
            "runpy.run_path('bench.py', run_name='__main__')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600).stdout.strip().splitlines()
    d = json.loads(out[-1])
    print(tag, d["forward_ms"], d["dense_forward_ms"], flush=True)
    open("gpurun_out/r6ax/lines.jsonl", "a").write(json.dumps({"table": tag, "forward_ms": d["forward_ms"], "dense_forward_ms": d["dense_forward_ms"]}) + "\n")
PY
done
