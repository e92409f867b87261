#!/bin/bash
# round 4, GPU session 8: DDPM attention in one launch; SD attention with 64 queries per workgroup
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4h; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -q -m gpu --tb=short -k "attention or sd_transformer or launch_plan_follows" > $OUT/pytest_round4.log 2>&1
echo "round4 attention tests rc=$?" >> $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_channels_last.py tests/test_gpu_round2.py -q -m gpu --tb=short -k "attention or ddpm" > $OUT/pytest_ddpm.log 2>&1
echo "ddpm / attention tests rc=$?" >> $OUT/summary.txt
timeout 600 python tools/attention_ab.py --out $OUT/attention_ab.json > $OUT/attention_ab.log 2>&1
echo "attention a/b rc=$?" >> $OUT/summary.txt
timeout 900 python bench.py --workload sd --steps 20 --warmup 5 > $OUT/bench_sd.json 2> $OUT/bench_sd.err
echo "bench sd rc=$?" >> $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/summary.txt
cat $OUT/summary.txt; tail -25 $OUT/pytest_round4.log; tail -6 $OUT/pytest_ddpm.log; tail -3 $OUT/attention_ab.log
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4h/bench_sd.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "forward_ms", "dense_forward_ms", "speedup_vs_dense", "parity_max_abs", "parity_ok", "hip_kernel_launches_per_forward")})
    print(json.dumps(d.get("attention_routing"), indent=1))
    print({k: v for k, v in (d.get("kernels") or {}).items()})
except Exception as e:
    print("sd bench parse failed", e)
try:
    d = json.loads(open("gpurun_out/r4h/bench.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "forward_ms", "forward_ms_eager", "launches_per_forward", "parity_max_abs", "parity_ok")})
    print(d.get("roofline", {}).get("frac"), d.get("kernels"))
    print({k: d["f16_compute"].get(k) for k in ("forward_ms", "parity_ok")}, d["dynamic"].get("mask_change_plan"))
except Exception as e:
    print("bench parse failed", e)
PY
tail -4 $OUT/bench_sd.err; tail -4 $OUT/bench.err
