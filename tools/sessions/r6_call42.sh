#!/bin/bash
# round 6, GPU session 42: the committed tree on one more fresh box after the experiment of session 41 was reverted and the library
# rebuilt -- the driver's test command, smoke, and the default bench line twice (box-to-box / run-to-run spread of the headline)
mkdir -p gpurun_out/r6ap
cd /root/repo
export TMPDIR=/tmp
python -c "import sige_amd.build as b; print('source_hash', b.source_hash())" > gpurun_out/r6ap/summary.txt
timeout 1500 python -m pytest tests -x -q -m gpu --tb=short > gpurun_out/r6ap/pytest_gpu.log 2>&1
echo "pytest -x -q -m gpu rc=$?" >> gpurun_out/r6ap/summary.txt
tail -n 1 gpurun_out/r6ap/pytest_gpu.log >> gpurun_out/r6ap/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6ap/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r6ap/summary.txt
for i in 1 2; do
/usr/bin/time -f "bench $i wall %e s" -a -o gpurun_out/r6ap/summary.txt timeout 900 python bench.py 2> gpurun_out/r6ap/bench_$i.err | tail -1 > gpurun_out/r6ap/bench_$i.json
python - <<PY >> gpurun_out/r6ap/summary.txt
import json
d = json.loads(open("gpurun_out/r6ap/bench_$i.json").read())
print("bench $i", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["parity_ok"], d.get("forward_ms_eager_launch_plan"))
PY
done
cat gpurun_out/r6ap/summary.txt
