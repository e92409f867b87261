#!/bin/bash
# round 6, GPU session 43 (the bench half of session 42, whose /usr/bin/time does not exist on the box): the committed tree on one more fresh box after the experiment of session 41 was reverted and the library
# rebuilt -- the driver's test command, smoke, and the default bench line twice (box-to-box / run-to-run spread of the headline)
mkdir -p gpurun_out/r6ap
cd /root/repo
export TMPDIR=/tmp
for i in 1 2; do
T0=$SECONDS
timeout 900 python bench.py 2> gpurun_out/r6ap/bench_$i.err | tail -1 > gpurun_out/r6ap/bench_$i.json
echo "bench $i wall $((SECONDS - T0)) s" >> gpurun_out/r6ap/summary.txt
python - <<PY >> gpurun_out/r6ap/summary.txt
import json
d = json.loads(open("gpurun_out/r6ap/bench_$i.json").read())
print("bench $i", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["parity_ok"], d.get("forward_ms_eager_launch_plan"))
PY
done
cat gpurun_out/r6ap/summary.txt
