#!/bin/bash
# One GPU-box session: K-loop ablations of the tile conv (python -m sige_amd.build --probe with SIGE_PROBE_TAG=_ablN
# SIGE_PROBE_DEFS=-DSIGE_ABL=N built beforehand; see conv_mfma.hpp SIGE_ABL_HAS).  Timing only: ablated results are wrong.
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-abl}
DEFAULT_CASES="f32:A,A',B,D"
CASES=${2:-$DEFAULT_CASES}
: > $OUT/${TAG}_conv_ablation.jsonl
for lib in sige_amd/lib/libsige_hip_probe_abl*.so; do
  SIGE_HIP_LIB=$PWD/$lib SIGE_PROBE_CASES="$CASES" timeout 300 python tools/conv_phase_probe.py >> $OUT/${TAG}_conv_ablation.jsonl 2>> $OUT/${TAG}_conv_ablation.err
done
python - <<PY
import json
rows=[json.loads(l) for l in open("$OUT/${TAG}_conv_ablation.jsonl") if l.startswith("{")]
for r in rows:
    m=r.get("median_ticks",{})
    print("%-28s %-60s launch %6.2f us  K loop %6d  total %6d" % (r["lib"], r["case"][:60], r.get("graph_launch_us",0), m.get("3-4 K loop",0), m.get("0-5 total",0)))
PY
