#!/bin/bash
# round 6, GPU session 46: runtime knobs that touch what the forward is made of (kernel-argument placement, how a graph's packets are
# submitted) on the benchmarked forward, same library
mkdir -p gpurun_out/r6as
cd /root/repo
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 300 python tools/forward_ab.py --ratios 0.012,0.05 --tag "$tag" >> gpurun_out/r6as/forward_ab.jsonl 2>> gpurun_out/r6as/err.log; }
for rep in 1 2; do
run default X=1
run dev-kernarg-1 HIP_FORCE_DEV_KERNARG=1
run dev-kernarg-0 HIP_FORCE_DEV_KERNARG=0
run graph-packet-capture-1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run graph-packet-capture-0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
done
python - <<'PY'
import json
for l in open("gpurun_out/r6as/forward_ab.jsonl"):
    d = json.loads(l)
    print(d["tag"], [(r["ratio"], r["forward_ms"], r["checksum"]) for r in d["rows"]])
PY
tail -n 2 gpurun_out/r6as/err.log
