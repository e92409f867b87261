#!/bin/bash
# The two traffic tables bench.py reads (profiles/pmc_traffic.json, pmc_traffic_f16.json) and nothing else: four rocprofv3 PMC
# passes (FETCH_SIZE / WRITE_SIZE, fp32 / f16 forward; each its own run, --kernel-trace only).  The raw counter tables of the
# conv kernels are kept (gzip) so that the reduction can be redone without the GPU.
set -u
TAG=${1:-r4}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
PF="python $ROOT/tools/profile_forward.py"
cd /tmp
for DT in f32 f16; do
  S=""; [ $DT = f16 ] && S="_f16"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_fetch$S" -o pmc -- $PF --mode eager --replays 4 --dtype $DT --manifest "$OUT/${TAG}_manifest$S.json" > "$OUT/${TAG}_pmc_fetch$S.log" 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_write$S" -o pmc -- $PF --mode eager --replays 4 --dtype $DT > "$OUT/${TAG}_pmc_write$S.log" 2>&1
done
cd "$ROOT"
for S in "" "_f16"; do
  F=$(ls "$OUT/${TAG}_pmc_fetch$S"/*counter_collection.csv | head -1); W=$(ls "$OUT/${TAG}_pmc_write$S"/*counter_collection.csv | head -1)
  python tools/pmc_traffic.py "$F" "$W" "$OUT/${TAG}_manifest$S.json" "$OUT/${TAG}_pmc_traffic$S.json" > "$OUT/${TAG}_pmc_traffic$S.txt" 2>&1
  grep -E "Counter_Name|sige::conv|sige::attn" "$F" | gzip > "$OUT/${TAG}_pmc_fetch_rows$S.csv.gz"
  grep -E "Counter_Name|sige::conv|sige::attn" "$W" | gzip > "$OUT/${TAG}_pmc_write_rows$S.csv.gz"
  find "$OUT/${TAG}_pmc_fetch$S" "$OUT/${TAG}_pmc_write$S" -type f -delete
  find "$OUT/${TAG}_pmc_fetch$S" "$OUT/${TAG}_pmc_write$S" -type d -empty -delete
  head -c 600 "$OUT/${TAG}_pmc_traffic$S.txt"; echo
done
ls -la "$OUT"/${TAG}_pmc_*rows*
