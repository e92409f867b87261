#!/bin/bash
# One GPU-box session producing the round's measured evidence (run through gpurun from the repo root):
#   tools/gpu_profile_round.sh r2c
# Writes under gpurun_out/<tag>_*: the bench line, rocprofv3 --kernel-trace --stats of 50 hipGraph replays of the sparse
# forward (fp32 and f16 compute), and three PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters; each in its own run, kernel-trace
# only) of eager forwards with the conv-family manifest tools/pmc_traffic.py needs.
set -u
TAG=${1:-r2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PF="python $ROOT/tools/profile_forward.py"
for DT in f32 f16; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace_$DT" -o fwd -- $PF --replays 50 --dtype $DT > "$OUT/${TAG}_trace_$DT.log" 2>&1
done
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_fetch" -o pmc -- $PF --mode eager --replays 4 --manifest "$OUT/${TAG}_manifest.json" > "$OUT/${TAG}_pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_write" -o pmc -- $PF --mode eager --replays 4 > "$OUT/${TAG}_pmc_write.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_sq" -o pmc -- $PF --mode eager --replays 4 > "$OUT/${TAG}_pmc_sq.log" 2>&1
cd "$ROOT"
# keep what is small: stats + counter tables (the raw kernel traces of 50 replays are a few MB)
find "$OUT" -name "*_agent_info.csv" -delete 2>/dev/null
du -sh "$OUT"/${TAG}_* 2>/dev/null | tail -20
