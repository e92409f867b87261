set -u
TAG=$1; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace_f32" -o fwd -- python $ROOT/tools/profile_forward.py --replays 50 --dtype f32 > "$OUT/${TAG}_trace_f32.log" 2>&1
cd $ROOT
T=$(ls "$OUT/${TAG}_trace_f32"/*kernel_trace.csv | head -1)
python tools/trace_summary.py "$T" --replays 50 --by-grid --top 12 --out "$OUT/${TAG}_kerneltrace_by_grid_sparse_fwd_1p2pct_f32.csv" --sequence "$OUT/${TAG}_kernel_sequence_sparse_fwd_1p2pct_f32.csv"
rm -rf "$OUT/${TAG}_trace_f32"
