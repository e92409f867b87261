export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o bench -- python $R/bench.py --steps 50 --warmup 5 --cpu-seconds 0 --no-roofline --sweep '' > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/stats.log
f=$(find /tmp/st -name '*kernel_stats.csv' | head -1); echo stats=$f; cp $f $R/gpurun_out/r1k_rocprofv3_kernel_stats_bench.csv
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p1 -o p -- python $R/tools/profile_forward.py --mode eager --replays 4 > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p2 -o p -- python $R/tools/profile_forward.py --mode eager --replays 4 > $R/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/p3 -o p -- python $R/tools/profile_forward.py --mode eager --replays 4 > $R/gpurun_out/pmc_sq.log 2>&1
for d in p1 p2 p3; do f=$(find /tmp/$d -name '*counter_collection.csv' | head -1); python $R/tools/pmc_summary.py $f $R/gpurun_out/r1k_pmc_$d.csv sige; done
head -5 $R/gpurun_out/r1k_rocprofv3_kernel_stats_bench.csv
tail -c 600 $R/gpurun_out/bench_under_rocprof.json
