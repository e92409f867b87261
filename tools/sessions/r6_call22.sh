#!/bin/bash
# round 6, GPU session 22: is the tile-list conv_in the one that runs, and what does it cost inside the forward (kernel trace)
mkdir -p gpurun_out/r6v
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q -k conv_in > gpurun_out/r6v/pytest.log 2>&1; tail -n 3 gpurun_out/r6v/pytest.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_v -o t -- python /root/repo/tools/forward_ab.py --tag trace > /dev/null 2> /tmp/prof_v.err
f=$(find /tmp/prof_v -name '*kernel_stats.csv' | head -1); cp "$f" /root/repo/gpurun_out/r6v/kernel_stats.csv
grep -i "small_cin\|conv_in" "$f" | cut -c 1-300
