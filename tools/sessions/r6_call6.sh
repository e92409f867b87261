#!/bin/bash
# round 6, GPU session 6: conv_in (branch-free loads, LDS weights, coalesced 16-byte stores) / conv_out (weights through LDS),
# write-through epilogue stores vs plain stores (A/B builds) on the whole forward
mkdir -p gpurun_out/r6f
cd /root/repo
export TMPDIR=/tmp
timeout 300 python tools/headtail_bench.py > gpurun_out/r6f/headtail.jsonl 2> gpurun_out/r6f/headtail.err
cat gpurun_out/r6f/headtail.jsonl
for rep in 1 2; do
  timeout 300 python tools/forward_ab.py --tag write-through >> gpurun_out/r6f/forward_ab.jsonl 2>> gpurun_out/r6f/forward_ab.err
  SIGE_HIP_LIB=$PWD/sige_amd/lib/libsige_hip_plain.so timeout 300 python tools/forward_ab.py --tag plain >> gpurun_out/r6f/forward_ab.jsonl 2>> gpurun_out/r6f/forward_ab.err
done
cat gpurun_out/r6f/forward_ab.jsonl
tail -n 3 gpurun_out/r6f/forward_ab.err
timeout 900 python -m pytest tests -q -x -m gpu -k "small_cout or small_cin or input_conv2d or benchmarked_forward or ddpm_unet_gpu or example or conv_img or gaugan_generator or wide_conv_vs or golden_cases or tile_conv3 or twins or f16_compute" > gpurun_out/r6f/pytest_subset.log 2>&1
tail -n 5 gpurun_out/r6f/pytest_subset.log
