#!/bin/bash
# round 6, GPU session 4: conv_in / conv_out rewritten: head / tail launches, 11 vs 8 vs 4 waves in conv_out, their tests
mkdir -p gpurun_out/r6d
cd /root/repo
export TMPDIR=/tmp
timeout 300 python tools/headtail_bench.py > gpurun_out/r6d/headtail_w11.jsonl 2> gpurun_out/r6d/headtail_w11.err
for W in 8 4; do
  SIGE_HIP_LIB=$PWD/sige_amd/lib/libsige_hip_convout_w$W.so timeout 300 python tools/headtail_bench.py > gpurun_out/r6d/headtail_w$W.jsonl 2> gpurun_out/r6d/headtail_w$W.err
done
tail -n +1 gpurun_out/r6d/headtail_w*.jsonl
tail -n 3 gpurun_out/r6d/headtail_w11.err
timeout 900 python -m pytest tests -q -m gpu -k "small_cout or small_cin or input_conv2d or benchmarked_forward or ddpm_unet_gpu or example or conv_img or gaugan_generator" > gpurun_out/r6d/pytest_subset.log 2>&1
tail -n 5 gpurun_out/r6d/pytest_subset.log
