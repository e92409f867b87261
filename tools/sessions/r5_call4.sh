#!/bin/bash
# round 5, GPU session 4: which aten ops are left in the GauGAN / SD sparse forwards; the 85 ms host stalls per way of waiting
mkdir -p gpurun_out/r5d
cd /root/repo
export TMPDIR=/tmp
timeout 300 python tools/torch_ops_probe.py --workload gaugan --out gpurun_out/r5d/torch_ops_gaugan.json > gpurun_out/r5d/probe_gaugan.log 2>&1
timeout 300 python tools/torch_ops_probe.py --workload sd --out gpurun_out/r5d/torch_ops_sd.json > gpurun_out/r5d/probe_sd.log 2>&1
timeout 300 python tools/sync_spike_probe.py --out gpurun_out/r5d/sync_spikes.json > gpurun_out/r5d/sync.log 2>&1
tail -n 5 gpurun_out/r5d/probe_gaugan.log gpurun_out/r5d/probe_sd.log gpurun_out/r5d/sync.log
