#!/bin/bash
# round 6, GPU session 5: store flavours (tools/probe/store_probe.hip)
mkdir -p gpurun_out/r6e
cd /root/repo
timeout 600 ./tools/probe/store_probe gpurun_out/r6e/store_probe.json > gpurun_out/r6e/store_probe.log 2>&1
cat gpurun_out/r6e/store_probe.log
