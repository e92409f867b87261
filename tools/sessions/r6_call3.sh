#!/bin/bash
# round 6, GPU session 3: instruction fetch per launch (tools/probe/icache_probe.hip)
mkdir -p gpurun_out/r6c
cd /root/repo
timeout 600 ./tools/probe/icache_probe gpurun_out/r6c/icache_probe.json > gpurun_out/r6c/icache_probe.log 2>&1
cat gpurun_out/r6c/icache_probe.log
