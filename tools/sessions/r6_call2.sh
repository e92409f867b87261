#!/bin/bash
# round 6, GPU session 2: what a dependent kernel boundary is made of (tools/probe/boundary_split.hip)
mkdir -p gpurun_out/r6b
cd /root/repo
timeout 600 ./tools/probe/boundary_split gpurun_out/r6b/boundary_split.json > gpurun_out/r6b/boundary_split.log 2>&1
cat gpurun_out/r6b/boundary_split.log
