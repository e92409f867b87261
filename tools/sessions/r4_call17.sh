#!/bin/bash
# round 4, GPU session 17: host-time profile of the eager forward
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4r; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/probe/eager_profile.py --out $OUT/eager_profile.txt > $OUT/eager_profile.log 2>&1
echo "rc=$?"; tail -5 $OUT/eager_profile.log
