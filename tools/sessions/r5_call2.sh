#!/bin/bash
# round 5, GPU session 2: the whole GPU suite (no -x), GauGAN's 80 ms spikes vs Python's cyclic GC, the 2-rank job (gloo, one GPU)
mkdir -p gpurun_out/r5b
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r5b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5b/pytest.log
tail -n 8 gpurun_out/r5b/pytest.log
for m in default freeze off; do
  timeout 200 python tools/gaugan_latency.py --gc $m --out gpurun_out/r5b/gaugan_latency_gc_$m.json > gpurun_out/r5b/gl_$m.log 2>&1
done
timeout 900 python bench.py --gpus 2 --backend gloo --oversubscribe --steps 20 --warmup 5 --no-extras --cpu-seconds 0 > gpurun_out/r5b/bench_2ranks_gloo.out 2> gpurun_out/r5b/bench_2ranks_gloo.err; echo "bench2 rc=$?"
tail -c 3000 gpurun_out/r5b/bench_2ranks_gloo.out; tail -n 5 gpurun_out/r5b/bench_2ranks_gloo.err
