#!/bin/bash
# round 5, GPU session 3: GauGAN on the library (new helper kernels, launch plan), the 80 ms host-wait spikes vs the interrupt path
mkdir -p gpurun_out/r5c
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_models_golden.py tests/test_c_abi.py -m gpu -q --tb=short > gpurun_out/r5c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5c/pytest.log
tail -n 30 gpurun_out/r5c/pytest.log
timeout 200 python tools/gaugan_latency.py --out gpurun_out/r5c/gaugan_latency_default.json > gpurun_out/r5c/gl_default.log 2>&1
HSA_ENABLE_INTERRUPT=0 timeout 200 python tools/gaugan_latency.py --out gpurun_out/r5c/gaugan_latency_nointerrupt.json > gpurun_out/r5c/gl_nointr.log 2>&1
timeout 200 python tools/gaugan_latency.py --spin --out gpurun_out/r5c/gaugan_latency_spin.json > gpurun_out/r5c/gl_spin.log 2>&1
mkdir -p /tmp/prof && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gaugan -- python /root/repo/tools/profile_gaugan.py --replays 10 > /root/repo/gpurun_out/r5c/prof_gaugan.log 2>&1)
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r5c/gaugan_kernel_stats.csv \;
head -n 40 gpurun_out/r5c/gaugan_kernel_stats.csv | cut -c1-160
timeout 600 python bench.py > gpurun_out/r5c/bench.out 2> gpurun_out/r5c/bench.err; echo "bench rc=$?"
tail -n 1 gpurun_out/r5c/bench.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('gaugan'))); print(d['forward_ms'], d['roofline'])"
cp bench_detail.json gpurun_out/r5c/ 2>/dev/null
