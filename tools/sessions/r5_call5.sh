#!/bin/bash
# round 5, GPU session 5: GauGAN launch plan (all-library forward), stacked standalone gathers, resident vs host-built edit inputs
mkdir -p gpurun_out/r5e
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_channels_last.py -m gpu -q --tb=short > gpurun_out/r5e/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5e/pytest.log
tail -n 25 gpurun_out/r5e/pytest.log
timeout 200 python tools/gaugan_latency.py --out gpurun_out/r5e/gaugan_latency_resident.json > gpurun_out/r5e/gl_res.log 2>&1
timeout 200 python tools/gaugan_latency.py --host-inputs --out gpurun_out/r5e/gaugan_latency_host_inputs.json > gpurun_out/r5e/gl_host.log 2>&1
timeout 200 python tools/torch_ops_probe.py --workload gaugan --out gpurun_out/r5e/torch_ops_gaugan.json > gpurun_out/r5e/probe_gaugan.log 2>&1
timeout 600 python bench.py > gpurun_out/r5e/bench.out 2> gpurun_out/r5e/bench.err; echo "bench rc=$?"
tail -n 1 gpurun_out/r5e/bench.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('gaugan'))); print(d['forward_ms'], d['roofline'])"
cp bench_detail.json gpurun_out/r5e/ 2>/dev/null
tail -n 3 gpurun_out/r5e/bench.err
