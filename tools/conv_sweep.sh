mkdir -p gpurun_out
python tools/conv_bench.py --dense --cl --tiles auto,16x1,16x2,32x1,32x2 --waves 4,8 --ksplit 0,2 > gpurun_out/r2o_conv_sweep_dense.jsonl 2> gpurun_out/r2o_conv_sweep_dense.err
python tools/conv_bench.py --sparse --cl --ratio 0.012 --tiles auto,16x1,16x2,32x1,32x2 --waves 4,8 --ksplit 0,2 > gpurun_out/r2o_conv_sweep_sparse.jsonl 2> gpurun_out/r2o_conv_sweep_sparse.err
tail -3 gpurun_out/r2o_conv_sweep_*.err
wc -l gpurun_out/r2o_conv_sweep_*.jsonl
