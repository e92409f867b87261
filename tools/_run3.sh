set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
timeout 600 python tools/conv_probe.py > gpurun_out/conv_probe.jsonl 2> gpurun_out/conv_probe.err; echo "probe rc=$?"
cd /tmp
rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc1 -o p -- python $R/tools/conv_probe.py --pmc --tiles 16x2,32x1 --n 6 > $R/gpurun_out/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmc2 -o p -- python $R/tools/conv_probe.py --pmc --tiles 16x2,32x1 --n 6 > $R/gpurun_out/pmc2.log 2>&1
for d in pmc1 pmc2; do f=$(find /tmp/$d -name '*counter_collection.csv' | head -1); echo $f; python - $f $R/gpurun_out/$d.csv <<'PY'
import sys,csv,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name']
    if 'conv_mfma' not in k: continue
    key=(k, r.get('Grid_Size'), r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('LDS_Block_Size'))
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
w=csv.writer(open(sys.argv[2],'w'))
names=sorted({c for v in agg.values() for c in v})
w.writerow(['kernel','grid','vgpr','agpr','lds']+names+['n'])
for key,v in agg.items():
    w.writerow(list(key)+[round(sum(v[c])/len(v[c]),1) if c in v else '' for c in names]+[len(next(iter(v.values())))])
PY
done
tail -3 $R/gpurun_out/pmc1.log
