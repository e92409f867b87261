#!/bin/bash
# round 6, GPU session 27: where the transposed-score attention kernel's wave cycles go at SD's self-attention shape (SQ and TCC
# counters, separate passes, kernel trace only)
mkdir -p gpurun_out/r6aa
cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r6aa
CMD="python $R/tools/attention_tokens_bench.py --eager 4 --only self_64,self_64_dense"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/p_sq -o pmc -- $CMD > $O/sq.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace --output-format csv -d /tmp/p_tcc -o pmc -- $CMD > $O/tcc.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d /tmp/p_sq2 -o pmc -- $CMD > $O/sq2.log 2>&1
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace --output-format csv -d /tmp/p_tcp -o pmc -- $CMD > $O/tcp.log 2>&1
for d in sq tcc sq2 tcp; do
  f=$(ls /tmp/p_$d/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/tools/pmc_rows.py "$f" attention > $O/$d.txt 2>&1 || tail -n 5 $O/$d.log > $O/$d.txt
  echo "== $d"; cat $O/$d.txt
done
