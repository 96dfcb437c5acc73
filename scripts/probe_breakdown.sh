#!/bin/bash
# Where the instructions and the time of the list kernel (k_clq_build_lists) go: the headline search with one chunk
# in flight, once as shipped and once per combination of parts left out (GPM_CLQ_PROBE bits: 1 no membership tests,
# 4 no hub-matrix gathers, 8 no rows written; counts are wrong with them).  Per run: wave-level VALU and SALU
# instructions of the kernel over one search (rocprofv3 counter pass, scripts/pmc_one.sh) and its time by size class.
# usage (on the GPU box): bash scripts/probe_breakdown.sh  -> gpurun_out/list_kernel_breakdown.txt
out=gpurun_out/list_kernel_breakdown.txt
: > $out
for p in 0 1 4 5 8 13; do
  bash scripts/pmc_one.sh p$p GPM_CLQ_PROBE=$p > /dev/null 2>&1
  GPM_CLQ_PROBE=$p timeout -k 10 100 python scripts/trace_classes.py 3 1 > gpurun_out/tcp$p.log 2>&1
  echo "GPM_CLQ_PROBE=$p: $(grep build_lists gpurun_out/pmc_p$p.txt | sed 's/.*SQ_INSTS_SALU/SQ_INSTS_SALU/;s/SQ_WAIT.*//') ms by class 2/3/4: $(grep k_clq_build gpurun_out/tcp$p.log | awk '{printf "%s ", $NF}')" >> $out
done
cat $out
