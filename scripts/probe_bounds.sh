#!/bin/bash
# What binds the two big kernels of the headline search, measured by leaving work out (the probes give wrong
# counts; they are diagnostics): family times with one chunk in flight
#   - as shipped,
#   - the list kernel reading its lists without testing them, whole lists and half of every list,
#   - the count kernel copying tiles and listing set bits without ANDing rows,
# and the streaming rates a plain library reduction / copy reaches on this GPU.
# usage (on the GPU box): bash scripts/probe_bounds.sh   -> gpurun_out/bound_probes.txt
out=gpurun_out/bound_probes.txt
: > $out
run() {
  echo "== ${*:-as shipped}" >> $out
  env "$@" timeout -k 10 120 python scripts/trace_classes.py 3 1 > gpurun_out/probe_one.log 2>&1 || { echo "failed" >> $out; tail -5 gpurun_out/probe_one.log >> $out; exit 1; }
  grep -E "k_clq_build|k_clq_count4|sum ms" gpurun_out/probe_one.log >> $out
}
run
run GPM_CLQ_PROBE=1
run GPM_CLQ_PROBE=2
run GPM_COUNT4_PROBE=1
echo "== streaming rates (scripts/bw_probe.py)" >> $out
timeout -k 10 120 python scripts/bw_probe.py >> $out 2>&1
