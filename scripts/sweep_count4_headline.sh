#!/bin/bash
# Headline time (chunks overlapping as in the product) under count-kernel tile settings.
out=gpurun_out/sweep_vt_headline.log
: > $out
run() {
  echo "== $*" >> $out
  env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-search > gpurun_out/sweep_vt_one.log 2>&1 || { echo "failed" >> $out; tail -5 gpurun_out/sweep_vt_one.log >> $out; exit 1; }
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/sweep_vt_one.log | head -1 >> $out
}
run GPM_COUNT4_VT_3=1024
run GPM_COUNT4_VT_3=512 GPM_COUNT4_VT_4=512 GPM_COUNT4_WIDE_MIN=512
run GPM_COUNT4_VT_3=512 GPM_COUNT4_VT_4=512
run GPM_COUNT4_VT_3=512 GPM_COUNT4_VT_4=512 GPM_COUNT4_WIDE_MIN=512 GPM_CLIQUE_INFLIGHT=1
