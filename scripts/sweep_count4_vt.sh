#!/bin/bash
# Tile widths of the 4-clique count kernel, per size class: family times of the headline with one chunk in flight.
# usage: bash scripts/sweep_count4_vt.sh   (on the GPU box; writes gpurun_out/sweep_vt.log)
out=gpurun_out/sweep_vt.log
: > $out
run() {
  echo "== $*" >> $out
  env "$@" timeout -k 10 120 python scripts/trace_classes.py 3 1 > gpurun_out/sweep_vt_one.log 2>&1 || { echo "failed" >> $out; tail -5 gpurun_out/sweep_vt_one.log >> $out; exit 1; }
  grep -o '"value": [0-9.e+]*' gpurun_out/sweep_vt_one.log | head -1 >> $out
  grep -E "count4|sum ms" gpurun_out/sweep_vt_one.log >> $out
}
run GPM_COUNT4_VT_2=1024
run GPM_COUNT4_VT_2=256 GPM_COUNT4_VT_3=512 GPM_COUNT4_VT_4=512
run GPM_COUNT4_VT_2=256 GPM_COUNT4_VT_3=512 GPM_COUNT4_VT_4=512 GPM_COUNT4_WIDE_MIN=256
run GPM_COUNT4_VT_3=256 GPM_COUNT4_VT_4=256 GPM_COUNT4_WIDE_MIN=512
