#!/bin/bash
mkdir -p gpurun_out/cyc
run() { timeout -k 10 120 python bench.py --config 2 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-search 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', 'ms', round(j['ms_per_step'],2), j['count_ok'], j['chunks_per_step_by_rank'])" || exit 1; }
for f in 4096 8192 16384 32768; do GPM_CYC_RANGES_FROM=$f run "ranges_from $f"; done > gpurun_out/cyc/from.txt 2>&1
for w in 16 32 128 256; do GPM_CYC_RANGE_WORK=$w run "range_work $w"; done >> gpurun_out/cyc/from.txt 2>&1
for s in 8192 32768; do GPM_CYC_RANGE_IDS=$s run "range_ids $s"; done >> gpurun_out/cyc/from.txt 2>&1
cat gpurun_out/cyc/from.txt
