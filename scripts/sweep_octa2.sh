#!/bin/bash
mkdir -p gpurun_out/octa
run() { timeout -k 10 120 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-search 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', 'ms', round(j['ms_per_step'],2), j['count_ok'], j['chunks_per_step_by_rank'])" || exit 1; }
for sg in 64 128 256; do for b in 2048 4096; do
  GPM_OCTA_SEG=$sg GPM_OCTA_BLOCKS=$b GPM_OCTA_MEM_GB=64 run "seg $sg slabs $b"
done; done > gpurun_out/octa/seg.txt 2>&1
cat gpurun_out/octa/seg.txt
