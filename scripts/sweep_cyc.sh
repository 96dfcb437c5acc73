#!/bin/bash
# Configuration 2 (4-cycles, R-MAT scale 20) over chunk counts: wall time per search, count checked by bench.py.
mkdir -p gpurun_out/cyc
run() { timeout -k 10 120 python bench.py --config 2 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-search 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', 'ms', round(j['ms_per_step'],2), j['count_ok'], j['chunks_per_step_by_rank'])" || exit 1; }
for c in 16 32 64 128 256; do GPM_BENCH_CHUNKS=$c run "chunks_hint $c"; done > gpurun_out/cyc/chunks.txt 2>&1
cat gpurun_out/cyc/chunks.txt
