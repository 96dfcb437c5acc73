#!/bin/bash
# Configuration 5 (octahedra) over chunk counts, streams and the wide-launch bar: wall time per search, count
# checked by bench.py.
mkdir -p gpurun_out/octa
run() { timeout -k 10 120 python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-search 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', 'ms', round(j['ms_per_step'],2), j['count_ok'], j['chunks_per_step_by_rank'])" || exit 1; }
for st in 4 2 1; do for c in 4 16 64 256; do
  GPM_OCTA_STREAMS=$st GPM_BENCH_CHUNKS=$c run "streams $st chunks_hint $c"
done; done > gpurun_out/octa/chunks.txt 2>&1
for w in 0 512 65536; do GPM_OCTA_WIDE=$w GPM_BENCH_CHUNKS=64 run "wide_bar $w chunks_hint 64"; done >> gpurun_out/octa/chunks.txt 2>&1
cat gpurun_out/octa/chunks.txt
