#!/bin/bash
mkdir -p gpurun_out/cyc
run() { timeout -k 10 120 python bench.py --config 2 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-search 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', 'ms', round(j['ms_per_step'],2), j['count_ok'], j['chunks_per_step_by_rank'])" || exit 1; }
for k in 1 2 3 4 8; do GPM_CYC_PIECES=$k run "pieces $k"; done > gpurun_out/cyc/pieces.txt 2>&1
for k in 1 2 3; do GPM_CYC_PIECES=1 GPM_CYC_INFLIGHT=$k run "pieces 1 inflight $k"; done >> gpurun_out/cyc/pieces.txt 2>&1
for c in 16 32; do GPM_CYC_PIECES=1 GPM_BENCH_CHUNKS=$c run "pieces 1 chunks $c"; done >> gpurun_out/cyc/pieces.txt 2>&1
cat gpurun_out/cyc/pieces.txt
