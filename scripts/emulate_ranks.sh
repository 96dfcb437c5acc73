#!/bin/bash
# Per-rank search time at N ranks, emulated on one GPU: rank r takes every N-th chunk.
N=${1:-8}; CH=${2:-$((64*N))}; shift; shift  # further arguments go to bench.py (e.g. --config 5)
for r in $(seq 0 $((N-1))); do
  GPM_EMULATE_RANKS=$N GPM_EMULATE_RANK=$r GPM_BENCH_CHUNKS=$CH timeout -k 10 200 python bench.py --steps ${STEPS:-3} --warmup ${WARMUP:-1} --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ranks $N chunks $CH rank $r ms', round(d['ms_per_step'],2), 'matches', d['matches'])" || exit 1
done
