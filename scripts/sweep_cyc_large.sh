#!/bin/bash
# 4-cycles on graphs larger than configuration 2: two-paths per range of ids from which a root counts in LDS
run() { s=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --config 2 --scale $s --steps 2 --warmup 1 --no-cpu-baseline --no-extra-search > gpurun_out/c2s.json 2> gpurun_out/c2s.err || { tail -3 gpurun_out/c2s.err; exit 1; }; python3 -c "import json,sys;d=json.loads(open('gpurun_out/c2s.json').read().strip().splitlines()[-1]);print(sys.argv[1:],round(d['ms_per_step'],1),d['matches'])" $s "$@"; }
for s in 22 24; do
  for w in 64 16 4 1; do run $s GPM_CYC_RANGE_WORK=$w; done
done
