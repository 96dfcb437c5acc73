#!/bin/bash
# sweep_env.sh CONFIG VAR v1 v2 ... : bench.py --config CONFIG once per value of an environment switch
CFG=$1; VAR=$2; shift 2
mkdir -p gpurun_out/sweep
for x in "$@"; do
  env $VAR=$x timeout -k 10 200 python bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/sweep/${VAR}_$x.json 2> gpurun_out/sweep/${VAR}_$x.err || exit 1
  python - "$VAR" "$x" <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/sweep/{sys.argv[1]}_{sys.argv[2]}.json'))
print(sys.argv[1], sys.argv[2], d['matches'], round(d['ms_per_step'],1), round(d['roofline']['frac'],3), {k:round(v['ms_per_step'],1) for k,v in d['kernel_families_rank0'].items()}, flush=True)
PY
done
