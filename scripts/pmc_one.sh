#!/bin/bash
# One rocprofv3 counter pass of the headline search with one chunk in flight; sums per kernel name.
# usage (on the GPU box): bash scripts/pmc_one.sh <tag> [ENV=VALUE ...]   -> gpurun_out/pmc_<tag>.txt
set -u
TAG=$1; shift
O=gpurun_out/pmc_$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for kv in "$@"; do export "$kv"; done
export GPM_CLIQUE_INFLIGHT=1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$O" -o t -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-search > "$O/out.json" 2> "$O/err.txt" || tail -2 "$O/err.txt"
python3 - "$O" <<'PY' > gpurun_out/pmc_$TAG.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_INSTS_VALU': calls[k] += 1
for k in sorted(acc, key=lambda k: -acc[k]['SQ_BUSY_CYCLES']):
    if 'clq' in k or 'clique' in k:
        print(k, calls[k], ' '.join(f"{c}={v:.4g}" for c, v in sorted(acc[k].items())))
PY
cat gpurun_out/pmc_$TAG.txt
