#!/bin/bash
# SQ counters of the headline search (one rocprofv3 pass) plus a kernel-trace pass, accounting off.
set -u
TAG=${1:-r2}
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPM_ACCOUNTING=0
BENCH="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-search"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d "$OUT/sq" -o t -- $BENCH > "$OUT/sq.json" 2> "$OUT/sq.err" || exit 1
echo "sq done"
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --output-format csv -d "$OUT/sq2" -o t -- $BENCH > "$OUT/sq2.json" 2> "$OUT/sq2.err" || exit 1
echo "sq2 done"
