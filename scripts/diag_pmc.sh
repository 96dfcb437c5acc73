#!/bin/bash
# Memory-path counters of the headline search, one rocprofv3 pass per counter group.
set -u
OUT=gpurun_out/diag
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-search"
timeout -s KILL 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/a" -o t -- $BENCH > "$OUT/a.json" 2> "$OUT/a.err" || exit 1
echo "a done"
timeout -s KILL 300 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum GRBM_GUI_ACTIVE TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --output-format csv -d "$OUT/b" -o t -- $BENCH > "$OUT/b.json" 2> "$OUT/b.err" || exit 1
echo "b done"
