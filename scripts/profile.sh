#!/bin/bash
# Profile the benchmark with rocprofv3: one kernel-trace pass for the per-kernel
# time table and separate counter passes (each block's slots are limited, and
# counters must not be mixed with the trace domains).  Results go to
# gpurun_out/prof_<tag>/ ; scripts/summarize_profile.py turns them into the
# files kept under profiles/.
set -u
TAG=${1:-r3}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-search"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $BENCH > "$OUT/trace.json" 2> "$OUT/trace.err" || exit 1
echo "trace done"
# the same with one chunk in flight: launches do not overlap, so a family's time is its own
GPM_CLIQUE_INFLIGHT=1 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_alone" -o t -- $BENCH > "$OUT/trace_alone.json" 2> "$OUT/trace_alone.err" || exit 1
echo "trace (one chunk in flight) done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o t -- $BENCH > "$OUT/fetch.json" 2> "$OUT/fetch.err" || exit 1
echo "fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o t -- $BENCH > "$OUT/write.json" 2> "$OUT/write.err" || exit 1
echo "write done"
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d "$OUT/sq" -o t -- $BENCH > "$OUT/sq.json" 2> "$OUT/sq.err" || exit 1
echo "sq done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib_fetch" -o t -- scripts/calib_fetch > "$OUT/calib.txt" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/calib_write" -o t -- scripts/calib_fetch >> "$OUT/calib.txt" 2>&1 || exit 1
echo "calib done"
ls "$OUT"/*
