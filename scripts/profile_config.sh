#!/bin/bash
# rocprofv3 passes of one benchmark configuration other than the headline one: kernel trace, LDS/VALU
# counters, HBM read and write sizes (separate runs: counter blocks are limited and must not be mixed with
# the trace).  Results go to gpurun_out/prof_<tag>_config<N>/ ; scripts/summarize_profile.py <tag>_config<N>
# turns them into profiles/<tag>_config<N>_* and profiles/config<N>_hbm_traffic.json.
set -u
CFG=${1:?configuration number}; TAG=${2:-r3}; NAME=${3:-config$CFG}; shift; shift; shift  # further arguments go to bench.py
O=gpurun_out/prof_${TAG}_$NAME
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-extra-search $*"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > $O/trace.json 2> $O/trace.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/sq -o t -- $B > $O/sq.json 2> $O/sq.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o t -- $B > $O/fetch.json 2> $O/fetch.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o t -- $B > $O/write.json 2> $O/write.err || exit 1
echo "$NAME profiled"
