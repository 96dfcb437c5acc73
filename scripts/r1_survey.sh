#!/bin/bash
# Full-size runs of the five configurations, a probe-bias sweep and a fresh kernel trace.
set -u
O=gpurun_out/survey
mkdir -p $O
for c in 1 3 4; do
  timeout -k 10 200 python scripts/run_config.py $c > $O/cfg$c.json 2> $O/cfg$c.err; echo "cfg $c rc=$?"; cat $O/cfg$c.json
done
for b in 0 64 256 1000000; do
  GPM_CLIQUE_PROBE_BIAS=$b timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bias$b.json 2> $O/bias$b.err; echo "bias $b rc=$?"
  python -c "import json;d=json.load(open('$O/bias$b.json'));print($b, d['ms_per_step'], d['matches'])"
done
timeout -k 10 280 python scripts/run_config.py 2 18 > $O/cfg2_18.json 2> $O/cfg2_18.err; echo "cfg 2@18 rc=$?"; cat $O/cfg2_18.json
timeout -k 10 280 python scripts/run_config.py 5 16 > $O/cfg5_16.json 2> $O/cfg5_16.err; echo "cfg 5@16 rc=$?"; cat $O/cfg5_16.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/trace.json 2> $O/trace.err; echo "trace rc=$?"
head -12 $O/trace/*/t_kernel_stats.csv 2>/dev/null || find $O/trace -name '*kernel_stats*'
