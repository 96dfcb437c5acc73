#!/bin/bash
# Cost of finer root chunks on one GPU (what every rank of an N-GPU run sees: bench.py asks for 64 chunks
# per rank), and the fixed cost of one search on a graph too small to matter.
out=gpurun_out/chunks
mkdir -p $out
for c in 64 128 256 512; do
  GPM_BENCH_CHUNKS=$c timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-extra-search --no-cpu-baseline \
      > $out/chunks_$c.json 2> $out/chunks_$c.err || exit 1
  python - $out/chunks_$c.json $c <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("chunks", sys.argv[2], "ms_per_step", round(d["ms_per_step"], 2), "taken", d["chunks_per_step_by_rank"])
PY
done
for s in 12 16 20; do
  timeout -k 10 200 python bench.py --scale $s --steps 20 --warmup 3 --no-extra-search --no-cpu-baseline \
      > $out/scale_$s.json 2> $out/scale_$s.err || exit 1
  python - $out/scale_$s.json $s <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("scale", sys.argv[2], "ms_per_step", round(d["ms_per_step"], 3), "count_ok", d.get("count_ok"))
PY
done
