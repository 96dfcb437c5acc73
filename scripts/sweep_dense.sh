#!/bin/bash
# Columns of the hub matrix probed without asking the summary first.
out=gpurun_out/dense
mkdir -p $out
for dn in 0 1024 4096 8192 16384 32768 65536; do
  GPM_HUB_DENSE=$dn timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-extra-search --no-cpu-baseline > $out/d$dn.json 2> $out/d$dn.err || { echo failed $dn; tail -3 $out/d$dn.err; continue; }
  python - $out/d$dn.json $dn <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("dense", sys.argv[2], "ms", round(d["ms_per_step"], 2), "ok", d["count_ok"], {k: round(v["ms_per_step"], 1) for k, v in d["kernel_families_rank0"].items()}, flush=True)
PY
done
