#!/bin/bash
# Sweep of the list-walking row builder: waves per workgroup, the per-row switch to hub-matrix gathers
# (kappa) and the largest size class it serves.  One bench line per setting under gpurun_out/lists/.
out=gpurun_out/lists
mkdir -p $out
run() {
  tag=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-extra-search --no-cpu-baseline \
      > $out/$tag.json 2> $out/$tag.err || { echo "$tag failed"; tail -3 $out/$tag.err; exit 1; }
  python - $out/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
f = d["kernel_families_rank0"]
print(sys.argv[2], "ms", round(d["ms_per_step"], 2), "ok", d["count_ok"], {k: round(v["ms_per_step"], 1) for k, v in f.items()}, flush=True)
PY
}
run base X=1
run waves4 GPM_CLQ_LISTS_WAVES=4
for k in 2 4 8 16 32; do run kappa$k GPM_CLQ_KAPPA=$k; done
for c in 3 4 5; do
  run class${c} GPM_CLQ_LISTS_MAX_CLASS=$c
  run class${c}_kappa4 GPM_CLQ_LISTS_MAX_CLASS=$c GPM_CLQ_KAPPA=4
  run class${c}_kappa16 GPM_CLQ_LISTS_MAX_CLASS=$c GPM_CLQ_KAPPA=16
done
