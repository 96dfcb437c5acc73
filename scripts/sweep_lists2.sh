#!/bin/bash
out=gpurun_out/lists2
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
for k in 0 1 2 3; do run kappa$k GPM_CLQ_KAPPA=$k; done
for k in 1 2 3; do run class5_kappa$k GPM_CLQ_LISTS_MAX_CLASS=5 GPM_CLQ_KAPPA=$k; done
run class3_kappa2 GPM_CLQ_LISTS_MAX_CLASS=3 GPM_CLQ_KAPPA=2
run kappa2_rb2 GPM_CLQ_KAPPA=2 GPM_CLQ_BUILD_RB=2
run kappa2_inflight1 GPM_CLQ_KAPPA=2 GPM_CLIQUE_INFLIGHT=1
