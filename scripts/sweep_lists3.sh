#!/bin/bash
out=gpurun_out/lists3
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
run default X=1
for k in 5 6 7 9 10; do run kappa4_$k GPM_CLQ_KAPPA4=$k; done
for r in 1 2 4 8 16; do run rb$r GPM_CLQ_BUILD_RB=$r; done
run default_again X=1
