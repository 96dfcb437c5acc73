#!/bin/bash
# Knob sweep on the kernels with the per-row list/gather row builder: one bench line per setting.
out=gpurun_out/sweep_r2b
mkdir -p $out
run() {
  tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-extra-search --no-cpu-baseline \
      > $out/$tag.json 2> $out/$tag.err || { echo "$tag failed"; tail -3 $out/$tag.err; return; }
  python - $out/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
f = d["kernel_families_rank0"]
print(sys.argv[2], "ms", round(d["ms_per_step"], 2), "ok", d["count_ok"], {k: round(v["ms_per_step"], 1) for k, v in f.items()}, flush=True)
PY
}
run base X=1
run hub17 GPM_HUB_ROWS=131072
run hub19 GPM_HUB_ROWS=524288
run inflight2 GPM_CLIQUE_INFLIGHT=2
run inflight3 GPM_CLIQUE_INFLIGHT=3
run chunks32 GPM_BENCH_CHUNKS=32
run chunks48 GPM_BENCH_CHUNKS=48
run chunks96 GPM_BENCH_CHUNKS=96
run kappa4 GPM_CLQ_KAPPA4=4
run kappa5 GPM_CLQ_KAPPA4=5
run kappa7 GPM_CLQ_KAPPA4=7
run small128 GPM_CLIQUE_SMALL_MAX=128
run small192 GPM_CLIQUE_SMALL_MAX=192
