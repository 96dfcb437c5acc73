mkdir -p gpurun_out/r2m
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-search"
run() { name=$1; shift; env "$@" timeout -k 10 120 $B > gpurun_out/r2m/$name.json 2> gpurun_out/r2m/$name.err; echo "$name $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2m/$name.json | head -1)"; }
run base GPM_X=0
run bias8 GPM_CLIQUE_PROBE_BIAS=8
run bias16 GPM_CLIQUE_PROBE_BIAS=16
run rb2 GPM_CLQ_BUILD_RB=2
run rb4 GPM_CLQ_BUILD_RB=4
run lists3 GPM_CLQ_LISTS_MAX_CLASS=3
run lists1 GPM_CLQ_LISTS_MAX_CLASS=1
run inflight2 GPM_CLIQUE_INFLIGHT=2
run inflight3 GPM_CLIQUE_INFLIGHT=3
run chunks32 GPM_BENCH_CHUNKS=32
run chunks128 GPM_BENCH_CHUNKS=128
run hub17 GPM_HUB_ROWS=131072
run hub19 GPM_HUB_ROWS=524288
