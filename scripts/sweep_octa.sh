run() { env "$@" timeout -k 10 300 python3 bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-search > gpurun_out/c5.json 2> gpurun_out/c5.err || { tail -3 gpurun_out/c5.err; exit 1; }; python3 -c "import json,sys;d=json.loads(open('gpurun_out/c5.json').read().strip().splitlines()[-1]);print(sys.argv[1:],round(d['ms_per_step'],2),d['count_ok'])" "$@"; }
run GPM_OCTA_BLOCKS=256
run GPM_OCTA_BLOCKS=512
run GPM_OCTA_BLOCKS=1024
run GPM_OCTA_BLOCKS=1280
run GPM_OCTA_BLOCKS=1536 GPM_OCTA_MEM_GB=32
run GPM_OCTA_BLOCKS=2048 GPM_OCTA_MEM_GB=32
