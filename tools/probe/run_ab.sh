mkdir -p gpurun_out/r2m
r() { python3 -c "import json; d=json.load(open('gpurun_out/r2m/b.json')); print('$1', d['value'], d['ms_per_step'], d['config']['step_kernels'])"; }
for cfg in "cfg5 32768" "cfg5 65536"; do set -- $cfg; python bench.py --workload $1 --batch $2 --steps 300 --no-cpu-baseline > gpurun_out/r2m/b.json 2>> gpurun_out/r2m/err.txt; r "$1 $2"; done
ORL_PERSIST_VARIANT=0 python bench.py --workload cfg5 --batch 32768 --steps 300 --no-cpu-baseline > gpurun_out/r2m/b.json 2>> gpurun_out/r2m/err.txt; r "cfg5 form0"
ORL_PERSIST_VARIANT=0 python bench.py --steps 300 --no-cpu-baseline > gpurun_out/r2m/b.json 2>> gpurun_out/r2m/err.txt; r "cfg2 form0"
timeout 900 python -m pytest tests -m gpu -x -q -k "persist or full_size or every_env or serial or corner" 2>&1 | tail -3
