mkdir -p gpurun_out/r2m
for st in 300 20; do python bench.py --steps $st --warmup 5 --no-cpu-baseline > gpurun_out/r2m/b.json 2> gpurun_out/r2m/err.txt; python3 -c "import json; d=json.load(open('gpurun_out/r2m/b.json')); print('cfg2 default steps $st', d['value'], d['ms_per_step'], d['config']['step_kernels'])"; done
for cfg in "cfg3 32768" "cfg4 8192" "cfg5 32768" "cfg1 32768"; do set -- $cfg; python bench.py --workload $1 --batch $2 --steps 300 --no-cpu-baseline > gpurun_out/r2m/b.json 2>> gpurun_out/r2m/err.txt; python3 -c "import json; d=json.load(open('gpurun_out/r2m/b.json')); print('$1 B=$2', d['value'], d['ms_per_step'], d['config']['step_kernels'])"; done
timeout 900 python -m pytest tests -m gpu -x -q -k "persist or full_size" 2>&1 | tail -3
