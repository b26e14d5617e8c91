mkdir -p gpurun_out/r2m
r() { python3 -c "import json; d=json.load(open('gpurun_out/r2m/b.json')); print('$1', d['value'], d['ms_per_step'], d['config']['step_kernels'])"; }
for st in 300 20; do python bench.py --steps $st --warmup 5 --no-cpu-baseline > gpurun_out/r2m/b.json 2> gpurun_out/r2m/err.txt; r "cfg2 steps $st"; done
ORL_PERSIST_INNER=1 python bench.py --steps 300 --warmup 5 --no-cpu-baseline > gpurun_out/r2m/b.json 2> gpurun_out/r2m/err.txt; r "cfg2 inner forced steps 300"
for cfg in "cfg3 32768" "cfg4 8192" "cfg5 32768" "cfg1 32768"; do set -- $cfg; python bench.py --workload $1 --batch $2 --steps 300 --no-cpu-baseline > gpurun_out/r2m/b.json 2>> gpurun_out/r2m/err.txt; r "$1 $2"; done
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2m/pytest.txt 2>&1; tail -c 600 gpurun_out/r2m/pytest.txt
