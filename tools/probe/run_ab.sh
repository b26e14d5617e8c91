mkdir -p gpurun_out/r2m
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2m/pytest.txt 2>&1; tail -c 1500 gpurun_out/r2m/pytest.txt
for st in 300 20; do python bench.py --steps $st --warmup 5 --no-cpu-baseline > gpurun_out/r2m/b.json 2> gpurun_out/r2m/err.txt; python3 -c "import json; d=json.load(open('gpurun_out/r2m/b.json')); print('cfg2 default steps $st', d['value'], d['ms_per_step'], d['config']['step_kernels'])"; done
