mkdir -p gpurun_out/r2m
./tools/probe/lds_occ > gpurun_out/r2m/lds_occ.txt 2>&1; head -40 gpurun_out/r2m/lds_occ.txt
for v in 0 4; do for st in "300 10" "20 5"; do set -- $st; ORL_PERSIST_VARIANT=$v python bench.py --steps $1 --warmup $2 --no-cpu-baseline > gpurun_out/r2m/bench_cfg2_v${v}_$1.json 2> gpurun_out/r2m/err.txt; python3 -c "import json; d=json.load(open('gpurun_out/r2m/bench_cfg2_v${v}_$1.json')); print('cfg2 variant $v steps $1', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done; done
for cfg in "cfg1 32768" "cfg3 32768" "cfg4 8192"; do set -- $cfg; python bench.py --workload $1 --batch $2 --steps 200 --no-cpu-baseline > gpurun_out/r2m/bench_$1_$2.json 2>> gpurun_out/r2m/err.txt; python3 -c "import json; d=json.load(open('gpurun_out/r2m/bench_$1_$2.json')); print('$1 B=$2', d['value'], d['ms_per_step'])"; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
