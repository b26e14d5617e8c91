#!/bin/bash
# The bench lines of every BASELINE configuration, run AFTER tools/collect_profiles.py has written profiles/traffic_<cfg>.json for
# these sources, so that each line carries the measured traffic / issue counters.  usage (GPU box): tools/bench_lines.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
python3 $R/bench.py --gpus 1 --steps 300 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_cfg2_steps20.json 2> $O/bench_cfg2_steps20.err
for cfg in "cfg1 65536" "cfg3 65536" "cfg4 16384" "cfg5 32768" "cfg1 4096" "cfg2 4096" "cfg3 4096"; do
  set -- $cfg
  python3 $R/bench.py --workload $1 --batch $2 --steps 200 --no-cpu-baseline --min-timed-s 2 > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
done
python3 -c "
import json, glob
for f in sorted(glob.glob('$O/bench_cfg*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f.split('/')[-1], d['value'], 'frac', r['frac'], 'traffic_frac', r['traffic_frac'], 'scaled', r['traffic_scaled_from_steps'], 'valu', r['valu'] and (r['valu']['frac'], r['valu']['valu_insts_per_wavefront_step']))
    except Exception as e: print(f, 'ERR', e)
"
