# A/B of the persistent kernel's issue-priority rotation (ORL_PERSIST_FAIR: 0 = the arbiter's oldest-first, n = rotation with
# 2^n x 10 ns per level) over the BASELINE workloads, 20-step and 300-step runs:  bash tools/fair_ab.sh  -> gpurun_out/fair/
mkdir -p gpurun_out/fair
run() {  # workload batch fair
  for s in 20 300; do
    ORL_PERSIST_FAIR=$3 python3 bench.py --workload $1 --batch $2 --steps $s --warmup 5 --no-cpu-baseline --min-timed-s 2 2>/dev/null | tail -1 > gpurun_out/fair/b_$1_f$3_s${s}.json
    python3 -c "
import json
d=json.load(open('gpurun_out/fair/b_$1_f$3_s${s}.json'))
print('$1 B=$2 fair=$3 steps=$s value %.4g ms_per_step %.4f' % (d['value'], d['ms_per_step']))"
  done
}
for f in 0 11 0 11; do run cfg2 65536 $f; done
for w in "cfg1 65536" "cfg3 65536" "cfg4 16384" "cfg5 32768"; do
  for f in 0 11 13; do run $w $f; done
done
