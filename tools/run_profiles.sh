#!/bin/bash
# Runs on the GPU box (via gpurun): bench lines, rocprofv3 kernel traces of the same commands, counter passes — for every
# BASELINE configuration.   usage: tools/run_profiles.sh <tag> [quick]      (quick: cfg2 only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r5}
QUICK=$2
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# ---- the two headline commands, and the kernel trace of each --------------------------------------------------------------
python3 $R/bench.py --gpus 1 --steps 300 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_cfg2_steps20.json 2> $O/bench_cfg2_steps20.err
python3 -c "import json; [print(n, d['value'], d['roofline']['frac'], d['roofline']['us_per_launch']) for n, d in ((n, json.load(open('$O/bench_cfg2%s.json' % n))) for n in ('', '_steps20'))]"
rm -rf $O/stats_cfg2 $O/stats_cfg2_steps20
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg2 -- python3 $R/bench.py --gpus 1 --steps 300 --no-cpu-baseline --min-timed-s 1 > $O/stats_cfg2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg2_steps20 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --min-timed-s 1 > $O/stats_cfg2_steps20.log 2>&1
python3 $R/tools/agent_loop_rate.py cfg2 65536 2> /dev/null | grep "^{" > $O/agent_loop_cfg2.json
python3 $R/tools/agent_loop_rate.py cfg3 65536 2> /dev/null | grep "^{" > $O/agent_loop_cfg3.json
cat $O/agent_loop_cfg2.json $O/agent_loop_cfg3.json
pass() {  # name, workload, batch, steps per launch, counters...
  n=$1; w=$2; b=$3; s=$4; shift 4
  rm -rf $O/$n
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python3 $R/tools/pmc_traffic.py $w $b $s > $O/$n.log 2>&1
}
counters() {  # workload, batch, steps per launch, suffix, full?
  w=$1; b=$2; s=$3; x=$4
  pass tr_f_$w$x $w $b $s FETCH_SIZE
  pass tr_w_$w$x $w $b $s WRITE_SIZE
  pass sq1_$w$x $w $b $s SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
  if [ "$5" = "full" ]; then
    pass ea_$w$x $w $b $s TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum
    pass sq2_$w$x $w $b $s SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT
    pass sq3_$w$x $w $b $s SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT
  fi
}
counters cfg2 65536 128 "" full
counters cfg2 65536 20 _s20
if [ "$QUICK" != "quick" ]; then
for cfg in "cfg1 65536" "cfg3 65536" "cfg4 16384" "cfg5 32768"; do
  set -- $cfg
  python3 $R/bench.py --workload $1 --batch $2 --steps 200 --no-cpu-baseline --min-timed-s 2 > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
  rm -rf $O/stats_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$1 -- python3 $R/bench.py --workload $1 --batch $2 --steps 200 --no-cpu-baseline --min-timed-s 1 > $O/stats_$1.log 2>&1
  counters $1 $2 128 ""
done
for cfg in "cfg1 4096" "cfg2 4096" "cfg3 4096"; do
  set -- $cfg
  python3 $R/bench.py --workload $1 --batch $2 --steps 200 --no-cpu-baseline --min-timed-s 1 > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
  counters $1 $2 128 _b4096   # (BASELINE's literal batch: the lines of tools/bench_lines.sh then carry traffic / valu too)
done
python3 -c "
import json, glob
for f in sorted(glob.glob('$O/bench_cfg*_*.json')):
    try:
        d = json.load(open(f)); print(f.split('/')[-1], d['value'], d['roofline']['frac'], d['state'])
    except Exception as e: print(f, 'ERR', e)
"
fi
python3 $R/tools/phase_prof.py cfg2 65536 256 > $O/phase_cfg2.txt 2>&1
# ---- small batches: the two-wavefront form against the one-wavefront form (state, rate), and where the pair's cycles go ----------
for n in 2048 4096 8192 12288; do python3 $R/tools/pair_check.py cfg2 cfg3 cfg1 cfg5 --envs $n --reps 3 2>&1 | grep -v amdgpu.ids; done > $O/pair_check.txt
for rw in 1 0; do ORL_PERSIST_RW=$rw python3 $R/tools/pair_prof.py cfg2 4096 256 2>&1 | grep -v amdgpu.ids; done > $O/pair_prof_cfg2_4096.txt
ls $O | head -80
