#!/bin/bash
# Runs on the GPU box (via gpurun): bench lines, rocprofv3 kernel stats of the same command, counter passes.
# usage: tools/run_profiles.sh <tag> [quick]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r2}
QUICK=$2
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --gpus 1 --steps 300 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
tail -c 1500 $O/bench_cfg2.json
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --min-timed-s 2 > $O/bench_cfg2_steps20.json 2> $O/bench_cfg2_steps20.err
python3 -c "import json; d=json.load(open('$O/bench_cfg2_steps20.json')); print('steps20', d['value'], d['timing'], d['roofline']['frac'])"
rm -rf $O/stats $O/stats20
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --gpus 1 --steps 300 --no-cpu-baseline --min-timed-s 1 > $O/stats.log 2>&1
# the driver's own invocation (20-step blocks: one launch each)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats20 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --min-timed-s 1 > $O/stats20.log 2>&1
python3 $R/tools/agent_loop_rate.py cfg2 65536 2> /dev/null | grep "^{" > $O/agent_loop_cfg2.json
python3 $R/tools/agent_loop_rate.py cfg3 65536 2> /dev/null | grep "^{" > $O/agent_loop_cfg3.json
cat $O/agent_loop_cfg2.json $O/agent_loop_cfg3.json
if [ "$QUICK" != "quick" ]; then
for cfg in "cfg1 4096" "cfg3 4096" "cfg4 16384" "cfg5 32768" "cfg2 4096" "cfg1 65536" "cfg3 65536"; do
  set -- $cfg
  python3 $R/bench.py --workload $1 --batch $2 --steps 200 --no-cpu-baseline --min-timed-s 1 > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
  python3 -c "import json; d=json.load(open('$O/bench_$1_$2.json')); print('$1 B=$2', d['value'], d['roofline']['frac'], d['state'])"
done
fi
pass() {  # name, counters...
  n=$1; shift
  rm -rf $O/$n
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python3 $R/tools/pmc_traffic.py cfg2 65536 > $O/$n.log 2>&1
}
pass tr_f FETCH_SIZE
pass tr_w WRITE_SIZE
pass ea TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT
pass sq3 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT
if [ "$QUICK" != "quick" ]; then
pass hit TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
fi
ls $O
python3 $R/tools/phase_prof.py cfg2 65536 256 > $O/phase_cfg2.txt 2>&1
