#!/bin/bash
# Runs on the GPU box (via gpurun): final bench line, rocprofv3 stats of the same command, other configs, PMC traffic.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --gpus 1 --steps 300 --warmup 1500 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
tail -c 2500 $O/bench_cfg2.json
rm -rf $O/stats $O/stats1 $O/tr_f $O/tr_w $O/sq $O/ea
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --gpus 1 --steps 300 --warmup 1500 --no-cpu-baseline > $O/stats.log 2>&1
export ORL_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 $R/bench.py --gpus 1 --steps 300 --warmup 1500 --no-cpu-baseline > $O/stats1.log 2>&1
unset ORL_STREAMS
for cfg in "cfg1 4096" "cfg3 4096" "cfg4 16384" "cfg4n 16384" "cfg5 32768" "cfg2 4096" "cfg1 32768" "cfg3 32768" "cfg2 32768"; do
  set -- $cfg
  python3 $R/bench.py --workload $1 --batch $2 --steps 200 --warmup 1500 --no-cpu-baseline > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
  python3 -c "import json,sys; d=json.load(open('$O/bench_$1_$2.json')); print('$1 B=$2', d['value'], {k:v['us_per_launch'] for k,v in d['roofline_by_kernel'].items()}, d['state'])"
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/tr_f -- python3 $R/tools/pmc_traffic.py > $O/tr_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/tr_w -- python3 $R/tools/pmc_traffic.py > $O/tr_w.log 2>&1
export ORL_STREAMS=1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/sq -- python3 $R/tools/pmc_traffic.py > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $O/ea -- python3 $R/tools/pmc_traffic.py > $O/ea.log 2>&1
ls $O
