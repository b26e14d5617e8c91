#!/bin/bash
# HBM traffic + request counters of the split pipeline's kernels (cfg2, one stream); runs on the GPU box
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_split
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ORL_STEP_IMPL=1
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU"; do
  d=$O/$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/tools/pmc_traffic.py > $d.log 2>&1
  python3 $R/tools/pmc_summary.py $d | grep -v "k_ctrl_a<\|k_ctrl_b1\|k_policy<\|k_seed\|k_reset\|k_init\|k_totals"
done
