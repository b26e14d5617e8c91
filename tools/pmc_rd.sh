#!/bin/bash
# counters of the rows-deferred form's kernels (k_persist form 7, k_rowstats): tools/pmc_rd.sh <tag> [steps per launch] [workload] [batch]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-rdpmc}
S=${2:-20}; W=${3:-cfg2}; B=${4:-65536}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ORL_PERSIST_VARIANT=${ORL_PERSIST_VARIANT-7}
pass() { n=$1; shift; rm -rf $O/$n; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python3 $R/tools/pmc_traffic.py $W $B $S > $O/$n.log 2>&1; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES
pass trf FETCH_SIZE
pass trw WRITE_SIZE
python3 $R/tools/pmc_summary.py $O/sq1 $O/sq2 $O/trf $O/trw | grep -v "k_calib\|k_policy\|k_seed\|k_init\|k_reset\|k_step"
