#!/bin/bash
# Round 6: the evidence behind DESIGN.md 4.6 (rows-deferred forms against the default form) in one call, into gpurun_out/<tag>/:
#   parity (tools/rd_check.py: every env), kernel durations against the launch length (default and RD), counter passes of the RD
#   kernels, the replay's phase profile, bench lines of both forms, overlap of the replay with the other half's loop.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r6_rd}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
{
  python3 tools/rd_check.py cfg2 4096 400 2>&1 | grep -v amdgpu.ids
  python3 tools/rd_check.py cfg3 2048 300 2>&1 | tail -1
  python3 tools/rd_check.py cfg1 2048 300 2>&1 | tail -1
  python3 tools/rd_check.py cfg2 16384 700 2>&1 | tail -1
} > $O/parity.txt
{
  echo "# default form (k_persist with the row phase in the loop + k_stats)"
  bash tools/launch_sweep.sh ${TAG}_sw_base "5 20 80 128" | grep "S="
  echo "# rows-deferred form 7 (k_persist without the row phase + k_rowstats + k_stats)"
  ORL_PERSIST_VARIANT=7 bash tools/launch_sweep.sh ${TAG}_sw_rd "2 5 10 20 40 80 128" | grep "S="
  echo "# cfg3 / cfg1, 20-step launches: default, then form 7"
  bash tools/launch_sweep.sh ${TAG}_sw_c3b "20" cfg3 | grep "S="; ORL_PERSIST_VARIANT=7 bash tools/launch_sweep.sh ${TAG}_sw_c3 "20" cfg3 | grep "S="
  bash tools/launch_sweep.sh ${TAG}_sw_c1b "20" cfg1 | grep "S="; ORL_PERSIST_VARIANT=7 bash tools/launch_sweep.sh ${TAG}_sw_c1 "20" cfg1 | grep "S="
} > $O/launch_sweep.txt 2>&1
bash tools/pmc_rd.sh ${TAG}_pmc 20 > $O/pmc_steps20.txt 2>&1
python3 tools/ab_run.py ${TAG}_ab "base20||--steps 20" "rd7_20|ORL_PERSIST_VARIANT=7|--steps 20" "base300||--steps 300" "rd7_300|ORL_PERSIST_VARIANT=7|--steps 300" \
  "cfg3_base300||--steps 300 --workload cfg3" "cfg3_rd8_300|ORL_PERSIST_VARIANT=8|--steps 300 --workload cfg3" \
  "cfg1_base300||--steps 300 --workload cfg1" "cfg1_rd8_300|ORL_PERSIST_VARIANT=8|--steps 300 --workload cfg1" > $O/bench_ab.txt 2>&1
cp $R/gpurun_out/${TAG}_ab/*.json $O/ 2>/dev/null
if [ -f $R/optical_rl_gym_amd/liborlgpu_exp.so ]; then
  ORL_JIT_SPEC=0 ORL_HIPCC_EXTRA=-DORL_RS_PROF ORL_LIB_VARIANT=exp python3 tools/rs_prof.py 20 20 2>&1 | grep -v amdgpu.ids > $O/rs_prof_steps20.txt
fi
cat $O/parity.txt $O/launch_sweep.txt $O/bench_ab.txt
