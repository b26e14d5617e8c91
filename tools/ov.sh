R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ov1
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
export ORL_PERSIST_VARIANT=7
for v in 128 104; do
  if [ $v = 104 ]; then export ORL_LIB_VARIANT=exp ORL_HIPCC_EXTRA=-DORL_PERSIST_VGPR=52 ORL_SPEC_PF_WAVES=3; fi
  rm -rf $O/t$v
  rocprofv3 --kernel-trace --output-format csv -d $O/t$v -- python3 $R/bench.py --gpus 1 --steps 300 --no-cpu-baseline --min-timed-s 1 > $O/t$v.log 2>&1
  echo "== $v"; python3 $R/tools/overlap_trace.py $O/t$v
done
