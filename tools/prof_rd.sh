R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/rd2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 7 base; do
  if [ $v = base ]; then unset ORL_PERSIST_VARIANT; else export ORL_PERSIST_VARIANT=$v; fi
  for s in 20 128; do
    rm -rf $O/st_${v}_$s
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_${v}_$s -- python3 $R/bench.py --gpus 1 --steps $s --warmup 5 --no-cpu-baseline --min-timed-s 1 > $O/st_${v}_$s.log 2>&1
    f=$(ls $O/st_${v}_$s/*/*kernel_stats.csv | head -1)
    echo "== $v steps $s"; head -8 $f | cut -c1-200
  done
done
