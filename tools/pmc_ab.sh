R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in 0 4; do
export ORL_PERSIST_VARIANT=$v
O=$R/gpurun_out/r2g_v$v
mkdir -p $O
pass() { n=$1; shift; rm -rf $O/$n; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python3 $R/tools/pmc_traffic.py cfg2 65536 > $O/$n.log 2>&1; }
pass tr_f FETCH_SIZE
pass tr_w WRITE_SIZE
pass ea TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
done
unset ORL_PERSIST_VARIANT
for c in 32 128 256; do ORL_PERSIST_CHUNK=$c python3 $R/bench.py --steps 512 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('chunk $c', d['value'], d['ms_per_step'])"; done
