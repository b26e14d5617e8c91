#!/bin/bash
# The 1 -> 8 GPU curve of BASELINE's headline metric on one node, in one command (for the day an 8-GPU node is at hand):
#   weak    cfg2 RMSA NSFNET 65 536 envs per GPU        python bench.py --gpus N
#   weak    cfg5 RMSA Germany50 32 768 envs per GPU     python bench.py --gpus N --workload cfg5 --batch 32768    (N = 8: BASELINE cfg5)
#   strong  cfg2, 65 536 envs in all                    python bench.py --gpus N --scaling strong --batch 65536
# Prints ONE JSON line: {"weak_cfg2": {N: line}, "weak_cfg5": {...}, "strong_cfg2": {...}} with value / per_rank of every run.
# usage: tools/scale_curve.sh [steps per block: 20] [GPU counts: "1 2 4 8"]      (bench.py starts its N ranks itself)
R=$(cd "$(dirname "$0")/.." && pwd)
STEPS=${1:-20}
NS=${2:-"1 2 4 8"}
HAVE=$(python3 - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
OUT=$(mktemp -d)
for N in $NS; do
  if [ "$N" -gt "$HAVE" ]; then echo "skipping N=$N: $HAVE GPU(s) visible" >&2; continue; fi
  python3 $R/bench.py --gpus $N --steps $STEPS --warmup 5 --no-cpu-baseline > $OUT/weak_cfg2_$N.json 2> $OUT/weak_cfg2_$N.err
  python3 $R/bench.py --gpus $N --steps $STEPS --warmup 5 --no-cpu-baseline --workload cfg5 --batch 32768 > $OUT/weak_cfg5_$N.json 2> $OUT/weak_cfg5_$N.err
  python3 $R/bench.py --gpus $N --steps $STEPS --warmup 5 --no-cpu-baseline --scaling strong --batch 65536 > $OUT/strong_cfg2_$N.json 2> $OUT/strong_cfg2_$N.err
done
python3 - $OUT <<'PY'
import glob, json, os, sys
res = {}
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    kind, n = os.path.basename(f)[:-5].rsplit("_", 1)
    lines = [ln for ln in open(f) if ln.startswith("{")]
    if not lines:
        res.setdefault(kind, {})[n] = {"error": open(f[:-5] + ".err").read()[-400:]}
        continue
    d = json.loads(lines[-1])
    res.setdefault(kind, {})[n] = {k: d.get(k) for k in ("value", "unit", "n_gpus", "ms_per_step", "scaling", "per_rank", "config")}
for kind, runs in res.items():
    base = runs.get("1", {}).get("value")
    for n, r in runs.items():
        if base and r.get("value"):
            r["vs_1gpu"] = r["value"] / base
print(json.dumps(res))
PY
