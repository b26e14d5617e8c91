#!/usr/bin/env python3
"""A/B runs of bench.py on the GPU box: each variant is a label, extra environment and bench arguments.

    tools/ab_run.py <tag> '<label>|ENV=VAL;ENV2=VAL2|--steps 20 --workload cfg2' ...
Results: gpurun_out/<tag>/<label>.json and a summary line per variant."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out", sys.argv[1])
os.makedirs(out, exist_ok=True)
for spec in sys.argv[2:]:
    label, envs, args = (spec.split("|") + ["", ""])[:3]
    env = dict(os.environ)
    for kv in envs.split(";"):
        if kv.strip():
            k, v = kv.split("=", 1)
            env[k.strip()] = v
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline", "--min-timed-s", "1.5"] + args.split()
    p = subprocess.run(cmd, env=env, capture_output=True, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        print(label, "FAILED", p.stderr[-600:], flush=True)
        continue
    open(os.path.join(out, label + ".json"), "w").write(lines[-1] + "\n")
    d = json.loads(lines[-1])
    print(label, d["value"], d["ms_per_step"], d["roofline"]["us_per_launch"], d["timing"]["block_s_min"], d["timing"]["block_s_max"], flush=True)
