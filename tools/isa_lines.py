#!/usr/bin/env python3
"""Where a kernel's instructions come from: static instruction counts per source function / line of one kernel of the
gfx950 assembly hipcc leaves with `--save-temps -gline-tables-only` (the .loc directives name the innermost inlined line).

    tools/isa_lines.py <file.s> <kernel name filter> [--lines N] [--func NAME]

Static counts: loops and branches are not weighted (a rare path counts like a hot one); use it to see what a source
construct costs in instructions, next to the dynamic per-wavefront counters of rocprofv3 (SQ_INSTS_VALU ...)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def functions_of(path):
    """[(first line, name)] of the function definitions in a source file (crude: a line that starts a definition)."""
    out = []
    try:
        lines = open(path).read().split("\n")
    except OSError:
        return out
    pat = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:static\s+)?(?:__device__|__global__|inline|static)[^;{]*?\b([A-Za-z_][A-Za-z_0-9]*)\s*\(")
    for i, ln in enumerate(lines, 1):
        m = pat.match(ln)
        if m and not ln.strip().startswith("//"):
            out.append((i, m.group(1)))
    return out


def kind(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_branch", "s_cbranch", "s_setpc", "s_sleep")):
            return "sctl"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    src, filt = sys.argv[1], sys.argv[2]
    n_lines = int(sys.argv[sys.argv.index("--lines") + 1]) if "--lines" in sys.argv else 25
    only = sys.argv[sys.argv.index("--func") + 1] if "--func" in sys.argv else None
    files = {}
    cur = None
    in_k = False
    per_line = collections.defaultdict(collections.Counter)
    ops = collections.defaultdict(collections.Counter)
    for ln in open(src):
        s = ln.split(";")[0].strip()
        m = re.match(r"\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", s)
        if m:
            files[int(m.group(1))] = os.path.join(m.group(2), m.group(3)) if m.group(3) else m.group(2)
            continue
        if s.endswith(":") and not s.startswith("."):
            name = s[:-1]
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() if name.startswith("_Z") else name
            in_k = filt in dem and "k_" in dem
            if in_k:
                print("kernel:", dem.split("(")[0])
            continue
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
            in_k = False
        if not in_k:
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((".", ";")) or s.endswith(":"):
            continue
        op = s.split()[0]
        k = kind(op)
        per_line[cur][k] += 1
        ops[k][op] += 1
    fn_tabs = {fid: functions_of(p) for fid, p in files.items()}

    def func_of(loc):
        if loc is None:
            return "?"
        fid, line = loc
        best = "?"
        for first, name in fn_tabs.get(fid, []):
            if first <= line:
                best = name
            else:
                break
        return "%s:%s" % (os.path.basename(files.get(fid, "?")), best)

    per_fn = collections.defaultdict(collections.Counter)
    for loc, c in per_line.items():
        per_fn[func_of(loc)].update(c)
    tot = collections.Counter()
    for c in per_fn.values():
        tot.update(c)
    print("total:", dict(tot))
    print("%-50s %6s %6s %5s %5s" % ("function", "valu", "salu", "lds", "vmem"))
    for fn, c in sorted(per_fn.items(), key=lambda kv: -kv[1]["valu"]):
        print("%-50s %6d %6d %5d %5d" % (fn, c["valu"], c["salu"], c["lds"], c["vmem"]))
    print("\nVALU opcodes:", ", ".join("%s %d" % kv for kv in ops["valu"].most_common(22)))
    print("\nhottest lines%s:" % (" of " + only if only else ""))
    rows = [(loc, c) for loc, c in per_line.items() if loc and (only is None or func_of(loc).endswith(":" + only))]
    for loc, c in sorted(rows, key=lambda kv: -kv[1]["valu"])[:n_lines]:
        path = files.get(loc[0], "?")
        try:
            text = open(path).read().split("\n")[loc[1] - 1].strip()[:110]
        except Exception:
            text = ""
        print("%5d valu %4d salu %3d lds  %s:%d  %s" % (c["valu"], c["salu"], c["lds"], os.path.basename(path), loc[1], text))


if __name__ == "__main__":
    main()
