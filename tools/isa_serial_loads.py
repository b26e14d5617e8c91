#!/usr/bin/env python3
"""Finds SERIALISED memory requests in the gfx950 code of a built library: runs of `load - s_waitcnt vmcnt(0) - load - s_waitcnt vmcnt(0) ...`
where the source asks for a batch of independent values.  (Round 6: the register allocator had turned the rebuild scan's batches of 8
release times into 8 dependent round trips each — 44 per scan instead of 6; it cost cfg2 7 %, cfg4 15 %.)

    python3 tools/isa_serial_loads.py <lib.so> [kernel name filter]        prints, per kernel, the runs with >= 3 loads and >= 3 full waits
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib, td):
    tmp = os.path.join(td, os.path.basename(lib))
    os.symlink(os.path.abspath(lib), tmp)
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", tmp], capture_output=True, check=True, cwd=td)
    return sorted(glob.glob(tmp + ".*gfx950"))


def analyse(co, flt):
    lines = subprocess.run([LLVM + "/llvm-objdump", "-d", co], capture_output=True, text=True).stdout.splitlines()
    funcs = [(i, l) for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <", l)]
    for fi, (start, name) in enumerate(funcs):
        if flt and flt not in name:
            continue
        end = funcs[fi + 1][0] if fi + 1 < len(funcs) else len(lines)
        code = [l.split("//")[0].strip() for l in lines[start + 1:end]]
        ev = sorted([(i, "L") for i, l in enumerate(code) if re.match(r"(global|buffer|flat|scratch)_load", l)] +
                    [(i, "W") for i, l in enumerate(code) if l.startswith("s_waitcnt") and "vmcnt(0)" in l])
        runs, cur = [], []
        for i, t in ev:
            if cur and i - cur[-1][0] < 14:
                cur.append((i, t))
            else:
                if cur:
                    runs.append(cur)
                cur = [(i, t)] if t == "L" else []
        if cur:
            runs.append(cur)
        out = []
        for r in runs:
            kinds = "".join(t for _, t in r)
            if kinds.count("LW") >= 3:  # three or more requests each followed by a full wait
                out.append("line %d: %s" % (r[0][0], kinds))
        short = re.sub(r"^[0-9a-f]+ <|>:$", "", name)[:70]
        print("%-70s %6d instructions, %d serialised runs" % (short, len(code), len(out)))
        for o in out:
            print("      " + o)


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as td:
        for co in code_objects(sys.argv[1], td):
            analyse(co, flt)
