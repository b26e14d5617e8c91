#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: how much of k_rowstats' / k_stats' run time overlaps a k_persist launch (two-stream runs)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(r["Kernel_Name"].split("<")[0].replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
ev = [e for e in ev if e[0] in ("k_persist", "k_rowstats", "k_stats")]
ev = ev[len(ev) // 2:]  # the timed part
P = [(s, e) for n, s, e in ev if n == "k_persist"]
for name in ("k_rowstats", "k_stats"):
    tot = ov = 0
    for n, s, e in ev:
        if n != name:
            continue
        tot += e - s
        for ps, pe in P:
            lo, hi = max(s, ps), min(e, pe)
            if hi > lo:
                ov += hi - lo
    print("%s: total %.1f ms, overlapped with k_persist %.1f ms (%.0f %%); mean duration %.1f us" % (name, tot / 1e6, ov / 1e6, 100.0 * ov / max(tot, 1), tot / 1e3 / max(1, sum(1 for n, _, _ in ev if n == name))))
span = max(e for _, _, e in ev) - min(s for _, s, _ in ev)
print("k_persist: total %.1f ms, mean %.1f us; span %.1f ms" % (sum(e - s for s, e in P) / 1e6, sum(e - s for s, e in P) / 1e3 / len(P), span / 1e6))
