#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel over the last N dispatches
(default 10 = the measured launches of tools/pmc_traffic.py; 20 for the stand-alone k_policy launches)."""
import csv
import glob
import sys
from collections import defaultdict


def main(paths, last=10):
    for path in paths:
        for f in sorted(glob.glob(path + "/**/*counter_collection.csv", recursive=True)):
            rows = list(csv.DictReader(open(f)))
            per = defaultdict(lambda: defaultdict(list))
            for r in rows:
                per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            print("==", f)
            for k, cs in per.items():
                if not (k.startswith("void k_") or k.startswith("k_")):
                    continue
                print("  %-50s" % k[:50], " ".join("%s=%.4g" % (c, sum(v[-last:]) / len(v[-last:])) for c, v in sorted(cs.items())))


if __name__ == "__main__":
    main(sys.argv[1:])
