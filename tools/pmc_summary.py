#!/usr/bin/env python3
"""tools/pmc_summary.py — summarise a rocprofv3 --pmc counter_collection.csv per kernel (sum over dispatches)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("void ", "").replace("mlkem::", "")[:34]
    agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    w = v.get("SQ_WAVES", 0) or 1
    print(k, "waves=%d" % w, " ".join("%s/wave=%.1f" % (a.replace("SQ_", ""), b / w) for a, b in sorted(v.items()) if a not in ("SQ_WAVES",)))
