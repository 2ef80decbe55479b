#!/usr/bin/env python3
"""tools/pmc_summary.py — summarise rocprofv3 --pmc counter_collection.csv files per kernel.

  python tools/pmc_summary.py <counter_collection.csv> [more.csv ...] [--json out.json] [--filter k_]

Per kernel (short name): number of dispatches, and for every counter the SUM over dispatches and the value per
dispatch.  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; `hbm_bytes_per_dispatch` applies the unit
(x1024) and, for FETCH_SIZE, lists both the raw value and the x2 correction MI355X_MICROARCH.md prescribes for
wide coalesced streaming reads (our loads are 4-16 B per lane, so the truth lies between the two)."""
import collections
import csv
import json
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else ""
if out_json in args:
    args.remove(out_json)
if flt in args:
    args.remove(flt)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for path in args:
    for r in csv.DictReader(open(path)):
        short = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mlkem::", "")[:40]
        agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[short].add((path, r["Dispatch_Id"]))
res = {}
for k in sorted(agg):
    if flt and flt not in k:
        continue
    passes = max(1, len({p for p, _ in disp[k]}))
    n = len(disp[k]) / passes
    row = {"dispatches": n, "per_dispatch": {c: v / n for c, v in agg[k].items()}}
    pd = row["per_dispatch"]
    if "FETCH_SIZE" in pd or "WRITE_SIZE" in pd:
        row["hbm_bytes_per_dispatch"] = {"read_raw": pd.get("FETCH_SIZE", 0) * 1024, "read_x2": pd.get("FETCH_SIZE", 0) * 2048,
                                         "write": pd.get("WRITE_SIZE", 0) * 1024}
    res[k] = row
    print(k, "dispatches=%d" % n, " ".join("%s=%.4g" % (c, v) for c, v in sorted(pd.items())))
if out_json:
    json.dump(res, open(out_json, "w"), indent=1)
