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
    if "SQ_INSTS_VALU" in pd and "GRBM_GUI_ACTIVE" in pd and pd["GRBM_GUI_ACTIVE"] > 0:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs in all
        simd_cycles = pd["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        row["valu_instr_per_simd_cycle"] = pd["SQ_INSTS_VALU"] / simd_cycles
        row["cycles_per_valu_instr"] = simd_cycles / pd["SQ_INSTS_VALU"]
        if "SQ_WAVE_CYCLES" in pd:
            row["resident_waves_per_simd"] = pd["SQ_WAVE_CYCLES"] * 4.0 / simd_cycles   # SQ_WAVE_CYCLES counts quad-cycles
    res[k] = row
    print(k, "dispatches=%d" % n, " ".join("%s=%.4g" % (c, v) for c, v in sorted(pd.items())),
          " ".join("%s=%.3g" % (c, row[c]) for c in ("valu_instr_per_simd_cycle", "cycles_per_valu_instr", "resident_waves_per_simd") if c in row))
if out_json:
    json.dump(res, open(out_json, "w"), indent=1)
