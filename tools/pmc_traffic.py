#!/usr/bin/env python3
"""tools/pmc_traffic.py — HBM bytes per bench step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

  python tools/pmc_traffic.py --fetch <counter_collection.csv> --write <counter_collection.csv> \
         --bench <bench line .json> --workload kem768 --out profiles/rNN_pmc_traffic_kem768.json

Per kernel the counters are averaged per dispatch; a step's traffic is sum(per-dispatch bytes x launches per step), with
the launches per step taken from the `kernels` object of the bench line produced by the same build.  FETCH_SIZE and
WRITE_SIZE are in KiB; `corrected` applies (2*FETCH + WRITE)*1024 as MI355X_MICROARCH.md prescribes for gfx950."""
import argparse
import collections
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--fetch", required=True)
ap.add_argument("--write", required=True)
ap.add_argument("--bench", required=True)
ap.add_argument("--workload", required=True)
ap.add_argument("--out", required=True)
a = ap.parse_args()


def per_dispatch(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return {k: tot[k] / len(n[k]) for k in tot}, {k: len(n[k]) for k in tot}


fetch, nf = per_dispatch(a.fetch, "FETCH_SIZE")
write, _ = per_dispatch(a.write, "WRITE_SIZE")
bench = json.loads(open(a.bench).read().strip().splitlines()[-1])
launches = {k: v["launches"] for k, v in bench["kernels"].items()}


def label_of(kernel_name):
    """bench labels: k_sample_main, k_sample_tail (= k_sample in list mode), k_encrypt / k_encrypt_cmp, ..."""
    short = re.sub(r"^void ", "", kernel_name).replace("mlkem::", "")
    base = short.split("<")[0].split("(")[0]
    if base == "k_sample":
        return "k_sample_restart", short
    if base == "k_ntt4_batch":
        return ("k_intt_batch" if "<true>" in short else "k_ntt_batch"), short
    if base in ("k_encrypt", "k_encrypt2"):
        return ("k_encrypt_cmp" if re.search(r",\s*true>", short) else "k_encrypt"), short
    if base == "k_keygen2":
        return "k_keygen", short
    if base == "k_decrypt4":
        return "k_decrypt", short
    if base == "k_sample_resume":
        return "k_sample_tail", short
    return base, short


raw = corr = 0.0
per_kernel = {}
per_label = {}   # keyed by bench.py's kernel labels: what bench.py attaches for its dominant kernel
for k in sorted(fetch):
    label, short = label_of(k)
    if label not in launches:
        continue
    rd, wr = fetch[k] * 1024.0, write.get(k, 0.0) * 1024.0
    per_kernel[short.split("(")[0]] = {"bench_label": label, "launches_per_step": launches[label], "dispatches_profiled": nf[k],
                                       "read_raw_per_dispatch": rd, "read_x2_per_dispatch": 2 * rd, "write_per_dispatch": wr}
    pl = per_label.setdefault(label, {"launches_per_step": launches[label], "bytes_per_launch_raw": 0.0, "bytes_per_launch_corrected": 0.0})
    pl["bytes_per_launch_raw"] += rd + wr
    pl["bytes_per_launch_corrected"] += 2 * rd + wr
    raw += launches[label] * (rd + wr)
    corr += launches[label] * (2 * rd + wr)
import bench as bench_mod  # noqa: E402  (source_id: the hash bench.py checks before it attaches this file)

try:
    git_head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "crystals-kyber_amd/csrc", "include"],
                                capture_output=True, text=True).stdout.strip())
except OSError:
    git_head, dirty = "?", False
out = {"workload": "%s batch %d chunk %s" % (a.workload, bench["config"]["batch_per_gpu"], bench["config"].get("chunk_items")),
       "batch": bench["config"]["batch_per_gpu"], "source_id": bench_mod.source_id(),
       "git_head": git_head + ("+uncommitted kernel changes" if dirty else ""),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace), tools/pmc_traffic.py",
       "hbm_bytes_per_step_raw": raw, "hbm_bytes_per_step_corrected": corr,
       "note": "corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 as MI355X_MICROARCH.md prescribes for gfx950; raw FETCH under-reports "
               "8-16 B/lane loads (DESIGN.md section 5)",
       "per_label": per_label, "per_kernel": per_kernel}
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("workload", "hbm_bytes_per_step_raw", "hbm_bytes_per_step_corrected")}))
