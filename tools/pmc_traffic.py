#!/usr/bin/env python3
"""tools/pmc_traffic.py — HBM bytes per bench step from rocprofv3 PMC passes.

  python tools/pmc_traffic.py --fetch <counter_collection.csv> --write <counter_collection.csv> \
         [--rdreq <counter_collection.csv> --wrreq <counter_collection.csv>] \
         --bench <bench line .json> --workload kem768 --out profiles/rNN_pmc_traffic_kem768.json

Per kernel the counters are averaged per dispatch; a step's traffic is sum(per-dispatch bytes x launches per step), with
the launches per step taken from the `kernels` object of the bench line produced by the same build.
  raw        (FETCH_SIZE + WRITE_SIZE) * 1024 as rocprofv3 reports them (KiB)
  corrected  (2*FETCH_SIZE + WRITE_SIZE) * 1024: MI355X_MICROARCH.md's gfx950 rule (FETCH_SIZE tallies 128-byte requests at 64)
  exact      from the request-size counters of two more passes (--rdreq: TCC_EA0_RDREQ_sum, TCC_EA0_RDREQ_32B_sum,
             TCC_EA0_RDREQ_128B_sum; --wrreq: TCC_EA0_WRREQ_sum, TCC_EA0_WRREQ_64B_sum):
             read = 128*RDREQ_128B + 32*RDREQ_32B + 64*(RDREQ - RDREQ_128B - RDREQ_32B), write = 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B).
Calibration (profiles/r04_traffic_calibration.txt): on gfx950 EVERY read request of these kernels is a 128-byte request,
whatever the width of the lane loads (4-byte LDS-DMA, 4-byte non-temporal, 16-byte) -- the L2 fetches whole lines -- so
`exact` equals `corrected` to within 0.1 %, and the stand-alone NTT reads exactly its 512 B per polynomial.  A kernel whose
read bytes exceed the span of its arguments really fetches lines twice (`over_fetch` below), it is not a counter artefact.
`expected_read` = the bytes of the kernel's input arguments per dispatch (each byte once)."""
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
ap.add_argument("--rdreq")
ap.add_argument("--wrreq")
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
exact_rd = exact_wr = None
if a.rdreq and a.wrreq:
    rq, _ = per_dispatch(a.rdreq, "TCC_EA0_RDREQ_sum")
    r32, _ = per_dispatch(a.rdreq, "TCC_EA0_RDREQ_32B_sum")
    r128, _ = per_dispatch(a.rdreq, "TCC_EA0_RDREQ_128B_sum")
    wq, _ = per_dispatch(a.wrreq, "TCC_EA0_WRREQ_sum")
    w64, _ = per_dispatch(a.wrreq, "TCC_EA0_WRREQ_64B_sum")
    exact_rd = {k: 128.0 * r128.get(k, 0) + 32.0 * r32.get(k, 0) + 64.0 * (rq[k] - r128.get(k, 0) - r32.get(k, 0)) for k in rq}
    exact_wr = {k: 64.0 * w64.get(k, 0) + 32.0 * (wq[k] - w64.get(k, 0)) for k in wq}
bench = json.loads(open(a.bench).read().strip().splitlines()[-1])
launches = {k: v["launches"] for k, v in bench["kernels"].items()}
batch = bench["config"]["batch_per_gpu"]
chunk = min(bench["config"].get("chunk_items") or batch, batch)


def expected_read(label, workload):
    """bytes of the kernel's input arguments per dispatch (each byte read once): the yardstick for over-fetch"""
    if workload == "ntt":
        return 512.0 * batch if label in ("k_ntt_batch", "k_intt_batch") else None
    k = {"kem512": 2, "kem768": 3, "kem1024": 4, "kem768_shared": 3}.get(workload)
    if not k:
        return None
    eta1 = 3 if k == 2 else 2
    du, dv = (11, 5) if k == 4 else (10, 4)
    ek, dk, c = 384 * k + 32, 768 * k + 96, 32 * (du * k + dv)
    A, ps = 512 * k * k, 64 * eta1
    prf_enc, prf_kg = k * ps + (k + 1) * 128, 2 * k * ps
    per_item = {"k_hash_encaps": (ek + 32, batch), "k_hash_decaps": (ek + 64 + c + 32, batch), "k_hash_keygen_fin": (ek + 32, batch),
                "k_hash_keygen_seed": (32, batch), "k_decrypt": (384 * k + c, batch),
                "k_encrypt": (A + ek + prf_enc + 32, chunk), "k_encrypt_cmp": (A + ek + prf_enc + 32 + c + 64, chunk),
                "k_keygen": (A + prf_kg + 32, chunk), "k_sample_main": (64, chunk)}.get(label)
    return float(per_item[0]) * per_item[1] if per_item else None


import bench as bench_mod  # noqa: E402  (kernel_label, and source_id: the hash bench.py checks before it attaches this file)


def label_of(kernel_name):
    short = re.sub(r"^void ", "", kernel_name).replace("mlkem::", "")
    return bench_mod.kernel_label(kernel_name), short


raw = corr = exact = 0.0
per_kernel = {}
per_label = {}   # keyed by bench.py's kernel labels: what bench.py attaches for its dominant kernel
for k in sorted(fetch):
    label, short = label_of(k)
    if label not in launches:
        continue
    rd, wr = fetch[k] * 1024.0, write.get(k, 0.0) * 1024.0
    row = {"bench_label": label, "launches_per_step": launches[label], "dispatches_profiled": nf[k],
           "read_raw_per_dispatch": rd, "read_x2_per_dispatch": 2 * rd, "write_per_dispatch": wr}
    pl = per_label.setdefault(label, {"launches_per_step": launches[label], "bytes_per_launch_raw": 0.0, "bytes_per_launch_corrected": 0.0})
    pl["bytes_per_launch_raw"] += rd + wr
    pl["bytes_per_launch_corrected"] += 2 * rd + wr
    raw += launches[label] * (rd + wr)
    corr += launches[label] * (2 * rd + wr)
    if exact_rd is not None:
        erd, ewr = exact_rd.get(k, 0.0), exact_wr.get(k, 0.0)
        row["read_exact_per_dispatch"], row["write_exact_per_dispatch"] = erd, ewr
        pl["bytes_per_launch_exact"] = pl.get("bytes_per_launch_exact", 0.0) + erd + ewr
        exact += launches[label] * (erd + ewr)
    exp = expected_read(label, a.workload)
    if exp:
        row["expected_read_per_dispatch"] = exp
        row["over_fetch"] = (row.get("read_exact_per_dispatch") or 2 * rd) / exp
    per_kernel[short.split("(")[0]] = row
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
       "hbm_bytes_per_step_exact": exact if exact_rd is not None else None,
       "note": "corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 as MI355X_MICROARCH.md prescribes for gfx950; exact = request-size "
               "weighted TCC_EA0_RDREQ* / WRREQ* counts (every read request of these kernels is a 128-byte request, so exact == "
               "corrected: profiles/r04_traffic_calibration.txt); over_fetch = read bytes / bytes of the kernel's input arguments",
       "per_label": per_label, "per_kernel": per_kernel}
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("workload", "hbm_bytes_per_step_raw", "hbm_bytes_per_step_corrected", "hbm_bytes_per_step_exact")}))
