#!/usr/bin/env python3
"""tools/power_probe.py — measurement aid: socket power, shader clock and throughput of each kernel family of the engine
when it runs alone for a few seconds (rocm-smi polled from a side thread).  The full KEM pass is power-capped on MI355X
(~1340 W at ~2.07 GHz instead of 2.4 GHz), so watts per kernel family matter as much as instruction counts."""
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
e = pkg.MLKEM(768, device=0)
n = 1 << 20
g = torch.Generator(device="cuda").manual_seed(1)
d, z, m = (torch.randint(0, 256, (n, 32), generator=g, device="cuda", dtype=torch.uint8) for _ in range(3))
ek, dk = e.keygen(d, z)
c, K = e.encaps(ek, m)
polys = torch.randint(0, 3329, (1 << 22, 256), generator=g, device="cuda", dtype=torch.int32).to(torch.uint16)
seeds34 = torch.randint(0, 256, (1 << 21, 34), generator=g, device="cuda", dtype=torch.uint8)
in33 = torch.randint(0, 256, (1 << 22, 33), generator=g, device="cuda", dtype=torch.uint8)

samples = []
stop = False


def poll():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        pw = re.search(r"Power \(W\): ([\d.]+)", out)
        if sclk and pw:
            samples.append((time.perf_counter(), int(sclk.group(1)), float(pw.group(1))))


th = threading.Thread(target=poll, daemon=True)
th.start()
phases = {
    "keygen": (lambda: e.keygen(d, z), n, "keygens"),
    "encaps": (lambda: e.encaps(ek, m), n, "encaps"),
    "decaps": (lambda: e.decaps(dk, c), n, "decaps"),
    "ntt+intt": (lambda: e.intt(e.ntt(polys)), polys.shape[0], "polys"),
    "sample_ntt(general kernel)": (lambda: e.sample_ntt(seeds34), seeds34.shape[0], "polys"),
    "prf eta2": (lambda: e.prf(in33, 2), in33.shape[0], "calls"),
    "H(ek)": (lambda: e.H(ek), n, "hashes"),
}
res = {}
for name, (fn, units, unit) in phases.items():
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = 0
    while time.perf_counter() - t0 < 3.0:
        fn()
        it += 1
        if it % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    mid = [(s, p) for (t, s, p) in samples if t0 + 1.0 < t < t1]
    res[name] = {"rate_per_s": it * units / (t1 - t0), "unit": unit,
                 "sclk_MHz": sum(s for s, _ in mid) / max(1, len(mid)), "power_W": sum(p for _, p in mid) / max(1, len(mid)),
                 "smi_samples": len(mid)}
    print(name, json.dumps(res[name]), flush=True)
    time.sleep(1.0)
stop = True
print(json.dumps(res))
