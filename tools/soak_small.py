#!/usr/bin/env python3
"""tools/soak_small.py [seconds [min_items max_items]] -- sustained run of the small-call kernels (mlkem_small.hpp: jobs and flag hand-overs in LDS): for
`seconds` (default 60) keygen -> encaps -> decaps calls of random sizes 1..896 and random parameter sets are queued back to back on
two streams, every round checked (K == K', status, a tampered ciphertext rejected), and a watchdog thread reports if any round
takes longer than 5 s (a wave that never finishes).  One line for profiles/rNN_soak_small.txt."""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, 896)   # optional size range (mid-size calls: 897 6000)
pkg = ge.load_package()
dev = torch.device("cuda", 0)
# one context per parameter set AND stream: a context's scratch serves one call at a time (sizes above a set's small-call limit take the batch path)
engines = {(s, i): pkg.MLKEM(s, device=0, chunk_items=8192) for s in (512, 768, 1024) for i in range(2)}
rng = np.random.default_rng(2024)
sizes = [1, 2, 3, 7, 64, 127, 128, 129, 255, 256, 257, 319, 320, 321, 500, 768, 895, 896]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
last_beat = [time.perf_counter()]
stop = threading.Event()
late = []


def watchdog():
    while not stop.is_set():
        if time.perf_counter() - last_beat[0] > 5.0:
            late.append(time.perf_counter() - last_beat[0])
            print("WATCHDOG: no round finished for %.1f s" % late[-1], flush=True)
            os._exit(3)
        time.sleep(0.25)


threading.Thread(target=watchdog, daemon=True).start()
t0 = time.perf_counter()
rounds = calls = items = 0
bad = 0
while time.perf_counter() - t0 < seconds:
    pend = []
    for si, s in enumerate(streams):
        with torch.cuda.stream(s):
            pset = int(rng.choice((512, 768, 1024)))
            n = int(rng.choice(sizes)) if (rng.random() < 0.7 and hi == 896) else int(rng.integers(lo, hi + 1))
            e = engines[(pset, si)]
            d, z, m = (torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)).to(dev, non_blocking=True) for _ in range(3))
            ek, dk = e.keygen(d, z)
            c, K = e.encaps(ek, m)
            cb = c.clone()
            t = int(rng.integers(0, n))
            cb[t, int(rng.integers(0, c.shape[1]))] ^= 1 << int(rng.integers(0, 8))
            Kd, st = e.decaps(dk, cb)
            pend.append((n, t, K, Kd, st))
            calls += 3
            items += n
    for s in streams:
        s.synchronize()
    for n, t, K, Kd, st in pend:
        same = (K == Kd).all(dim=1)
        keep = torch.ones(n, dtype=torch.bool, device=dev)
        keep[t] = False
        bad += int((~same[keep]).sum()) + int(same[t]) + int(st.abs().sum())
    rounds += 1
    last_beat[0] = time.perf_counter()
stop.set()
dt = time.perf_counter() - t0
print("small-call soak: %.0f s, %d rounds on 2 streams, %d calls (keygen / encaps / decaps of %d..%d items, all three parameter sets), %d items; "
      "mismatches %d; longest gap between rounds under the 5 s watchdog: yes" % (dt, rounds, calls, lo, hi, items, bad))
for e in engines.values():
    e.close()
sys.exit(1 if bad else 0)
