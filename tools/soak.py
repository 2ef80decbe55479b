#!/usr/bin/env python3
"""tools/soak.py [steps] — sustained run: ML-KEM-768 encaps + decaps of 2^20 pairs, `steps` steps back to back (default 1500, ~25 s),
shader clock and socket power sampled throughout, correctness gate after the soak.  One line for profiles/rNN_soak.txt."""
import glob
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda", 0)
p = torch.cuda.get_device_properties(dev)
hw = glob.glob("/sys/bus/pci/devices/%04x:%02x:%02x.0/hwmon/hwmon*" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))[0]
pkg = ge.load_package()
n = 1 << 20
eng = pkg.MLKEM(768, device=0)
d, z, m = (bench.device_seeds(lbl, 0, n, dev) for lbl in ("mlkem-bench-d", "mlkem-bench-z", "mlkem-bench-m"))
ek, dk = eng.keygen(d, z)
c = torch.empty((n, eng.c_len), dtype=torch.uint8, device=dev)
K, K2 = (torch.empty((n, 32), dtype=torch.uint8, device=dev) for _ in range(2))
st = torch.empty(n, dtype=torch.int32, device=dev)
samples, stop = [], threading.Event()


def poll():
    while not stop.is_set():
        samples.append((int(open(os.path.join(hw, "freq1_input")).read()) // 1000000, int(open(os.path.join(hw, "power1_input")).read()) / 1e6))
        time.sleep(0.1)


for _ in range(3):
    eng.encaps(ek, m, c=c, K=K)
    eng.decaps(dk, c, K=K2, status=st)
torch.cuda.synchronize()
th = threading.Thread(target=poll, daemon=True)
th.start()
t0 = time.perf_counter()
for i in range(steps):
    eng.encaps(ek, m, c=c, K=K)
    eng.decaps(dk, c, K=K2, status=st)
    if i % 50 == 49:
        torch.cuda.synchronize()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
stop.set()
th.join(timeout=2)
ok = bool(torch.equal(K, K2)) and int(st.abs().max()) == 0
clk, pw = sorted(s[0] for s in samples[5:]), sorted(s[1] for s in samples[5:])
cap = int(open(os.path.join(hw, "power1_cap")).read()) / 1e6
print("ML-KEM-768 encaps+decaps, %d steps of 2^20 pairs back to back (%.1f s): %.4g pairs/s, %.3f ms per step, correct=%s; shader clock min %d / median %d / max %d MHz, "
      "socket power min %.0f / median %.0f / max %.0f W (cap %.0f); %.2f uJ per pair" % (
          steps, dt, n * steps / dt, 1e3 * dt / steps, ok, clk[0], clk[len(clk) // 2], clk[-1], pw[0], pw[len(pw) // 2], pw[-1], cap,
          1e6 * pw[len(pw) // 2] * dt / steps / n))
eng.close()
sys.exit(0 if ok else 3)
