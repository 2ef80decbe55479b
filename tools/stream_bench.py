#!/usr/bin/env python3
"""tools/stream_bench.py — PCIe-inclusive rate of the host-resident streaming front-end and the
device converters' bandwidth.  Measurement aid, not part of bench.py's contract (its `value` is device-resident)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
lib = pkg.load_library()
e = pkg.MLKEM(768, device=0)
n = 1 << 19
g = torch.Generator(device="cuda").manual_seed(1)
d, z, m = (torch.randint(0, 256, (n, 32), generator=g, device="cuda", dtype=torch.uint8) for _ in range(3))
ek, dk = e.keygen(d, z)
ekh, dkh, mh = ek.cpu().numpy(), dk.cpu().numpy(), m.cpu().numpy()
c, K = np.zeros((n, 1088), np.uint8), np.zeros((n, 32), np.uint8)
K2, st = np.zeros((n, 32), np.uint8), np.ones(n, np.int32)
res = {}
for chunk in (1 << 13, 1 << 14, 1 << 15, 1 << 16, 1 << 17):
    lib.mlkem_encaps_stream(768, n, ekh.ctypes.data, mh.ctypes.data, c.ctypes.data, K.ctypes.data, chunk)   # warm
    t0 = time.perf_counter()
    assert lib.mlkem_encaps_stream(768, n, ekh.ctypes.data, mh.ctypes.data, c.ctypes.data, K.ctypes.data, chunk) == 0
    assert lib.mlkem_decaps_stream(768, n, dkh.ctypes.data, c.ctypes.data, K2.ctypes.data, st.ctypes.data, chunk) == 0
    dt = time.perf_counter() - t0
    assert (K == K2).all() and (st == 0).all()
    res["stream_pairs_per_s_chunk_%d" % chunk] = n / dt
    res["stream_pcie_GBps_chunk_%d" % chunk] = n * 5856 / dt / 1e9
# converters
nc = 1 << 28
b = torch.randint(0, 256, (nc,), generator=g, device="cuda", dtype=torch.uint8)
cells = e.bytes_to_cells(b)
torch.cuda.synchronize()
for name, fn, arg in (("cells_to_bytes", e.cells_to_bytes, cells), ("bytes_to_cells", e.bytes_to_cells, b)):
    fn(arg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn(arg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    res[name + "_GBps"] = nc * 5 / dt / 1e9
print(json.dumps(res))
