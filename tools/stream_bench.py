#!/usr/bin/env python3
"""tools/stream_bench.py — PCIe-inclusive rate of the host-resident streaming front-end for pageable
and for pinned caller buffers, and the device converters' bandwidth.  Measurement aid, not part of bench.py's contract
(its `value` is device-resident).  Gate of VERDICT r1 item 7: >= 9e6 pairs/s with pinned buffers."""
import json
import os
import sys
import time

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
c_ref, K_ref = e.encaps(ek, m)
torch.cuda.synchronize()
res = {"items": n}


def run(tag, ekh, dkh, mh, c, K, K2, st, chunks):
    for chunk in chunks:
        lib.mlkem_encaps_stream(768, n, ekh.data_ptr(), mh.data_ptr(), c.data_ptr(), K.data_ptr(), chunk)   # warm (allocations)
        lib.mlkem_decaps_stream(768, n, dkh.data_ptr(), c.data_ptr(), K2.data_ptr(), st.data_ptr(), chunk)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            assert lib.mlkem_encaps_stream(768, n, ekh.data_ptr(), mh.data_ptr(), c.data_ptr(), K.data_ptr(), chunk) == 0
            t1 = time.perf_counter()
            assert lib.mlkem_decaps_stream(768, n, dkh.data_ptr(), c.data_ptr(), K2.data_ptr(), st.data_ptr(), chunk) == 0
            t2 = time.perf_counter()
            if t2 - t0 < best:
                best, enc, dec = t2 - t0, t1 - t0, t2 - t1
        assert torch.equal(K, K2) and int(st.abs().max()) == 0 and torch.equal(c, c_ref.cpu()) and torch.equal(K, K_ref.cpu())
        res["%s_pairs_per_s_chunk_%d" % (tag, chunk)] = n / best
        res["%s_pcie_GBps_chunk_%d" % (tag, chunk)] = n * 5856 / best / 1e9
        res["%s_encaps_GBps_h2d_chunk_%d" % (tag, chunk)] = n * 1216 / enc / 1e9
        res["%s_decaps_GBps_h2d_chunk_%d" % (tag, chunk)] = n * 3488 / dec / 1e9


chunks = [int(x) for x in os.environ.get("CHUNKS", "8192,16384,32768,65536,131072").split(",")]
host = [t.cpu() for t in (ek, dk, m)]
outs = [torch.zeros((n, 1088), dtype=torch.uint8), torch.zeros((n, 32), dtype=torch.uint8), torch.zeros((n, 32), dtype=torch.uint8),
        torch.ones(n, dtype=torch.int32)]
run("pageable", *host, *outs, chunks)
run("pinned", *[t.pin_memory() for t in host], *[t.pin_memory() for t in outs], chunks)
lib.mlkem_stream_release()
# a 4-member in-process shard on this one device (rehearsal of mlkem_*_multi; pinned numpy views)
mm = pkg.MLKEMMulti(768, devices=[0, 0])
ekn, mn = host[0].numpy(), host[2].numpy()
mm.encaps(ekn, mn)
t0 = time.perf_counter()
c2, K2 = mm.encaps(ekn, mn)
res["multi2_pageable_encaps_items_per_s"] = n / (time.perf_counter() - t0)
assert (c2 == c_ref.cpu().numpy()).all()
mm.close()
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
