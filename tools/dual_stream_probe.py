#!/usr/bin/env python3
"""tools/dual_stream_probe.py — measurement aid: does splitting one device-resident batch over several engine contexts /
streams of the SAME GPU (mlkem_*_multi_dev with a repeated device) raise throughput?  (kernel tails of one member are
filled by the other's kernels).  Prints pairs/s for 1, 2, 3, 4 members."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
n = 1 << 20
g = torch.Generator(device="cuda").manual_seed(7)
d, z, m = (torch.randint(0, 256, (n, 32), generator=g, device="cuda", dtype=torch.uint8) for _ in range(3))
e = pkg.MLKEM(768, device=0)
ek, dk = e.keygen(d, z)
c_ref, K_ref = e.encaps(ek, m)
torch.cuda.synchronize()
res = {}
for members in (1, 2, 3, 4):
    mm = pkg.MLKEMMulti(768, devices=[0] * members)
    rg = mm.ranges(n)
    eks, dks, ms = ([t[lo:hi].contiguous() for lo, hi in rg] for t in (ek, dk, m))
    torch.cuda.synchronize()

    def step():
        cs, Ks = mm.encaps_dev(eks, ms)
        Kd, st = mm.decaps_dev(dks, cs)
        return cs, Ks, Kd, st
    for _ in range(3):
        out = step()
    mm.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        out = step()
    mm.sync()
    dt = (time.perf_counter() - t0) / 20
    cs, Ks, Kd, st = out
    ok = torch.equal(torch.cat(cs), c_ref) and torch.equal(torch.cat(Ks), K_ref) and torch.equal(torch.cat(Kd), K_ref) and int(torch.cat(st).abs().max()) == 0
    res["members_%d" % members] = {"pairs_per_s": n / dt, "ms_per_step": dt * 1e3, "correct": bool(ok)}
    mm.close()
    del eks, dks, ms, out, cs, Ks, Kd, st
    torch.cuda.empty_cache()
print(json.dumps(res))
