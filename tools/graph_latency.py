#!/usr/bin/env python3
"""tools/graph_latency.py -- device-resident Encaps + Decaps of a small batch: eager calls against the replay of a captured HIP
graph (torch.cuda.CUDAGraph around the *_dev calls; the capture includes the fork to the context's side stream)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
for n in (1, 64, 1024, 4096):
    e = pkg.MLKEM(768, device=0, chunk_items=8192)
    rng = np.random.default_rng(n)
    d, z, m = (torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)).cuda() for _ in range(3))
    ek, dk = e.keygen(d, z)
    c = torch.empty((n, 1088), dtype=torch.uint8, device="cuda")
    K = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    K2 = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    st = torch.empty(n, dtype=torch.int32, device="cuda")

    def pair():
        e.encaps(ek, m, c=c, K=K)
        e.decaps(dk, c, K=K2, status=st)

    for _ in range(5):
        pair()
    torch.cuda.synchronize()
    R = 300
    t0 = time.perf_counter()
    for _ in range(R):
        pair()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / R
    t0 = time.perf_counter()
    for _ in range(R):
        pair()
        torch.cuda.synchronize()
    eager_sync = (time.perf_counter() - t0) / R
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        pair()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(K, K2) and int(st.abs().sum()) == 0
    t0 = time.perf_counter()
    for _ in range(R):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / R
    t0 = time.perf_counter()
    for _ in range(R):
        g.replay()
        torch.cuda.synchronize()
    graph_sync = (time.perf_counter() - t0) / R
    print("n=%5d  eager %.1f us back to back, %.1f us with a sync per pair | graph replay %.1f us back to back, %.1f us with a sync per pair"
          % (n, eager * 1e6, eager_sync * 1e6, graph * 1e6, graph_sync * 1e6))
    e.close()
