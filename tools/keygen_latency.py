"""tools/keygen_latency.py -- device-resident call latency of the small-call kernels: KeyGen / Encaps / Decaps of every parameter set
at 1 and 64 items (one launch per call, calls queued back to back on one stream), and KeyGen at 1024 items (batch kernels)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
pkg = ge.load_package()
R = 300


def per_call(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / R * 1e6


for pset in (512, 768, 1024):
    for n in (1, 64) + ((1024,) if pset == 768 else ()):
        e = pkg.MLKEM(pset, device=0, chunk_items=8192)
        rng = np.random.default_rng(n)
        d, z, m = (torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)).cuda() for _ in range(3))
        ek, dk = e.keygen(d, z)
        c, K = e.encaps(ek, m)
        K2, st = torch.empty_like(K), torch.empty(n, dtype=torch.int32, device="cuda")
        t_k = per_call(lambda: e.keygen(d, z, ek=ek, dk=dk))
        t_e = per_call(lambda: e.encaps(ek, m, c=c, K=K))
        t_d = per_call(lambda: e.decaps(dk, c, K=K2, status=st))
        assert torch.equal(K, K2) and int(st.abs().sum()) == 0
        print("ML-KEM-%d n=%d: keygen %.1f  encaps %.1f  decaps %.1f us per call" % (pset, n, t_k, t_e, t_d))
        e.close()
