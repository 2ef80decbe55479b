import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
pkg = ge.load_package()
for n in (64, 1024):
    e = pkg.MLKEM(768, device=0, chunk_items=8192)
    rng = np.random.default_rng(n)
    d, z = (torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)).cuda() for _ in range(2))
    for _ in range(5): e.keygen(d, z)
    torch.cuda.synchronize(); t0 = time.perf_counter(); R = 300
    for _ in range(R): e.keygen(d, z)
    torch.cuda.synchronize()
    print("MLKEM_WIDE_HASH_ITEMS=%s keygen n=%d: %.1f us per call" % (os.environ.get("MLKEM_WIDE_HASH_ITEMS", "default"), n, (time.perf_counter() - t0) / R * 1e6))
    e.close()
