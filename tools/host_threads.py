"""tools/host_threads.py -- one-item host-pointer calls (what the ml_kem.h shim makes) from T host threads at once: ML-KEM-768
Encaps + Decaps pairs per second against T.  Every thread has its own buffers and checks K == K' on every pair; the library hands
each concurrent caller an engine of its own (HostState::lanes, mlkem_capi.hip; MLKEM_HOST_LANES=0 queues them on one)."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
lib = pkg.load_library()
for f in (lib.mlkem_keygen, lib.mlkem_encaps):
    f.argtypes = [C.c_int, C.c_size_t] + [C.c_void_p] * 4
lib.mlkem_decaps.argtypes = [C.c_int, C.c_size_t] + [C.c_void_p] * 4
R = int(os.environ.get("HOST_THREADS_ROUNDS", "400"))


def worker(tid, out, barrier):
    rng = np.random.default_rng(tid)
    d, z, m = (rng.integers(0, 256, (1, 32), dtype=np.uint8) for _ in range(3))
    ek, dk = np.zeros((1, 1184), np.uint8), np.zeros((1, 2400), np.uint8)
    c, K, K2 = np.zeros((1, 1088), np.uint8), np.zeros((1, 32), np.uint8), np.zeros((1, 32), np.uint8)
    st = np.zeros(1, np.int32)
    assert lib.mlkem_keygen(768, 1, d.ctypes.data, z.ctypes.data, ek.ctypes.data, dk.ctypes.data) == 0
    bad = 0
    barrier.wait()
    t0 = time.perf_counter()
    for r in range(R):
        m[0, 0] = r & 255
        bad += lib.mlkem_encaps(768, 1, ek.ctypes.data, m.ctypes.data, c.ctypes.data, K.ctypes.data) != 0
        bad += lib.mlkem_decaps(768, 1, dk.ctypes.data, c.ctypes.data, K2.ctypes.data, st.ctypes.data) != 0
        bad += int(not (K == K2).all()) + int(st[0] != 0)
    out[tid] = (time.perf_counter() - t0, bad)


for T in (1, 2, 4, 8, 16):
    out, barrier = {}, threading.Barrier(T)
    th = [threading.Thread(target=worker, args=(t, out, barrier)) for t in range(T)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = max(v[0] for v in out.values())
    bad = sum(v[1] for v in out.values())
    print("MLKEM_HOST_LANES=%s MLKEM_HOST_COMBINE=%s threads=%2d: %8.0f pairs/s  (%.1f us per pair and thread, errors %d)" % (
        os.environ.get("MLKEM_HOST_LANES", "default"), os.environ.get("MLKEM_HOST_COMBINE", "default"), T, T * R / wall, wall / R * 1e6, bad))
lib.mlkem_host_release()
