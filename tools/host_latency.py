import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
pkg = ge.load_package(); lib = pkg.load_library()
for n in (1, 64, 1024):
    d = np.random.default_rng(1).integers(0,256,(n,32)).astype(np.uint8); z = d.copy(); m = d.copy()
    ek, dk = np.zeros((n,1184),np.uint8), np.zeros((n,2400),np.uint8)
    c, K, K2, st = np.zeros((n,1088),np.uint8), np.zeros((n,32),np.uint8), np.zeros((n,32),np.uint8), np.zeros(n,np.int32)
    lib.mlkem_keygen(768, n, d.ctypes.data, z.ctypes.data, ek.ctypes.data, dk.ctypes.data)
    lib.mlkem_encaps(768, n, ek.ctypes.data, m.ctypes.data, c.ctypes.data, K.ctypes.data)
    t0 = time.perf_counter(); R = 200
    for _ in range(R):
        lib.mlkem_encaps(768, n, ek.ctypes.data, m.ctypes.data, c.ctypes.data, K.ctypes.data)
        lib.mlkem_decaps(768, n, dk.ctypes.data, c.ctypes.data, K2.ctypes.data, st.ctypes.data)
    dt = (time.perf_counter() - t0) / R
    assert (K == K2).all()
    print(f"n={n}: encaps+decaps host-pointer round trip {dt*1e6:.0f} us per call pair")
