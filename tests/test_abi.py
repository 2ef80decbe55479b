"""CPU tier: the C-ABI libraries load and export every symbol include/*.h declares (no compute calls),
and host-side logic that needs no GPU."""
import ctypes as C
import os
import re

import pytest

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    return ge.build()


def test_batch_header_symbols_exported(pkg):
    hdr = open(os.path.join(ROOT, "include", "mlkem_batch.h")).read()
    declared = set(re.findall(r"\b(mlkem_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(pkg.ABI_SYMBOLS), declared ^ set(pkg.ABI_SYMBOLS)
    lib = C.CDLL(pkg.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_library_exports_the_c_abi_and_nothing_else(pkg):
    """libmlkem_amd.so is built with -fvisibility=hidden + csrc/exports.map: every defined dynamic symbol is an entry point
    of include/mlkem_batch.h -- no template instantiations, kernel handles, worker classes or libstdc++ weak symbols that a
    host program's same-named symbol could interpose."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", pkg.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert names, "no dynamic symbols?"
    stray = [n for n in names if not n.startswith("mlkem_")]
    assert stray == [], stray
    assert set(names) == set(pkg.ABI_SYMBOLS), set(names) ^ set(pkg.ABI_SYMBOLS)


def test_shim_symbols_exported(pkg):
    lib = C.CDLL(pkg.SHIM_PATH)
    for name in pkg.SHIM_SYMBOLS:
        assert hasattr(lib, name), name
    assert C.c_int.in_dll(lib, "ml_errno").value == 0


def test_sizes_and_params_match_reference_init(pkg, golden):
    lib = pkg.load_library()
    for pset, (ekl, dkl, cl) in pkg.SIZES.items():
        a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
        assert lib.mlkem_sizes(pset, C.byref(a), C.byref(b), C.byref(c)) == 0
        assert (a.value, b.value, c.value) == (ekl, dkl, cl)
        v = (C.c_int * 5)()
        assert lib.mlkem_params(pset, v) == 0
        k = v[0]
        assert ekl == 384 * k + 32 and dkl == 768 * k + 96 and cl == 32 * (v[3] * k + v[4])
    assert list((lambda v: (lib.mlkem_params(512, v), list(v))[1])((C.c_int * 5)())) == [2, 3, 2, 10, 4]
    assert list((lambda v: (lib.mlkem_params(768, v), list(v))[1])((C.c_int * 5)())) == [3, 2, 2, 10, 4]
    assert list((lambda v: (lib.mlkem_params(1024, v), list(v))[1])((C.c_int * 5)())) == [4, 2, 2, 11, 5]
    # invalid parameter sets: same code as the reference's init() (golden G7)
    for s, rc in golden["G7_init_errno"].items():
        assert lib.mlkem_sizes(int(s), None, None, None) == rc


def test_shim_struct_layout_matches_reference_abi(pkg):
    """union byte is 4 bytes; sizeof(PARAMS)=20, PKE=24, KEM=144 (K at 0, c at 128, c_len at 136)."""
    class Byte(C.Union):
        _fields_ = [("s", C.c_uint, 7), ("e", C.c_uint, 8)]

    class PARAMS(C.Structure):
        _fields_ = [(n, Byte) for n in ("k", "n1", "n2", "du", "dv")]

    class PKE(C.Structure):
        _fields_ = [("ek", C.POINTER(Byte)), ("dk", C.POINTER(Byte)), ("ek_len", C.c_uint), ("dk_len", C.c_uint)]

    class KEM(C.Structure):
        _fields_ = [("K", Byte * 32), ("c", C.POINTER(Byte)), ("c_len", C.c_uint)]

    assert C.sizeof(Byte) == 4 and C.sizeof(PARAMS) == 20 and C.sizeof(PKE) == 24 and C.sizeof(KEM) == 144
    assert KEM.c.offset == 128 and KEM.c_len.offset == 136

    # libffi cannot pass bit-field unions by value, so the calls below use layout-identical plain-uint structs
    class RawParams(C.Structure):
        _fields_ = [(n, C.c_uint) for n in ("k", "n1", "n2", "du", "dv")]

    class RawKEM(C.Structure):
        _fields_ = [("K", C.c_uint * 32), ("c", C.POINTER(C.c_uint)), ("c_len", C.c_uint)]

    assert C.sizeof(RawParams) == C.sizeof(PARAMS) and C.sizeof(RawKEM) == C.sizeof(KEM)
    shim = C.CDLL(pkg.SHIM_PATH)
    shim.init.restype = RawParams
    p = shim.init(768)
    assert (p.k & 0xFF, p.n1 & 0xFF, p.n2 & 0xFF, p.du & 0xFF, p.dv & 0xFF) == (3, 2, 2, 10, 4)
    errno = C.c_int.in_dll(shim, "ml_errno")
    assert errno.value == 0
    shim.init(123)
    assert errno.value == -1   # ml_kem.c:1389-1391
    errno.value = 0
    # length checks happen before any device work (ml_kem.c:1267, :1320, :1328)
    shim.KEM_Encaps.restype = RawKEM
    shim.KEM_Encaps.argtypes = [C.POINTER(RawParams), C.POINTER(C.c_uint), C.c_uint]
    buf = (C.c_uint * 4000)()
    r = shim.KEM_Encaps(C.byref(p), buf, 1)   # EncapsDecaps_test.c passes ek_len = 1
    assert errno.value == -3 and not r.c and r.c_len == 0
    errno.value = 0
    shim.KEM_Decaps.restype = C.POINTER(C.c_uint)
    shim.KEM_Decaps.argtypes = [C.POINTER(RawParams), C.POINTER(C.c_uint), C.c_uint, C.POINTER(C.c_uint), C.c_uint]
    assert not shim.KEM_Decaps(C.byref(p), buf, 2400, buf, 1087) and errno.value == -3
    errno.value = 0
    assert not shim.KEM_Decaps(C.byref(p), buf, 2399, buf, 1088) and errno.value == -3
    errno.value = 0


def test_engine_fails_loudly_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.MLKEMError) as e:
        pkg.MLKEM(768)
    assert e.value.code == -100
    lib = pkg.load_library()
    assert lib.mlkem_device_count() == 0
    import numpy as np
    d = np.zeros(32, np.uint8)
    ek = np.zeros(1184, np.uint8)
    dk = np.zeros(2400, np.uint8)
    rc = lib.mlkem_keygen(768, 1, d.ctypes.data, d.ctypes.data, ek.ctypes.data, dk.ctypes.data)
    assert rc == -100 and not ek.any()   # no CPU fallback: nothing was computed


def test_shard_range(pkg):
    for n, w in ((1 << 23, 8), (1000, 3), (5, 8), (0, 2)):
        spans = [pkg.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    assert pkg.shard_range(1 << 23, 3, 8) == (3 << 20, 4 << 20)   # config 5: item i -> GPU i >> 20


def test_caller_supplied_outputs_are_validated(pkg):
    """outputs reach the C-ABI as raw pointers, so a wrong device / dtype / shape / stride must be rejected on
    the host (here on CPU tensors through a bare instance: no engine can be created without a GPU)."""
    import torch
    e = pkg.MLKEM.__new__(pkg.MLKEM)
    e.torch, e.device = torch, torch.device("cpu")
    good = torch.empty((4, 32), dtype=torch.uint8)
    assert e._out(4, 32, given=good) is good
    assert e._out(4, None, torch.int32, given=torch.empty(4, dtype=torch.int32)).shape == (4,)
    bad = [torch.empty((3, 32), dtype=torch.uint8),                    # too few rows
           torch.empty((4, 31), dtype=torch.uint8),                    # short rows
           torch.empty((4, 32), dtype=torch.int64),                    # wrong element type
           torch.empty((4, 64), dtype=torch.uint8)[:, ::2],            # non-contiguous
           torch.empty((4, 32), dtype=torch.uint8, device="meta"),     # wrong device
           [0] * 128]                                                  # not a tensor
    for t in bad:
        with pytest.raises(pkg.MLKEMError) as ex:
            e._out(4, 32, given=t)
        assert ex.value.code == -101
    with pytest.raises(pkg.MLKEMError):
        e._out(4, None, torch.int32, given=torch.empty(4, dtype=torch.int64))   # status must be int32
    e._ctx = None   # nothing to destroy


def test_multi_and_host_state_entry_points_fail_loudly_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = pkg.load_library()
    h = C.c_void_p()
    assert lib.mlkem_multi_create(C.byref(h), 2, (C.c_int * 2)(0, 0), 0) == -100 and not h.value
    with pytest.raises(pkg.MLKEMError):
        pkg.MLKEMMulti(768, devices=[0, 0])
    lib.mlkem_host_release()     # nothing cached: a no-op, must not crash
    lib.mlkem_stream_release()
    a, b = C.c_size_t(), C.c_size_t()
    assert lib.mlkem_shard_range(1 << 23, 3, 8, C.byref(a), C.byref(b)) == 0 and (a.value, b.value) == (3 << 20, 4 << 20)
    assert lib.mlkem_shard_range(10, 3, 3, C.byref(a), C.byref(b)) == -101
    for n, w in ((1000, 3), (5, 8), (0, 2)):
        for r in range(w):
            lib.mlkem_shard_range(n, r, w, C.byref(a), C.byref(b))
            assert (a.value, b.value) == pkg.shard_range(n, r, w)


def test_sha3_pad_suffix_host_helper(pkg):
    """mlkem_sha3_pad_suffix appends the caller's suffix bits verbatim, then pad10*1 (pure host code)."""
    import numpy as np
    lib = pkg.load_library()
    bits = np.array([1, 0, 1], np.uint8)
    out = np.zeros(168, np.uint8)
    sfx = np.array([1, 1], np.uint8)   # RawSHAKE
    assert lib.mlkem_sha3_pad_suffix(bits.ctypes.data, 3, sfx.ctypes.data, 2, 168, out.ctypes.data, out.size) == 1
    assert out[0] == 0b00111101 and out[167] == 0x80 and not out[1:167].any()    # 101 | 11 | 1, LSB first
    out2 = np.zeros(168, np.uint8)
    assert lib.mlkem_sha3_pad_bits(bits.ctypes.data, 3, 1, 168, out2.ctypes.data, out2.size) == 1
    assert out2[0] == 0b11111101 and out2[167] == 0x80
    assert lib.mlkem_sha3_pad_suffix(bits.ctypes.data, 3, sfx.ctypes.data, 9, 168, out.ctypes.data, out.size) == -101


def test_rng_failure_sets_ml_errno_minus_2(pkg, tmp_path):
    """ml_kem.c:458-478, :1245-1251, :1298-1304: when the random source fails, KEM_KeyGen / KEM_Encaps set ml_errno = -2
    (the shim returns zeroed structs where the reference returns uninitialised ones).  The random draw comes before any
    device work, so this runs without a GPU: a child process preloads an interposer whose getrandom() always fails."""
    import subprocess
    import sys
    src = tmp_path / "norandom.c"
    src.write_text("#include <errno.h>\n#include <sys/types.h>\n"
                   "ssize_t getrandom(void* b, size_t n, unsigned f) { (void)b; (void)n; (void)f; errno = ENOSYS; return -1; }\n")
    so = tmp_path / "libnorandom.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    child = r'''
import ctypes as C, sys
class RawParams(C.Structure):
    _fields_ = [(n, C.c_uint) for n in ("k", "n1", "n2", "du", "dv")]
class RawPKE(C.Structure):
    _fields_ = [("ek", C.POINTER(C.c_uint)), ("dk", C.POINTER(C.c_uint)), ("ek_len", C.c_uint), ("dk_len", C.c_uint)]
class RawKEM(C.Structure):
    _fields_ = [("K", C.c_uint * 32), ("c", C.POINTER(C.c_uint)), ("c_len", C.c_uint)]
shim = C.CDLL(sys.argv[1])
shim.init.restype = RawParams
errno = C.c_int.in_dll(shim, "ml_errno")
for pset, ek_len in ((512, 800), (768, 1184), (1024, 1568)):
    p = shim.init(pset)
    shim.KEM_KeyGen.restype = RawPKE
    shim.KEM_KeyGen.argtypes = [C.POINTER(RawParams)]
    errno.value = 0
    r = shim.KEM_KeyGen(C.byref(p))
    assert errno.value == -2 and not r.ek and not r.dk and r.ek_len == 0 and r.dk_len == 0, (pset, errno.value)
    shim.KEM_Encaps.restype = RawKEM
    shim.KEM_Encaps.argtypes = [C.POINTER(RawParams), C.POINTER(C.c_uint), C.c_uint]
    errno.value = 0
    k = shim.KEM_Encaps(C.byref(p), (C.c_uint * ek_len)(), ek_len)
    assert errno.value == -2 and not k.c and k.c_len == 0 and not any(k.K), (pset, errno.value)
print("rng-failure OK")
'''
    env = dict(os.environ, LD_PRELOAD=str(so))
    r = subprocess.run([sys.executable, "-c", child, pkg.SHIM_PATH], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "rng-failure OK" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
    assert "Random bit generation failed" in r.stderr   # the reference's message (ERR_MSG, ml_kem.c:11-13)


def test_release_while_other_threads_call_is_safe_without_gpu(pkg):
    """Two threads inside host-pointer calls while a third keeps releasing the cached host state: no crash, and without a
    device every call still fails loudly.  (The same race on a real device, where the registry holds engines and contexts,
    is tests/test_gpu_round3.py::test_host_release_races_with_calls_in_flight.)"""
    import threading
    import numpy as np
    lib = pkg.load_library()
    if lib.mlkem_device_count() > 0:
        pytest.skip("covered by the GPU tier on a box with a device")
    stop = threading.Event()
    rcs = []

    def caller():
        f = np.zeros((4, 256), np.uint16)
        d = np.zeros((4, 32), np.uint8)
        ek, dk = np.zeros((4, 1184), np.uint8), np.zeros((4, 2400), np.uint8)
        for _ in range(300):
            rcs.append(lib.mlkem_ntt(4, f.ctypes.data, f.ctypes.data))
            rcs.append(lib.mlkem_keygen(768, 4, d.ctypes.data, d.ctypes.data, ek.ctypes.data, dk.ctypes.data))

    def releaser():
        while not stop.is_set():
            lib.mlkem_host_release()
            lib.mlkem_stream_release()

    ts = [threading.Thread(target=caller) for _ in range(2)]
    rel = threading.Thread(target=releaser)
    rel.start()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    stop.set()
    rel.join()
    assert len(rcs) == 1200 and set(rcs) == {-100}
