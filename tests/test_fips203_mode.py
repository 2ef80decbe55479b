"""the FIPS-203-conformant mode (PRF and J on SHAKE256, real modulus check).

The reference has no vectors for it (its PRF/J are SHAKE128), and no external KAT file is available
offline, so the mode is pinned by an independent pure-Python restatement of FIPS 203 (tests/fips203_py.py, hashlib
underneath): restatement -> oracle (FIPS switch) in the CPU tier, oracle -> emulated kernels (CPU) and HIP engine (GPU)."""
import ctypes as C

import numpy as np
import pytest

import __graft_entry__ as ge
import fips203_py as F
from conftest import seeds
from oracle.loader import SIZES


@pytest.fixture()
def fips_oracle(oracle):
    oracle.set_conformance(1)
    yield oracle
    oracle.set_conformance(0)


@pytest.mark.parametrize("pset", (512, 768, 1024))
def test_oracle_fips_mode_equals_independent_python_restatement(fips_oracle, pset):
    d, z, m = seeds("fips-d", 1, pset)[0], seeds("fips-z", 1, pset)[0], seeds("fips-m", 1, pset)[0]
    ek_p, dk_p = F.keygen(pset, bytes(d), bytes(z))
    c_p, K_p = F.encaps(pset, ek_p, bytes(m))
    assert F.decaps(pset, dk_p, c_p) == K_p
    ek, dk = fips_oracle.keygen(pset, d, z)
    assert bytes(ek[0]) == ek_p and bytes(dk[0]) == dk_p
    c, K = fips_oracle.encaps(pset, ek, m)
    assert bytes(c[0]) == c_p and bytes(K[0]) == K_p
    cb = c.copy()
    cb[0, 11] ^= 8
    Kr, st = fips_oracle.decaps(pset, dk, cb)
    assert st[0] == 0 and bytes(Kr[0]) == F.decaps(pset, dk_p, bytes(cb[0])) != K_p
    # modulus check (FIPS 203 7.2): a coefficient >= q is rejected in FIPS mode, accepted by the reference (F3)
    bad = ek[0].copy()
    bad[0], bad[1] = 0xFF, bad[1] | 0x0F
    assert not F.modulus_check(pset, bytes(bad)) and F.modulus_check(pset, ek_p)
    assert fips_oracle.kem_encaps_check(pset, bad) == -4 and fips_oracle.kem_encaps_check(pset, ek[0]) == 0


def test_fips_mode_differs_from_reference_mode(oracle):
    d, z = seeds("fips-d", 1, 1), seeds("fips-z", 1, 1)
    ek_ref, _ = oracle.keygen(768, d, z)
    oracle.set_conformance(1)
    try:
        ek_fips, _ = oracle.keygen(768, d, z)
    finally:
        oracle.set_conformance(0)
    assert not (ek_ref == ek_fips).all()
    assert (ek_ref[0, -32:] == ek_fips[0, -32:]).all()   # rho = G(d || k)[:32] does not depend on the PRF


@pytest.mark.parametrize("pset", (512, 768, 1024))
def test_emulated_kernels_fips_mode(fips_oracle, pset):
    emu = C.CDLL(ge.build_emulator())
    p8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))   # noqa: E731
    ekl, dkl, cl = SIZES[pset]
    n = 4
    d, z, m = seeds("fe-d", n, pset), seeds("fe-z", n, pset), seeds("fe-m", n, pset)
    emu.emu_conformance(1)
    try:
        ek, dk = np.zeros((n, ekl), np.uint8), np.zeros((n, dkl), np.uint8)
        assert emu.emu_keygen(pset, C.c_size_t(n), p8(d), p8(z), p8(ek), p8(dk)) == 0
        ek_o, dk_o = fips_oracle.keygen(pset, d, z)
        assert (ek == ek_o).all() and (dk == dk_o).all()
        bad = ek.copy()
        bad[2, 383] = 0xFF   # last coefficient of t-hat[0] = 0xFF? -> >= q
        bad[2, 382] |= 0xF0
        c, K, st = np.zeros((n, cl), np.uint8), np.zeros((n, 32), np.uint8), np.ones(n, np.int32)
        assert emu.emu_encaps(pset, C.c_size_t(n), p8(bad), p8(m), p8(c), p8(K), st.ctypes.data_as(C.POINTER(C.c_int32))) == 0
        assert st.tolist() == [0, 0, -4, 0]
        c_o, K_o = fips_oracle.encaps(pset, ek_o, m)
        ok = [0, 1, 3]
        assert (c[ok] == c_o[ok]).all() and (K[ok] == K_o[ok]).all()
        cb = c_o.copy()
        cb[1, 3] ^= 1
        Kd, sd = np.zeros((n, 32), np.uint8), np.ones(n, np.int32)
        assert emu.emu_decaps(pset, C.c_size_t(n), p8(dk), p8(cb), p8(Kd), sd.ctypes.data_as(C.POINTER(C.c_int32)), 1) == 0
        Kd_o, sd_o = fips_oracle.decaps(pset, dk_o, cb)
        assert (Kd == Kd_o).all() and (sd == sd_o).all() and not (Kd[1] == K_o[1]).all()
    finally:
        emu.emu_conformance(0)


@pytest.mark.gpu
@pytest.mark.parametrize("path", ("chunked", "lane-sliced", "one-workgroup-per-item"))
@pytest.mark.parametrize("pset", (512, 768, 1024))
def test_gpu_fips_mode(fips_oracle, pset, path, monkeypatch):
    """FIPS 203 mode on every kernel family: `chunked` = the default switches with 96-item chunks (300 items: one sponge per
    wave for the hashes, the direct sampler per chunk), `lane-sliced` = the full-batch kernels every call of more than 2048
    items takes (k_hash_decaps<.., 136>, k_sample_main at rate 136), `one-workgroup-per-item` = mlkem_small.hpp."""
    import torch
    pkg = ge.load_package()
    chunk = 96
    if path == "lane-sliced":
        monkeypatch.setenv("MLKEM_WIDE_HASH_ITEMS", "0")
        monkeypatch.setenv("MLKEM_SMALL_ITEMS", "0")
    elif path == "one-workgroup-per-item":
        monkeypatch.setenv("MLKEM_SMALL_ITEMS", "100000")
        chunk = 512
    e = pkg.MLKEM(pset, device=0, chunk_items=chunk, conformance="fips203")
    n = 300
    d, z, m = seeds("fg-d", n, pset), seeds("fg-z", n, pset), seeds("fg-m", n, pset)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
    ek, dk = e.keygen(dev(d), dev(z))
    ek_o, dk_o = fips_oracle.keygen(pset, d, z)
    assert (ek.cpu().numpy() == ek_o).all() and (dk.cpu().numpy() == dk_o).all()
    bad = ek_o.copy()
    bad[7, 0], bad[7, 1] = 0xFF, bad[7, 1] | 0x0F
    c, K, st = e.encaps(dev(bad), dev(m), return_status=True)
    st = st.cpu().numpy()
    assert st[7] == -4 and (np.delete(st, 7) == 0).all()
    c_o, K_o = fips_oracle.encaps(pset, ek_o, m)
    keep = np.arange(n) != 7
    assert (c.cpu().numpy()[keep] == c_o[keep]).all() and (K.cpu().numpy()[keep] == K_o[keep]).all()
    cb = c_o.copy()
    cb[::9, 40] ^= 2
    Kd, sd = e.decaps(dk, dev(cb))
    Kd_o, sd_o = fips_oracle.decaps(pset, dk_o, cb)
    assert (Kd.cpu().numpy() == Kd_o).all() and (sd.cpu().numpy() == sd_o).all()
    # item 0 against the independent Python restatement directly
    ek_p, dk_p = F.keygen(pset, bytes(d[0]), bytes(z[0]))
    assert bytes(ek.cpu().numpy()[0]) == ek_p and bytes(dk.cpu().numpy()[0]) == dk_p
    # shared-key batches in FIPS mode (PRF and J on SHAKE256 there too): key 0 for everything
    c_s, K_s = e.encaps_shared(ek[0], dev(m))
    c_so, K_so = fips_oracle.encaps(pset, np.repeat(ek_o[:1], n, axis=0), m)
    assert (c_s.cpu().numpy() == c_so).all() and (K_s.cpu().numpy() == K_so).all()
    cs = c_so.copy()
    cs[::5, 3] ^= 1
    Kd_s, st_s = e.decaps_shared(dk[0], dev(cs))
    Kd_so, st_so = fips_oracle.decaps(pset, np.repeat(dk_o[:1], n, axis=0), cs)
    assert (Kd_s.cpu().numpy() == Kd_so).all() and (st_s.cpu().numpy() == st_so).all()
    e.close()
    # the default (reference-compatible) mode reports status 0 for the same bad key (F3)
    r = pkg.MLKEM(pset, device=0)
    _, _, st_ref = r.encaps(dev(bad[:16]), dev(m[:16]), return_status=True)
    assert (st_ref.cpu().numpy() == 0).all()
    r.close()
