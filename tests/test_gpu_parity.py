"""GPU tier (-m gpu): the HIP path, called through the C-ABI (ctypes -> libmlkem_amd.so), against the oracle on
seeded inputs, against the committed golden vectors (generated from the real reference), and — at BASELINE's
full batch sizes — through size-independent properties.  Integer/byte work: the bar is bit-exact."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import __graft_entry__ as ge
from conftest import expand_seed, seeds, sha256, unhex
from oracle.loader import SIZES

pytestmark = pytest.mark.gpu
SETS = (512, 768, 1024)


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tier needs a HIP device"
    return torch


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    p.load_library()   # fails loudly if the HIP extension is missing
    return p


@pytest.fixture(scope="module")
def engines(pkg, torch):
    return {s: pkg.MLKEM(s, device=0) for s in SETS}


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def as_u16(t):
    return host(t).view(np.uint16)


# ---- primitives ------------------------------------------------------------------------------------------
def test_g1_ntt_test08_fixture(engines, torch, golden, golden_npz):
    """BASELINE config 2 anchor: Test_Archive/NTT_test08.c recipe (B[i] = 2i)."""
    e = engines[768]
    g = golden["G1_test08"]
    f1 = as_u16(e.sample_ntt(dev(torch, unhex(g["B"])[None])))[0]
    assert f1[:8].tolist() == g["f1_first8"] and sha256(f1) == g["f1_sha256"]
    fh = as_u16(e.ntt(dev(torch, f1.view(np.int16)[None])))[0]
    assert fh[:8].tolist() == g["fh_first8"] and sha256(fh) == g["fh_sha256"]
    f2 = as_u16(e.intt(dev(torch, fh.view(np.int16)[None])))[0]
    assert (f2 == f1).all()


def test_ntt_intt_multiply_golden(engines, torch, golden_npz):
    e = engines[768]
    a, b = golden_npz["rand_a"], golden_npz["rand_b"]
    da, db = dev(torch, a.view(np.int16)), dev(torch, b.view(np.int16))
    assert (as_u16(e.ntt(da)) == golden_npz["rand_a_ntt"]).all()
    assert (as_u16(e.intt(da)) == golden_npz["rand_a_intt"]).all()
    assert (as_u16(e.multiply_ntts(da, db)) == golden_npz["rand_ab_mul"]).all()
    na, nb = golden_npz["nc_a"], golden_npz["nc_b"]   # raw 12-bit inputs as ByteDecode_12 yields (F3)
    assert (as_u16(e.multiply_ntts(dev(torch, na.view(np.int16)), dev(torch, nb.view(np.int16)))) == golden_npz["nc_ab_mul"]).all()


def test_ntt_random_vs_oracle_and_ragged_sizes(engines, torch, oracle):
    e = engines[768]
    rng = np.random.default_rng(11)
    for n in (1, 3, 4, 5, 63, 257, 4099):
        a = rng.integers(0, 3329, (n, 256)).astype(np.uint16)
        da = dev(torch, a.view(np.int16))
        assert (as_u16(e.ntt(da)) == oracle.ntt(a)).all(), n
        assert (as_u16(e.intt(da)) == oracle.intt(a)).all(), n
    assert e.ntt(torch.empty((0, 256), dtype=torch.int16, device="cuda")).shape[0] == 0   # empty batch


def test_sample_ntt_and_cbd_golden(engines, torch, golden_npz, oracle):
    e = engines[768]
    assert (as_u16(e.sample_ntt(dev(torch, golden_npz["g2_in"]))) == golden_npz["g2_out"]).all()
    for eta in (2, 3):
        assert (as_u16(e.sample_cbd(dev(torch, golden_npz[f"g3_eta{eta}_in"]), eta)) == golden_npz[f"g3_eta{eta}_out"]).all()
    rng = np.random.default_rng(12)
    s = rng.integers(0, 256, (1000, 34)).astype(np.uint8)
    got = as_u16(e.sample_ntt(dev(torch, s)))
    for i in range(1000):
        assert (got[i] == oracle.sample_ntt(s[i])).all(), i
    assert got.max() < 3329


def test_prf_and_hashes_vs_oracle_and_golden(engines, torch, golden, oracle):
    e = engines[768]
    for p in golden["G5_prf"]:
        x = np.concatenate([unhex(p["s"]), [p["b"]]]).astype(np.uint8)[None]
        for eta in (2, 3):
            assert bytes(host(e.prf(dev(torch, x), eta))[0]).hex() == p[f"eta{eta}"]
    for h in golden["G5_hashes"]:
        x = unhex(h["in"])
        msgs = dev(torch, np.tile(x, (3, 1))) if x.size else torch.empty((3, 0), dtype=torch.uint8, device="cuda")
        assert bytes(host(e.H(msgs))[2]).hex() == h["H"]
        assert bytes(host(e.G(msgs))[1]).hex() == h["G"]
        assert bytes(host(e.J(msgs))[0]).hex() == h["J"]
    rng = np.random.default_rng(13)
    for ln in (4, 7, 12, 100, 132, 136, 137, 140, 168, 172, 800, 1120, 1184, 1600):   # % 4 == 0 and full waves: LDS-DMA staging
        m = rng.integers(0, 256, (130, ln)).astype(np.uint8)
        H, G, J = host(e.H(dev(torch, m))), host(e.G(dev(torch, m))), host(e.J(dev(torch, m)))
        for i in (0, 63, 64, 129):
            assert (H[i] == oracle.H(m[i])).all() and (G[i] == oracle.G(m[i])).all() and (J[i] == oracle.J(m[i])).all()


# ---- full KEM ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pset", SETS)
def test_g6_recipe_and_seeded_triples(engines, torch, golden, pset):
    """Golden vectors produced by the real reference (ml_kem.c) in the build container."""
    e = engines[pset]
    g = golden["G6_kem"][str(pset)]
    rec, trip = g["recipe"], g["seeded"]
    n = len(trip)
    d = np.stack([unhex(rec["d"])] + [np.frombuffer(expand_seed("mlkem-golden-d", i, 0x203), np.uint8) for i in range(n)])
    z = np.stack([unhex(rec["z"])] + [np.frombuffer(expand_seed("mlkem-golden-z", i, 0x203), np.uint8) for i in range(n)])
    m = np.stack([unhex(rec["m"])] + [np.frombuffer(expand_seed("mlkem-golden-m", i, 0x203), np.uint8) for i in range(n)])
    ek, dk = e.keygen(dev(torch, d), dev(torch, z))
    c, K = e.encaps(ek, dev(torch, m))
    Kd, st = e.decaps(dk, c)
    ekh, dkh, ch, Kh = host(ek), host(dk), host(c), host(K)
    assert bytes(ekh[0]).hex() == rec["ek"] and bytes(dkh[0]).hex() == rec["dk"]
    assert bytes(ch[0]).hex() == rec["c"] and bytes(Kh[0]).hex() == rec["K"]
    assert (host(st) == 0).all() and (host(Kd) == Kh).all()
    cb = ch.copy()
    cb[0, 5] ^= 1
    for t in trip:
        i = t["i"] + 1
        assert sha256(ekh[i]) == t["ek_sha256"] and sha256(dkh[i]) == t["dk_sha256"]
        assert sha256(ch[i]) == t["c_sha256"] and bytes(Kh[i]).hex() == t["K"]
        cb[i, t["tamper_pos"]] ^= t["tamper_mask"]
    Kr, st = e.decaps(dk, dev(torch, cb))
    Kr = host(Kr)
    assert (host(st) == 0).all()
    assert bytes(Kr[0]).hex() == rec["K_reject_c5_xor1"]
    for t in trip:
        assert bytes(Kr[t["i"] + 1]).hex() == t["K_reject"]
    # PKE_EncryptDecrypt_test.c recipe is covered through Decaps_internal == KEM_Decaps on valid keys
    assert (host(e.Decaps_internal(dk, c)) == Kh).all()


@pytest.mark.parametrize("pset", SETS)
def test_g7_negative_paths(engines, torch, golden, pset):
    e = engines[pset]
    g = golden["G6_kem"][str(pset)]
    g7, rec = g["G7"], g["recipe"]
    ekl, dkl, cl = SIZES[pset]
    # F3: an ek with a coefficient >= q is accepted and processed exactly like the reference does
    c, K = e.encaps(dev(torch, unhex(g7["bad_ek"])[None]), dev(torch, unhex(g7["bad_ek_m"])[None]))
    assert sha256(host(c)[0]) == g7["bad_ek_c_sha256"] and bytes(host(K)[0]).hex() == g7["bad_ek_K"]
    dk, cc = unhex(rec["dk"]), unhex(rec["c"])
    dks = np.stack([dk, dk, dk, dk])
    dks[1, (ekl - 32) + 7] ^= 0x10     # embedded ek corrupted -> -5
    dks[2, dkl - 64] ^= 1              # stored H(ek) corrupted -> -5
    dks[3, 3] ^= 0x40                  # dk_pke corrupted: not covered by the hash -> implicit rejection
    Kd, st = e.decaps(dev(torch, dks), dev(torch, np.stack([cc] * 4)))
    st, Kd = host(st), host(Kd)
    assert st.tolist() == [0, g7["errno_dk_hash_ek"], g7["errno_dk_hash_h"], g7["bad_dkpke_errno"]] == [0, -5, -5, 0]
    assert bytes(Kd[0]).hex() == rec["K"] and bytes(Kd[3]).hex() == g7["bad_dkpke_K"]
    # length (type) checks: reference ml_errno -3
    with pytest.raises(Exception) as ex:
        e.encaps(dev(torch, unhex(rec["ek"])[None, :-1]), dev(torch, unhex(rec["m"])[None]))
    assert getattr(ex.value, "code", None) == -3 == g7["errno_ek_len"]
    with pytest.raises(Exception) as ex:
        e.decaps(dev(torch, dk[None]), dev(torch, cc[None, :-1]))
    assert getattr(ex.value, "code", None) == -3 == g7["errno_c_len"]
    with pytest.raises(Exception) as ex:
        e.decaps(dev(torch, dk[None, :-1]), dev(torch, cc[None]))
    assert getattr(ex.value, "code", None) == -3 == g7["errno_dk_len"]


@pytest.mark.parametrize("pset", SETS)
@pytest.mark.parametrize("n", (1, 63, 64, 65, 300))
def test_kem_vs_oracle_ragged_batches(engines, torch, oracle, pset, n):
    e = engines[pset]
    d, z, m = seeds("gpu-d", n, pset + n), seeds("gpu-z", n, pset + n), seeds("gpu-m", n, pset + n)
    ek, dk = e.keygen(dev(torch, d), dev(torch, z))
    c, K = e.encaps(ek, dev(torch, m))
    ek_o, dk_o = oracle.keygen(pset, d, z)
    c_o, K_o = oracle.encaps(pset, ek_o, m)
    assert (host(ek) == ek_o).all() and (host(dk) == dk_o).all()
    assert (host(c) == c_o).all() and (host(K) == K_o).all()
    cb = c_o.copy()
    cb[::3, (np.arange(0, n, 3) * 37) % SIZES[pset][2]] ^= 0x04
    Kd, st = e.decaps(dk, dev(torch, cb))
    Kd_o, st_o = oracle.decaps(pset, dk_o, cb)
    assert (host(st) == st_o).all() and (host(Kd) == Kd_o).all()


@pytest.mark.parametrize("d", (1, 4, 5, 10, 11, 12))
def test_codec_primitives_vs_reference_tables(engines, torch, oracle, golden_npz, d):
    """SURVEY 8a rows a4/a5 as stand-alone entry points: every x in [0, q) through Compress_d + ByteEncode_d and every
    y in [0, 2^d) through ByteDecode_d + Decompress_d, against the reference's full tables (golden G4)."""
    from test_emulated_kernels import _codec_cases
    xs, enc_want, dec_in, dec_want = _codec_cases(oracle, golden_npz, d)
    e = engines[768]
    assert (host(e.compress_encode(dev(torch, xs.view(np.int16)), d)) == enc_want).all()
    assert (as_u16(e.decode_decompress(dev(torch, dec_in), d)) == dec_want).all()
    with pytest.raises(Exception):
        e.compress_encode(dev(torch, xs.view(np.int16)), 7)


@pytest.mark.parametrize("pset", SETS)
def test_k_pke_alone_vs_oracle(engines, torch, oracle, pset):
    """SURVEY 8a rows a21-a23 as stand-alone entry points (PKE_KeyGen / PKE_Encrypt with the caller's randomness /
    PKE_Decrypt); 150 items = two full waves + a ragged one for the lane-per-item hash stage of PKE_KeyGen."""
    e = engines[pset]
    n = 150
    d, m, r = seeds("pke-d", n, pset), seeds("pke-m", n, pset), seeds("pke-r", n, pset)
    ek, dkp = e.PKE_KeyGen(dev(torch, d))
    c = e.PKE_Encrypt(ek, dev(torch, m), dev(torch, r))
    m2 = e.PKE_Decrypt(dkp, c)
    ekh, dkh, ch = host(ek), host(dkp), host(c)
    for i in (0, 1, 63, 64, 127, 128, 149):
        ek_o, dk_o = oracle.pke_keygen(pset, d[i])
        assert (ekh[i] == ek_o).all() and (dkh[i] == dk_o).all()
        assert (ch[i] == oracle.pke_encrypt(pset, ek_o, m[i], r[i])).all()
    assert (host(m2) == m).all()
    # the KEM layers on top: dk = dk_pke || ek || H(ek) || z and c = Encrypt(ek, m, r) with (K, r) = G(m || H(ek))
    z = seeds("pke-z", n, pset)
    ek2, dk2 = e.keygen(dev(torch, d), dev(torch, z))
    assert (host(ek2) == ekh).all() and (host(dk2)[:, : 384 * e.k] == dkh).all()


def test_random_shapes_vs_oracle(pkg, torch, oracle):
    """Seeded random cases: parameter set, batch size, chunk size (chunk / h-chunk / wave boundaries fall anywhere),
    tampered ciphertext positions, corrupted dk hashes -- every output byte against the oracle."""
    rng = np.random.default_rng(20260401)
    for case in range(10):
        pset = int(rng.choice(SETS))
        n = int(rng.integers(1, 600))
        chunk = int(rng.choice([64, 100, 192, 1000]))
        ekl, dkl, cl = SIZES[pset]
        e = pkg.MLKEM(pset, device=0, chunk_items=chunk)
        d, z, m = seeds(f"rnd-d{case}", n, pset), seeds(f"rnd-z{case}", n, pset), seeds(f"rnd-m{case}", n, pset)
        ek, dk = e.keygen(dev(torch, d), dev(torch, z))
        c, K = e.encaps(ek, dev(torch, m))
        cb, dkb = host(c).copy(), host(dk).copy()
        bad_c = rng.choice(n, size=max(1, n // 7), replace=False)
        cb[bad_c, rng.integers(0, cl, bad_c.size)] ^= rng.integers(1, 256, bad_c.size).astype(np.uint8)
        bad_h = rng.choice(n, size=max(1, n // 11), replace=False)
        dkb[bad_h, dkl - 64 + rng.integers(0, 32, bad_h.size)] ^= 0x40
        Kd, st = e.decaps(dev(torch, dkb), dev(torch, cb))
        ek_o, dk_o = oracle.keygen(pset, d, z)
        c_o, K_o = oracle.encaps(pset, ek_o, m)
        Kd_o, st_o = oracle.decaps(pset, dkb, cb)
        assert (host(ek) == ek_o).all() and (host(dk) == dk_o).all(), (case, pset, n, chunk)
        assert (host(c) == c_o).all() and (host(K) == K_o).all(), (case, pset, n, chunk)
        sth = host(st)
        assert (sth == st_o).all() and set(np.nonzero(sth)[0]) == set(bad_h.tolist()), (case, pset, n, chunk)
        ok = st_o == 0
        assert (host(Kd)[ok] == Kd_o[ok]).all(), (case, pset, n, chunk)
        e.close()


@pytest.mark.parametrize("pset", SETS)
def test_shared_key_batches(pkg, torch, oracle, pset):
    """One ek / dk for the whole batch: bytes equal to the per-item path on the replicated key (oracle on a sample, the
    per-item GPU path on everything), implicit rejection, once-per-call hash check, chunk boundaries crossed."""
    ekl, dkl, cl = SIZES[pset]
    e = pkg.MLKEM(pset, device=0, chunk_items=700)
    n = 3000
    d, z, m = seeds("shg-d", 1, pset), seeds("shg-z", 1, pset), seeds("shg-m", n, pset)
    ek1, dk1 = e.keygen(dev(torch, d), dev(torch, z))
    c, K = e.encaps_shared(ek1, dev(torch, m))
    c_r, K_r = e.encaps(ek1.expand(n, ekl).contiguous(), dev(torch, m))
    assert torch.equal(c, c_r) and torch.equal(K, K_r)
    idx = [0, 1, 63, 64, 699, 700, 2999]
    c_o, K_o = oracle.encaps(pset, np.repeat(host(ek1), len(idx), axis=0), m[idx])
    assert (host(c)[idx] == c_o).all() and (host(K)[idx] == K_o).all()
    cb = c.clone()
    bad = torch.arange(0, n, 37, device="cuda")
    cb[bad, (bad * 7) % cl] ^= 0x20
    Kd, st = e.decaps_shared(dk1, cb)
    Kd_r, st_r = e.decaps(dk1.expand(n, dkl).contiguous(), cb)
    assert torch.equal(Kd, Kd_r) and torch.equal(st, st_r) and int(st.abs().max()) == 0
    same = (Kd == K).all(dim=1)
    assert not bool(same[bad].any()) and int(same.sum()) == n - bad.numel()
    dkb = dk1.clone()
    dkb[0, dkl - 40] ^= 2
    Kd2, st2 = e.decaps_shared(dkb, cb)
    Kd2_r, st2_r = e.decaps(dkb.expand(n, dkl).contiguous(), cb)
    assert bool((st2 == -5).all()) and torch.equal(st2, st2_r) and torch.equal(Kd2, Kd2_r)
    e.close()


def test_empty_batches(engines, torch):
    e = engines[768]
    u8 = torch.uint8
    ek, dk = e.keygen(torch.empty((0, 32), dtype=u8, device="cuda"), torch.empty((0, 32), dtype=u8, device="cuda"))
    assert ek.shape == (0, 1184) and dk.shape == (0, 2400)
    c, K = e.encaps(ek, torch.empty((0, 32), dtype=u8, device="cuda"))
    assert c.shape == (0, 1088) and K.shape == (0, 32)


def test_chunked_execution_equals_single_chunk(pkg, torch, oracle):
    """The engine processes batches in chunks of ctx.chunk items through one scratch area."""
    small = pkg.MLKEM(768, device=0, chunk_items=100)
    n = 333
    d, z, m = seeds("chunk-d", n, 1), seeds("chunk-z", n, 1), seeds("chunk-m", n, 1)
    ek, dk = small.keygen(dev(torch, d), dev(torch, z))
    c, K = small.encaps(ek, dev(torch, m))
    Kd, st = small.decaps(dk, c)
    ek_o, dk_o = oracle.keygen(768, d, z)
    c_o, K_o = oracle.encaps(768, ek_o, m)
    assert (host(ek) == ek_o).all() and (host(dk) == dk_o).all() and (host(c) == c_o).all()
    assert (host(K) == K_o).all() and (host(Kd) == K_o).all() and (host(st) == 0).all()
    small.close()


# ---- full-size properties (BASELINE configs 2-4) -------------------------------------------------------------
def test_config2_ntt_roundtrip_full_batch(engines, torch):
    """2^20 polynomials: InverseNTT(NTT(f)) == f, outputs canonical, linearity NTT(a+b) = NTT(a)+NTT(b) mod q."""
    e = engines[768]
    n = 1 << 20
    g = torch.Generator(device="cuda").manual_seed(2)
    a = torch.randint(0, 3329, (n, 256), generator=g, device="cuda", dtype=torch.int16)
    b = torch.randint(0, 3329, (n, 256), generator=g, device="cuda", dtype=torch.int16)
    ah, bh = e.ntt(a), e.ntt(b)
    assert int(ah.min()) >= 0 and int(ah.max()) < 3329
    assert torch.equal(e.intt(ah), a)
    s = ((a.int() + b.int()) % 3329).short()
    assert torch.equal(e.ntt(s), ((ah.int() + bh.int()) % 3329).short())


@pytest.mark.parametrize("pset,n", ((768, 1 << 20), (1024, 1 << 20), (512, 1 << 20)))
def test_config3_4_full_batch_roundtrip(pkg, torch, oracle, pset, n):
    """BASELINE configs[2] / configs[3] at their stated batch (2^20): K_encaps == K_decaps for every item, every tampered
    ciphertext is rejected (one per 1024), and a fixed 1024-item subset (SURVEY 8d's gate) is compared byte-for-byte
    (ek, dk, c, K, and the implicit-rejection keys of its tampered members) with the oracle."""
    e = pkg.MLKEM(pset, device=0)
    g = torch.Generator(device="cuda").manual_seed(pset)
    d, z, m = (torch.randint(0, 256, (n, 32), generator=g, device="cuda", dtype=torch.uint8) for _ in range(3))
    ek, dk = e.keygen(d, z)
    c, K = e.encaps(ek, m)
    Kd, st = e.decaps(dk, c)
    assert int(st.abs().max()) == 0 and torch.equal(Kd, K)
    ct = c.clone()
    idx = torch.arange(0, n, 1024, device="cuda")
    ct[idx, (idx * 7) % c.shape[1]] ^= 1
    Kt, st = e.decaps(dk, ct)
    same = (Kt == K).all(dim=1)
    assert int(st.abs().max()) == 0 and not bool(same[idx].any()) and int(same.sum()) == n - idx.numel()
    sub = torch.arange(0, n, n // 1024, device="cuda")[:1024]          # stride 1024 = the tamper stride: all of these are tampered
    sub = torch.cat([sub[:512], sub[512:] + 513])                       # ... so shift half of them: 512 tampered, 512 intact
    ek_o, dk_o = oracle.keygen(pset, host(d[sub]), host(z[sub]))
    c_o, K_o = oracle.encaps(pset, ek_o, host(m[sub]))
    assert (host(ek[sub]) == ek_o).all() and (host(dk[sub]) == dk_o).all()
    assert (host(c[sub]) == c_o).all() and (host(K[sub]) == K_o).all()
    Kt_o, st_o = oracle.decaps(pset, dk_o, host(ct[sub]))
    assert (host(Kt[sub]) == Kt_o).all() and (st_o == 0).all()
    assert int((Kt_o != K_o).any(axis=1).sum()) == 512                 # the tampered half took the implicit-rejection path
    del ek, dk, c, ct, K, Kd, Kt
    e.close()
    torch.cuda.empty_cache()


# ---- host-pointer C-ABI and the ml_kem.h drop-in shim ---------------------------------------------------------
def test_host_pointer_abi(pkg, torch, oracle):
    lib = pkg.load_library()
    n = 70
    d, z, m = seeds("host-d", n, 5), seeds("host-z", n, 5), seeds("host-m", n, 5)
    ek, dk = np.zeros((n, 1184), np.uint8), np.zeros((n, 2400), np.uint8)
    assert lib.mlkem_keygen(768, n, d.ctypes.data, z.ctypes.data, ek.ctypes.data, dk.ctypes.data) == 0
    c, K = np.zeros((n, 1088), np.uint8), np.zeros((n, 32), np.uint8)
    assert lib.mlkem_encaps(768, n, ek.ctypes.data, m.ctypes.data, c.ctypes.data, K.ctypes.data) == 0
    Kd, st = np.zeros((n, 32), np.uint8), np.ones(n, np.int32)
    assert lib.mlkem_decaps(768, n, dk.ctypes.data, c.ctypes.data, Kd.ctypes.data, st.ctypes.data) == 0
    ek_o, dk_o = oracle.keygen(768, d, z)
    c_o, K_o = oracle.encaps(768, ek_o, m)
    assert (ek == ek_o).all() and (dk == dk_o).all() and (c == c_o).all() and (K == K_o).all()
    assert (Kd == K).all() and (st == 0).all()
    f = np.random.default_rng(1).integers(0, 3329, (9, 256)).astype(np.uint16)
    fh, f2 = np.zeros_like(f), np.zeros_like(f)
    assert lib.mlkem_ntt(9, f.ctypes.data, fh.ctypes.data) == 0 and lib.mlkem_intt(9, fh.ctypes.data, f2.ctypes.data) == 0
    assert (fh == oracle.ntt(f)).all() and (f2 == f).all()
    # randomised wrappers: KEM_KeyGen / KEM_Encaps semantics at batch scale
    assert lib.mlkem_keygen_random(768, n, ek.ctypes.data, dk.ctypes.data) == 0
    assert lib.mlkem_encaps_random(768, n, ek.ctypes.data, 1184, c.ctypes.data, K.ctypes.data) == 0
    assert lib.mlkem_encaps_random(768, n, ek.ctypes.data, 1183, c.ctypes.data, K.ctypes.data) == -3
    Ko, sto = oracle.decaps(768, dk, c)
    assert (sto == 0).all() and (Ko == K).all()
    # large host batches are routed through the streaming front-end: same bytes as the device-resident path
    big = 33000
    e = pkg.MLKEM(512, device=0)
    db, zb, mb = seeds("host-bd", big, 5), seeds("host-bz", big, 5), seeds("host-bm", big, 5)
    ekb, dkb = np.zeros((big, 800), np.uint8), np.zeros((big, 1632), np.uint8)
    assert lib.mlkem_keygen(512, big, db.ctypes.data, zb.ctypes.data, ekb.ctypes.data, dkb.ctypes.data) == 0
    cb, Kb = np.zeros((big, 768), np.uint8), np.zeros((big, 32), np.uint8)
    assert lib.mlkem_encaps(512, big, ekb.ctypes.data, mb.ctypes.data, cb.ctypes.data, Kb.ctypes.data) == 0
    cb[::1000, 3] ^= 1
    Kdb = np.zeros((big, 32), np.uint8)
    assert lib.mlkem_decaps(512, big, dkb.ctypes.data, cb.ctypes.data, Kdb.ctypes.data, None) == 0
    ek_d, dk_d = e.keygen(dev(torch, db), dev(torch, zb))
    cb[::1000, 3] ^= 1
    c_d, K_d = e.encaps(ek_d, dev(torch, mb))
    assert (host(ek_d) == ekb).all() and (host(dk_d) == dkb).all() and (host(c_d) == cb).all() and (host(K_d) == Kb).all()
    same = (Kdb == Kb).all(axis=1)
    assert not same[::1000].any() and same.sum() == big - len(range(0, big, 1000))
    lib.mlkem_stream_release()
    e.close()


SHIM_TEST_C = r"""
/* A program written against the reference's ml_kem.h API (same flow as Test_Archive/EncapsDecaps_test.c and
 * KeyGen_test.c, but with the correct ek_len), linked against the drop-in shim. */
#include "mlkem_compat.h"
#include <stdio.h>
#include <stdlib.h>
int main(void) {
    int sets[3] = {ML_KEM_512, ML_KEM_768, ML_KEM_1024};
    for (int s = 0; s < 3; s++) {
        struct PARAMS params = init(sets[s]);
        if (ml_errno != 0) return 10;
        struct PKE keys = KEM_KeyGen(&params);
        if (ml_errno != 0) return 11;
        struct KEM kem = KEM_Encaps(&params, keys.ek, keys.ek_len);
        if (ml_errno != 0) return 12;
        /* poison the upper 24 bits of every cell: the shim must never read them (SURVEY F1) */
        for (unsigned i = 0; i < keys.dk_len; i++) ((unsigned*)keys.dk)[i] |= 0xABCDEF00u;
        for (unsigned i = 0; i < kem.c_len; i++) ((unsigned*)kem.c)[i] |= 0x12345600u;
        union byte* K = KEM_Decaps(&params, keys.dk, keys.dk_len, kem.c, kem.c_len);
        if (ml_errno != 0 || !K) return 13;
        for (int i = 0; i < 32; i++) if (kem.K[i].e != K[i].e) return 14;
        printf("ML-KEM-%d ek_len=%u dk_len=%u c_len=%u K=", sets[s], keys.ek_len, keys.dk_len, kem.c_len);
        for (int i = 0; i < 32; i++) printf("%02x", K[i].e);
        printf("\n");
        /* error paths: ek_len = 1 (what EncapsDecaps_test.c passes) -> -3 ; corrupted dk hash -> -5, NULL */
        (void)KEM_Encaps(&params, keys.ek, 1);
        if (ml_errno != -3) return 15;
        ml_errno = 0;
        keys.dk[keys.dk_len - 64].e ^= 1;
        free(K);
        K = KEM_Decaps(&params, keys.dk, keys.dk_len, kem.c, kem.c_len);
        if (K != NULL || ml_errno != -5) return 16;
        ml_errno = 0;
        free(keys.ek); free(keys.dk); free(kem.c);
    }
    (void)init(999);
    if (ml_errno != -1) return 17;
    printf("Test successful\n");
    return 0;
}
"""


def test_ml_kem_h_dropin_shim(pkg, tmp_path):
    """BASELINE config[0] flow (KeyGen+Encaps+Decaps round trip through the public ml_kem.h API), against the shim."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "shim_test.c"
    src.write_text(SHIM_TEST_C)
    exe = tmp_path / "shim_test"
    libdir = os.path.dirname(pkg.SHIM_PATH)
    subprocess.run(["gcc", "-O1", "-I", os.path.join(root, "include"), str(src), "-o", str(exe), "-L", libdir, "-lml_kem",
                    "-lmlkem_amd", f"-Wl,-rpath,{libdir}"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "Test successful" in r.stdout and r.stdout.count("ML-KEM-") == 3
    assert "Type check failed" in r.stderr and "Hash check failed" in r.stderr


PRIM_TEST_C = r"""
/* The recipes of the reference's own primitive tests (Test_Archive/SampleNTT_test06.c, SampleCBD_test07.c,
 * NTT_test08.c), written against the symbols ml_kem.o exports: SampleNTT / SamplePolyCBD / NTT / InverseNTT. */
#include "mlkem_compat.h"
#include <stdio.h>
#include <stdlib.h>
int main(void) {
    union byte B[192];
    /* test08: B[i] = 2 i ; f1 = SampleNTT(B) ; fh = NTT(f1) ; f2 = InverseNTT(fh) ; f1 == f2 */
    for (int i = 0; i < 34; i++) { B[i].e = i * 2; }
    B[3].s = B[3].s;                               /* both views of the cell alias bits 0.. */
    union integer* f1 = SampleNTT(B);
    union integer* fh = NTT(f1);
    union integer* f2 = InverseNTT(fh);
    if (!f1 || !fh || !f2) return 2;
    for (int i = 0; i < 256; i++) if (f1[i].t != f2[i].t) { printf("ERROR :: f1[%d] = %d :: f2[%d] = %d\n", i, f1[i].t, i, f2[i].t); return 3; }
    printf("f1"); for (int i = 0; i < 8; i++) printf(" %d", f1[i].t); printf("\n");
    printf("fh"); for (int i = 0; i < 8; i++) printf(" %d", fh[i].t); printf("\n");
    for (int i = 0; i < 34; i++) if (B[i].e != (unsigned)(i * 2)) return 4;   /* input untouched */
    free(f1); free(fh); free(f2);
    /* test06: B[i] = it*i + i, it = 0..6 */
    for (int it = 0; it < 7; it++) {
        for (int i = 0; i < 34; i++) B[i].e = it * i + i;
        union integer* a = SampleNTT(B);
        if (!a) return 5;
        printf("s%d", it); for (int i = 0; i < 256; i++) printf(" %d", a[i].t); printf("\n");
        free(a);
    }
    /* test07: B[i] = i, eta = 3 (and eta = 2) */
    for (int i = 0; i < 192; i++) B[i].e = i;
    for (unsigned eta = 3; eta >= 2; eta--) {
        union integer* f = SamplePolyCBD(B, eta);
        if (!f) return 6;
        printf("c%u", eta); for (int i = 0; i < 256; i++) printf(" %d", f[i].t); printf("\n");
        free(f);
    }
    if (SamplePolyCBD(B, 4) != NULL) return 7;     /* unsupported eta: message + ml_errno, NULL */
    printf("Test Complete!\n");
    return 0;
}
"""


def test_reference_primitive_symbols_through_the_shim(pkg, tmp_path, golden, golden_npz, oracle):
    """the reference's objects also export SampleNTT / SamplePolyCBD / NTT / InverseNTT (4-byte `union
    integer` cells) and its own test programs call them.  Same recipes, against the goldens G1-G3."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "prim_test.c"
    src.write_text(PRIM_TEST_C)
    exe = tmp_path / "prim_test"
    libdir = os.path.dirname(pkg.SHIM_PATH)
    subprocess.run(["gcc", "-O1", "-I", os.path.join(root, "include"), str(src), "-o", str(exe), "-L", libdir, "-lml_kem",
                    "-lmlkem_amd", f"-Wl,-rpath,{libdir}"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    rows = {ln.split()[0]: np.array(ln.split()[1:], dtype=np.int64) for ln in r.stdout.splitlines() if ln and ln[0] in "fsc"}
    assert rows["f1"].tolist() == [2931, 2112, 3266, 1044, 1856, 3090, 520, 2900]
    assert rows["fh"].tolist() == [192, 1622, 2207, 476, 2400, 883, 637, 686]
    for it in range(7):
        seed = np.array([(it * i + i) & 0xFF for i in range(34)], np.uint8)
        assert (rows[f"s{it}"] == oracle.sample_ntt(seed)).all(), it
    assert (rows["s1"][:8] == rows["f1"]).all()                                         # it = 1 is G1's f1
    b = np.arange(192, dtype=np.uint8)
    assert rows["c3"][:8].tolist() == [0, 1, 3328, 0, 2, 3328, 3328, 1]
    assert (rows["c3"] == oracle.sample_cbd(b[:192], 3)).all() and (rows["c2"] == oracle.sample_cbd(b[:128], 2)).all()
    assert "eta must be 2 or 3" in r.stderr and "Test Complete!" in r.stdout


@pytest.mark.parametrize("env", ({"MLKEM_HCHUNK_ITEMS": "300"}, {"MLKEM_CHUNK_ITEMS": "77", "MLKEM_HCHUNK_ITEMS": "154"}))
def test_engine_options_do_not_change_results(pkg, torch, oracle, env, monkeypatch):
    """The two sizing knobs read at context creation (chunk and h-chunk capacity): same bytes out."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    e = pkg.MLKEM(768, device=0, chunk_items=0 if "MLKEM_CHUNK_ITEMS" in env else 128)     # several chunks (and h-chunks) per call
    n = 700
    d, z, m = seeds("opt-d", n, 3), seeds("opt-z", n, 3), seeds("opt-m", n, 3)
    ek, dk = e.keygen(dev(torch, d), dev(torch, z))
    c, K = e.encaps(ek, dev(torch, m))
    cb = host(c).copy()
    cb[::50, 9] ^= 0x10
    Kd, st = e.decaps(dk, dev(torch, cb))
    ek_o, dk_o = oracle.keygen(768, d, z)
    c_o, K_o = oracle.encaps(768, ek_o, m)
    Kd_o, st_o = oracle.decaps(768, dk_o, cb)
    assert (host(ek) == ek_o).all() and (host(dk) == dk_o).all() and (host(c) == c_o).all() and (host(K) == K_o).all()
    assert (host(Kd) == Kd_o).all() and (host(st) == st_o).all()
    e.close()


def test_sha3_surface_nist_examples(pkg, engines, torch, golden):
    """SURVEY 8f row 2 on the GPU: (1) the batched sponge through the Python binding, (2) the reference-ABI front-ends
    sha3_b / sha3_h / sha3_s / h2b / b2h of the drop-in shim (4-byte cells), against the NIST examples."""
    import hashlib
    e = engines[768]
    for ex in golden["G8_nist_sha3"]:
        bits = [int(ch) for ch in ex["msg_bits"]]
        want = bytes.fromhex(ex["out"])
        got = host(e.sha3_bits([bits, bits], ex["xof"], ex["rate_bytes"], len(want)))
        assert bytes(got[1]) == want, ex["file"]
    shim = C.CDLL(pkg.SHIM_PATH)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    cell = C.c_uint
    shim.sha3_b.restype = C.POINTER(cell)
    shim.sha3_b.argtypes = [C.POINTER(cell), C.c_uint, C.c_uint, C.c_uint, C.POINTER(cell)]
    sfx_hash, sfx_xof = (cell * 4)(0, 1, 0, 0), (cell * 4)(1, 1, 1, 1)
    for ex in golden["G8_nist_sha3"]:
        bits = [int(ch) for ch in ex["msg_bits"]]
        want = bytes.fromhex(ex["out"])
        cells = (cell * max(1, len(bits)))(*[b | 0xABCD0000 for b in bits])   # garbage above bit 0 must be ignored
        cap = 1600 - 8 * ex["rate_bytes"]
        r = shim.sha3_b(cells, len(bits), 8 * len(want), cap, sfx_xof if ex["xof"] else sfx_hash)
        assert r, ex["file"]
        got = np.packbits(np.array([r[i] & 1 for i in range(8 * len(want))], np.uint8), bitorder="little")
        libc.free(r)
        assert bytes(got) == want, ex["file"]
    # character front-end (sha3_s) and the hex helpers
    shim.sha3_s.restype = C.POINTER(C.c_ubyte)
    shim.sha3_s.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_uint, C.POINTER(cell)]
    msg = b"MI355X batched ML-KEM"
    r = shim.sha3_s(msg, len(msg), 256, 512, sfx_hash)
    assert bytes(r[i] for i in range(32)) == hashlib.sha3_256(msg).digest()
    libc.free(r)
    r = shim.sha3_s(msg, len(msg), 8 * 200, 256, sfx_xof)
    assert bytes(r[i] for i in range(200)) == hashlib.shake_128(msg).digest(200)
    libc.free(r)
    shim.h2b.restype = C.POINTER(cell)
    shim.h2b.argtypes = [C.POINTER(cell), C.c_uint, C.c_uint]
    shim.b2h.restype = C.POINTER(cell)
    shim.b2h.argtypes = [C.POINTER(cell), C.c_uint]
    hx = (cell * 4)(0xA, 0x3, 0x0, 0xF)          # bytes A3 0F
    b = shim.h2b(hx, 2, 13)
    assert [b[i] & 1 for i in range(13)] == [1, 1, 0, 0, 0, 1, 0, 1, 1, 1, 1, 1, 0]   # LSB-first within each byte
    h = shim.b2h(b, 13)
    assert [h[i] & 15 for i in range(4)] == [0xA, 0x3, 0x0, 0xF]
    libc.free(b)
    libc.free(h)


@pytest.mark.parametrize("chunk", (64, 1024), ids=("four-chunks", "one-chunk-forked"))
def test_dev_calls_are_graph_capturable(pkg, torch, oracle, chunk):
    """The *_dev entry points neither allocate nor synchronise: a whole encaps+decaps pass is captured into a HIP graph
    on a side stream and replayed on new inputs.  With one chunk per call the capture contains the fork to the context's side
    stream and the join (SideFork); the context creates that stream with its first one-chunk call, here inside the capture."""
    e = pkg.MLKEM(768, device=0, chunk_items=chunk)
    n = 200
    d, z = seeds("graph-d", n, 1), seeds("graph-z", n, 1)
    ek, dk = e.keygen(dev(torch, d), dev(torch, z))
    m = dev(torch, seeds("graph-m", n, 1))
    c = torch.empty((n, 1088), dtype=torch.uint8, device="cuda")
    K = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    K2 = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        e.encaps(ek, m, c=c, K=K)
        e.decaps(dk, c, K=K2, status=st)
    m2 = seeds("graph-m2", n, 2)
    m.copy_(dev(torch, m2))
    c.zero_(); K.zero_(); K2.zero_(); st.fill_(7)
    g.replay()
    torch.cuda.synchronize()
    c_o, K_o = oracle.encaps(768, host(ek), m2)
    assert (host(c) == c_o).all() and (host(K) == K_o).all() and (host(K2) == K_o).all() and (host(st) == 0).all()
    e.close()


def test_row4_converters_and_streaming_front_end(pkg, engines, torch, oracle):
    """device cell<->byte converters and the pinned double-buffered host-resident front-end."""
    e = engines[768]
    g = torch.Generator(device="cuda").manual_seed(4)
    n = 1 << 22
    b = torch.randint(0, 256, (n + 5,), generator=g, device="cuda", dtype=torch.uint8)
    junk = torch.randint(0, 1 << 23, (n + 5,), generator=g, device="cuda", dtype=torch.int32) << 8
    cells = b.to(torch.int32) | junk
    assert torch.equal(e.cells_to_bytes(cells), b)
    assert torch.equal(e.bytes_to_cells(b), b.to(torch.int32))
    lib = pkg.load_library()
    n = 1000   # 4 chunks of 256 + a ragged tail, two slots
    d, z, m = seeds("st-d", n, 1), seeds("st-z", n, 1), seeds("st-m", n, 1)
    ek, dk = np.zeros((n, 1184), np.uint8), np.zeros((n, 2400), np.uint8)
    assert lib.mlkem_keygen_stream(768, n, d.ctypes.data, z.ctypes.data, ek.ctypes.data, dk.ctypes.data, 256) == 0
    c, K = np.zeros((n, 1088), np.uint8), np.zeros((n, 32), np.uint8)
    assert lib.mlkem_encaps_stream(768, n, ek.ctypes.data, m.ctypes.data, c.ctypes.data, K.ctypes.data, 256) == 0
    cb = c.copy()
    cb[::97, 5] ^= 1
    Kd, st = np.zeros((n, 32), np.uint8), np.ones(n, np.int32)
    assert lib.mlkem_decaps_stream(768, n, dk.ctypes.data, cb.ctypes.data, Kd.ctypes.data, st.ctypes.data, 256) == 0
    ek_o, dk_o = oracle.keygen(768, d, z)
    c_o, K_o = oracle.encaps(768, ek_o, m)
    Kd_o, st_o = oracle.decaps(768, dk_o, cb)
    assert (ek == ek_o).all() and (dk == dk_o).all() and (c == c_o).all() and (K == K_o).all()
    assert (Kd == Kd_o).all() and (st == st_o).all()
    # single-chunk path (one slot) and empty batch
    assert lib.mlkem_encaps_stream(768, 10, ek.ctypes.data, m.ctypes.data, c.ctypes.data, K.ctypes.data, 0) == 0
    assert (c[:10] == c_o[:10]).all()
    assert lib.mlkem_encaps_stream(768, 0, None, None, None, None, 0) == 0
