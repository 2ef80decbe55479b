"""CPU tier: pin the oracle (oracle/mlkem_oracle.c) against the golden vectors produced by the real
reference (tests/golden/*, generator oracle/gen_golden.py) and against the NIST SHA-3 examples the
reference ships (Test_Examples/SHA, parsed into the fixture)."""
import numpy as np
import pytest

from conftest import expand_seed, sha256, unhex
from oracle.loader import SIZES

SETS = (512, 768, 1024)


def test_g8_nist_sha3_examples(oracle, golden):
    assert len(golden["G8_nist_sha3"]) == 16
    for ex in golden["G8_nist_sha3"]:
        bits = np.array([int(c) for c in ex["msg_bits"]], np.uint8)
        want = bytes.fromhex(ex["out"])
        got = oracle.sponge_bits(ex["rate_bytes"], ex["xof"], bits, len(want))
        assert bytes(got) == want, ex["file"]
        if len(bits) % 8 == 0:  # byte-aligned examples also through the byte sponge
            msg = np.packbits(bits, bitorder="little") if len(bits) else np.zeros(0, np.uint8)
            got = oracle.sponge(ex["rate_bytes"], 0x1F if ex["xof"] else 0x06, msg, len(want))
            assert bytes(got) == want, ex["file"]


def test_g1_ntt_test08(oracle, golden, golden_npz):
    g = golden["G1_test08"]
    f1 = oracle.sample_ntt(unhex(g["B"]))
    assert f1[:8].tolist() == g["f1_first8"] == [2931, 2112, 3266, 1044, 1856, 3090, 520, 2900]
    assert sha256(f1) == g["f1_sha256"]
    fh = oracle.ntt(f1)[0]
    assert fh[:8].tolist() == g["fh_first8"] == [192, 1622, 2207, 476, 2400, 883, 637, 686]
    assert sha256(fh) == g["fh_sha256"]
    assert (oracle.intt(fh)[0] == f1).all()
    assert (golden_npz["g1_f1"] == f1).all() and (golden_npz["g1_fh"] == fh).all()


def test_g1_random_polys(oracle, golden_npz):
    a, b = golden_npz["rand_a"], golden_npz["rand_b"]
    assert (oracle.ntt(a) == golden_npz["rand_a_ntt"]).all()
    assert (oracle.intt(a) == golden_npz["rand_a_intt"]).all()
    for i in range(a.shape[0]):
        assert (oracle.multiply_ntts(a[i], b[i]) == golden_npz["rand_ab_mul"][i]).all()
        assert (oracle.poly_add(a[i], b[i]) == golden_npz["rand_ab_add"][i]).all()
        assert (oracle.poly_sub(a[i], b[i]) == golden_npz["rand_ab_sub"][i]).all()
    # round trips
    assert (oracle.intt(oracle.ntt(a)) == a).all()


def test_g1_noncanonical_inputs(oracle, golden_npz):
    a, b = golden_npz["nc_a"], golden_npz["nc_b"]
    assert (oracle.ntt(a) == golden_npz["nc_a_ntt"]).all()
    assert (oracle.intt(a) == golden_npz["nc_a_intt"]).all()
    for i in range(a.shape[0]):
        assert (oracle.multiply_ntts(a[i], b[i]) == golden_npz["nc_ab_mul"][i]).all()


def test_g2_sample_ntt(oracle, golden_npz):
    for x, want in zip(golden_npz["g2_in"], golden_npz["g2_out"]):
        assert (oracle.sample_ntt(x) == want).all()


def test_g3_cbd(oracle, golden, golden_npz):
    c3 = oracle.sample_cbd(np.arange(192, dtype=np.uint8), 3)
    assert c3[:8].tolist() == golden["G3_test07"]["first8"] == [0, 1, 3328, 0, 2, 3328, 3328, 1]
    for eta in (2, 3):
        for x, want in zip(golden_npz[f"g3_eta{eta}_in"], golden_npz[f"g3_eta{eta}_out"]):
            assert (oracle.sample_cbd(x, eta) == want).all()


def test_g4_compress_tables(oracle, golden, golden_npz):
    comp, dec = golden_npz["g4_compress"], golden_npz["g4_decompress"]
    for d in range(1, 12):
        got = np.array([oracle.compress(x, d) for x in range(3329)], np.uint16)
        assert (got == comp[d - 1]).all(), d
        got = np.array([oracle.decompress(y, d) for y in range(1 << d)], np.uint16)
        assert (got == dec[d - 1, : 1 << d]).all(), d
        # CompressDecompress_test04 property: Compress(Decompress(y)) == y
        assert all(oracle.compress(oracle.decompress(y, d), d) == y for y in range(1 << d))
    assert [oracle.compress(x, 12) for x in (0, 1, 3328)] == golden["G4_identity_d12"]["compress"]
    assert [oracle.decompress(x, 12) for x in (0, 1, 3328)] == golden["G4_identity_d12"]["decompress"]
    assert [oracle.bitrev7(i) for i in range(128)] == golden["G4_bitrev7"]


@pytest.mark.parametrize("d", (1, 4, 5, 10, 11, 12))
def test_g4_byte_codec(oracle, golden_npz, d):
    for F, want in zip(golden_npz[f"g4_enc{d}_in"], golden_npz[f"g4_enc{d}_out"]):
        assert (oracle.byte_encode(F, d) == want).all()
        assert (oracle.byte_decode(want, d) == F).all()  # EncodeDecode_test03 round trip (incl. >= q at d=12)
    for B, want in zip(golden_npz[f"g4_dec{d}_in"], golden_npz[f"g4_dec{d}_out"]):
        assert (oracle.byte_decode(B, d) == want).all()


def test_g5_hashes(oracle, golden):
    import hashlib
    for h in golden["G5_hashes"]:
        x = unhex(h["in"])
        assert bytes(oracle.H(x)).hex() == h["H"] == hashlib.sha3_256(bytes(x)).hexdigest()
        assert bytes(oracle.G(x)).hex() == h["G"] == hashlib.sha3_512(bytes(x)).hexdigest()
        # F2: the reference's J is SHAKE128, not SHAKE256
        assert bytes(oracle.J(x)).hex() == h["J"] == hashlib.shake_128(bytes(x)).hexdigest(32)
    for p in golden["G5_prf"]:
        s = unhex(p["s"])
        for eta in (2, 3):
            want = hashlib.shake_128(bytes(s) + bytes([p["b"]])).hexdigest(64 * eta)
            assert bytes(oracle.prf(s, p["b"], eta)).hex() == p[f"eta{eta}"] == want


@pytest.mark.parametrize("pset", SETS)
def test_g6_recipe_triple(oracle, golden, pset):
    g = golden["G6_kem"][str(pset)]["recipe"]
    ek, dk = oracle.keygen(pset, unhex(g["d"]), unhex(g["z"]))
    assert bytes(ek).hex() == g["ek"] and bytes(dk).hex() == g["dk"]
    c, K = oracle.encaps(pset, ek, unhex(g["m"]))
    assert bytes(c).hex() == g["c"] and bytes(K).hex() == g["K"]
    Kd, st = oracle.decaps(pset, dk, c)
    assert st[0] == 0 and bytes(Kd).hex() == g["K"]
    cb = c.copy()
    cb[0, 5] ^= 1
    Kr, st = oracle.decaps(pset, dk, cb)
    assert st[0] == 0 and bytes(Kr).hex() == g["K_reject_c5_xor1"]
    # PKE_EncryptDecrypt_test.c recipe
    t = golden["G6_kem"][str(pset)]["pke_test10"]
    ekp, dkp = oracle.pke_keygen(pset, unhex(t["randomness"]))
    cp = oracle.pke_encrypt(pset, ekp, unhex(t["message"]), unhex(t["randomness"]))
    assert bytes(cp).hex() == t["c"]
    assert bytes(oracle.pke_decrypt(pset, dkp, cp)).hex() == t["message"]


def test_g6_survey_K_values(golden):
    """The recipe shared secrets quoted in SURVEY.md section 8c."""
    assert golden["G6_kem"]["512"]["recipe"]["K"] == "215d3b38605da46a4ae9ed6c70037797fda977c4226db3a5516f3c7e70c6a824"
    assert golden["G6_kem"]["768"]["recipe"]["K"] == "ca49ed38f11d513390bb0db10b9bf900eb6ce82f1ca0c71acca7947ad0dd2c37"
    assert golden["G6_kem"]["1024"]["recipe"]["K"] == "8e9df1ffbff5244b0d561328f081bf1d3b578362b526dc790fa839dc3c838710"
    assert golden["G6_kem"]["768"]["recipe"]["K_reject_c5_xor1"] == \
        "1ff209d0da6ec725d8513af357049d0cb065caa7fd3fd2b038aa4c2487e962b3"


@pytest.mark.parametrize("pset", SETS)
def test_g6_seeded_triples(oracle, golden, pset):
    trip = golden["G6_kem"][str(pset)]["seeded"]
    n = len(trip)
    d = np.frombuffer(b"".join(expand_seed("mlkem-golden-d", i, 0x203) for i in range(n)), np.uint8)
    z = np.frombuffer(b"".join(expand_seed("mlkem-golden-z", i, 0x203) for i in range(n)), np.uint8)
    m = np.frombuffer(b"".join(expand_seed("mlkem-golden-m", i, 0x203) for i in range(n)), np.uint8)
    ek, dk = oracle.keygen(pset, d, z)
    c, K = oracle.encaps(pset, ek, m)
    Kd, st = oracle.decaps(pset, dk, c)
    assert (st == 0).all() and (Kd == K).all()
    cb = c.copy()
    for t in trip:
        i = t["i"]
        assert sha256(ek[i]) == t["ek_sha256"] and sha256(dk[i]) == t["dk_sha256"]
        assert sha256(c[i]) == t["c_sha256"] and bytes(K[i]).hex() == t["K"]
        cb[i, t["tamper_pos"]] ^= t["tamper_mask"]
    Kr, st = oracle.decaps(pset, dk, cb)
    assert (st == 0).all()
    for t in trip:
        assert bytes(Kr[t["i"]]).hex() == t["K_reject"]


@pytest.mark.parametrize("pset", SETS)
def test_g7_negative_paths(oracle, golden, pset):
    g = golden["G6_kem"][str(pset)]
    g7, rec = g["G7"], g["recipe"]
    ekl, dkl, cl = SIZES[pset]
    # F3: the reference accepts an ek with a coefficient >= q (ml_errno stays 0)
    assert g7["bad_ek_errno"] == 0
    bad = unhex(g7["bad_ek"])
    assert oracle.kem_encaps_check(pset, bad) == 0
    c, K = oracle.encaps(pset, bad, unhex(g7["bad_ek_m"]))
    assert sha256(c) == g7["bad_ek_c_sha256"] and bytes(K).hex() == g7["bad_ek_K"]
    ek, dk, c = unhex(rec["ek"]), unhex(rec["dk"]), unhex(rec["c"])
    assert g7["errno_ek_len"] == -3 == oracle.kem_encaps_check(pset, ek[:-1])
    assert g7["errno_c_len"] == -3 == oracle.kem_decaps(pset, dk, c[:-1])[0]
    assert g7["errno_dk_len"] == -3 == oracle.kem_decaps(pset, dk[:-1], c)[0]
    bad_dk = dk.copy()
    bad_dk[(ekl - 32) + 7] ^= 0x10
    assert g7["errno_dk_hash_ek"] == -5 == oracle.kem_decaps(pset, bad_dk, c)[0]
    bad_h = dk.copy()
    bad_h[dkl - 64] ^= 1
    assert g7["errno_dk_hash_h"] == -5 == oracle.kem_decaps(pset, bad_h, c)[0]
    bad_s = dk.copy()
    bad_s[3] ^= 0x40
    rc, K = oracle.kem_decaps(pset, bad_s, c)
    assert rc == g7["bad_dkpke_errno"] == 0 and bytes(K).hex() == g7["bad_dkpke_K"]


def test_g7_init_errno(oracle, golden):
    from oracle.loader import C
    for s, rc in golden["G7_init_errno"].items():
        class P(C.Structure):
            _fields_ = [("f", C.c_uint * 9)]
        assert oracle.lib.orc_params_init(int(s), C.byref(P())) == rc


def test_g4_full_field_tables_and_g9_suffix_vectors(oracle, golden, golden_npz):
    """The fixtures added in round 2 (generated from the real reference by oracle/gen_golden.py): Compress / Decompress
    over the whole 12-bit field for d = 1..12, and sha3_b with caller-chosen suffix bits."""
    comp, dec = golden_npz["g4_compress_full"], golden_npz["g4_decompress_full"]
    for d in range(1, 13):
        assert [oracle.compress(v, d) for v in range(4096)] == comp[d - 1].tolist()
        assert [oracle.decompress(v, d) for v in range(4096)] == dec[d - 1].tolist()
        if d < 12:   # the older tables are the sub-ranges
            assert (comp[d - 1, :3329] == golden_npz["g4_compress"][d - 1]).all()
            assert (dec[d - 1, : 1 << d] == golden_npz["g4_decompress"][d - 1, : 1 << d]).all()
    for g in golden["G9_sha3_suffix"]:
        bits = np.array([int(ch) for ch in g["msg_bits"]], np.uint8)
        sfx = g["sfx"][: 4 if g["sfx"][2] == 1 else 2]
        want = np.frombuffer(bytes.fromhex(g["out"]), np.uint8)
        assert (oracle.sponge_bits_sfx(g["rate_bytes"], sfx, bits, want.size) == want).all()
