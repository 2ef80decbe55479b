#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (oracle/_ref, built from /root/reference).

Run in the build container only:   python oracle/gen_golden.py
The reference itself cannot travel to the GPU box, so the vectors produced here are committed
as data fixtures (inputs + expected outputs) together with this script.

Fixture families:
  G1 NTT          Test_Archive/NTT_test08.c recipe  B[i] = 2i -> f1 = SampleNTT(B), fh = NTT(f1),
                  f2 = InverseNTT(fh); plus seeded random polynomials through NTT / InverseNTT /
                  MultiplyNTTs (canonical and non-canonical 12-bit inputs)
  G2 SampleNTT    Test_Archive/SampleNTT_test06.c recipe  B[i] = it*i + i, it = 0..6
  G3 CBD          Test_Archive/SampleCBD_test07.c recipe  B[i] = i (eta = 3) and eta = 2
  G4 codec        full Compress tables d = 1..11 over [0,q), Decompress over [0,2^d),
                  ByteEncode/ByteDecode for d in {1,4,5,10,11,12} (EncodeDecode_test03 recipe F[i]=16i too)
  G5 hashes       G / H / J / PRF on fixed inputs (J and PRF are SHAKE128 in the reference)
  G6 end-to-end   per parameter set: recipe seeds d=i, z=32+i, m=64+i in full, plus SEEDED_TRIPLES
                  triples from the documented SHAKE128 seed expander (digests + K in full)
  G7 negative     bad-ek accepted (F3), ml_errno codes of the public API
  G9 suffix       sha3_b with other suffix cells than ml_kem.c passes (RawSHAKE "11", "10", "00", four bits "1011")
  G8 SHA-3        the 16 NIST FIPS-202 examples held by the reference's Test_Examples/SHA
                  (message bits + expected output, parsed from the data files), cross-checked
                  against the reference's sha3_b
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.loader import Ref, SIZES  # noqa: E402

REF_DIR = os.environ.get("MLKEM_REF_DIR", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
SEEDED_TRIPLES = 24
GOLDEN_SEED = 0x203


def expand_seed(label: str, index: int, seed: int = GOLDEN_SEED) -> bytes:
    """Documented seed expander shared by fixtures, tests and bench:
    SHAKE128(label || LE64(index) || LE64(seed))[:32]."""
    return hashlib.shake_128(label.encode() + index.to_bytes(8, "little") + seed.to_bytes(8, "little")).digest(32)


def hx(a) -> str:
    return bytes(np.asarray(a, dtype=np.uint8).ravel()).hex()


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def parse_nist_example(path):
    """Test_Examples/SHA file format: 'Msg as bit string' block, then 'Hash val is' / 'Output val is'."""
    lines = open(path).read().split("\n")
    bits, out = "", ""
    i = 0
    while i < len(lines):
        if lines[i].strip() == "Msg as bit string":
            i += 1
            while i < len(lines) and lines[i].strip() != "":
                bits += lines[i]
                i += 1
        elif lines[i].strip() in ("Hash val is", "Output val is"):
            out = "".join(lines[i + 1:])
            break
        i += 1
    bits = "".join(ch for ch in bits if ch in "01") if "empty" not in bits else ""
    out = "".join(out.split()).lower()
    return bits, out


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = Ref()
    rng = np.random.default_rng(GOLDEN_SEED)
    J = {"generator": "oracle/gen_golden.py", "source": "reference build oracle/_ref/libmlkem_ref.so (-O2)",
         "seed_expander": "SHAKE128(label || LE64(index) || LE64(0x203))[:32]"}
    npz = {}

    # ---- G1 ------------------------------------------------------------------------------
    B = np.array([(2 * i) & 0xFF for i in range(34)], np.uint8)
    f1 = ref.sample_ntt(B)
    fh = ref.ntt(f1)[0]
    f2 = ref.intt(fh)[0]
    assert (f1 == f2).all()
    J["G1_test08"] = {"B": hx(B), "f1_first8": f1[:8].tolist(), "fh_first8": fh[:8].tolist(),
                      "f1_sha256": sha(f1), "fh_sha256": sha(fh)}
    npz["g1_f1"], npz["g1_fh"] = f1, fh
    polys = rng.integers(0, 3329, (64, 256)).astype(np.uint16)
    polys[0] = 0
    polys[1] = 3328
    polys[2] = np.arange(256) * 13 % 3329
    polys_b = rng.integers(0, 3329, (64, 256)).astype(np.uint16)
    npz["rand_a"], npz["rand_b"] = polys, polys_b
    npz["rand_a_ntt"] = ref.ntt(polys)
    npz["rand_a_intt"] = ref.intt(polys)
    npz["rand_ab_mul"] = np.stack([ref.multiply_ntts(polys[i], polys_b[i]) for i in range(64)])
    npz["rand_ab_add"] = np.stack([ref.poly_add(polys[i], polys_b[i]) for i in range(64)])
    npz["rand_ab_sub"] = np.stack([ref.poly_sub(polys[i], polys_b[i]) for i in range(64)])
    nc_a = rng.integers(0, 4096, (16, 256)).astype(np.uint16)  # non-canonical 12-bit (F3 path)
    nc_b = rng.integers(0, 4096, (16, 256)).astype(np.uint16)
    npz["nc_a"], npz["nc_b"] = nc_a, nc_b
    npz["nc_ab_mul"] = np.stack([ref.multiply_ntts(nc_a[i], nc_b[i]) for i in range(16)])
    npz["nc_a_ntt"] = ref.ntt(nc_a)
    npz["nc_a_intt"] = ref.intt(nc_a)

    # ---- G2 ------------------------------------------------------------------------------
    g2_in = np.array([[(it * i + i) & 0xFF for i in range(34)] for it in range(7)], np.uint8)
    g2_out = np.stack([ref.sample_ntt(g2_in[it]) for it in range(7)])
    assert (g2_out[1] == f1).all()
    xs = rng.integers(0, 256, (24, 34)).astype(np.uint8)
    npz["g2_in"] = np.concatenate([g2_in, xs])
    npz["g2_out"] = np.concatenate([g2_out, np.stack([ref.sample_ntt(x) for x in xs])])

    # ---- G3 ------------------------------------------------------------------------------
    b3 = np.arange(192, dtype=np.uint8)
    c3 = ref.sample_cbd(b3, 3)
    J["G3_test07"] = {"eta": 3, "first8": c3[:8].tolist(), "sha256": sha(c3)}
    npz["g3_eta3_in"] = np.concatenate([b3[None], rng.integers(0, 256, (15, 192)).astype(np.uint8)])
    npz["g3_eta3_out"] = np.stack([ref.sample_cbd(x, 3) for x in npz["g3_eta3_in"]])
    npz["g3_eta2_in"] = np.concatenate([np.arange(128, dtype=np.uint8)[None],
                                        rng.integers(0, 256, (15, 128)).astype(np.uint8)])
    npz["g3_eta2_out"] = np.stack([ref.sample_cbd(x, 2) for x in npz["g3_eta2_in"]])

    # ---- G4 ------------------------------------------------------------------------------
    comp = np.zeros((11, 3329), np.uint16)
    for d in range(1, 12):
        comp[d - 1] = [ref.compress(x, d) for x in range(3329)]
    npz["g4_compress"] = comp
    dec = np.zeros((11, 2048), np.uint16)
    for d in range(1, 12):
        dec[d - 1, : 1 << d] = [ref.decompress(y, d) for y in range(1 << d)]
    npz["g4_decompress"] = dec
    J["G4_identity_d12"] = {"compress": [ref.compress(x, 12) for x in (0, 1, 3328)],
                            "decompress": [ref.decompress(x, 12) for x in (0, 1, 3328)]}
    for d in (1, 4, 5, 10, 11, 12):
        F = rng.integers(0, 1 << d, (4, 256)).astype(np.uint16)
        if d == 12:
            F[0] = np.arange(256) * 16  # EncodeDecode_test03 recipe (includes values >= q)
        npz[f"g4_enc{d}_in"] = F
        npz[f"g4_enc{d}_out"] = np.stack([ref.byte_encode(f, d) for f in F])
        Bd = rng.integers(0, 256, (4, 32 * d)).astype(np.uint8)
        npz[f"g4_dec{d}_in"] = Bd
        npz[f"g4_dec{d}_out"] = np.stack([ref.byte_decode(b, d) for b in Bd])
    J["G4_bitrev7"] = [ref.bitrev7(i) for i in range(128)]

    # ---- G5 ------------------------------------------------------------------------------
    g5 = []
    for n in (0, 1, 32, 33, 64, 71, 72, 73, 135, 136, 137, 167, 168, 169, 800, 1120, 1184, 1568, 1600):
        x = rng.integers(0, 256, n).astype(np.uint8)
        g5.append({"in": hx(x), "H": hx(ref.H(x)), "G": hx(ref.G(x)), "J": hx(ref.J(x))})
    J["G5_hashes"] = g5
    prf = []
    for b in (0, 1, 6, 255):
        s = rng.integers(0, 256, 32).astype(np.uint8)
        prf.append({"s": hx(s), "b": b, "eta2": hx(ref.prf(s, b, 2)), "eta3": hx(ref.prf(s, b, 3))})
    J["G5_prf"] = prf

    # ---- G6 / G7 ---------------------------------------------------------------------------
    kem = {}
    for pset in (512, 768, 1024):
        ekl, dkl, cl = SIZES[pset]
        d = np.arange(32, dtype=np.uint8)
        z = d + 32
        m = d + 64
        ek, dk = ref.keygen(pset, d, z)
        c, K = ref.encaps(pset, ek, m)
        Kd, st = ref.decaps(pset, dk, c)
        assert st[0] == 0 and (Kd == K).all()
        cb = c.copy()
        cb[0, 5] ^= 1
        Kbar, st = ref.decaps(pset, dk, cb)
        assert st[0] == 0 and not (Kbar == K).all()
        ek_pke, dk_pke = ref.pke_keygen(pset, d)
        assert (ek_pke == ek[0]).all() and (dk_pke == dk[0, : ekl - 32]).all()
        rec = {"recipe": {"d": hx(d), "z": hx(z), "m": hx(m), "ek": hx(ek), "dk": hx(dk), "c": hx(c), "K": hx(K),
                          "K_reject_c5_xor1": hx(Kbar)}}
        # PKE_EncryptDecrypt_test.c recipe: randomness[i] = i, message[i] = i % 5
        msg = np.array([i % 5 for i in range(32)], np.uint8)
        cp = ref.pke_encrypt(pset, ek_pke, msg, d)
        assert (ref.pke_decrypt(pset, dk_pke, cp) == msg).all()
        rec["pke_test10"] = {"randomness": hx(d), "message": hx(msg), "c": hx(cp)}
        trip = []
        for i in range(SEEDED_TRIPLES):
            di = np.frombuffer(expand_seed("mlkem-golden-d", i), np.uint8)
            zi = np.frombuffer(expand_seed("mlkem-golden-z", i), np.uint8)
            mi = np.frombuffer(expand_seed("mlkem-golden-m", i), np.uint8)
            eki, dki = ref.keygen(pset, di, zi)
            ci, Ki = ref.encaps(pset, eki, mi)
            Kdi, sti = ref.decaps(pset, dki, ci)
            assert sti[0] == 0 and (Kdi == Ki).all()
            # tamper one byte chosen by index so that c1 and c2 regions both get hit over the set
            pos = (i * 131) % cl
            cbi = ci.copy()
            cbi[0, pos] ^= 1 << (i % 8)
            Kri, sti = ref.decaps(pset, dki, cbi)
            trip.append({"i": i, "ek_sha256": sha(eki), "dk_sha256": sha(dki), "c_sha256": sha(ci), "K": hx(Ki),
                         "tamper_pos": pos, "tamper_mask": 1 << (i % 8), "K_reject": hx(Kri)})
        rec["seeded"] = trip
        # G7: F3 — an ek whose first coefficient is 0xFFF (>= q) is accepted by the public KEM_Encaps
        bad = ek[0].copy()
        bad[0] = 0xFF
        bad[1] |= 0x0F
        rc, c_pub, K_pub = ref.kem_encaps_public(pset, bad)
        cdet, Kdet = ref.encaps(pset, bad, m)
        rec["G7"] = {"bad_ek_errno": rc, "bad_ek": hx(bad), "bad_ek_m": hx(m), "bad_ek_c_sha256": sha(cdet),
                     "bad_ek_K": hx(Kdet)}
        rc_len, _, _ = ref.kem_encaps_public(pset, ek[0][:-1])
        rc_clen, _ = ref.kem_decaps(pset, dk[0], c[0][:-1])
        rc_dklen, _ = ref.kem_decaps(pset, dk[0][:-1], c[0])
        bad_dk = dk[0].copy()
        bad_dk[(ekl - 32) + 7] ^= 0x10  # flip a bit inside the embedded ek -> hash check fails
        rc_hash, _ = ref.kem_decaps(pset, bad_dk, c[0])
        bad_h = dk[0].copy()
        bad_h[dkl - 64] ^= 1  # flip a bit of the stored H(ek)
        rc_hash2, _ = ref.kem_decaps(pset, bad_h, c[0])
        bad_s = dk[0].copy()
        bad_s[3] ^= 0x40  # corrupt dk_pke (not covered by the hash check) -> implicit rejection path
        rc_s, K_s = ref.kem_decaps(pset, bad_s, c[0])
        rec["G7"].update({"errno_ek_len": rc_len, "errno_c_len": rc_clen, "errno_dk_len": rc_dklen,
                          "errno_dk_hash_ek": rc_hash, "errno_dk_hash_h": rc_hash2,
                          "bad_dkpke_errno": rc_s, "bad_dkpke_K": hx(K_s)})
        kem[str(pset)] = rec
        print(f"  ML-KEM-{pset}: recipe K = {hx(K)}", flush=True)
    J["G6_kem"] = kem
    J["G7_init_errno"] = {str(s): ref.init_errno(s) for s in (512, 768, 1024, 0, 256, 2048)}

    # ---- G8 --------------------------------------------------------------------------------
    nist = []
    exdir = os.path.join(REF_DIR, "Test_Examples", "SHA")
    for name in sorted(os.listdir(exdir)):
        bits, out = parse_nist_example(os.path.join(exdir, name))
        kind, rest = name.split("-")
        size = int(rest.split("_")[0])
        xof = kind == "XOF"
        cap = 2 * size
        d_bits = 4096 if xof else size  # sha_testing.sh:20 squeezes 4096 bits from the XOFs
        got = ref.sha3_bits(np.array([int(ch) for ch in bits], np.uint8), d_bits, cap, xof)
        got_hex = bytes(np.packbits(got, bitorder="little")).hex()
        assert got_hex == out[: len(got_hex)] and len(out) >= len(got_hex), name
        nist.append({"file": name, "xof": xof, "size": size, "rate_bytes": (1600 - cap) // 8, "msg_bits": bits,
                     "out": out[: len(got_hex)]})
    J["G8_nist_sha3"] = nist
    print(f"  {len(nist)} NIST SHA-3 examples reproduced by the reference sha3_b")

    # ---- G4 over the whole 12-bit field, every d (Test_Archive/CompressDecompress_test04.c sweeps d = 1..12) ---------
    npz["g4_compress_full"] = np.array([[ref.compress(x, d) for x in range(4096)] for d in range(1, 13)], np.uint16)
    npz["g4_decompress_full"] = np.array([[ref.decompress(y, d) for y in range(4096)] for d in range(1, 13)], np.uint16)

    # ---- G9: sha3_b appends the caller's suffix cells verbatim (sha3.c:414-429) -----------------------------------
    # two-bit suffixes other than the hash "01" (RawSHAKE "11", and "10" / "00" to pin "verbatim"), NIST's message
    # lengths; the lengths avoid the reference's latent pad bug (n + suffix + 2 = 0 mod r)
    rng9 = np.random.default_rng(GOLDEN_SEED + 9)
    raw = []
    for cap in (256, 512):
        for sfx in ((1, 1, 0, 0), (1, 0, 0, 0), (0, 0, 0, 0), (1, 0, 1, 1)):
            for nbits in (0, 5, 30, 1600, 1605, 1630):
                bits = rng9.integers(0, 2, nbits).astype(np.uint8)
                got = ref.sha3_bits_sfx(bits, 512, cap, np.array(sfx, np.uint8))
                raw.append({"cap": cap, "rate_bytes": (1600 - cap) // 8, "sfx": list(sfx), "msg_bits": "".join(map(str, bits)),
                            "out": bytes(np.packbits(got, bitorder="little")).hex()})
    J["G9_sha3_suffix"] = raw

    with open(os.path.join(OUT, "mlkem_golden.json"), "w") as f:
        json.dump(J, f, indent=1)
    np.savez_compressed(os.path.join(OUT, "mlkem_golden.npz"), **npz)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
