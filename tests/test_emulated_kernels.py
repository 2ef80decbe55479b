"""CPU tier: the PRODUCT's kernel source (crystals-kyber_amd/csrc/mlkem_kernels.hpp + pipeline) compiled for
the host wave emulator (tests/emu) and checked against the oracle and the golden vectors.  This validates
kernel logic (layouts, indexing, codecs, sponge padding, rejection sampling, kernel sequencing) before any
GPU time is spent; the GPU tier (-m gpu) then checks the same things on the real gfx950 build."""
import ctypes as C

import numpy as np
import pytest

import __graft_entry__ as ge
from conftest import seeds, unhex
from oracle.loader import SIZES

u8p, u16p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16)


def p8(a):
    return a.ctypes.data_as(u8p)


def p16(a):
    return a.ctypes.data_as(u16p)


@pytest.fixture(scope="module")
def emu():
    return C.CDLL(ge.build_emulator())


def test_emu_ntt_intt_multiply(emu, oracle, golden_npz):
    a, b = golden_npz["rand_a"][:6].copy(), golden_npz["rand_b"][:6].copy()
    out = np.zeros_like(a)
    emu.emu_ntt(0, C.c_size_t(6), p16(a), p16(out))
    assert (out == golden_npz["rand_a_ntt"][:6]).all()
    emu.emu_ntt(1, C.c_size_t(6), p16(a), p16(out))
    assert (out == golden_npz["rand_a_intt"][:6]).all()
    emu.emu_basemul(C.c_size_t(6), p16(a), p16(b), p16(out))
    assert (out == golden_npz["rand_ab_mul"][:6]).all()
    na, nb = golden_npz["nc_a"][:4].copy(), golden_npz["nc_b"][:4].copy()   # non-canonical 12-bit inputs (F3 path)
    out = np.zeros_like(na)
    emu.emu_basemul(C.c_size_t(4), p16(na), p16(nb), p16(out))
    assert (out == golden_npz["nc_ab_mul"][:4]).all()


def test_emu_fp32_twiddle_product_exhaustive(emu):
    """Every (twiddle, b) pair the NTT can meet: 258 multipliers x 20165 values through the product's 3-FMA modular product."""
    emu.emu_fmulmod_exhaustive.restype = C.c_long
    assert emu.emu_fmulmod_exhaustive() == 0


def test_emu_fp32_compress_exhaustive(emu):
    """Compress_d on the fp32 pipe, every |x| <= 4095 (any representative) x d in {1,4,5,10,11}, against the integer form."""
    emu.emu_compress_f_exhaustive.restype = C.c_long
    assert emu.emu_compress_f_exhaustive() == 0


def test_emu_cbd2_exhaustive(emu):
    """The byte-parallel eta = 2 CBD evaluation over all 65536 inputs of a lane against the definition."""
    emu.emu_cbd2_exhaustive.restype = C.c_long
    assert emu.emu_cbd2_exhaustive() == 0


def test_emu_sampling(emu, oracle, golden_npz):
    s = golden_npz["g2_in"].copy()
    out = np.zeros((s.shape[0], 256), np.uint16)
    emu.emu_sample_ntt(C.c_size_t(s.shape[0]), p8(s), p16(out))
    assert (out == golden_npz["g2_out"]).all()
    for eta in (2, 3):
        b = golden_npz[f"g3_eta{eta}_in"].copy()
        out = np.zeros((b.shape[0], 256), np.uint16)
        assert emu.emu_cbd(eta, C.c_size_t(b.shape[0]), p8(b), p16(out)) == 0
        assert (out == golden_npz[f"g3_eta{eta}_out"]).all()
        rng = np.random.default_rng(eta)
        x = rng.integers(0, 256, (65, 33)).astype(np.uint8)
        o = np.zeros((65, 64 * eta), np.uint8)
        assert emu.emu_prf(eta, C.c_size_t(65), p8(x), p8(o)) == 0
        assert all((o[i] == oracle.prf(x[i, :32], int(x[i, 32]), eta)).all() for i in range(65))


def test_emu_hashes_all_block_boundaries(emu, oracle):
    rng = np.random.default_rng(3)
    for kind, fn, ol in ((0, oracle.H, 32), (1, oracle.G, 64), (2, oracle.J, 32)):
        for ln in (0, 1, 3, 4, 33, 64, 71, 72, 73, 135, 136, 137, 167, 168, 169, 800, 1120, 1184):
            n = 65
            stride = (ln + 7) // 8 * 8 + 8
            msg = rng.integers(0, 256, (n, stride)).astype(np.uint8)
            out = np.zeros((n, ol), np.uint8)
            assert emu.emu_hash(kind, C.c_size_t(n), p8(msg), ln, C.c_size_t(stride), p8(out)) == 0
            for i in (0, 1, 63, 64):
                assert (out[i] == fn(msg[i, :ln])).all(), (kind, ln, i)


@pytest.mark.parametrize("pset", (512, 768, 1024))
def test_emu_kem_matches_oracle_and_golden(emu, oracle, golden, pset):
    ekl, dkl, cl = SIZES[pset]
    n = 5
    d, z, m = seeds("emu-d", n, pset), seeds("emu-z", n, pset), seeds("emu-m", n, pset)
    g = golden["G6_kem"][str(pset)]["recipe"]
    d[0], z[0], m[0] = unhex(g["d"]), unhex(g["z"]), unhex(g["m"])
    ek, dk = np.zeros((n, ekl), np.uint8), np.zeros((n, dkl), np.uint8)
    assert emu.emu_keygen(pset, C.c_size_t(n), p8(d), p8(z), p8(ek), p8(dk)) == 0
    ek_o, dk_o = oracle.keygen(pset, d, z)
    assert (ek == ek_o).all() and (dk == dk_o).all()
    assert bytes(ek[0]).hex() == g["ek"] and bytes(dk[0]).hex() == g["dk"]
    c, K = np.zeros((n, cl), np.uint8), np.zeros((n, 32), np.uint8)
    assert emu.emu_encaps(pset, C.c_size_t(n), p8(ek), p8(m), p8(c), p8(K), None) == 0
    c_o, K_o = oracle.encaps(pset, ek, m)
    assert (c == c_o).all() and (K == K_o).all()
    assert bytes(c[0]).hex() == g["c"] and bytes(K[0]).hex() == g["K"]
    cb = c.copy()
    cb[0, 5] ^= 1            # golden implicit-rejection case
    cb[3, cl - 1] ^= 0x80    # tamper inside c2
    dkb = dk.copy()
    dkb[2, dkl - 64] ^= 1    # stored H(ek) corrupted -> status -5
    Kd, st = np.zeros((n, 32), np.uint8), np.zeros(n, np.int32)
    assert emu.emu_decaps(pset, C.c_size_t(n), p8(dkb), p8(cb), p8(Kd), st.ctypes.data_as(C.POINTER(C.c_int32)), 1) == 0
    Ko, sto = oracle.decaps(pset, dkb, cb)
    assert (st == sto).all() and st.tolist() == [0, 0, -5, 0, 0]
    assert (Kd[sto == 0] == Ko[sto == 0]).all()
    assert bytes(Kd[0]).hex() == g["K_reject_c5_xor1"]
    assert (Kd[1] == K[1]).all() and (Kd[4] == K[4]).all() and not (Kd[3] == K[3]).all()


def _codec_cases(oracle, golden_npz, d):
    """(polys in, bytes expected) for Compress+ByteEncode and (bytes in, polys expected) for ByteDecode+Decompress, from
    the reference's full Compress / Decompress tables (G4) and its ByteEncode / ByteDecode (oracle = pinned restatement)."""
    xs = np.zeros(14 * 256, np.uint16)
    xs[:3329] = np.arange(3329)
    xs = xs.reshape(14, 256)
    if d == 12:
        enc_want = np.stack([oracle.byte_encode(f, 12) for f in xs])
        dec_in = golden_npz["g4_dec12_in"]                      # includes values >= q: stay unreduced (F3)
        dec_want = golden_npz["g4_dec12_out"]
    else:
        comp, dec = golden_npz["g4_compress"][d - 1], golden_npz["g4_decompress"][d - 1]
        enc_want = np.stack([oracle.byte_encode(comp[f], d) for f in xs])
        ys = (np.arange(8 * 256) % (1 << d)).astype(np.uint16).reshape(8, 256)
        dec_in = np.stack([oracle.byte_encode(y, d) for y in ys])
        dec_want = dec[ys]
    return xs, enc_want, np.ascontiguousarray(dec_in), dec_want


@pytest.mark.parametrize("d", (1, 4, 5, 10, 11, 12))
def test_emu_codec_primitives(emu, oracle, golden_npz, d):
    xs, enc_want, dec_in, dec_want = _codec_cases(oracle, golden_npz, d)
    out = np.zeros((xs.shape[0], 32 * d), np.uint8)
    assert emu.emu_codec(1, d, C.c_size_t(xs.shape[0]), p16(xs), p8(out)) == 0
    assert (out == enc_want).all()
    f = np.zeros((dec_in.shape[0], 256), np.uint16)
    assert emu.emu_codec(0, d, C.c_size_t(dec_in.shape[0]), p8(dec_in), p16(f)) == 0
    assert (f == dec_want).all()
    assert emu.emu_codec(1, 7, C.c_size_t(1), p16(xs), p8(out)) == -1


@pytest.mark.parametrize("pset", (512, 768, 1024))
def test_emu_shared_key_batches(emu, oracle, pset):
    """One key for the whole batch: same bytes as the per-item path on the replicated key (oracle), incl. implicit
    rejection and the once-per-call dk hash check; chunk loops crossed (cap 3, hcap 7, 17 items)."""
    emu.emu_config(C.c_size_t(3), C.c_size_t(7))
    try:
        ekl, dkl, cl = SIZES[pset]
        n = 17
        d, z, m = seeds("sh-d", 1, pset), seeds("sh-z", 1, pset), seeds("sh-m", n, pset)
        ek1, dk1 = oracle.keygen(pset, d, z)
        c, K = np.zeros((n, cl), np.uint8), np.zeros((n, 32), np.uint8)
        assert emu.emu_encaps_shared(pset, C.c_size_t(n), p8(ek1), p8(m), p8(c), p8(K)) == 0
        c_o, K_o = oracle.encaps(pset, np.repeat(ek1, n, axis=0), m)
        assert (c == c_o).all() and (K == K_o).all()
        cb = c.copy()
        cb[[0, 5, 16], [1, cl - 1, 77]] ^= 8
        Kd, st = np.zeros((n, 32), np.uint8), np.ones(n, np.int32)
        assert emu.emu_decaps_shared(pset, C.c_size_t(n), p8(dk1), p8(cb), p8(Kd), st.ctypes.data_as(C.POINTER(C.c_int32))) == 0
        Kd_o, st_o = oracle.decaps(pset, np.repeat(dk1, n, axis=0), cb)
        assert (st == 0).all() and (st_o == 0).all() and (Kd == Kd_o).all()
        dkb = dk1.copy()
        dkb[0, dkl - 50] ^= 1                              # stored H(ek) corrupted: every item reports -5
        assert emu.emu_decaps_shared(pset, C.c_size_t(n), p8(dkb), p8(cb), p8(Kd), st.ctypes.data_as(C.POINTER(C.c_int32))) == 0
        assert (st == -5).all()
    finally:
        emu.emu_config(C.c_size_t(0), C.c_size_t(0))


@pytest.mark.parametrize("pset", (512, 768, 1024))
def test_emu_k_pke_alone(emu, oracle, pset):
    """PKE_KeyGen / PKE_Encrypt (caller's randomness) / PKE_Decrypt as stand-alone entry points."""
    ekl, dkl, cl = SIZES[pset]
    k = {512: 2, 768: 3, 1024: 4}[pset]
    n = 6
    d, m, r = seeds("pke-d", n, pset), seeds("pke-m", n, pset), seeds("pke-r", n, pset)
    ek, dkp = np.zeros((n, ekl), np.uint8), np.zeros((n, 384 * k), np.uint8)
    assert emu.emu_pke_keygen(pset, C.c_size_t(n), p8(d), p8(ek), p8(dkp)) == 0
    c = np.zeros((n, cl), np.uint8)
    assert emu.emu_pke_encrypt(pset, C.c_size_t(n), p8(ek), p8(m), p8(r), p8(c)) == 0
    m2 = np.zeros((n, 32), np.uint8)
    assert emu.emu_pke_decrypt(pset, C.c_size_t(n), p8(dkp), p8(c), p8(m2)) == 0
    for i in range(n):
        ek_o, dk_o = oracle.pke_keygen(pset, d[i])
        assert (ek[i] == ek_o).all() and (dkp[i] == dk_o).all()
        assert (c[i] == oracle.pke_encrypt(pset, ek_o, m[i], r[i])).all()
        assert (m2[i] == oracle.pke_decrypt(pset, dk_o, c[i])).all()
    assert (m2 == m).all()


@pytest.mark.parametrize("pset", (512, 768, 1024))
def test_emu_decrypt_four_items_per_wave_every_quad_shape(emu, oracle, pset):
    """mlkem_kpke4.hpp: K-PKE.Decrypt with four items per wave.  Every batch size mod 4 (rows beyond n redo the last item and
    store nothing), several quads per workgroup, random ciphertext BYTES (not encryptions: every 10/11/4/5-bit field value and
    every position of the unaligned pieces shows up) and raw 12-bit key coefficients >= q (F3), against the oracle."""
    ekl, dkl, cl = SIZES[pset]
    k = {512: 2, 768: 3, 1024: 4}[pset]
    rng = np.random.default_rng(pset)
    for n in (1, 2, 3, 4, 5, 18):
        c = rng.integers(0, 256, (n, cl)).astype(np.uint8)
        dkp = rng.integers(0, 256, (n, 384 * k)).astype(np.uint8)   # 12-bit fields up to 4095
        m = np.full((n + 1, 32), 0xA5, np.uint8)                     # one guard row behind the batch
        assert emu.emu_pke_decrypt(pset, C.c_size_t(n), p8(dkp), p8(c), p8(m)) == 0
        for i in range(n):
            assert (m[i] == oracle.pke_decrypt(pset, dkp[i], c[i])).all(), (n, i)
        assert (m[n] == 0xA5).all()


@pytest.mark.parametrize("pset", (512, 768, 1024))
def test_emu_full_wave_hash_kernels_take_the_dma_staging(emu, oracle, pset):
    """64 + 3 items: the first wave of k_hash_encaps / k_hash_decaps / k_hash_keygen_fin is complete, so its sponges are
    fed by the LDS-DMA staging (one- and two-segment messages, partial last column at ML-KEM-512: 800 = 5*136 + 120);
    the second, ragged wave takes the synchronous staging.  Both must agree with the oracle."""
    ekl, dkl, cl = SIZES[pset]
    n = 67
    d, z, m = seeds("emu-fw-d", n, pset), seeds("emu-fw-z", n, pset), seeds("emu-fw-m", n, pset)
    ek, dk = np.zeros((n, ekl), np.uint8), np.zeros((n, dkl), np.uint8)
    assert emu.emu_keygen(pset, C.c_size_t(n), p8(d), p8(z), p8(ek), p8(dk)) == 0
    ek_o, dk_o = oracle.keygen(pset, d, z)
    assert (ek == ek_o).all() and (dk == dk_o).all()
    c, K = np.zeros((n, cl), np.uint8), np.zeros((n, 32), np.uint8)
    assert emu.emu_encaps(pset, C.c_size_t(n), p8(ek), p8(m), p8(c), p8(K), None) == 0
    c_o, K_o = oracle.encaps(pset, ek, m)
    assert (c == c_o).all() and (K == K_o).all()
    cb, dkb = c.copy(), dk.copy()
    cb[[0, 17, 63, 66], [3, cl - 1, 100, 7]] ^= 4      # implicit rejection: K = J(z || c) from the DMA-fed sponge
    dkb[[5, 64], dkl - 40] ^= 1                        # stored H(ek) corrupted -> -5
    Kd, st = np.zeros((n, 32), np.uint8), np.zeros(n, np.int32)
    assert emu.emu_decaps(pset, C.c_size_t(n), p8(dkb), p8(cb), p8(Kd), st.ctypes.data_as(C.POINTER(C.c_int32)), 1) == 0
    Ko, sto = oracle.decaps(pset, dkb, cb)
    assert (st == sto).all() and sorted(np.nonzero(st)[0].tolist()) == [5, 64]
    assert (Kd[sto == 0] == Ko[sto == 0]).all()
    for i in (0, 17, 63, 66):
        assert not (Kd[i] == K[i]).all()


@pytest.mark.parametrize("resume_cap", (64, 2, 0))
def test_emu_matrix_sampler_with_leftover_pass(emu, oracle, resume_cap):
    """Production SampleNTT path: three-block main kernel + the leftover passes.  With 576 sponges about 0.8 % (4-5) need
    a 4th squeeze block; the test requires that the leftover path was taken.  Leftovers are handed over with their sponge
    state (k_sample_resume); a small resume capacity pushes the overflow (or, with capacity 0, everything) onto the restart
    list that k_sample redoes from the seed."""
    emu.emu_config(C.c_size_t(0), C.c_size_t(0))
    emu.emu_resume_cap(resume_cap)
    k, n = 3, 64
    rho = seeds("emu-rho", n, 31)
    A = np.zeros((n, k * k, 256), np.uint16)
    left = emu.emu_sample_matrix(k, C.c_size_t(n), p8(rho), 1, p16(A))
    emu.emu_resume_cap(64)
    total, restarted = left & 0xFFFF, left >> 16
    assert total >= 2, "seed set must exercise the leftover passes"
    assert restarted == max(0, total - resume_cap), (total, restarted)    # records beyond the capacity restart from the seed
    for i in range(n):
        for a in range(k):
            for b in range(k):
                seed = np.concatenate([rho[i], [a, b]]).astype(np.uint8)   # Encrypt order: B[32] = row, B[33] = col
                assert (A[i, a * k + b] == oracle.sample_ntt(seed)).all(), (i, a, b)
    assert A.max() < 3329


def test_emu_chunk_and_hchunk_loops(emu, oracle):
    """cap = 3 items per chunk, hcap = 7 items per h-chunk, 17 items: every loop boundary is crossed."""
    emu.emu_config(C.c_size_t(3), C.c_size_t(7))
    try:
        pset, n = 768, 17
        ekl, dkl, cl = SIZES[pset]
        d, z, m = seeds("emu-cd", n, 9), seeds("emu-cz", n, 9), seeds("emu-cm", n, 9)
        ek, dk = np.zeros((n, ekl), np.uint8), np.zeros((n, dkl), np.uint8)
        assert emu.emu_keygen(pset, C.c_size_t(n), p8(d), p8(z), p8(ek), p8(dk)) == 0
        ek_o, dk_o = oracle.keygen(pset, d, z)
        assert (ek == ek_o).all() and (dk == dk_o).all()
        c, K = np.zeros((n, cl), np.uint8), np.zeros((n, 32), np.uint8)
        assert emu.emu_encaps(pset, C.c_size_t(n), p8(ek), p8(m), p8(c), p8(K), None) == 0
        c_o, K_o = oracle.encaps(pset, ek, m)
        assert (c == c_o).all() and (K == K_o).all()
        cb = c.copy()
        cb[[2, 8, 16], [0, 500, 1087]] ^= 1
        Kd, st = np.zeros((n, 32), np.uint8), np.zeros(n, np.int32)
        assert emu.emu_decaps(pset, C.c_size_t(n), p8(dk), p8(cb), p8(Kd), st.ctypes.data_as(C.POINTER(C.c_int32)), 1) == 0
        Ko, sto = oracle.decaps(pset, dk, cb)
        assert (st == 0).all() and (sto == 0).all() and (Kd == Ko).all()
    finally:
        emu.emu_config(C.c_size_t(0), C.c_size_t(0))


def test_emu_raw_sponge_nist_examples_and_bit_lengths(emu, oracle, golden):
    """the bare sponge kernel + the host padding helper (mlkem_sha3_pad_bits in libmlkem_amd.so,
    a pure host function) against the 16 NIST FIPS-202 examples the reference ships, incl. 5/30/1605/1630-bit messages."""
    lib = ge.load_package().load_library()
    for ex in golden["G8_nist_sha3"]:
        bits = np.array([int(ch) for ch in ex["msg_bits"]], np.uint8)
        rate, want = ex["rate_bytes"], bytes.fromhex(ex["out"])
        padded = np.zeros(((bits.size + 6) // (8 * rate) + 2) * rate, np.uint8)
        nb = lib.mlkem_sha3_pad_bits(bits.ctypes.data, bits.size, int(ex["xof"]), rate, padded.ctypes.data, padded.size)
        assert nb > 0
        n = 3   # same message in three lanes
        msgs = np.tile(padded[: nb * rate], (n, 1))
        stride = (len(want) + 3) // 4 * 4
        out = np.zeros((n, stride), np.uint8)
        assert emu.emu_sponge_raw(rate, C.c_size_t(n), p8(msgs), nb, p8(out), len(want), C.c_size_t(stride)) == 0
        assert bytes(out[2, : len(want)]) == want, ex["file"]
    rng = np.random.default_rng(8)
    for rate, xof in ((136, False), (168, True), (72, False), (104, False), (144, False), (136, True)):
        for nbits in (0, 1, 7, 8 * rate - 3, 8 * rate - 2, 8 * rate - 1, 8 * rate, 8 * rate + 1, 2501):
            bits = rng.integers(0, 2, nbits).astype(np.uint8)
            padded = np.zeros((nbits // (8 * rate) + 2) * rate, np.uint8)
            nb = lib.mlkem_sha3_pad_bits(bits.ctypes.data, nbits, int(xof), rate, padded.ctypes.data, padded.size)
            outlen = 200 if xof else 32
            out = np.zeros((1, outlen), np.uint8)
            m1 = padded[: nb * rate][None].copy()
            assert emu.emu_sponge_raw(rate, C.c_size_t(1), p8(m1), nb, p8(out), outlen, C.c_size_t(outlen)) == 0
            assert (out[0] == oracle.sponge_bits(rate, xof, bits, outlen)).all(), (rate, xof, nbits)


def test_emu_wave_sponge_any_rate(emu, oracle):
    """k_sponge_raw_w (mlkem_small.hpp): one sponge per wave, ANY rate of 1..199 bytes -- what sha3_b(..., c, ...) of the shim runs
    for capacities other than the SHA-3 / SHAKE ones, and what every small call runs.  Host padding (mlkem_sha3_pad_suffix, any
    byte rate) + the kernel against the oracle's bit-granular sponge (itself pinned against the live reference at these rates:
    tests/test_oracle_vs_reference.py), incl. outputs of several blocks and an unaligned output stride."""
    lib = ge.load_package().load_library()
    rng = np.random.default_rng(9)
    # (64 host threads per sponge and a barrier per cross-lane operation: ~0.2 s per permutation, so few blocks per case)
    for rate, nbits, sfx in ((1, 0, (0, 1)), (1, 13, (1, 1, 1, 1)), (13, 8 * 13 + 5, (1, 1)), (72, 40, (0, 1)), (137, 8 * 137 - 2, (1, 1, 1, 1)),
                             (137, 1203, (0, 1)), (199, 8 * 199 + 5, (1, 1))):
        if True:
            bits = rng.integers(0, 2, nbits).astype(np.uint8)
            sb = np.array(sfx, np.uint8)
            padded = np.zeros((nbits // (8 * rate) + 2) * rate, np.uint8)
            nb = lib.mlkem_sha3_pad_suffix(bits.ctypes.data, nbits, sb.ctypes.data, sb.size, rate, padded.ctypes.data, padded.size)
            assert nb > 0, (rate, nbits)
            outlen, n = max(3, rate + rate // 2 + 1), 2 if rate == 13 else 1
            stride = outlen + 3
            msgs = np.tile(padded[: nb * rate], (n, 1))
            out = np.zeros((n, stride), np.uint8)
            assert emu.emu_sponge_raw_wave(rate, C.c_size_t(n), p8(msgs), nb, p8(out), outlen, C.c_size_t(stride)) == 0
            want = oracle.sponge_bits_sfx(rate, sb, bits, outlen)
            assert (out[:, :outlen] == want).all() and not out[:, outlen:].any(), (rate, nbits)
    assert lib.mlkem_sha3_pad_suffix(None, 0, None, 0, 200, None, 0) == -101 and lib.mlkem_sha3_pad_suffix(None, 0, None, 0, 0, None, 0) == -101


def test_emu_cell_converters(emu):
    """4-byte `union byte` cells <-> packed bytes; garbage in the upper 24 bits is ignored (F1)."""
    rng = np.random.default_rng(4)
    for n in (0, 1, 15, 16, 17, 1000, 4099):
        b = rng.integers(0, 256, n).astype(np.uint8)
        cells = b.astype(np.uint32) | (rng.integers(0, 1 << 24, n).astype(np.uint32) << 8)
        out = np.full(n + 1, 0xEE, np.uint8)
        emu.emu_cells(1, C.c_size_t(n), cells.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        assert (out[:n] == b).all() and out[n] == 0xEE
        back = np.full(n + 1, 0xFFFFFFFF, np.uint32)
        emu.emu_cells(0, C.c_size_t(n), b.ctypes.data_as(C.c_void_p), back.ctypes.data_as(C.c_void_p))
        assert (back[:n] == b).all() and back[n] == 0xFFFFFFFF


def test_emu_compress_values_every_d_whole_field(emu, golden_npz):
    """k_compress_values (general Compress / Decompress entry, any d in 1..12, any 12-bit input) against the tables the
    real reference produced (golden G4, whole 12-bit field); junk above bit 11 is ignored like the reference's bit-field."""
    x = (np.arange(4096, dtype=np.uint16) | np.uint16(0xF000 & (np.arange(4096) * 4096))).astype(np.uint16)
    x = np.concatenate([x, x[:5]])   # a ragged tail past the 8-value groups
    for d in range(1, 13):
        out = np.zeros_like(x)
        assert emu.emu_compress_values(0, d, C.c_size_t(x.size), p16(x), p16(out)) == 0
        assert (out[:4096] == golden_npz["g4_compress_full"][d - 1]).all() and (out[4096:] == out[:5]).all(), d
        assert emu.emu_compress_values(1, d, C.c_size_t(x.size), p16(x), p16(out)) == 0
        assert (out[:4096] == golden_npz["g4_decompress_full"][d - 1]).all(), d
    assert emu.emu_compress_values(0, 13, C.c_size_t(8), p16(x), p16(out)) != 0


def test_emu_fcanon_floor_exhaustive(emu):
    emu.emu_fcanon_floor_exhaustive.restype = C.c_long
    assert emu.emu_fcanon_floor_exhaustive() == 0


def test_emu_register_ntt_four_polys_per_wave(emu, oracle, golden_npz):
    """k_ntt4_batch (mlkem_rntt.hpp: four polynomials per wave in registers, DPP cross-lane butterflies, emulated with
    shuffles): every batch size mod 4 (rows beyond n idle), raw 12-bit inputs, extreme-magnitude inputs that
    drive the lazy bounds to their limits."""
    rng = np.random.default_rng(5)
    cases = [rng.integers(0, 3329, (n, 256)).astype(np.uint16) for n in (1, 2, 3, 4, 5, 7, 9)]
    cases.append(rng.integers(0, 4096, (4, 256)).astype(np.uint16))                       # non-canonical 12-bit values
    ext = np.zeros((6, 256), np.uint16)
    ext[0] = 4095; ext[1] = 3328; ext[2, ::2] = 4095; ext[3, 1::2] = 4095
    ext[4] = np.where(np.arange(256) & 128, 4095, 0); ext[5] = np.where(np.arange(256) & 2, 0, 4095)
    cases.append(ext)
    cases.append((rng.integers(0, 3329, (2, 256)) | 0xF000).astype(np.uint16))           # bits 12..15 are ignored
    one = np.zeros((3, 256), np.uint16)
    one[1, 0] = 4095                                                                      # NTT(4095 x^0)[254] = 4095 in the reference
    cases.append(one)
    for a in cases:
        # forward: the reference's own (non-modular, ml_kem.c:317) behaviour on coefficients >= q is reproduced exactly;
        # inverse: inputs >= q are outside the contract (the reference overflows a signed int there): reduced first
        want_f, want_i = oracle.ntt(a & 0xFFF).reshape(a.shape), oracle.intt((a & 0xFFF) % 3329).reshape(a.shape)
        out = np.zeros_like(a)
        emu.emu_ntt(0, C.c_size_t(a.shape[0]), p16(a), p16(out))
        assert (out == want_f).all(), a.shape
        emu.emu_ntt(1, C.c_size_t(a.shape[0]), p16(a), p16(out))
        assert (out == want_i).all(), a.shape
    assert want_f[1, 254] == 4095


def _vecmul_expect(oracle, u, v):
    """VectorMultiply as the reference composes it (ml_kem.c:618-638): MultiplyNTTs per term, PolyAddition of the partial sums"""
    out = np.zeros((u.shape[0], 256), np.uint16)
    for i in range(u.shape[0]):
        w = oracle.multiply_ntts(u[i, 0], v[i, 0])
        for j in range(1, u.shape[1]):
            w = oracle.poly_add(w, oracle.multiply_ntts(u[i, j], v[i, j]))
        out[i] = w
    return out


def test_emu_poly_add_sub_and_vector_multiply(emu, oracle, golden_npz):
    """SURVEY 8a rows a12 / a13 as stand-alone entries: PolyAddition / PolySubtraction against the reference-generated goldens
    and (raw 12-bit inputs, ragged lengths) the oracle; VectorMultiply for k = 1..4 against MultiplyNTTs + PolyAddition."""
    a, b = golden_npz["rand_a"].copy(), golden_npz["rand_b"].copy()
    out = np.zeros_like(a)
    emu.emu_poly_addsub(0, C.c_size_t(a.size), p16(a), p16(b), p16(out))
    assert (out == golden_npz["rand_ab_add"]).all()
    emu.emu_poly_addsub(1, C.c_size_t(a.size), p16(a), p16(b), p16(out))
    assert (out == golden_npz["rand_ab_sub"]).all()
    na, nb = golden_npz["nc_a"].copy(), golden_npz["nc_b"].copy()
    junk = (na | 0xF000).astype(np.uint16)   # bits 12..15 are not part of the reference's field
    for sub, fn in ((0, oracle.poly_add), (1, oracle.poly_sub)):
        want = np.stack([fn(na[i], nb[i]) for i in range(na.shape[0])])
        out = np.zeros_like(na)
        emu.emu_poly_addsub(sub, C.c_size_t(na.size), p16(junk), p16(nb), p16(out))
        assert (out == want).all(), sub
        flat_a, flat_b = na.reshape(-1)[:1003].copy(), nb.reshape(-1)[:1003].copy()   # not a multiple of 8: scalar tail
        o2 = np.zeros(1003, np.uint16)
        emu.emu_poly_addsub(sub, C.c_size_t(1003), p16(flat_a), p16(flat_b), p16(o2))
        assert (o2 == want.reshape(-1)[:1003]).all(), sub
    rng = np.random.default_rng(618)
    for k in (1, 2, 3, 4):
        for hi in (3329, 4096):
            u = rng.integers(0, hi, (5, k, 256)).astype(np.uint16)
            v = rng.integers(0, hi, (5, k, 256)).astype(np.uint16)
            w = np.zeros((5, 256), np.uint16)
            assert emu.emu_vecmul(k, C.c_size_t(5), p16(u), p16(v), p16(w)) == 0
            assert (w == _vecmul_expect(oracle, u, v)).all(), (k, hi)
    assert emu.emu_vecmul(5, C.c_size_t(1), p16(a), p16(b), p16(out)) == -1


@pytest.mark.parametrize("pset", (512, 768, 1024))
def test_emu_two_items_per_wave_every_batch_parity_and_piece_compare(emu, oracle, pset):
    """mlkem_kpke2.hpp: two items per wave, so odd batches leave the upper half-wave without an item (it recomputes the last
    item and must store nothing), and the re-encryption compare works on register pieces whose dwords neighbouring lanes share
    (d = 10: lane pairs, d = 11 / 5: quads).  For n = 1, 2, 3 (+ guard rows behind every output): K-PKE.KeyGen / Encrypt bytes
    against the oracle; then one flipped bit in a byte of EVERY dword class of the ciphertext — first / shared / last dword of
    a lane group in every polynomial row and in v — must turn K into the implicit-rejection key."""
    ekl, dkl, cl = SIZES[pset]
    k = {512: 2, 768: 3, 1024: 4}[pset]
    du, dv = (10, 4) if k < 4 else (11, 5)
    for n in (1, 2, 3):
        d, z, m = seeds("k2-d", n, pset), seeds("k2-z", n, pset), seeds("k2-m", n, pset)
        ek, dk = np.full((n + 1, ekl), 0xA5, np.uint8), np.full((n + 1, dkl), 0xA5, np.uint8)
        assert emu.emu_keygen(pset, C.c_size_t(n), p8(d), p8(z), p8(ek), p8(dk)) == 0
        ek_o, dk_o = oracle.keygen(pset, d, z)
        assert (ek[:n] == ek_o).all() and (dk[:n] == dk_o).all() and (ek[n] == 0xA5).all() and (dk[n] == 0xA5).all(), n
        c, K = np.full((n + 1, cl), 0x5A, np.uint8), np.full((n + 1, 32), 0x5A, np.uint8)
        assert emu.emu_encaps(pset, C.c_size_t(n), p8(ek), p8(m), p8(c), p8(K), None) == 0
        c_o, K_o = oracle.encaps(pset, ek_o, m)
        assert (c[:n] == c_o).all() and (K[:n] == K_o).all() and (c[n] == 0x5A).all() and (K[n] == 0x5A).all(), n
    # byte positions: for every polynomial row, the first / second / last bytes of the first and last lane groups and one in
    # the middle; for v likewise
    pos = []
    for row in range(k):
        base = row * 32 * du
        pos += [base, base + 4, base + 9, base + 10, base + 11, base + 19, base + 20, base + 16 * du + 3, base + 32 * du - 12, base + 32 * du - 1]
    vb = k * 32 * du
    pos += [vb, vb + 2, vb + 4, vb + 5, vb + 19, vb + 16 * dv, vb + 32 * dv - 6, vb + 32 * dv - 1]
    n = len(pos)
    d, z, m = seeds("k2p-d", 1, pset), seeds("k2p-z", 1, pset), seeds("k2p-m", n, pset)
    ek1, dk1 = oracle.keygen(pset, d, z)
    ekr, dkr = np.repeat(ek1, n, 0), np.repeat(dk1, n, 0)
    c, K = oracle.encaps(pset, ekr, m)
    cb = c.copy()
    for i, p in enumerate(pos):
        cb[i, p] ^= 1 << (i % 8)
    Kd, st = np.zeros((n, 32), np.uint8), np.zeros(n, np.int32)
    assert emu.emu_decaps(pset, C.c_size_t(n), p8(dkr), p8(cb), p8(Kd), st.ctypes.data_as(C.POINTER(C.c_int32)), 1) == 0
    Ko, sto = oracle.decaps(pset, dkr, cb)
    assert (st == 0).all() and (sto == 0).all() and (Kd == Ko).all()
    assert not (Kd == K).all(axis=1).any()            # every one of them rejected
    Kd2 = np.zeros((n, 32), np.uint8)
    assert emu.emu_decaps(pset, C.c_size_t(n), p8(dkr), p8(c), p8(Kd2), st.ctypes.data_as(C.POINTER(C.c_int32)), 1) == 0
    assert (Kd2 == K).all()                            # and the untouched ciphertexts accepted


@pytest.mark.parametrize("pset,fips", ((768, 0), (1024, 1)))
def test_emu_one_sponge_per_wave_hash_kernels(emu, oracle, pset, fips):
    """mlkem_wkeccak.hpp (calls of at most `wide_max` items): H(ek) -> G and the dk hash check | J -> G with one sponge per
    wave, cross-lane fetches emulated by shuffles.  Two items (one in FIPS mode): keygen (its dk tail H(ek) || z per wave) -> encaps
    -> decaps of a corrupted stored H(ek) and a tampered ciphertext, bit for bit against the oracle."""
    n = 1 if fips else 2                                  # 64 host threads shuffle slowly: the FIPS variant runs one item
    ekl, dkl, cl = SIZES[pset]
    emu.emu_conformance(fips)
    emu.emu_wide_hash(C.c_size_t(16))
    try:
        d, z, m = seeds("wk-d", n, pset), seeds("wk-z", n, pset), seeds("wk-m", n, pset)
        oracle.set_conformance(bool(fips))
        ek, dk = oracle.keygen(pset, d, z)
        ek_e, dk_e = np.zeros((n, ekl), np.uint8), np.zeros((n, dkl), np.uint8)
        assert emu.emu_keygen(pset, C.c_size_t(n), p8(d), p8(z), p8(ek_e), p8(dk_e)) == 0     # dk tail: H(ek) per wave, then z
        assert (ek_e == ek).all() and (dk_e == dk).all()
        c, K = np.zeros((n, cl), np.uint8), np.zeros((n, 32), np.uint8)
        assert emu.emu_encaps(pset, C.c_size_t(n), p8(ek), p8(m), p8(c), p8(K), None) == 0
        c_o, K_o = oracle.encaps(pset, ek, m)
        assert (c == c_o).all() and (K == K_o).all()
        cb, dkb = c.copy(), dk.copy()
        cb[n - 1, 40] ^= 4                                 # last item: implicit rejection
        if n == 2:
            dkb[0, dkl - 40] ^= 1                          # item 0: stored H(ek) corrupted -> status -5
        Kd, st = np.zeros((n, 32), np.uint8), np.ones(n, np.int32)
        assert emu.emu_decaps(pset, C.c_size_t(n), p8(dkb), p8(cb), p8(Kd), st.ctypes.data_as(C.POINTER(C.c_int32)), 1) == 0
        Kd_o, st_o = oracle.decaps(pset, dkb, cb)
        assert st.tolist() == ([-5, 0] if n == 2 else [0]) and (st == st_o).all(), (st, st_o)
        assert (Kd[st_o == 0] == Kd_o[st_o == 0]).all()
        if n == 2:
            assert (Kd[0] == oracle.decaps_internal(pset, dkb[0], cb[0])).all()   # the -5 row still carries Decaps_internal's key (G on the stored h)
        assert (Kd[n - 1] != K[n - 1]).any()
    finally:
        emu.emu_wide_hash(C.c_size_t(0))
        emu.emu_conformance(0)
        oracle.set_conformance(False)


@pytest.mark.parametrize("pset,fips,waves", ((512, 0, 4), (768, 0, 8), (768, 0, 4), (1024, 1, 12)))
def test_emu_one_workgroup_per_item_kernels(emu, oracle, pset, fips, waves):
    """mlkem_small.hpp (calls of at most `small_max` items): KeyGen, Encaps and Decaps each as ONE launch, a workgroup of eight
    (latency form; twelve for Decaps of 1024) or four (dense form) waves per item -- wave-level SampleNTT (ballot + prefix-count
    compaction) and the PRF rows dealt out as jobs by a counter in LDS, H / G / J and the K-PKE bodies, handed over between the waves
    by flag counters in LDS (flag_signal / flag_wait).  An untouched and a tampered ciphertext (two items for 768, one for the others), then a corrupted stored
    H(ek) with the untouched ciphertext; bit for bit against the oracle (reference mode for 512 / 768, FIPS 203 mode -- PRF and J on SHAKE256 -- for 1024)."""
    n = 1
    ekl, dkl, cl = SIZES[pset]
    emu.emu_conformance(fips)
    emu.emu_small(C.c_size_t(16))
    emu.emu_small_latency(C.c_size_t(16 if waves >= 8 else 0))
    emu.emu_small_wide(C.c_size_t(16 if waves == 12 else 0))     # Decaps of k >= 3 with twelve waves
    try:
        d, z, m = seeds("sm-d", n, pset), seeds("sm-z", n, pset), seeds("sm-m", n, pset)
        oracle.set_conformance(bool(fips))
        ek, dk = oracle.keygen(pset, d, z)
        ek_e, dk_e = np.zeros((n, ekl), np.uint8), np.zeros((n, dkl), np.uint8)
        assert emu.emu_keygen(pset, C.c_size_t(n), p8(d), p8(z), p8(ek_e), p8(dk_e)) == 0
        assert (ek_e == ek).all() and (dk_e == dk).all()
        c, K = np.zeros((n, cl), np.uint8), np.zeros((n, 32), np.uint8)
        st = np.ones(n, np.int32)
        assert emu.emu_encaps(pset, C.c_size_t(n), p8(ek), p8(m), p8(c), p8(K), st.ctypes.data_as(C.POINTER(C.c_int32)) if fips else None) == 0
        c_o, K_o = oracle.encaps(pset, ek, m)
        assert (c == c_o).all() and (K == K_o).all()
        if fips:
            assert (st == 0).all()
        cb, dkb = c.copy(), dk.copy()
        cb[n - 1, 40] ^= 4                                 # last item: implicit rejection (the only item where n = 1)
        Kd, st = np.zeros((n, 32), np.uint8), np.ones(n, np.int32)
        assert emu.emu_decaps(pset, C.c_size_t(n), p8(dkb), p8(cb), p8(Kd), st.ctypes.data_as(C.POINTER(C.c_int32)), 1) == 0
        Kd_o, st_o = oracle.decaps(pset, dkb, cb)
        assert (st == 0).all() and (st_o == 0).all() and (Kd == Kd_o).all()
        assert (Kd[n - 1] != K[n - 1]).any() and (n == 1 or (Kd[0] == K[0]).all())
        if pset != 768:
            return
        dkb[0, dkl - 40] ^= 1                              # stored H(ek) corrupted -> status -5, key = Decaps_internal's
        assert emu.emu_decaps(pset, C.c_size_t(1), p8(dkb), p8(c), p8(Kd), st.ctypes.data_as(C.POINTER(C.c_int32)), 1) == 0
        assert st[0] == -5 and (Kd[0] == oracle.decaps_internal(pset, dkb[0], c[0])).all()   # G ran on the corrupted stored h
    finally:
        emu.emu_small(C.c_size_t(0))
        emu.emu_small_latency(C.c_size_t(256))
        emu.emu_small_wide(C.c_size_t(0))
        emu.emu_conformance(0)
        oracle.set_conformance(False)


@pytest.mark.parametrize("variant,bound,cap", ((1, 1900, 278), (2, 2100, 250)))
def test_emu_sampler_fifth_block_cap_and_seed_mutation_retry(emu, oracle, variant, bound, cap):
    """The branches of SampleNTT that real SHAKE output never reaches (ml_kem.c:221-242: a fifth squeeze block needs < 256
    accepted candidates out of 448, probability ~ e^-40; the 279-triple limit and the B[32]++, B[33]++ retry ~ e^-220).  The
    product's sampler templates are instantiated with a LOWER acceptance bound and triple cap so that about half of the sponges
    exhaust their triples: every sponge then runs a fourth and a fifth block, the cap inside the fifth, and a large share the
    retry -- some twice or more.  All four forms against the oracle's parameterised restatement (orc_sample_ntt_bounded; its
    (3329, 279) instance is the oracle every other test uses), ~900 sponges per variant:
      * batch path: k_sample_main (three blocks) -> k_sample_resume (blocks four, five, cap) -> restart list -> k_sample
        (list mode, the retry), with a large and a small resume capacity;
      * k_sample in direct mode (calls of <= 2048 items);
      * stand-alone seeds, lane-sliced (k_sample) and one sponge per wave (k_sample_ntt_w = the sampler of mlkem_small.hpp).
    Parity with the reference ON THIS BRANCH rests on reading ml_kem.c:221-242 (oracle and kernels were written from it);
    no reference run can reach it."""
    limit = cap + 1                                           # the reference counts the triple that trips the limit
    k, n = 3, 24
    rho = seeds("cap-rho%d" % variant, n, 31)
    want = np.zeros((n, k * k, 256), np.uint16)
    retries = []
    for i in range(n):
        for a in range(k):
            for b in range(k):
                seed = np.concatenate([rho[i], [a, b]]).astype(np.uint8)
                want[i, a * k + b], r, Bo = oracle.sample_ntt_bounded(seed, bound, limit)
                assert Bo[32] == (a + r) & 0xFF and Bo[33] == (b + r) & 0xFF and (Bo[:32] == rho[i]).all()
                retries.append(r)
    retries = np.array(retries)
    assert (retries >= 1).sum() >= 40 and (retries >= 2).sum() >= 8, np.bincount(retries)
    assert want.max() < bound
    emu.emu_config(C.c_size_t(0), C.c_size_t(0))
    for resume_cap, direct in ((256, 0), (16, 0), (64, 1)):
        emu.emu_resume_cap(resume_cap)
        A = np.zeros((n, k * k, 256), np.uint16)
        left = emu.emu_sample_matrix_bounded(variant, k, C.c_size_t(n), p8(rho), 1, p16(A), direct)
        emu.emu_resume_cap(64)
        assert left >= 0
        if not direct:
            handed, restarted = left & 0xFFFF, left >> 16
            assert handed == n * k * k                        # after three blocks nobody has 256 coefficients under this bound
            assert restarted >= max(0, handed - resume_cap) + (retries >= 1).sum() * (resume_cap >= handed)
        assert (A == want).all(), (resume_cap, direct, np.argwhere((A != want).any(axis=2))[:5])
    # stand-alone seeds, arbitrary bytes 32 and 33 (the mutation wraps modulo 256)
    ns = 256
    s34 = np.concatenate([seeds("cap-s%d" % variant, ns, 7), seeds("cap-t%d" % variant, ns, 8)[:, :2]], axis=1).astype(np.uint8)
    s34[:8, 32] = 255
    s34[8:16, 33] = 254
    ws = [oracle.sample_ntt_bounded(s34[i], bound, limit) for i in range(ns)]
    wpoly = np.stack([w[0] for w in ws])
    assert sum(w[1] >= 1 for w in ws) >= 60
    out, rt = np.zeros((ns, 256), np.uint16), np.full(ns, 77, np.uint8)
    assert emu.emu_sample_ntt_bounded(variant, 0, C.c_size_t(ns), p8(s34), p16(out), p8(rt)) == 0
    assert (out == wpoly).all()
    # the retry counts the shim uses to advance the caller's B[32], B[33] like the reference does (ml_kem.c:237-242)
    assert rt.tolist() == [w[1] for w in ws]
    assert all(((int(s34[i, 32]) + int(rt[i])) & 0xFF, (int(s34[i, 33]) + int(rt[i])) & 0xFF) == (int(ws[i][2][32]), int(ws[i][2][33])) for i in range(ns))
    # one sponge per wave (64 host threads per sponge, a barrier per cross-lane operation: slow): 4 seeds chosen by their
    # retry count -- none, one (twice), two or more
    r_of = np.array([w[1] for w in ws])
    pick = np.concatenate([np.nonzero(r_of == 0)[0][:1], np.nonzero(r_of == 1)[0][:2], np.nonzero(r_of >= 2)[0][:1]])
    assert len(pick) == 4
    sw = np.ascontiguousarray(s34[pick])
    out, rt = np.zeros((len(pick), 256), np.uint16), np.full(len(pick), 77, np.uint8)
    assert emu.emu_sample_ntt_bounded(variant, 1, C.c_size_t(len(pick)), p8(sw), p16(out), p8(rt)) == 0
    assert (out == wpoly[pick]).all() and rt.tolist() == r_of[pick].tolist()


def test_emu_sampler_templates_at_product_parameters_and_wave_form(emu, oracle):
    """Variant 0 = (3329, 278): the same entry points at the product's parameters against the plain oracle, incl. the
    one-sponge-per-wave SampleNTT on the test06 / test08 recipes' seeds."""
    ns = 8
    s34 = np.concatenate([seeds("capp-s", ns, 7), seeds("capp-t", ns, 8)[:, :2]], axis=1).astype(np.uint8)
    for it in range(7):
        s34[it] = [(it * i + i) & 0xFF for i in range(34)]   # Test_Archive/SampleNTT_test06.c
    s34[7] = [2 * i for i in range(34)]                       # Test_Archive/NTT_test08.c
    want = np.stack([oracle.sample_ntt(s34[i]) for i in range(ns)])
    for wave in (0, 1):
        out, rt = np.zeros((ns, 256), np.uint16), np.full(ns, 9, np.uint8)
        assert emu.emu_sample_ntt_bounded(0, wave, C.c_size_t(ns), p8(s34), p16(out), p8(rt)) == 0
        assert (out == want).all() and not rt.any(), wave
