"""ctypes bindings for the CHECKERS under oracle/ — test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(the product package crystals-kyber_amd never does).

  Oracle  -> oracle/liboracle_mlkem.so   (our C restatement, oracle/mlkem_oracle.c)
  Ref     -> oracle/_ref/libmlkem_ref.so (the real reference, built by oracle/Makefile from
             $MLKEM_REF_DIR when that directory exists; the built .so travels to the GPU box)
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SIZES = {512: (800, 1632, 768), 768: (1184, 2400, 1088), 1024: (1568, 3168, 1568)}
PARAMS = {512: (2, 3, 2, 10, 4), 768: (3, 2, 2, 10, 4), 1024: (4, 2, 2, 11, 5)}

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
i32p = C.POINTER(C.c_int32)


def build(verbose=False):
    """Compile the checkers (idempotent).  The reference build is attempted only when its
    sources are present (this container); elsewhere the prebuilt oracle/_ref is used."""
    r = subprocess.run(["make", "-s", "-C", HERE, "all"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(r.stdout)


def _p8(a):
    return a.ctypes.data_as(u8p)


def _p16(a):
    return a.ctypes.data_as(u16p)


def _u8(x, n=None):
    a = np.ascontiguousarray(np.frombuffer(bytes(x), dtype=np.uint8) if isinstance(x, (bytes, bytearray)) else x,
                             dtype=np.uint8)
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


def _u16(x):
    a = np.ascontiguousarray(x, dtype=np.uint16)
    assert a.size == 256
    return a


class Oracle:
    """Our CPU restatement (oracle/mlkem_oracle.c)."""

    def __init__(self):
        path = os.path.join(HERE, "liboracle_mlkem.so")
        if not os.path.exists(path):
            build()
        self.lib = L = C.CDLL(path)
        L.orc_keygen_batch.argtypes = [C.c_int, C.c_size_t, u8p, u8p, u8p, u8p]
        L.orc_encaps_batch.argtypes = [C.c_int, C.c_size_t, u8p, u8p, u8p, u8p]
        L.orc_decaps_batch.argtypes = [C.c_int, C.c_size_t, u8p, u8p, u8p, i32p]
        L.orc_ntt_batch.argtypes = [C.c_size_t, u16p, u16p]
        L.orc_intt_batch.argtypes = [C.c_size_t, u16p, u16p]
        L.orc_sponge.argtypes = [C.c_uint, C.c_uint8, u8p, C.c_size_t, u8p, C.c_size_t]
        L.orc_sponge_bits.argtypes = [C.c_uint, C.c_int, u8p, C.c_size_t, u8p, C.c_size_t]
        L.orc_compress.restype = C.c_uint
        L.orc_decompress.restype = C.c_uint
        L.orc_bitrev7.restype = C.c_uint
        L.orc_sample_ntt.restype = C.c_int

    def set_conformance(self, fips203):
        """0 = reference-compatible (default), 1 = FIPS 203 (PRF/J on SHAKE256, real modulus check)."""
        self.lib.orc_set_conformance(int(bool(fips203)))

    # -- batch KEM ---------------------------------------------------------------------
    def keygen(self, pset, d, z):
        ekl, dkl, _ = SIZES[pset]
        d, z = _u8(d), _u8(z)
        n = d.size // 32
        ek = np.zeros((n, ekl), np.uint8)
        dk = np.zeros((n, dkl), np.uint8)
        self.lib.orc_keygen_batch(pset, n, _p8(d), _p8(z), _p8(ek), _p8(dk))
        return ek, dk

    def encaps(self, pset, ek, m):
        ekl, _, cl = SIZES[pset]
        ek, m = _u8(ek), _u8(m)
        n = m.size // 32
        assert ek.size == n * ekl
        c = np.zeros((n, cl), np.uint8)
        K = np.zeros((n, 32), np.uint8)
        self.lib.orc_encaps_batch(pset, n, _p8(ek), _p8(m), _p8(c), _p8(K))
        return c, K

    def decaps(self, pset, dk, c):
        """KEM_Decaps semantics (hash check included): returns (K, status)."""
        _, dkl, cl = SIZES[pset]
        dk, c = _u8(dk), _u8(c)
        n = c.size // cl
        assert dk.size == n * dkl
        K = np.zeros((n, 32), np.uint8)
        st = np.zeros(n, np.int32)
        self.lib.orc_decaps_batch(pset, n, _p8(dk), _p8(c), _p8(K), st.ctypes.data_as(i32p))
        return K, st

    def decaps_internal(self, pset, dk, c):
        dk, c = _u8(dk), _u8(c)
        K = np.zeros(32, np.uint8)
        self.lib.orc_decaps_internal(pset, _p8(dk), _p8(c), _p8(K))
        return K

    def kem_decaps(self, pset, dk, c):
        dk, c = _u8(dk), _u8(c)
        K = np.zeros(32, np.uint8)
        rc = self.lib.orc_kem_decaps(pset, _p8(dk), dk.size, _p8(c), c.size, _p8(K))
        return rc, K

    def kem_encaps_check(self, pset, ek):
        ek = _u8(ek)
        return self.lib.orc_kem_encaps_check(pset, _p8(ek), ek.size)

    # -- K-PKE ---------------------------------------------------------------------------
    def _params(self, pset):
        class P(C.Structure):
            _fields_ = [("set", C.c_int)] + [(n, C.c_uint) for n in
                                             ("k", "eta1", "eta2", "du", "dv", "ek_len", "dk_len", "c_len")]
        p = P()
        assert self.lib.orc_params_init(pset, C.byref(p)) == 0
        return p

    def pke_keygen(self, pset, d):
        p = self._params(pset)
        d = _u8(d, 32)
        ek = np.zeros(p.ek_len, np.uint8)
        dk = np.zeros(384 * p.k, np.uint8)
        self.lib.orc_pke_keygen(C.byref(p), _p8(d), _p8(ek), _p8(dk))
        return ek, dk

    def pke_encrypt(self, pset, ek, m, r):
        p = self._params(pset)
        ek, m, r = _u8(ek), _u8(m, 32), _u8(r, 32)
        c = np.zeros(p.c_len, np.uint8)
        self.lib.orc_pke_encrypt(C.byref(p), _p8(ek), _p8(m), _p8(r), _p8(c))
        return c

    def pke_decrypt(self, pset, dk_pke, c):
        p = self._params(pset)
        dk_pke, c = _u8(dk_pke), _u8(c)
        m = np.zeros(32, np.uint8)
        self.lib.orc_pke_decrypt(C.byref(p), _p8(dk_pke), _p8(c), _p8(m))
        return m

    # -- primitives ------------------------------------------------------------------------
    def ntt(self, f):
        f = np.ascontiguousarray(f, np.uint16).reshape(-1, 256)
        out = np.zeros_like(f)
        self.lib.orc_ntt_batch(f.shape[0], _p16(f), _p16(out))
        return out

    def intt(self, f):
        f = np.ascontiguousarray(f, np.uint16).reshape(-1, 256)
        out = np.zeros_like(f)
        self.lib.orc_intt_batch(f.shape[0], _p16(f), _p16(out))
        return out

    def multiply_ntts(self, a, b):
        a, b = _u16(a), _u16(b)
        h = np.zeros(256, np.uint16)
        self.lib.orc_multiply_ntts(_p16(a), _p16(b), _p16(h))
        return h

    def poly_add(self, a, b):
        a, b = _u16(a), _u16(b)
        h = np.zeros(256, np.uint16)
        self.lib.orc_poly_add(_p16(a), _p16(b), _p16(h))
        return h

    def poly_sub(self, a, b):
        a, b = _u16(a), _u16(b)
        h = np.zeros(256, np.uint16)
        self.lib.orc_poly_sub(_p16(a), _p16(b), _p16(h))
        return h

    def sample_ntt(self, B):
        B = _u8(B, 34)
        a = np.zeros(256, np.uint16)
        self.lib.orc_sample_ntt(_p8(B), _p16(a))
        return a

    def sample_ntt_bounded(self, B, bound, limit):
        """Test-only generalisation of SampleNTT (acceptance bound, triple limit; the reference is (3329, 279)).
        Returns (polynomial, retries, the 34 seed bytes of the attempt that succeeded)."""
        B = _u8(B, 34)
        a = np.zeros(256, np.uint16)
        Bo = np.zeros(34, np.uint8)
        self.lib.orc_sample_ntt_bounded.restype = C.c_int
        r = self.lib.orc_sample_ntt_bounded(_p8(B), _p16(a), C.c_uint(bound), C.c_uint(limit), _p8(Bo))
        return a, r, Bo

    def sample_cbd(self, B, eta):
        B = _u8(B, 64 * eta)
        f = np.zeros(256, np.uint16)
        self.lib.orc_sample_cbd(_p8(B), eta, _p16(f))
        return f

    def compress(self, x, d):
        return self.lib.orc_compress(int(x), int(d))

    def decompress(self, y, d):
        return self.lib.orc_decompress(int(y), int(d))

    def bitrev7(self, r):
        return self.lib.orc_bitrev7(int(r))

    def byte_encode(self, F, d):
        F = _u16(F)
        B = np.zeros(32 * d, np.uint8)
        self.lib.orc_byte_encode(_p16(F), d, _p8(B))
        return B

    def byte_decode(self, B, d):
        B = _u8(B, 32 * d)
        F = np.zeros(256, np.uint16)
        self.lib.orc_byte_decode(_p8(B), d, _p16(F))
        return F

    # -- hashes --------------------------------------------------------------------------
    def sponge(self, rate, suffix, msg, outlen):
        msg = _u8(msg)
        out = np.zeros(outlen, np.uint8)
        self.lib.orc_sponge(rate, suffix, _p8(msg), msg.size, _p8(out), outlen)
        return out

    def sponge_bits(self, rate, xof, bits, outlen):
        bits = _u8(bits)
        out = np.zeros(outlen, np.uint8)
        self.lib.orc_sponge_bits(rate, int(xof), _p8(bits), bits.size, _p8(out), outlen)
        return out

    def sponge_bits_sfx(self, rate, sfx_bits, bits, outlen):
        """bit-granular sponge with the caller's suffix bits appended verbatim (sha3.c:414-429)"""
        bits, sfx = _u8(bits), _u8(sfx_bits)
        out = np.zeros(outlen, np.uint8)
        self.lib.orc_sponge_bits_sfx(rate, _p8(sfx), sfx.size, _p8(bits), bits.size, _p8(out), outlen)
        return out

    def prf(self, s, b, eta):
        s = _u8(s, 32)
        out = np.zeros(64 * eta, np.uint8)
        self.lib.orc_prf(_p8(s), C.c_uint8(b), eta, _p8(out))
        return out

    def H(self, x):
        return self.sponge(136, 0x06, x, 32)

    def G(self, x):
        return self.sponge(72, 0x06, x, 64)

    def J(self, x):
        return self.sponge(168, 0x1F, x, 32)


class Ref:
    """The real reference compiled from /root/reference (oracle/_ref/libmlkem_ref.so)."""

    @staticmethod
    def path(o0=False):
        return os.path.join(HERE, "_ref", "libmlkem_ref_O0.so" if o0 else "libmlkem_ref.so")

    @staticmethod
    def available(o0=False):
        return os.path.exists(Ref.path(o0))

    def __init__(self, o0=False):
        if not Ref.available(o0):
            build()
        if not Ref.available(o0):
            raise FileNotFoundError("oracle/_ref not built and reference sources absent")
        self.lib = L = C.CDLL(Ref.path(o0))
        L.ref_compress.restype = C.c_uint
        L.ref_decompress.restype = C.c_uint
        L.ref_bitrev7.restype = C.c_uint
        L.ref_time_encaps_decaps.restype = C.c_double
        L.ref_time_encaps_decaps.argtypes = [C.c_int, C.c_int, u8p, u8p, u8p, u8p, u8p, C.POINTER(C.c_int)]
        if hasattr(L, "ref_time_triples"):   # absent from an oracle/_ref built by an earlier round
            L.ref_time_triples.restype = C.c_double
            L.ref_time_triples.argtypes = [C.c_int, C.c_int, u8p, u8p, u8p, u8p, u8p, u8p, u8p, C.POINTER(C.c_int)]

    def keygen(self, pset, d, z):
        ekl, dkl, _ = SIZES[pset]
        d, z = _u8(d).reshape(-1, 32), _u8(z).reshape(-1, 32)
        n = d.shape[0]
        ek = np.zeros((n, ekl), np.uint8)
        dk = np.zeros((n, dkl), np.uint8)
        for i in range(n):
            self.lib.ref_keygen(pset, _p8(d[i]), _p8(z[i]), _p8(ek[i]), _p8(dk[i]))
        return ek, dk

    def encaps(self, pset, ek, m):
        ekl, _, cl = SIZES[pset]
        m = _u8(m).reshape(-1, 32)
        n = m.shape[0]
        ek = _u8(ek).reshape(n, ekl)
        c = np.zeros((n, cl), np.uint8)
        K = np.zeros((n, 32), np.uint8)
        for i in range(n):
            self.lib.ref_encaps(pset, _p8(ek[i]), _p8(m[i]), _p8(c[i]), _p8(K[i]))
        return c, K

    def decaps(self, pset, dk, c):
        _, dkl, cl = SIZES[pset]
        c = _u8(c).reshape(-1, cl)
        n = c.shape[0]
        dk = _u8(dk).reshape(n, dkl)
        K = np.zeros((n, 32), np.uint8)
        st = np.zeros(n, np.int32)
        for i in range(n):
            st[i] = self.lib.ref_kem_decaps(pset, _p8(dk[i]), dkl, _p8(c[i]), cl, _p8(K[i]))
        return K, st

    def kem_decaps(self, pset, dk, c):
        dk, c = _u8(dk), _u8(c)
        K = np.zeros(32, np.uint8)
        rc = self.lib.ref_kem_decaps(pset, _p8(dk), dk.size, _p8(c), c.size, _p8(K))
        return rc, K

    def decaps_internal(self, pset, dk, c):
        dk, c = _u8(dk), _u8(c)
        K = np.zeros(32, np.uint8)
        self.lib.ref_decaps_internal(pset, _p8(dk), _p8(c), _p8(K))
        return K

    def kem_encaps_public(self, pset, ek):
        _, _, cl = SIZES[pset]
        ek = _u8(ek)
        c = np.zeros(cl, np.uint8)
        K = np.zeros(32, np.uint8)
        rc = self.lib.ref_kem_encaps_public(pset, _p8(ek), ek.size, _p8(c), _p8(K))
        return rc, c, K

    def kem_keygen_public(self, pset):
        ekl, dkl, _ = SIZES[pset]
        ek = np.zeros(ekl, np.uint8)
        dk = np.zeros(dkl, np.uint8)
        rc = self.lib.ref_kem_keygen_public(pset, _p8(ek), _p8(dk))
        return rc, ek, dk

    def init_errno(self, pset):
        return self.lib.ref_init_errno(int(pset))

    def pke_keygen(self, pset, d):
        ekl = SIZES[pset][0]
        d = _u8(d, 32)
        ek = np.zeros(ekl, np.uint8)
        dk = np.zeros(ekl - 32, np.uint8)
        self.lib.ref_pke_keygen(pset, _p8(d), _p8(ek), _p8(dk))
        return ek, dk

    def pke_encrypt(self, pset, ek, m, r):
        ek, m, r = _u8(ek), _u8(m, 32), _u8(r, 32)
        c = np.zeros(SIZES[pset][2], np.uint8)
        self.lib.ref_pke_encrypt(pset, _p8(ek), _p8(m), _p8(r), _p8(c))
        return c

    def pke_decrypt(self, pset, dk_pke, c):
        dk_pke, c = _u8(dk_pke), _u8(c)
        m = np.zeros(32, np.uint8)
        self.lib.ref_pke_decrypt(pset, _p8(dk_pke), _p8(c), _p8(m))
        return m

    def ntt(self, f):
        f = np.ascontiguousarray(f, np.uint16).reshape(-1, 256)
        out = np.zeros_like(f)
        for i in range(f.shape[0]):
            self.lib.ref_ntt(_p16(f[i]), _p16(out[i]))
        return out

    def intt(self, f):
        f = np.ascontiguousarray(f, np.uint16).reshape(-1, 256)
        out = np.zeros_like(f)
        for i in range(f.shape[0]):
            self.lib.ref_intt(_p16(f[i]), _p16(out[i]))
        return out

    def _bin(self, fn, a, b):
        a, b = _u16(a), _u16(b)
        h = np.zeros(256, np.uint16)
        fn(_p16(a), _p16(b), _p16(h))
        return h

    def multiply_ntts(self, a, b):
        return self._bin(self.lib.ref_multiply_ntts, a, b)

    def poly_add(self, a, b):
        return self._bin(self.lib.ref_poly_add, a, b)

    def poly_sub(self, a, b):
        return self._bin(self.lib.ref_poly_sub, a, b)

    def sample_ntt(self, B):
        B = _u8(B, 34)
        a = np.zeros(256, np.uint16)
        self.lib.ref_sample_ntt(_p8(B), _p16(a))
        return a

    def sample_cbd(self, B, eta):
        B = _u8(B, 64 * eta)
        f = np.zeros(256, np.uint16)
        self.lib.ref_sample_cbd(_p8(B), eta, _p16(f))
        return f

    def compress(self, x, d):
        return self.lib.ref_compress(int(x), int(d))

    def decompress(self, y, d):
        return self.lib.ref_decompress(int(y), int(d))

    def bitrev7(self, r):
        return self.lib.ref_bitrev7(int(r))

    def byte_encode(self, F, d):
        F = _u16(F)
        B = np.zeros(32 * d, np.uint8)
        self.lib.ref_byte_encode(_p16(F), d, _p8(B))
        return B

    def byte_decode(self, B, d):
        B = _u8(B, 32 * d)
        F = np.zeros(256, np.uint16)
        self.lib.ref_byte_decode(_p8(B), d, _p16(F))
        return F

    def prf(self, s, b, eta):
        s = _u8(s, 32)
        out = np.zeros(64 * eta, np.uint8)
        self.lib.ref_prf(_p8(s), C.c_uint8(b), eta, _p8(out))
        return out

    def _hash(self, fn, x, n):
        x = _u8(x)
        out = np.zeros(n, np.uint8)
        fn(_p8(x), x.size, _p8(out))
        return out

    def H(self, x):
        return self._hash(self.lib.ref_H, x, 32)

    def J(self, x):
        return self._hash(self.lib.ref_J, x, 32)

    def G(self, x):
        return self._hash(self.lib.ref_G, x, 64)

    def sha3_bits(self, bits, d_bits, c_bits, xof):
        bits = _u8(bits)
        out = np.zeros(d_bits, np.uint8)
        self.lib.ref_sha3_bits(_p8(bits), bits.size, d_bits, c_bits, int(xof), _p8(out))
        return out

    def sha3_bits_sfx(self, bits, d_bits, c_bits, sfx4):
        """sha3_b with the caller's four suffix cells (sfx4[2] == 1: four suffix bits, else sfx4[0], sfx4[1])"""
        bits, sfx = _u8(bits), _u8(sfx4, 4)
        out = np.zeros(d_bits, np.uint8)
        self.lib.ref_sha3_bits_sfx(_p8(bits), bits.size, d_bits, c_bits, _p8(sfx), _p8(out))
        return out

    def time_encaps_decaps(self, pset, ek, dk, m):
        """Encaps_internal + KEM_Decaps over the given items on ONE core; returns
        (seconds, c, K, pairs_that_agree)."""
        ekl, dkl, cl = SIZES[pset]
        m = _u8(m).reshape(-1, 32)
        n = m.shape[0]
        ek = _u8(ek).reshape(n, ekl)
        dk = _u8(dk).reshape(n, dkl)
        c = np.zeros((n, cl), np.uint8)
        K = np.zeros((n, 32), np.uint8)
        agree = C.c_int(0)
        secs = self.lib.ref_time_encaps_decaps(pset, n, _p8(ek), _p8(dk), _p8(m), _p8(c), _p8(K), C.byref(agree))
        return secs, c, K, agree.value

    def time_triples(self, pset, d, z, m):
        """KeyGen_internal + Encaps_internal + KEM_Decaps over the given seed sets on ONE core; returns
        (seconds, ek, dk, c, K, items_that_agree)."""
        ekl, dkl, cl = SIZES[pset]
        d, z, m = (_u8(x).reshape(-1, 32) for x in (d, z, m))
        n = m.shape[0]
        ek, dk = np.zeros((n, ekl), np.uint8), np.zeros((n, dkl), np.uint8)
        c, K = np.zeros((n, cl), np.uint8), np.zeros((n, 32), np.uint8)
        agree = C.c_int(0)
        secs = self.lib.ref_time_triples(pset, n, _p8(d), _p8(z), _p8(m), _p8(ek), _p8(dk), _p8(c), _p8(K), C.byref(agree))
        return secs, ek, dk, c, K, agree.value
