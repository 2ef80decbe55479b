"""crystals-kyber_amd — Python host side of the MI355X batched ML-KEM engine.

Thin ctypes binding over the C-ABI of libmlkem_amd.so (include/mlkem_batch.h).  PyTorch is used only as
plumbing: device memory (uint8 / int16 CUDA tensors) and the current HIP stream.  The class mirrors the
reference's operator surface for the hot path (rsjahnige/CRYSTALS-Kyber, ml_kem.c):

    KeyGen_internal(d, z)   ml_kem.c:1034      Encaps_internal(ek, m)  ml_kem.c:1093
    KEM_Decaps(dk, c)       ml_kem.c:1310      Decaps_internal(dk, c)  ml_kem.c:1136
    NTT / InverseNTT / MultiplyNTTs / SampleNTT / SamplePolyCBD / PRF / H / G / J

There is NO CPU fallback: importing works anywhere, but constructing `MLKEM` without a HIP device (or
without the built extension) raises.  The directory name contains a hyphen, so import it through
`__graft_entry__.load_package()` (importlib) — the module is registered as `crystals_kyber_amd`.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MLKEM_LIB_PATH") or os.path.join(HERE, "libmlkem_amd.so")   # override: A/B builds in experiments
SHIM_PATH = os.path.join(HERE, "libml_kem.so")

SIZES = {512: (800, 1632, 768), 768: (1184, 2400, 1088), 1024: (1568, 3168, 1568)}  # ek, dk, c (ml_kem.h:52-59)
ERR_HASH = -5

# every symbol include/mlkem_batch.h declares (checked by tests/test_abi.py without a GPU)
ABI_SYMBOLS = (
    "mlkem_sizes", "mlkem_params", "mlkem_device_count", "mlkem_strerror", "mlkem_last_hip_error",
    "mlkem_ctx_create", "mlkem_ctx_destroy", "mlkem_ctx_scratch_bytes", "mlkem_timing_begin", "mlkem_timing_end",
    "mlkem_keygen_dev", "mlkem_encaps_dev", "mlkem_decaps_dev", "mlkem_encaps_status_dev", "mlkem_ctx_set_conformance", "mlkem_ctx_debug_stages",
    "mlkem_encaps_shared_dev", "mlkem_decaps_shared_dev",
    "mlkem_pke_keygen_dev", "mlkem_pke_encrypt_dev", "mlkem_pke_decrypt_dev",
    "mlkem_ntt_dev", "mlkem_intt_dev", "mlkem_multiply_ntts_dev", "mlkem_sample_ntt_dev", "mlkem_sample_cbd_dev",
    "mlkem_compress_encode_dev", "mlkem_decode_decompress_dev",
    "mlkem_prf_dev", "mlkem_hash_dev", "mlkem_keccak_sponge_dev", "mlkem_sha3_pad_bits",
    "mlkem_keygen", "mlkem_encaps", "mlkem_decaps", "mlkem_ntt", "mlkem_intt", "mlkem_keccak_sponge",
    "mlkem_sample_ntt", "mlkem_sample_ntt_retries", "mlkem_sample_ntt_retries_dev", "mlkem_sample_cbd",
    "mlkem_keygen_random", "mlkem_encaps_random",
    "mlkem_cells_to_bytes_dev", "mlkem_bytes_to_cells_dev", "mlkem_keygen_stream", "mlkem_encaps_stream", "mlkem_decaps_stream", "mlkem_stream_release",
    "mlkem_compress_dev", "mlkem_decompress_dev", "mlkem_compress", "mlkem_decompress", "mlkem_sha3_pad_suffix",
    "mlkem_selftest_count", "mlkem_selftest", "mlkem_host_release", "mlkem_host_register", "mlkem_host_unregister",
    "mlkem_multi_create", "mlkem_multi_destroy", "mlkem_multi_members", "mlkem_multi_device", "mlkem_shard_range",
    "mlkem_keygen_multi", "mlkem_encaps_multi", "mlkem_decaps_multi",
    "mlkem_keygen_multi_dev", "mlkem_encaps_multi_dev", "mlkem_decaps_multi_dev", "mlkem_multi_sync", "mlkem_multi_stream", "mlkem_stream_last_staged",
    "mlkem_vector_multiply_dev", "mlkem_poly_add_dev", "mlkem_poly_sub_dev",
)
SHIM_SYMBOLS = ("init", "KEM_KeyGen", "KEM_Encaps", "KEM_Decaps", "ml_errno", "sha3_b", "sha3_h", "sha3_s", "h2b", "b2h",
                "SampleNTT", "SamplePolyCBD", "NTT", "InverseNTT")


class MLKEMError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mlkem error {code}: {msg}")
        self.code = code


_lib = None


def load_library():
    """dlopen libmlkem_amd.so (fails loudly when the HIP extension has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # torch must load its HIP runtime BEFORE this library is mapped: the process then has exactly one
    # libamdhip64/ROCr instance, shared by torch (allocator, streams) and the engine (measured: the other
    # order leaves the second runtime without a visible device).
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    L.mlkem_strerror.restype = C.c_char_p
    L.mlkem_last_hip_error.restype = C.c_char_p
    L.mlkem_ctx_create.argtypes = [C.POINTER(vp), i32, sz]
    L.mlkem_ctx_destroy.argtypes = [vp]
    L.mlkem_ctx_destroy.restype = None
    L.mlkem_ctx_scratch_bytes.argtypes = [vp]
    L.mlkem_ctx_scratch_bytes.restype = sz
    L.mlkem_keygen_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp, vp]
    L.mlkem_encaps_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp, vp]
    L.mlkem_decaps_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp, vp]
    L.mlkem_encaps_status_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp, vp, vp]
    L.mlkem_ctx_set_conformance.argtypes = [vp, i32]
    L.mlkem_ctx_debug_stages.argtypes = [vp, C.c_uint]
    L.mlkem_encaps_shared_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp, vp]
    L.mlkem_decaps_shared_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp, vp]
    L.mlkem_pke_keygen_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp]
    L.mlkem_pke_encrypt_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp, vp]
    L.mlkem_pke_decrypt_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp]
    L.mlkem_ntt_dev.argtypes = [vp, sz, vp, vp, vp]
    L.mlkem_intt_dev.argtypes = [vp, sz, vp, vp, vp]
    L.mlkem_multiply_ntts_dev.argtypes = [vp, sz, vp, vp, vp, vp]
    L.mlkem_sample_ntt_dev.argtypes = [vp, sz, vp, vp, vp]
    L.mlkem_sample_cbd_dev.argtypes = [vp, i32, sz, vp, vp, vp]
    L.mlkem_prf_dev.argtypes = [vp, i32, sz, vp, vp, vp]
    L.mlkem_compress_encode_dev.argtypes = [vp, i32, sz, vp, vp, vp]
    L.mlkem_decode_decompress_dev.argtypes = [vp, i32, sz, vp, vp, vp]
    L.mlkem_hash_dev.argtypes = [vp, i32, sz, vp, C.c_uint, sz, vp, vp]
    L.mlkem_keygen.argtypes = [i32, sz, vp, vp, vp, vp]
    L.mlkem_encaps.argtypes = [i32, sz, vp, vp, vp, vp]
    L.mlkem_decaps.argtypes = [i32, sz, vp, vp, vp, vp]
    L.mlkem_ntt.argtypes = [sz, vp, vp]
    L.mlkem_intt.argtypes = [sz, vp, vp]
    L.mlkem_keygen_random.argtypes = [i32, sz, vp, vp]
    L.mlkem_encaps_random.argtypes = [i32, sz, vp, C.c_uint, vp, vp]
    L.mlkem_timing_end.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int), i32]
    L.mlkem_keccak_sponge_dev.argtypes = [vp, C.c_uint, sz, vp, C.c_uint, vp, C.c_uint, sz, vp]
    L.mlkem_keccak_sponge.argtypes = [C.c_uint, sz, vp, C.c_uint, vp, C.c_uint]
    L.mlkem_sha3_pad_bits.argtypes = [vp, sz, i32, C.c_uint, vp, sz]
    L.mlkem_cells_to_bytes_dev.argtypes = [vp, sz, vp, vp, vp]
    L.mlkem_bytes_to_cells_dev.argtypes = [vp, sz, vp, vp, vp]
    L.mlkem_keygen_stream.argtypes = [i32, sz, vp, vp, vp, vp, sz]
    L.mlkem_encaps_stream.argtypes = [i32, sz, vp, vp, vp, vp, sz]
    L.mlkem_decaps_stream.argtypes = [i32, sz, vp, vp, vp, vp, sz]
    L.mlkem_compress_dev.argtypes = [vp, i32, sz, vp, vp, vp]
    L.mlkem_decompress_dev.argtypes = [vp, i32, sz, vp, vp, vp]
    L.mlkem_compress.argtypes = [i32, sz, vp, vp]
    L.mlkem_decompress.argtypes = [i32, sz, vp, vp]
    L.mlkem_sha3_pad_suffix.argtypes = [vp, sz, vp, C.c_uint, C.c_uint, vp, sz]
    L.mlkem_selftest.argtypes = [vp, i32, C.POINTER(C.c_ulonglong)]
    L.mlkem_host_release.restype = None
    L.mlkem_stream_release.restype = None
    L.mlkem_host_register.argtypes = [vp, sz]
    L.mlkem_host_unregister.argtypes = [vp]
    L.mlkem_multi_create.argtypes = [C.POINTER(vp), i32, C.POINTER(C.c_int), sz]
    L.mlkem_multi_destroy.argtypes = [vp]
    L.mlkem_multi_destroy.restype = None
    L.mlkem_multi_members.argtypes = [vp]
    L.mlkem_multi_device.argtypes = [vp, i32]
    L.mlkem_shard_range.argtypes = [sz, i32, i32, C.POINTER(sz), C.POINTER(sz)]
    L.mlkem_keygen_multi.argtypes = [vp, i32, sz, vp, vp, vp, vp, sz]
    L.mlkem_encaps_multi.argtypes = [vp, i32, sz, vp, vp, vp, vp, sz]
    L.mlkem_decaps_multi.argtypes = [vp, i32, sz, vp, vp, vp, vp, sz]
    pp = C.POINTER(vp)
    L.mlkem_keygen_multi_dev.argtypes = [vp, i32, C.POINTER(sz), pp, pp, pp, pp]
    L.mlkem_encaps_multi_dev.argtypes = [vp, i32, C.POINTER(sz), pp, pp, pp, pp]
    L.mlkem_decaps_multi_dev.argtypes = [vp, i32, C.POINTER(sz), pp, pp, pp, pp]
    L.mlkem_multi_sync.argtypes = [vp]
    L.mlkem_multi_stream.argtypes = [vp, i32]
    L.mlkem_multi_stream.restype = vp
    L.mlkem_vector_multiply_dev.argtypes = [vp, i32, sz, vp, vp, vp, vp]
    L.mlkem_poly_add_dev.argtypes = [vp, sz, vp, vp, vp, vp]
    L.mlkem_poly_sub_dev.argtypes = [vp, sz, vp, vp, vp, vp]
    _lib = L
    return L


class kernel_timing:
    """Context manager: per-kernel HIP-event timing of everything launched inside (bench.py roofline leg).
    After exit, `.rows` = {label: (total_ms, launches)}."""

    def __enter__(self):
        self.lib = load_library()
        rc = self.lib.mlkem_timing_begin()
        if rc:
            raise MLKEMError(rc, "timing already active")
        self.rows = {}
        return self

    def __exit__(self, *exc):
        labels = C.create_string_buffer(32 * 64)
        ms = (C.c_double * 64)()
        cnt = (C.c_int * 64)()
        n = self.lib.mlkem_timing_end(labels, ms, cnt, 64)
        for i in range(max(n, 0)):
            name = labels.raw[32 * i:32 * i + 32].split(b"\0")[0].decode()
            self.rows[name] = (ms[i], cnt[i])
        return False


def sizes(param_set):
    if param_set not in SIZES:
        raise MLKEMError(-1, "invalid parameter set (reference ml_errno -1, ml_kem.c:1389)")
    return SIZES[param_set]


class MLKEM:
    """One engine context = one device + its scratch HBM.  Work is enqueued on torch's current stream."""

    def __init__(self, param_set=768, device=0, chunk_items=0, conformance="reference"):
        """conformance: "reference" (default; bit-identical to ml_kem.c incl. PRF/J on SHAKE128) or "fips203"
        (PRF/J on SHAKE256, encapsulation-key modulus check reported by `encaps(..., return_status=True)`)."""
        import torch
        self.torch = torch
        self.lib = load_library()
        self.ek_len, self.dk_len, self.c_len = sizes(param_set)
        self.k = (self.ek_len - 32) // 384
        self.param_set = param_set
        if not torch.cuda.is_available():
            raise MLKEMError(-100, "no HIP device visible (the engine has no CPU fallback)")
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        self._check(self.lib.mlkem_ctx_create(C.byref(h), device, chunk_items))
        self._ctx = h
        if conformance not in ("reference", "fips203"):
            raise MLKEMError(-101, "conformance must be 'reference' or 'fips203'")
        self.conformance = conformance
        self._check(self.lib.mlkem_ctx_set_conformance(self._ctx, 1 if conformance == "fips203" else 0))

    def close(self):
        if getattr(self, "_ctx", None):
            self.lib.mlkem_ctx_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    # -- helpers -------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            msg = self.lib.mlkem_strerror(rc).decode()
            hip = self.lib.mlkem_last_hip_error().decode()
            raise MLKEMError(rc, msg + (f" [{hip}]" if hip and rc == -100 else ""))

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t, dtype, last):
        torch = self.torch
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(t)
        t = t.to(device=self.device, dtype=dtype).contiguous()
        if t.dim() == 1:
            t = t.reshape(-1, last)
        if t.dim() != 2 or t.shape[1] != last:
            raise MLKEMError(-3, f"type check failed: expected rows of {last}, got {tuple(t.shape)} (reference ml_errno -3)")
        return t

    def _out(self, n, last, dtype=None, given=None):
        """A fresh [n, last] output (or [n] when last is None), or the caller's tensor after checking that the raw pointer
        handed to the C-ABI really is n x last elements of `dtype`, contiguous, on this engine's device."""
        torch = self.torch
        dtype = dtype or torch.uint8
        shape = (n,) if last is None else (n, last)
        if given is None:
            return torch.empty(shape, dtype=dtype, device=self.device)
        if not isinstance(given, torch.Tensor):
            raise MLKEMError(-101, "output must be a torch tensor")
        if given.device != self.device or given.dtype != dtype or tuple(given.shape) != shape or not given.is_contiguous():
            raise MLKEMError(-101, f"output tensor must be contiguous {dtype} {shape} on {self.device}, got "
                                   f"{given.dtype} {tuple(given.shape)} on {given.device}"
                                   f"{'' if given.is_contiguous() else ' (non-contiguous)'}")
        return given

    @property
    def scratch_bytes(self):
        return self.lib.mlkem_ctx_scratch_bytes(self._ctx)

    # -- batched KEM (device resident) -------------------------------------------------------------
    def keygen(self, d, z, ek=None, dk=None):
        """KeyGen_internal (ml_kem.c:1034): d, z [n,32] -> ek [n,ek_len], dk [n,dk_len]."""
        u8 = self.torch.uint8
        d, z = self._dev(d, u8, 32), self._dev(z, u8, 32)
        n = d.shape[0]
        if z.shape[0] != n:
            raise MLKEMError(-101, "d and z batch sizes differ")
        ek = self._out(n, self.ek_len, given=ek)
        dk = self._out(n, self.dk_len, given=dk)
        self._check(self.lib.mlkem_keygen_dev(self._ctx, self.param_set, n, d.data_ptr(), z.data_ptr(), ek.data_ptr(),
                                              dk.data_ptr(), self._stream()))
        return ek, dk

    def encaps(self, ek, m, c=None, K=None, return_status=False):
        """Encaps_internal (ml_kem.c:1093): ek [n,ek_len], m [n,32] -> c [n,c_len], K [n,32]
        (+ status [n]: 0 or -4 = modulus check of KEM_Encaps / FIPS 203 7.2, when return_status)."""
        u8 = self.torch.uint8
        ek, m = self._dev(ek, u8, self.ek_len), self._dev(m, u8, 32)
        n = m.shape[0]
        if ek.shape[0] != n:
            raise MLKEMError(-101, "ek and m batch sizes differ")
        c = self._out(n, self.c_len, given=c)
        K = self._out(n, 32, given=K)
        if return_status:
            st = self.torch.empty(n, dtype=self.torch.int32, device=self.device)
            self._check(self.lib.mlkem_encaps_status_dev(self._ctx, self.param_set, n, ek.data_ptr(), m.data_ptr(), c.data_ptr(),
                                                         K.data_ptr(), st.data_ptr(), self._stream()))
            return c, K, st
        self._check(self.lib.mlkem_encaps_dev(self._ctx, self.param_set, n, ek.data_ptr(), m.data_ptr(), c.data_ptr(),
                                              K.data_ptr(), self._stream()))
        return c, K

    def decaps(self, dk, c, K=None, status=None, hash_check=True):
        """KEM_Decaps (ml_kem.c:1310) incl. the dk hash check: -> K [n,32], status [n] (0 or -5).
        hash_check=False gives Decaps_internal (ml_kem.c:1136) and status None."""
        torch = self.torch
        dk, c = self._dev(dk, torch.uint8, self.dk_len), self._dev(c, torch.uint8, self.c_len)
        n = c.shape[0]
        if dk.shape[0] != n:
            raise MLKEMError(-101, "dk and c batch sizes differ")
        K = self._out(n, 32, given=K)
        if hash_check:
            status = self._out(n, None, torch.int32, given=status)
        sp = status.data_ptr() if hash_check else None
        self._check(self.lib.mlkem_decaps_dev(self._ctx, self.param_set, n, dk.data_ptr(), c.data_ptr(), K.data_ptr(), sp,
                                              self._stream()))
        return K, (status if hash_check else None)

    # reference-style aliases
    KeyGen_internal = keygen
    Encaps_internal = encaps
    KEM_Decaps = decaps

    def Decaps_internal(self, dk, c):
        return self.decaps(dk, c, hash_check=False)[0]

    # -- shared-key batches: one ek / dk for all items ----------------------------------------------
    def encaps_shared(self, ek, m):
        """n encapsulations to ONE key: ek [ek_len] or [1,ek_len], m [n,32] -> c [n,c_len], K [n,32]; same bytes as
        encaps() on the replicated key, H(ek) and the matrix computed once."""
        u8 = self.torch.uint8
        ek = self._dev(self.torch.as_tensor(ek).reshape(1, -1), u8, self.ek_len)
        m = self._dev(m, u8, 32)
        n = m.shape[0]
        c, K = self._out(n, self.c_len), self._out(n, 32)
        self._check(self.lib.mlkem_encaps_shared_dev(self._ctx, self.param_set, n, ek.data_ptr(), m.data_ptr(), c.data_ptr(),
                                                     K.data_ptr(), self._stream()))
        return c, K

    def decaps_shared(self, dk, c, hash_check=True):
        """n decapsulations under ONE key: dk [dk_len] or [1,dk_len], c [n,c_len] -> K [n,32], status [n] (0 / -5)."""
        torch = self.torch
        dk = self._dev(torch.as_tensor(dk).reshape(1, -1), torch.uint8, self.dk_len)
        c = self._dev(c, torch.uint8, self.c_len)
        n = c.shape[0]
        K = self._out(n, 32)
        status = torch.empty(n, dtype=torch.int32, device=self.device) if hash_check else None
        self._check(self.lib.mlkem_decaps_shared_dev(self._ctx, self.param_set, n, dk.data_ptr(), c.data_ptr(), K.data_ptr(),
                                                     status.data_ptr() if hash_check else None, self._stream()))
        return K, status

    # -- K-PKE on its own ---------------------------------------------------
    def PKE_KeyGen(self, d):
        """K-PKE.KeyGen (ml_kem.c:651): d [n,32] -> ek [n,ek_len], dk_pke [n,384k]."""
        d = self._dev(d, self.torch.uint8, 32)
        n = d.shape[0]
        ek, dk = self._out(n, self.ek_len), self._out(n, 384 * self.k)
        self._check(self.lib.mlkem_pke_keygen_dev(self._ctx, self.param_set, n, d.data_ptr(), ek.data_ptr(), dk.data_ptr(), self._stream()))
        return ek, dk

    def PKE_Encrypt(self, ek, m, r):
        """K-PKE.Encrypt (ml_kem.c:776): ek [n,ek_len], m, r [n,32] -> c [n,c_len]."""
        u8 = self.torch.uint8
        ek, m, r = self._dev(ek, u8, self.ek_len), self._dev(m, u8, 32), self._dev(r, u8, 32)
        n = m.shape[0]
        if ek.shape[0] != n or r.shape[0] != n:
            raise MLKEMError(-101, "ek, m and r batch sizes differ")
        c = self._out(n, self.c_len)
        self._check(self.lib.mlkem_pke_encrypt_dev(self._ctx, self.param_set, n, ek.data_ptr(), m.data_ptr(), r.data_ptr(),
                                                   c.data_ptr(), self._stream()))
        return c

    def PKE_Decrypt(self, dk_pke, c):
        """K-PKE.Decrypt (ml_kem.c:942): dk_pke [n,384k], c [n,c_len] -> m [n,32]."""
        u8 = self.torch.uint8
        dk, c = self._dev(dk_pke, u8, 384 * self.k), self._dev(c, u8, self.c_len)
        n = c.shape[0]
        if dk.shape[0] != n:
            raise MLKEMError(-101, "dk_pke and c batch sizes differ")
        m = self._out(n, 32)
        self._check(self.lib.mlkem_pke_decrypt_dev(self._ctx, self.param_set, n, dk.data_ptr(), c.data_ptr(), m.data_ptr(), self._stream()))
        return m

    # -- batched primitives --------------------------------------------------------------------------
    def _poly(self, f):
        torch = self.torch
        if not isinstance(f, torch.Tensor):
            f = torch.as_tensor(f)
        if f.dtype == torch.uint16:
            f = f.view(torch.int16)
        return self._dev(f, torch.int16, 256)

    def ntt(self, f):
        """NTT (ml_kem.c:287): [n,256] coefficients in [0,q) (int16 storage of uint16 values)."""
        f = self._poly(f)
        out = self.torch.empty_like(f)
        self._check(self.lib.mlkem_ntt_dev(self._ctx, f.shape[0], f.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def intt(self, fh):
        """InverseNTT (ml_kem.c:336)."""
        fh = self._poly(fh)
        out = self.torch.empty_like(fh)
        self._check(self.lib.mlkem_intt_dev(self._ctx, fh.shape[0], fh.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def multiply_ntts(self, a, b):
        """MultiplyNTTs (ml_kem.c:415)."""
        a, b = self._poly(a), self._poly(b)
        out = self.torch.empty_like(a)
        self._check(self.lib.mlkem_multiply_ntts_dev(self._ctx, a.shape[0], a.data_ptr(), b.data_ptr(), out.data_ptr(),
                                                     self._stream()))
        return out

    def vector_multiply(self, u, v):
        """VectorMultiply (ml_kem.c:618): u, v [n,k,256] -> [n,256] = sum_i MultiplyNTTs(u[:,i], v[:,i])."""
        torch = self.torch
        u, v = (t.view(torch.int16) if t.dtype == torch.uint16 else t for t in (torch.as_tensor(u), torch.as_tensor(v)))
        u = u.to(device=self.device, dtype=torch.int16).contiguous()
        v = v.to(device=self.device, dtype=torch.int16).contiguous()
        if u.dim() != 3 or u.shape != v.shape or u.shape[2] != 256 or not 1 <= u.shape[1] <= 4:
            raise MLKEMError(-101, f"vector_multiply expects two [n, k, 256] operands with k in 1..4, got {tuple(u.shape)} and {tuple(v.shape)}")
        out = self._out(u.shape[0], 256, torch.int16)
        self._check(self.lib.mlkem_vector_multiply_dev(self._ctx, u.shape[1], u.shape[0], u.data_ptr(), v.data_ptr(), out.data_ptr(),
                                                       self._stream()))
        return out

    def _addsub(self, fn, a, b):
        a, b = self._poly(a), self._poly(b)
        if a.shape != b.shape:
            raise MLKEMError(-101, "operands differ in shape")
        out = self.torch.empty_like(a)
        self._check(fn(self._ctx, a.numel(), a.data_ptr(), b.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def poly_add(self, a, b):
        """PolyAddition (ml_kem.c:580): [n,256] + [n,256] coefficient-wise mod q."""
        return self._addsub(self.lib.mlkem_poly_add_dev, a, b)

    def poly_sub(self, a, b):
        """PolySubtraction (ml_kem.c:599)."""
        return self._addsub(self.lib.mlkem_poly_sub_dev, a, b)

    def sample_ntt(self, seeds34):
        """SampleNTT (ml_kem.c:189): [n,34] -> [n,256]."""
        s = self._dev(seeds34, self.torch.uint8, 34)
        out = self._out(s.shape[0], 256, self.torch.int16)
        self._check(self.lib.mlkem_sample_ntt_dev(self._ctx, s.shape[0], s.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def sample_cbd(self, data, eta):
        """SamplePolyCBD (ml_kem.c:253): [n,64*eta] -> [n,256]."""
        b = self._dev(data, self.torch.uint8, 64 * eta)
        out = self._out(b.shape[0], 256, self.torch.int16)
        self._check(self.lib.mlkem_sample_cbd_dev(self._ctx, eta, b.shape[0], b.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def compress_encode(self, f, d):
        """ByteEncode_d(Compress_d(f)) (ml_kem.c:83, :125) for d in {1,4,5,10,11}; d = 12: ByteEncode_12 of canonical f."""
        f = self._poly(f)
        out = self._out(f.shape[0], 32 * d)
        self._check(self.lib.mlkem_compress_encode_dev(self._ctx, d, f.shape[0], f.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def decode_decompress(self, data, d):
        """Decompress_d(ByteDecode_d(B)) (ml_kem.c:153, :104); d = 12: raw 12-bit values (no reduction, F3)."""
        b = self._dev(data, self.torch.uint8, 32 * d)
        out = self._out(b.shape[0], 256, self.torch.int16)
        self._check(self.lib.mlkem_decode_decompress_dev(self._ctx, d, b.shape[0], b.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def compress(self, x, d):
        """Compress(x, d) (ml_kem.c:83) value by value, any d in 1..12, any 12-bit x: tensor of int16/uint16 -> same shape."""
        return self._compress(x, d, False)

    def decompress(self, y, d):
        """Decompress(y, d) (ml_kem.c:104) value by value, any d in 1..12."""
        return self._compress(y, d, True)

    def _compress(self, v, d, inverse):
        torch = self.torch
        if not isinstance(v, torch.Tensor):
            v = torch.as_tensor(v)
        if v.dtype == torch.uint16:
            v = v.view(torch.int16)
        v = v.to(device=self.device, dtype=torch.int16).contiguous()
        out = torch.empty_like(v)
        fn = self.lib.mlkem_decompress_dev if inverse else self.lib.mlkem_compress_dev
        self._check(fn(self._ctx, int(d), v.numel(), v.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def selftest(self):
        """On-device exhaustive sweeps of the exact fp32-pipe arithmetic (include/mlkem_batch.h: mlkem_selftest);
        returns the list of violation counts, one per sweep (all zero = every helper exact over its whole domain)."""
        self.torch.cuda.synchronize(self.device)
        out = []
        for w in range(self.lib.mlkem_selftest_count()):
            v = C.c_ulonglong(12345)
            self._check(self.lib.mlkem_selftest(self._ctx, w, C.byref(v)))
            out.append(v.value)
        return out

    def prf(self, in33, eta):
        """PRF (ml_kem.c:496; SHAKE128 in the reference): [n,33] (s || b) -> [n,64*eta]."""
        s = self._dev(in33, self.torch.uint8, 33)
        out = self._out(s.shape[0], 64 * eta)
        self._check(self.lib.mlkem_prf_dev(self._ctx, eta, s.shape[0], s.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def _hash(self, kind, msgs, outlen):
        torch = self.torch
        if not isinstance(msgs, torch.Tensor):
            msgs = torch.as_tensor(msgs)
        msgs = msgs.to(device=self.device, dtype=torch.uint8)
        n, ln = msgs.shape
        stride = (ln + 7) // 8 * 8 + 8
        padded = torch.zeros((n, stride), dtype=torch.uint8, device=self.device)
        padded[:, :ln] = msgs
        out = self._out(n, outlen)
        self._check(self.lib.mlkem_hash_dev(self._ctx, kind, n, padded.data_ptr(), ln, stride, out.data_ptr(), self._stream()))
        return out

    def H(self, msgs):
        """H = SHA3-256 (ml_kem.c:521) over equal-length messages [n,len] -> [n,32]."""
        return self._hash(0, msgs, 32)

    def G(self, msgs):
        """G = SHA3-512 (ml_kem.c:559) -> [n,64]."""
        return self._hash(1, msgs, 64)

    def J(self, msgs):
        """J (ml_kem.c:540; SHAKE128 in the reference) -> [n,32]."""
        return self._hash(2, msgs, 32)

    def cells_to_bytes(self, cells):
        """`union byte` cells (int32 storage, value in bits 0-7) -> packed uint8, on the device."""
        torch = self.torch
        cells = cells.to(device=self.device).contiguous().view(torch.int32)
        out = torch.empty(cells.numel(), dtype=torch.uint8, device=self.device)
        self._check(self.lib.mlkem_cells_to_bytes_dev(self._ctx, cells.numel(), cells.data_ptr(), out.data_ptr(), self._stream()))
        return out.reshape(cells.shape)

    def bytes_to_cells(self, data):
        torch = self.torch
        data = data.to(device=self.device, dtype=torch.uint8).contiguous()
        out = torch.empty(data.numel(), dtype=torch.int32, device=self.device)
        self._check(self.lib.mlkem_bytes_to_cells_dev(self._ctx, data.numel(), data.data_ptr(), out.data_ptr(), self._stream()))
        return out.reshape(data.shape)

    def sha3_bits(self, bits_list, xof, rate, outlen, suffix=None):
        """SHA-3 / SHAKE of bit-granular messages (sha3_b, sha3.c:408): `bits_list` = equal-length sequences of 0/1;
        suffix + pad10*1 on the host (mlkem_sha3_pad_suffix), sponge on the device.  -> [n, outlen] bytes.
        `suffix`: explicit suffix bits (e.g. (1, 1) = RawSHAKE) instead of the hash "01" / XOF "1111"."""
        import numpy as np
        torch = self.torch
        n = len(bits_list)
        nbits = len(bits_list[0]) if n else 0
        sfx = np.ascontiguousarray(suffix if suffix is not None else ((1, 1, 1, 1) if xof else (0, 1)), np.uint8)
        nblocks = (nbits + sfx.size + 2 + 8 * rate - 1) // (8 * rate)
        padded = np.zeros((n, nblocks * rate), np.uint8)
        for i, b in enumerate(bits_list):
            b = np.ascontiguousarray(b, np.uint8)
            assert b.size == nbits
            rc = self.lib.mlkem_sha3_pad_suffix(b.ctypes.data, nbits, sfx.ctypes.data, sfx.size, rate, padded[i].ctypes.data,
                                                padded[i].size)
            if rc != nblocks:
                raise MLKEMError(rc, "sha3 padding failed")
        dp = torch.from_numpy(padded).to(self.device)
        stride = (outlen + 3) // 4 * 4
        out = torch.zeros((n, stride), dtype=torch.uint8, device=self.device)
        self._check(self.lib.mlkem_keccak_sponge_dev(self._ctx, rate, n, dp.data_ptr(), nblocks, out.data_ptr(), outlen, stride,
                                                     self._stream()))
        return out[:, :outlen]

    # NTT-only workload of BASELINE config 2
    NTT = ntt
    InverseNTT = intt
    MultiplyNTTs = multiply_ntts
    VectorMultiply = vector_multiply
    PolyAddition = poly_add
    PolySubtraction = poly_sub
    SampleNTT = sample_ntt
    SamplePolyCBD = sample_cbd
    PRF = prf


class MLKEMMulti:
    """In-process sharding over several devices (include/mlkem_batch.h: mlkem_multi_*): member r of R works on
    the contiguous item range shard_range(n, r, R) on its own device; no exchange between members, no collective.
    `devices` may repeat a device (rehearsal of the sharded path on one GPU)."""

    def __init__(self, param_set=768, devices=None, chunk_items=0):
        import torch
        self.torch = torch
        self.lib = load_library()
        self.ek_len, self.dk_len, self.c_len = sizes(param_set)
        self.param_set = param_set
        if not torch.cuda.is_available():
            raise MLKEMError(-100, "no HIP device visible (the engine has no CPU fallback)")
        devices = list(range(torch.cuda.device_count())) if devices is None else list(devices)
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self.lib.mlkem_multi_create(C.byref(h), len(devices), arr, chunk_items)
        if rc:
            raise MLKEMError(rc, self.lib.mlkem_strerror(rc).decode())
        self._mm = h
        self.devices = devices
        self.members = len(devices)

    def close(self):
        if getattr(self, "_mm", None):
            self._ext = None
            self.lib.mlkem_multi_destroy(self._mm)
            self._mm = None

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise MLKEMError(rc, self.lib.mlkem_strerror(rc).decode() + " [" + self.lib.mlkem_last_hip_error().decode() + "]")

    def ranges(self, n):
        return [shard_range(n, r, self.members) for r in range(self.members)]

    # -- host-resident batches (numpy uint8 arrays), one host thread + streaming engine per member --------------------
    def _host(self, a, last):
        import numpy as np
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, last)
        return a

    def keygen(self, d, z, chunk_items=0):
        import numpy as np
        d, z = self._host(d, 32), self._host(z, 32)
        n = d.shape[0]
        ek, dk = np.empty((n, self.ek_len), np.uint8), np.empty((n, self.dk_len), np.uint8)
        self._check(self.lib.mlkem_keygen_multi(self._mm, self.param_set, n, d.ctypes.data, z.ctypes.data, ek.ctypes.data,
                                                dk.ctypes.data, chunk_items))
        return ek, dk

    def encaps(self, ek, m, chunk_items=0):
        import numpy as np
        ek, m = self._host(ek, self.ek_len), self._host(m, 32)
        n = m.shape[0]
        c, K = np.empty((n, self.c_len), np.uint8), np.empty((n, 32), np.uint8)
        self._check(self.lib.mlkem_encaps_multi(self._mm, self.param_set, n, ek.ctypes.data, m.ctypes.data, c.ctypes.data,
                                                K.ctypes.data, chunk_items))
        return c, K

    def decaps(self, dk, c, chunk_items=0):
        import numpy as np
        dk, c = self._host(dk, self.dk_len), self._host(c, self.c_len)
        n = c.shape[0]
        K, st = np.empty((n, 32), np.uint8), np.empty(n, np.int32)
        self._check(self.lib.mlkem_decaps_multi(self._mm, self.param_set, n, dk.ctypes.data, c.ctypes.data, K.ctypes.data,
                                                st.ctypes.data, chunk_items))
        return K, st

    # -- device-resident shards: lists (one tensor per member, on that member's device) ---------------------------------
    def _shards(self, ts, last, dtype=None):
        torch = self.torch
        dtype = dtype or torch.uint8
        if len(ts) != self.members:
            raise MLKEMError(-101, f"need one shard per member ({self.members}), got {len(ts)}")
        for r, t in enumerate(ts):
            want_dev = torch.device("cuda", self.devices[r])
            ok = isinstance(t, torch.Tensor) and t.device == want_dev and t.dtype == dtype and t.is_contiguous() and \
                (t.dim() == 2 and t.shape[1] == last if last is not None else t.dim() == 1)
            if not ok:
                raise MLKEMError(-101, f"shard {r}: expected a contiguous {dtype} tensor [n, {last}] on {want_dev}")
        return (C.c_void_p * self.members)(*[t.data_ptr() for t in ts])

    def _sizes(self, ts, *others):
        ns = [int(t.shape[0]) for t in ts]
        for o in others:
            if [int(t.shape[0]) for t in o] != ns:
                raise MLKEMError(-101, "shard sizes differ between operands")
        return (C.c_size_t * self.members)(*ns), ns

    def _empty(self, ns, last, dtype=None):
        torch = self.torch
        return [torch.empty((n, last) if last is not None else (n,), dtype=dtype or torch.uint8, device=torch.device("cuda", dv))
                for n, dv in zip(ns, self.devices)]

    # Stream discipline of the device-resident calls.  The members enqueue on their own non-blocking streams (C-ABI:
    # mlkem_multi_stream), torch allocates and produces on ITS current stream.  Around every call the wrapper therefore
    #   1. makes each member stream wait for torch's current stream on that device (the producers of the input shards and
    #      whatever last used the memory the caching allocator hands out for the outputs),
    #   2. tells the allocator that inputs and outputs are in use on the member stream (record_stream), so that dropping a
    #      tensor before sync() cannot hand its memory to other work while the member still reads or writes it.
    # Results are ordered for torch by sync() (or by streams()[r] events).
    def streams(self):
        """torch views (ExternalStream) of the members' HIP streams, one per member."""
        torch = self.torch
        if getattr(self, "_ext", None) is None:
            ext = []
            for r, dv in enumerate(self.devices):
                h = self.lib.mlkem_multi_stream(self._mm, r)
                if not h:
                    raise MLKEMError(-100, "member stream unavailable [" + self.lib.mlkem_last_hip_error().decode() + "]")
                ext.append(torch.cuda.ExternalStream(h, device=torch.device("cuda", dv)))
            self._ext = ext
        return self._ext

    def _order(self, *tensor_lists):
        torch = self.torch
        for r, (dv, ext) in enumerate(zip(self.devices, self.streams())):
            ext.wait_stream(torch.cuda.current_stream(torch.device("cuda", dv)))
            for ts in tensor_lists:
                ts[r].record_stream(ext)

    def keygen_dev(self, d, z):
        """shards d[r], z[r] [n_r, 32] on device r -> ek[r], dk[r].  Enqueued on the member streams behind torch's current
        streams; call sync() (or wait on streams()[r]) before reading the outputs from another stream."""
        ns_c, ns = self._sizes(d, z)
        ek, dk = self._empty(ns, self.ek_len), self._empty(ns, self.dk_len)
        pd, pz, pek, pdk = self._shards(d, 32), self._shards(z, 32), self._shards(ek, self.ek_len), self._shards(dk, self.dk_len)
        self._order(d, z, ek, dk)
        self._check(self.lib.mlkem_keygen_multi_dev(self._mm, self.param_set, ns_c, pd, pz, pek, pdk))
        return ek, dk

    def encaps_dev(self, ek, m, c=None, K=None):
        ns_c, ns = self._sizes(m, ek)
        c = self._empty(ns, self.c_len) if c is None else c
        K = self._empty(ns, 32) if K is None else K
        pek, pm, pc, pK = self._shards(ek, self.ek_len), self._shards(m, 32), self._shards(c, self.c_len), self._shards(K, 32)
        self._sizes(m, c, K)
        self._order(ek, m, c, K)
        self._check(self.lib.mlkem_encaps_multi_dev(self._mm, self.param_set, ns_c, pek, pm, pc, pK))
        return c, K

    def decaps_dev(self, dk, c, K=None, status=None):
        ns_c, ns = self._sizes(c, dk)
        K = self._empty(ns, 32) if K is None else K
        st = self._empty(ns, None, self.torch.int32) if status is None else status
        pdk, pc, pK = self._shards(dk, self.dk_len), self._shards(c, self.c_len), self._shards(K, 32)
        pst = self._shards(st, None, self.torch.int32)
        self._sizes(c, K, st)
        self._order(dk, c, K, st)
        self._check(self.lib.mlkem_decaps_multi_dev(self._mm, self.param_set, ns_c, pdk, pc, pK, pst))
        return K, st

    def sync(self):
        self._check(self.lib.mlkem_multi_sync(self._mm))


def shard_range(n_total, rank, world):
    """Contiguous shard of a batch for rank `rank` of `world` (item i -> GPU floor(i / (B/G)));
    the remainder is spread over the first ranks.  Returns (start, stop)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
