/*
 * mlkem_batch.h — C-ABI of libmlkem_amd.so, the MI355X (gfx950) batched ML-KEM engine.
 *
 * This is the drop-in boundary for the hot path of rsjahnige/CRYSTALS-Kyber: every entry point is
 * `extern "C"`, takes plain pointers + sizes, and names the reference interface it replaces.  All byte
 * strings are PACKED uint8 (one byte per byte — not the reference's 4-byte `union byte`, SURVEY.md F1;
 * the ml_kem.h-compatible shim in mlkem_compat.h widens/narrows at its edge), polynomials are uint16[256]
 * with coefficients in [0, q), item i of a batch lives at base + i * item_size.
 *
 * Results are bit-identical to the reference's ml_kem.c on the same inputs, including its deviations
 * from FIPS 203: PRF and J are SHAKE128 (ml_kem.c:508, :546) and ByteDecode_12 performs no reduction, so
 * the Encaps modulus check never fires (ml_kem.c:170, :1273-1291).
 *
 * Two families:
 *   *_dev : pointers are DEVICE pointers (HBM resident), the work is enqueued on `stream` (a hipStream_t
 *           passed as void*; NULL = default stream) and the call returns without synchronising.
 *           Base pointers must be 16-byte aligned.
 *   plain : pointers are HOST pointers; the call stages through device memory, runs the *_dev path and
 *           synchronises before returning.
 * There is no CPU fallback: without a usable HIP device every call returns MLKEM_ERR_NO_DEVICE.
 *
 * Return value: 0 on success, negative MLKEM_ERR_* otherwise.  Per-item conditions (the reference's
 * ml_errno = -5 hash-check failure) are reported through the `status` array, never through a global.
 */
#ifndef MLKEM_BATCH_H
#define MLKEM_BATCH_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: only the entry points declared here (MLKEM_API) are exported. */
#ifndef MLKEM_API
#define MLKEM_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define MLKEM_OK 0
#define MLKEM_ERR_PARAM_SET (-1)    /* unknown parameter set   (reference: ml_errno = -1, ml_kem.c:1389-1391) */
#define MLKEM_ERR_RNG (-2)          /* entropy source failed   (reference: ml_errno = -2, ml_kem.c:1243, :1297) */
#define MLKEM_ERR_LENGTH (-3)       /* ek/dk/c length mismatch (reference: ml_errno = -3, ml_kem.c:1269, :1323, :1331) */
#define MLKEM_ERR_MODULUS (-4)      /* never produced: the reference's check is a no-op (ml_kem.c:1273-1291) */
#define MLKEM_ERR_HASH (-5)         /* per-item status: H(dk.ek) != dk.h (reference: ml_errno = -5, ml_kem.c:1347) */
#define MLKEM_ERR_NO_DEVICE (-100)  /* no HIP device / HIP runtime error */
#define MLKEM_ERR_ARG (-101)        /* NULL or misaligned pointer, bad argument */
#define MLKEM_ERR_ALLOC (-102)      /* device or host allocation failed */

/* Lengths of ek / dk / c for a parameter set in {512, 768, 1024} (ml_kem.h:52-59, ml_kem.c:1363-1395). */
MLKEM_API int mlkem_sizes(int param_set, unsigned* ek_len, unsigned* dk_len, unsigned* c_len);
/* (k, eta1, eta2, du, dv) as the reference's init() fills struct PARAMS (ml_kem.c:1363-1395). */
MLKEM_API int mlkem_params(int param_set, int out_k_eta1_eta2_du_dv[5]);
/* Number of visible HIP devices (0 when none / runtime unusable). */
MLKEM_API int mlkem_device_count(void);
/* Text for an MLKEM_ERR_* code. */
MLKEM_API const char* mlkem_strerror(int code);
/* Last HIP runtime error string seen by this thread's calls ("" if none). */
MLKEM_API const char* mlkem_last_hip_error(void);

/* ---- engine context: owns the scratch HBM of one device --------------------------------------------- */
typedef struct mlkem_ctx mlkem_ctx;
/* `chunk_items` = items per sampler/arithmetic chunk (0 = default 2^18, env MLKEM_CHUNK_ITEMS); scratch is
 * ~10 KiB x chunk_items, allocated once here so that no *_dev call allocates.  A context is bound to `device`: *_dev
 * calls launch on the caller's current HIP device and return MLKEM_ERR_ARG when that is not the context's device.
 * The context gets one side stream with its first encaps / decaps call of at most `chunk_items` items (env
 * MLKEM_SIDE_STREAM=0: never): such a call samples its matrix there while the hash kernels run on the caller's stream; fork and
 * join are events inside the call, so the caller's stream is ordered after all of the call's work exactly as without it
 * (stream capture sees a fork/join).  Contexts that only ever see larger calls create no stream.
 * A context's scratch serves ONE call at a time: queue the *_dev calls of a context on one stream (or order the streams
 * yourself); work that runs side by side on several streams needs a context per stream. */
MLKEM_API int mlkem_ctx_create(mlkem_ctx** out, int device, size_t chunk_items);
MLKEM_API void mlkem_ctx_destroy(mlkem_ctx* ctx);
MLKEM_API size_t mlkem_ctx_scratch_bytes(const mlkem_ctx* ctx);

/* Conformance of a context.  REFERENCE (default): bit-identical to ml_kem.c including its deviations
 * from FIPS 203 (PRF and J on SHAKE128: ml_kem.c:508, :546; modulus check that never fires: :1273-1291).
 * FIPS203: PRF = SHAKE256(s || b), J = SHAKE256(z || c) (FIPS 203 section 4.1) and mlkem_encaps_status_dev reports
 * MLKEM_ERR_MODULUS per item when ByteEncode_12(ByteDecode_12(ek)) != ek (FIPS 203 section 7.2). */
#define MLKEM_CONFORMANCE_REFERENCE 0
#define MLKEM_CONFORMANCE_FIPS203 1
MLKEM_API int mlkem_ctx_set_conformance(mlkem_ctx* ctx, int mode);

/* ---- per-kernel timing (measurement aid used by bench.py) ---------------------------------------------
 * Between begin and end, every kernel launched by this thread is bracketed by HIP events on its launch
 * stream.  mlkem_timing_end synchronises and returns per-kernel-label rows: labels[32*i..] (NUL-terminated),
 * total_ms[i], counts[i]; return value = number of rows (<= max) or a negative error (MLKEM_ERR_NO_DEVICE when an event
 * could not be created or recorded: incomplete rows are not reported). */
/* Measurement aid (tools/energy_probe.py): restrict the batch path of this context's *_dev KEM calls (more than 256 items) to some
 * of its kernel families -- mask bits: 1 = hash kernels, 2 = sampler, 4 = K-PKE.Encrypt, 8 = K-PKE.Decrypt; 15 = all (default).
 * The outputs of a call with stages missing are meaningless; only time and power of the remaining launches are of interest. */
MLKEM_API int mlkem_ctx_debug_stages(mlkem_ctx* ctx, unsigned mask);
MLKEM_API int mlkem_timing_begin(void);
MLKEM_API int mlkem_timing_end(char* labels, double* total_ms, int* counts, int max);

/* ---- batched KEM, device pointers ----------------------------------------------------------------- */
/* replaces KeyGen_internal(params, d, z)          ml_kem.c:1034-1084   (d, z : n x 32 ; ek : n x ek_len ; dk : n x dk_len) */
MLKEM_API int mlkem_keygen_dev(mlkem_ctx* ctx, int param_set, size_t n, const uint8_t* d, const uint8_t* z, uint8_t* ek, uint8_t* dk,
                     void* stream);
/* replaces Encaps_internal(params, ek, m)         ml_kem.c:1093-1130   (m : n x 32 ; c : n x c_len ; K : n x 32) */
MLKEM_API int mlkem_encaps_dev(mlkem_ctx* ctx, int param_set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K,
                     void* stream);
/* mlkem_encaps_dev + the encapsulation-key modulus check of KEM_Encaps (ml_kem.c:1273-1291).  REFERENCE conformance:
 * status[i] = 0 always, because the reference's check cannot fail (F3).  FIPS203 conformance: status[i] = 0, or
 * MLKEM_ERR_MODULUS when a 12-bit coefficient of ek_i is >= q (c_i / K_i are then still written but must be discarded). */
MLKEM_API int mlkem_encaps_status_dev(mlkem_ctx* ctx, int param_set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K,
                            int32_t* status, void* stream);
/* replaces KEM_Decaps(params, dk, dk_len, c, c_len) ml_kem.c:1310-1359 incl. the dk hash check:
 *   status[i] = 0, or MLKEM_ERR_HASH when H(dk[384k : 768k+32]) != dk[768k+32 : 768k+64]
 *   K[i]      = Decaps_internal(dk_i, c_i) (ml_kem.c:1136-1225) — for status -5 the reference returns NULL;
 *               K[i] then holds the value Decaps_internal would have produced.
 * `status` may be NULL (check skipped = Decaps_internal semantics). */
MLKEM_API int mlkem_decaps_dev(mlkem_ctx* ctx, int param_set, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* K, int32_t* status,
                     void* stream);

/* ---- shared-key batches, device pointers ---------------------------------------------------------------------------
 * n encapsulations to ONE encapsulation key / n decapsulations under ONE decapsulation key (a server's long-lived key):
 * the same bytes as mlkem_encaps_dev / mlkem_decaps_dev on n replicated keys, but H(ek), the dk hash check and the
 * k x k matrix are computed once per call instead of per item (35 of 44 / 36 of 51 Keccak-f per item at k = 3).
 * ek : 384k+32 bytes, dk : 768k+96 bytes (one key); m, c, K, status as in the per-item calls. */
MLKEM_API int mlkem_encaps_shared_dev(mlkem_ctx* ctx, int param_set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K,
                            void* stream);
MLKEM_API int mlkem_decaps_shared_dev(mlkem_ctx* ctx, int param_set, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* K,
                            int32_t* status, void* stream);

/* ---- K-PKE on its own, device pointers -----------------------------------------------------
 * replaces PKE_KeyGen(params, d)          ml_kem.c:651-769  d : n x 32  ->  ek : n x (384k+32), dk_pke : n x 384k
 * replaces PKE_Encrypt(params, ek, m, r)  ml_kem.c:776-936  m, r : n x 32  ->  c : n x 32(du k + dv)
 * replaces PKE_Decrypt(params, dk, c)     ml_kem.c:942-1023 dk_pke : n x 384k  ->  m : n x 32
 * (static in the reference: reached there through KeyGen_internal / Encaps_internal / Decaps_internal) */
MLKEM_API int mlkem_pke_keygen_dev(mlkem_ctx* ctx, int param_set, size_t n, const uint8_t* d, uint8_t* ek, uint8_t* dk_pke, void* stream);
MLKEM_API int mlkem_pke_encrypt_dev(mlkem_ctx* ctx, int param_set, size_t n, const uint8_t* ek, const uint8_t* m, const uint8_t* r,
                          uint8_t* c, void* stream);
MLKEM_API int mlkem_pke_decrypt_dev(mlkem_ctx* ctx, int param_set, size_t n, const uint8_t* dk_pke, const uint8_t* c, uint8_t* m,
                          void* stream);

/* ---- batched primitives, device pointers (BASELINE config 2 and component parity tests) ------------ */
/* replaces NTT(f)              ml_kem.c:287-329 ; in/out : n x uint16[256].  Inputs are taken mod 2^12 like the reference's
 * `union integer.t`; the result is bit-identical to the reference for EVERY 12-bit input: coefficients in [0, q) (FIPS 203's
 * domain, all the reference's own call sites) give the canonical transform, and coefficients >= q take an exact integer
 * path that reproduces the reference's non-modular step (ml_kem.c:317 stores f[j] - t unreduced), whose outputs can stay
 * >= q (NTT of 4095 x^0 has 4095 at index 254).  f_hat may not alias f partially (equal pointers are fine). */
MLKEM_API int mlkem_ntt_dev(mlkem_ctx* ctx, size_t n, const uint16_t* f, uint16_t* f_hat, void* stream);
/* replaces InverseNTT(fh)      ml_kem.c:336-384 ; coefficients in [0, q).  For coefficients >= q the reference's own result
 * is undefined (ml_kem.c:364-367 overflows a signed int: its -O0 and -O2 builds disagree); this entry then returns the
 * inverse transform of the input reduced mod q. */
MLKEM_API int mlkem_intt_dev(mlkem_ctx* ctx, size_t n, const uint16_t* f_hat, uint16_t* f, void* stream);
/* replaces MultiplyNTTs(f, g)  ml_kem.c:415-442 ; inputs may be any 12-bit value (as ByteDecode_12 yields) */
MLKEM_API int mlkem_multiply_ntts_dev(mlkem_ctx* ctx, size_t n, const uint16_t* f_hat, const uint16_t* g_hat, uint16_t* h_hat, void* stream);
/* replaces VectorMultiply(u, v, k) ml_kem.c:618-638 ; u, v : n x k x uint16[256] (k = 1..4, any 12-bit values),
 * w : n x uint16[256] = sum_i MultiplyNTTs(u[i], v[i]), every partial sum reduced as PolyAddition does */
MLKEM_API int mlkem_vector_multiply_dev(mlkem_ctx* ctx, int k, size_t n, const uint16_t* u, const uint16_t* v, uint16_t* w, void* stream);
/* replaces PolyAddition(u, v) / PolySubtraction(u, v)  ml_kem.c:580-592 / :599-613, coefficient by coefficient over n_values
 * uint16 values (256 per polynomial), inputs taken mod 2^12 like the reference's `union integer.t`:
 *   add: (u + v) % q ;  sub: u < v ? q - (v - u) : u - v, stored into the reference's 12-bit field.  In-place allowed. */
MLKEM_API int mlkem_poly_add_dev(mlkem_ctx* ctx, size_t n_values, const uint16_t* u, const uint16_t* v, uint16_t* z, void* stream);
MLKEM_API int mlkem_poly_sub_dev(mlkem_ctx* ctx, size_t n_values, const uint16_t* u, const uint16_t* v, uint16_t* z, void* stream);
/* replaces ByteEncode(Compress(f, d), d)   ml_kem.c:83-97 + :125-145 ; f : n x uint16[256] -> bytes : n x 32d ;
 *          d in {1, 4, 5, 10, 11} (coefficients taken mod 2^12, then mod q) ; d = 12 : ByteEncode_12 alone (ml_kem.c:736-756) */
MLKEM_API int mlkem_compress_encode_dev(mlkem_ctx* ctx, int d, size_t n, const uint16_t* f, uint8_t* bytes, void* stream);
/* replaces Decompress(ByteDecode(B, d), d) ml_kem.c:153-177 + :104-119 ; d = 12 : the raw 12-bit values, NOT reduced mod q
 *          (ml_kem.c:170) */
MLKEM_API int mlkem_decode_decompress_dev(mlkem_ctx* ctx, int d, size_t n, const uint8_t* bytes, uint16_t* f, void* stream);
/* replaces Compress(x, d) / Decompress(y, d)  ml_kem.c:83-97 / :104-119, value by value, for ANY d in 1..12 and any 12-bit
 * input (taken mod 2^12 like the reference's `union integer.t`; d = 12 is the identity; the rounded quotient wraps at 12
 * bits like the reference's field does).  n uint16 values in, n out; in-place allowed. */
MLKEM_API int mlkem_compress_dev(mlkem_ctx* ctx, int d, size_t n, const uint16_t* x, uint16_t* y, void* stream);
MLKEM_API int mlkem_decompress_dev(mlkem_ctx* ctx, int d, size_t n, const uint16_t* y, uint16_t* x, void* stream);
/* replaces SampleNTT(B)        ml_kem.c:189-245 ; seeds : n x 34 bytes (packed) */
MLKEM_API int mlkem_sample_ntt_dev(mlkem_ctx* ctx, size_t n, const uint8_t* seeds34, uint16_t* a_hat, void* stream);
/* The same with retries[i] (n bytes, may be NULL) = how often the reference's SampleNTT would have incremented B[32] and B[33] of
 * seed i before it succeeded (ml_kem.c:221-242: more than 278 triples consumed; probability < 2^-300 per seed, so always 0 in
 * practice).  a_hat[i] is the polynomial of the seed with bytes 32, 33 advanced by retries[i]; the shim writes the advanced bytes
 * back into the caller's buffer like the reference does. */
MLKEM_API int mlkem_sample_ntt_retries_dev(mlkem_ctx* ctx, size_t n, const uint8_t* seeds34, uint16_t* a_hat, uint8_t* retries, void* stream);
/* replaces SamplePolyCBD(B, eta) ml_kem.c:253-275 ; bytes : n x 64*eta */
MLKEM_API int mlkem_sample_cbd_dev(mlkem_ctx* ctx, int eta, size_t n, const uint8_t* bytes, uint16_t* f, void* stream);
/* replaces PRF(s, b, eta)      ml_kem.c:496-515 (SHAKE128!) ; in : n x 33 bytes (s || b) ; out : n x 64*eta */
MLKEM_API int mlkem_prf_dev(mlkem_ctx* ctx, int eta, size_t n, const uint8_t* in33, uint8_t* out, void* stream);
/* replaces H / G / J           ml_kem.c:521-572 on n equal-length messages; message i starts at
 * msg + i*stride (stride % 8 == 0, stride >= len).  kind: 0 = H (32 B out), 1 = G (64 B), 2 = J (32 B, SHAKE128) */
MLKEM_API int mlkem_hash_dev(mlkem_ctx* ctx, int kind, size_t n, const uint8_t* msg, unsigned len, size_t stride, uint8_t* out, void* stream);

/* replaces Sponge[Keccak-f[1600], pad10*1, r] sha3.c:257-317 on n PRE-PADDED messages of `nblocks` rate blocks each,
 * `outlen` bytes squeezed per message into rows of `out_stride`.  rate in bytes: ANY value 1..199 (the reference's Sponge takes
 * any capacity; byte-aligned ones are served: sha3_b(..., c, ...) with (1600 - c) % 8 == 0).  The five SHA-3 / SHAKE rates
 * 72 / 104 / 136 / 144 / 168 have lane-sliced kernels for large n (out_stride % 4 == 0 there); every other rate, and every call
 * of at most MLKEM_WIDE_HASH_ITEMS messages, runs one sponge per wavefront.  Used by the sha3.h front-ends of the drop-in shim. */
MLKEM_API int mlkem_keccak_sponge_dev(mlkem_ctx* ctx, unsigned rate, size_t n, const uint8_t* padded, unsigned nblocks, uint8_t* out,
                            unsigned outlen, size_t out_stride, void* stream);
/* host helper, no device work: message bits (one per byte) + suffix ("01" hash / "1111" XOF: sha3.c:408-436) + pad10*1
 * (sha3.c:226-240) -> whole rate blocks in `padded`; returns the number of blocks or a negative error */
MLKEM_API int mlkem_sha3_pad_bits(const uint8_t* msg_bits, size_t nbits, int xof, unsigned rate, uint8_t* padded, size_t padded_cap);
/* the same with the caller's own suffix bits, appended verbatim as sha3_b does (sha3.c:414-429: two bits sfx[0], sfx[1],
 * or four bits sfx[0..3]; e.g. "11" = RawSHAKE): nsfx in 0..8, bit i of the suffix = sfx_bits[i] & 1 */
MLKEM_API int mlkem_sha3_pad_suffix(const uint8_t* msg_bits, size_t nbits, const uint8_t* sfx_bits, unsigned nsfx, unsigned rate,
                          uint8_t* padded, size_t padded_cap);

/* ---- on-device exhaustive self-test of the exact fp32-pipe arithmetic ---------------------------------------------
 * The kernels evaluate `% q` (ml_kem.c:83-97, :253-275, :287-442) with fp32 FMAs on integers below 2^24.  Sweep `which`
 * (0 .. mlkem_selftest_count() - 1) checks one helper over its WHOLE input domain on the context's device against integer
 * arithmetic (fred: |x| <= 2^24; twiddle products: 258 multipliers x |b| <= 10082; Compress_d: d = 1..11 x |x| <= 4095;
 * CBD eta = 2 / 3: all 2^16 / 2^24 lane inputs; the base-case multiply-accumulate at the corners of its bound;
 * canonicalisation).  Synchronises; *violations == 0 means the property holds on this device and build. */
MLKEM_API int mlkem_selftest_count(void);
MLKEM_API int mlkem_selftest(mlkem_ctx* ctx, int which, unsigned long long* violations);

/* ---- batched KEM, host pointers (stage + run + synchronise) ----------------------------------------
 * These are the streaming front-end below with its default chunking: staging buffers, stream and context are cached
 * between calls (mlkem_stream_release() frees them), so a call allocates nothing after the first. */
MLKEM_API int mlkem_keygen(int param_set, size_t n, const uint8_t* d, const uint8_t* z, uint8_t* ek, uint8_t* dk);
MLKEM_API int mlkem_encaps(int param_set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K);
MLKEM_API int mlkem_decaps(int param_set, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* K, int32_t* status);
MLKEM_API int mlkem_ntt(size_t n, const uint16_t* f, uint16_t* f_hat);
MLKEM_API int mlkem_intt(size_t n, const uint16_t* f_hat, uint16_t* f);
/* SampleNTT (ml_kem.c:189-245) and SamplePolyCBD (ml_kem.c:253-275) over host buffers: n x 34 seed bytes /
 * n x 64*eta bytes -> n x uint16[256].  They back the reference's externally visible primitives in the drop-in shim. */
MLKEM_API int mlkem_sample_ntt(size_t n, const uint8_t* seeds34, uint16_t* a_hat);
MLKEM_API int mlkem_sample_ntt_retries(size_t n, const uint8_t* seeds34, uint16_t* a_hat, uint8_t* retries);
MLKEM_API int mlkem_sample_cbd(int eta, size_t n, const uint8_t* bytes, uint16_t* f);
MLKEM_API int mlkem_keccak_sponge(unsigned rate, size_t n, const uint8_t* padded, unsigned nblocks, uint8_t* out, unsigned outlen);
/* Compress / Decompress (ml_kem.c:83-119) over n host values, any d in 1..12 (reference test Test_Archive/CompressDecompress_test04.c) */
MLKEM_API int mlkem_compress(int d, size_t n, const uint16_t* x, uint16_t* y);
MLKEM_API int mlkem_decompress(int d, size_t n, const uint16_t* y, uint16_t* x);
/* All state the host-pointer calls cache is kept PER DEVICE (the HIP device current in the calling thread): threads that
 * work on different devices share nothing.  mlkem_host_release() zeroes and frees all of it (contexts, streams, pinned and
 * device staging) on every device; mlkem_stream_release() only the streaming engines.  Both may be called from any thread
 * at any time: the cached state is reference-counted, a release waits for calls in flight on the same device's locks, and
 * a call that overlaps a release simply rebuilds (and afterwards frees) what it needs.
 * Threads that call the host-pointer KEM entry points on the SAME device at the same time each get a streaming engine of
 * their own (streams, staging buffers, context; up to 1 + MLKEM_HOST_LANES = 8 per device, further callers queue), so that
 * the one-item calls of a multi-threaded host run side by side on the GPU; the host-pointer primitives (mlkem_ntt, ...)
 * share one context per device and queue.  KEM calls of at most 4 items (the ml_kem.h shim makes calls of one) that arrive
 * while others of the same operation and parameter set are in flight are COMBINED into one launch by whichever caller finds
 * fewer than MLKEM_HOST_COMBINE (default 2; 0: never) batches in flight; a lone caller never waits.  Results and error codes
 * are those of the separate calls. */
MLKEM_API void mlkem_host_release(void);

/* ---- layout converters and streaming front-end --------------------------------------------------
 * The reference stores every byte in a 4-byte `union byte` cell (ml_kem.h:35-38; value in bits 0-7, upper bits
 * undefined).  Device-side converters at memory bandwidth: */
MLKEM_API int mlkem_cells_to_bytes_dev(mlkem_ctx* ctx, size_t n_cells, const uint32_t* cells, uint8_t* bytes, void* stream);
MLKEM_API int mlkem_bytes_to_cells_dev(mlkem_ctx* ctx, size_t n_cells, const uint8_t* bytes, uint32_t* cells, void* stream);
/* Host-resident batches of any size in chunks (chunk_items = 0 -> 2^15, env MLKEM_STREAM_CHUNK_ITEMS): three streams
 * (H2D / kernels / D2H) and three buffer sets ordered by events, so that H2D(i+1), kernels(i) and D2H(i-1) overlap (PCIe is
 * full duplex).  Caller buffers that are pinned (hipHostMalloc, hipHostRegister or mlkem_host_register below) are handed to
 * the DMA engines directly; pageable buffers go through pinned staging with threaded copies.  Same results as the
 * plain host-pointer calls; PCIe-bound. */
MLKEM_API int mlkem_keygen_stream(int param_set, size_t n, const uint8_t* d, const uint8_t* z, uint8_t* ek, uint8_t* dk, size_t chunk_items);
MLKEM_API int mlkem_encaps_stream(int param_set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K, size_t chunk_items);
MLKEM_API int mlkem_decaps_stream(int param_set, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* K, int32_t* status, size_t chunk_items);
/* The streaming calls keep their engine (streams, events, a context, pinned + device staging buffers) cached per device
 * between calls; this zeroes and frees the engines of all devices. */
MLKEM_API void mlkem_stream_release(void);
/* pin / unpin caller memory (hipHostRegister, portable across devices) so that the streaming and *_multi calls skip the
 * staging copy for it */
/* Which operands of the calling thread's last *_stream / host-pointer KEM call were copied through the engine's pinned staging
 * buffers: bit j = span j in argument order (keygen: d z ek dk ; encaps: ek m c K ; decaps: dk c K status); 0 = every operand
 * was handed to the DMA engines in place; -1 before the first call.  Calls of at most 16 KB per operand always copy (cheaper than
 * asking the runtime what kind of memory each pointer is). */
MLKEM_API int mlkem_stream_last_staged(void);
MLKEM_API int mlkem_host_register(void* p, size_t bytes);
MLKEM_API int mlkem_host_unregister(void* p);

/* ---- in-process sharding over several devices (BASELINE configs[4]: 2^23 items over 8 x MI355X) ------------
 * The batch dimension is embarrassingly parallel: an mlkem_multi is a list of MEMBERS, each bound to one HIP device
 * (`devices[r]`; NULL = devices 0 .. n_members-1; n_members = 0 and NULL = every visible device once).  A device may be
 * listed more than once, which is how the sharded path is rehearsed on one GPU.  Member r of R takes the contiguous item
 * range mlkem_shard_range(n, r, R) (item i -> member floor(i / (n/R)), remainder spread over the first members); there is
 * no exchange between members, no collective and no RCCL.  The reference has no counterpart (single-threaded C). */
typedef struct mlkem_multi mlkem_multi;
MLKEM_API int mlkem_multi_create(mlkem_multi** out, int n_members, const int* devices, size_t chunk_items);
MLKEM_API void mlkem_multi_destroy(mlkem_multi* mm);
MLKEM_API int mlkem_multi_members(const mlkem_multi* mm);
MLKEM_API int mlkem_multi_device(const mlkem_multi* mm, int member);
MLKEM_API int mlkem_shard_range(size_t n, int member, int n_members, size_t* start, size_t* stop);
/* host-resident batch (same arguments as the *_stream calls): one host thread per member drives that member's own
 * streaming engine on its device over its item range; returns when all members are done */
MLKEM_API int mlkem_keygen_multi(mlkem_multi* mm, int param_set, size_t n, const uint8_t* d, const uint8_t* z, uint8_t* ek, uint8_t* dk,
                       size_t chunk_items);
MLKEM_API int mlkem_encaps_multi(mlkem_multi* mm, int param_set, size_t n, const uint8_t* ek, const uint8_t* m, uint8_t* c, uint8_t* K,
                       size_t chunk_items);
MLKEM_API int mlkem_decaps_multi(mlkem_multi* mm, int param_set, size_t n, const uint8_t* dk, const uint8_t* c, uint8_t* K, int32_t* status,
                       size_t chunk_items);
/* device-resident shards: arrays of length mlkem_multi_members(); shard r (n_shard[r] items, layout as in the *_dev calls)
 * lives on member r's device.  The work is enqueued on the member's own stream and the call returns without
 * synchronising; the member streams are not ordered after any other stream, so the inputs must be complete when the
 * call is made.  mlkem_multi_sync waits for every member. */
MLKEM_API int mlkem_keygen_multi_dev(mlkem_multi* mm, int param_set, const size_t* n_shard, const uint8_t* const* d, const uint8_t* const* z,
                           uint8_t* const* ek, uint8_t* const* dk);
MLKEM_API int mlkem_encaps_multi_dev(mlkem_multi* mm, int param_set, const size_t* n_shard, const uint8_t* const* ek, const uint8_t* const* m,
                           uint8_t* const* c, uint8_t* const* K);
MLKEM_API int mlkem_decaps_multi_dev(mlkem_multi* mm, int param_set, const size_t* n_shard, const uint8_t* const* dk, const uint8_t* const* c,
                           uint8_t* const* K, int32_t* const* status);
MLKEM_API int mlkem_multi_sync(mlkem_multi* mm);
/* The hipStream_t (as void*) member `member` enqueues its device-resident work on.  A caller orders it after the producers
 * of its inputs (hipStreamWaitEvent on it) and times or consumes the member's work with events recorded on it.  Member
 * streams come from a per-device pool that lives as long as the process: the handle remains a valid stream after
 * mlkem_multi_destroy (which synchronises it and returns it to the pool), so allocator bookkeeping or events that still
 * refer to it stay harmless.  NULL on a bad argument. */
MLKEM_API void* mlkem_multi_stream(mlkem_multi* mm, int member);

/* ---- randomised wrappers: seeds drawn on the host with getrandom(2) ---------------- */
/* replaces KEM_KeyGen(params)  ml_kem.c:1233-1252 for n key pairs */
MLKEM_API int mlkem_keygen_random(int param_set, size_t n, uint8_t* ek, uint8_t* dk);
/* replaces KEM_Encaps(params, ek, ek_len) ml_kem.c:1257-1305 for n encapsulations; ek_len is checked (-3) */
MLKEM_API int mlkem_encaps_random(int param_set, size_t n, const uint8_t* ek, unsigned ek_len, uint8_t* c, uint8_t* K);

#ifdef __cplusplus
}
#endif
#endif
