/*
 * oracle/mlkem_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C11, packed bytes, uint16 polynomials) of the algorithm implemented by
 * the reference rsjahnige/CRYSTALS-Kyber (ml_kem.c + sha3.c).  It is the checker the HIP path is
 * compared against; it is never linked into, called from, or used as a fallback by the product
 * library.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py) against
 *   (1) the 16 NIST FIPS-202 example vectors the reference ships (the Test_Examples/SHA text files),
 *   (2) golden vectors produced by the real reference compiled in the build container
 *       (oracle/_ref, generator oracle/gen_golden.py, fixtures tests/golden/),
 *   (3) the live reference build whenever oracle/_ref/libmlkem_ref.so is present.
 *
 * Deliberate reference quirks reproduced:
 *   F2  PRF and J are SHAKE128 (ml_kem.c:508, :546), not SHAKE256.
 *   F3  ByteDecode_12 does not reduce mod q (ml_kem.c:170), so the Encaps modulus check never fires.
 */
#ifndef MLKEM_ORACLE_H
#define MLKEM_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#define ORC_N 256
#define ORC_Q 3329

typedef struct {
    int set;      /* 512 / 768 / 1024 */
    unsigned k, eta1, eta2, du, dv;
    unsigned ek_len, dk_len, c_len;
} orc_params;

/* Conformance switch of the CHECKER (process-global; tests are single-threaded): 0 = reference-compatible (default:
 * PRF and J on SHAKE128 like ml_kem.c:508, :546), 1 = FIPS 203 (PRF and J on SHAKE256, real modulus check).  The FIPS
 * mode has no vectors in the reference; it is pinned by hashlib (SHAKE256) and an independent pure-Python restatement
 * of FIPS 203 in tests/test_fips203_mode.py. */
void orc_set_conformance(int fips203);

/* ml_kem.c:1363-1395 (init).  Returns 0, or -1 for an unknown parameter set. */
int orc_params_init(int set, orc_params *p);

/* sha3.c:207 Keccak-f[1600] on 25 little-endian 64-bit lanes. */
void orc_keccak_f1600(uint64_t st[25]);
/* sha3.c:257/408 sponge, byte-aligned messages.  rate in bytes, suffix = 0x06 (hash) / 0x1F (XOF). */
void orc_sponge(unsigned rate, uint8_t suffix, const uint8_t *in, size_t inlen, uint8_t *out, size_t outlen);
/* Bit-granular sponge: msg as one bit per byte, nbits arbitrary (NIST 5-/30-/1605-/1630-bit examples). */
void orc_sponge_bits(unsigned rate, int xof, const uint8_t *msg_bits, size_t nbits, uint8_t *out, size_t outlen);
void orc_sponge_bits_sfx(unsigned rate, const uint8_t *sfx_bits, unsigned nsfx, const uint8_t *msg_bits, size_t nbits,
                         uint8_t *out, size_t outlen);
void orc_sha3_256(const uint8_t *in, size_t n, uint8_t out[32]);
void orc_sha3_512(const uint8_t *in, size_t n, uint8_t out[64]);
void orc_shake128(const uint8_t *in, size_t n, uint8_t *out, size_t outlen);
void orc_shake256(const uint8_t *in, size_t n, uint8_t *out, size_t outlen);

/* ml_kem.c:496 PRF, :521 H, :540 J, :559 G (reference-compatible: PRF/J on SHAKE128). */
void orc_prf(const uint8_t s[32], uint8_t b, unsigned eta, uint8_t *out);
void orc_H(const uint8_t *in, size_t n, uint8_t out[32]);
void orc_J(const uint8_t *in, size_t n, uint8_t out[32]);
void orc_G(const uint8_t *in, size_t n, uint8_t out[64]);

unsigned orc_bitrev7(unsigned r);                         /* ml_kem.c:26  */
unsigned orc_compress(unsigned x, unsigned d);            /* ml_kem.c:83  */
unsigned orc_decompress(unsigned y, unsigned d);          /* ml_kem.c:104 */
void orc_byte_encode(const uint16_t F[256], unsigned d, uint8_t *B);   /* ml_kem.c:125 */
void orc_byte_decode(const uint8_t *B, unsigned d, uint16_t F[256]);   /* ml_kem.c:153 */
/* ml_kem.c:189; returns the number of seed-mutation retries taken (0 in practice). */
int orc_sample_ntt(const uint8_t B[34], uint16_t a[256]);
/* test-only generalisation: acceptance bound and triple limit as parameters ((q, 279) = the reference) */
int orc_sample_ntt_bounded(const uint8_t B[34], uint16_t a[256], unsigned bound, unsigned limit, uint8_t B_out[34]);
void orc_sample_cbd(const uint8_t *B, unsigned eta, uint16_t f[256]);  /* ml_kem.c:253 */
void orc_ntt(const uint16_t f[256], uint16_t fh[256]);                 /* ml_kem.c:287 */
void orc_intt(const uint16_t fh[256], uint16_t f[256]);                /* ml_kem.c:336 */
void orc_multiply_ntts(const uint16_t a[256], const uint16_t b[256], uint16_t h[256]); /* ml_kem.c:415 */
void orc_poly_add(const uint16_t a[256], const uint16_t b[256], uint16_t h[256]);      /* ml_kem.c:580 */
void orc_poly_sub(const uint16_t a[256], const uint16_t b[256], uint16_t h[256]);      /* ml_kem.c:599 */

/* K-PKE: ml_kem.c:651, :776, :942 */
void orc_pke_keygen(const orc_params *p, const uint8_t d[32], uint8_t *ek, uint8_t *dk_pke);
void orc_pke_encrypt(const orc_params *p, const uint8_t *ek, const uint8_t m[32], const uint8_t r[32], uint8_t *c);
void orc_pke_decrypt(const orc_params *p, const uint8_t *dk_pke, const uint8_t *c, uint8_t m[32]);

/* ML-KEM internal: ml_kem.c:1034, :1093, :1136 */
int orc_keygen(int set, const uint8_t d[32], const uint8_t z[32], uint8_t *ek, uint8_t *dk);
int orc_encaps(int set, const uint8_t *ek, const uint8_t m[32], uint8_t *c, uint8_t K[32]);
int orc_decaps_internal(int set, const uint8_t *dk, const uint8_t *c, uint8_t K[32]);
/* Public-API checks of KEM_Decaps (ml_kem.c:1310-1359): returns 0, -3 (length), -5 (hash). */
int orc_kem_decaps(int set, const uint8_t *dk, unsigned dk_len, const uint8_t *c, unsigned c_len, uint8_t K[32]);
/* Public-API checks of KEM_Encaps (ml_kem.c:1257-1291): returns 0 or -3; never -4 (F3). */
int orc_kem_encaps_check(int set, const uint8_t *ek, unsigned ek_len);

/* Batch drivers (plain loops) used by tests and by bench.py's cpu_baseline "port" timing. */
void orc_keygen_batch(int set, size_t n, const uint8_t *d, const uint8_t *z, uint8_t *ek, uint8_t *dk);
void orc_encaps_batch(int set, size_t n, const uint8_t *ek, const uint8_t *m, uint8_t *c, uint8_t *K);
void orc_decaps_batch(int set, size_t n, const uint8_t *dk, const uint8_t *c, uint8_t *K, int32_t *status);
void orc_ntt_batch(size_t n, const uint16_t *in, uint16_t *out);
void orc_intt_batch(size_t n, const uint16_t *in, uint16_t *out);

#endif
